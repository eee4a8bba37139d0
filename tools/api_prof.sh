cd /tmp; export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/apiprof; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o api -- python3 $R/tools/api_rate.py c3 > $O/out.txt 2>&1
python3 - "$O/api_kernel_stats.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Name"].lstrip("void ").startswith("k_")]
for r in rows[:10]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:10.2f} min {float(r["MinNs"])/1e3:8.2f} max {float(r["MaxNs"])/1e3:8.2f}')
PY
tail -3 $O/out.txt
