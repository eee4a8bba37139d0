# kernel-trace stats of one bench configuration (GPU box): tools/prof_stats.sh <config> [steps] -> gpurun_out/ps_<config>/
cd /tmp; export TMPDIR=/tmp
R=/root/repo; C=$1; S=${2:-20}; O=$R/gpurun_out/ps_$C; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $C -- python3 $R/bench.py --config $C --steps $S --warmup 3 --no-cpu-baseline --no-api > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - "$O/${C}_kernel_stats.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").startswith("k_")]
for r in rows[:22]:
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:10.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f}')
PY
tail -n1 $O/bench.json | cut -c1-120
