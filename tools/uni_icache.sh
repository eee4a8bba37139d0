# instruction-cache behaviour of the unimodal kernel (throughput form, config-5 scale): bash tools/uni_icache.sh -> gpurun_out/uni_ic/
cd /tmp; export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/uni_ic; mkdir -p $O
B="python3 $R/tools/unimodal_bench.py --I 8192 --data peak --reps 2"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O/a -o a -- $B > $O/a.log 2>&1 || { tail -5 $O/a.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/b -o b -- $B > $O/b.log 2>&1 || { tail -5 $O/b.log; exit 1; }
python3 - $O <<'PY'
import csv, glob, os, sys, collections
acc = collections.OrderedDict()
for path in glob.glob(os.path.join(sys.argv[1], "*", "*counter_collection.csv")) + glob.glob(os.path.join(sys.argv[1], "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        if "unimodal" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0][-28:], r["Counter_Name"])
            a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for (name, cn), (n, v) in acc.items():
    print(f"{name:30s} {cn:28s} launches {n:3d}  per launch {v / n:16.0f}")
PY
