# SQ counters of the unimodal kernels on a late (smooth) iterate of the 1/8 shard of config 5 and on the throughput form
# (tools/unimodal_bench.py): bash tools/uni_counters.sh -> gpurun_out/uni_sq/
cd /tmp; export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/uni_sq; mkdir -p $O
B="python3 $R/tools/unimodal_bench.py --I 4096 --data peak --reps 2"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/a -o a -- $B > /dev/null 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/b -o b -- $B > /dev/null 2>&1 || exit 1
python3 - $O <<'PY'
import csv, glob, os, sys, collections
acc = collections.OrderedDict()
for path in glob.glob(os.path.join(sys.argv[1], "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        if "unimodal" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0][-28:], r["Counter_Name"])
            a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for (name, cn), (n, v) in acc.items():
    print(f"{name:30s} {cn:24s} launches {n:3d}  per launch {v / n:16.0f}")
PY
