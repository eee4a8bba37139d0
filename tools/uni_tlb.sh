# address-translation counters of the unimodal kernel (throughput form, config-5 scale): bash tools/uni_tlb.sh -> gpurun_out/uni_tlb/
cd /tmp; export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/uni_tlb; mkdir -p $O
B="python3 $R/tools/unimodal_bench.py --I 8192 --data peak --reps 2"
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/$1 -o $1 -- $B > $O/$1.log 2>&1 || { tail -5 $O/$1.log; exit 1; }; }
run a "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_PERMISSION_MISS_sum"
run b "TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
run c "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum"
# (a pass with TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum never returned on
# this pool - rocprofv3 sat silent until the box's watchdog ended the call: the TA block's counters are left out)
python3 - $O <<'PY'
import csv, glob, os, sys, collections
acc = collections.OrderedDict()
for path in sorted(glob.glob(os.path.join(sys.argv[1], "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(path)):
        if "unimodal" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0][-28:], r["Counter_Name"])
            a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for (name, cn), (n, v) in acc.items():
    print(f"{name:30s} {cn:48s} launches {n:3d}  per launch {v / n:18.0f}")
PY
