# A round's evidence on one GPU box (run via gpurun from the repo root):  bash tools/evidence_round.sh <round> <config> [steps] [sq] [warmup]
# (one script for every round: rounds 2-5 kept four copies that differed in the round number - VERDICT r5)
# For ONE bench configuration: un-profiled bench line, kernel-trace stats, --pmc FETCH_SIZE and --pmc WRITE_SIZE passes
# (separate runs, counters never combined with other trace domains), and for c3 the three SQ-counter passes.
# Output: gpurun_out/ev<round>/<config>/...; `python tools/collect_profiles_round.py <round> [config ...]` turns it into
# profiles/r<round>_<config>_*.
cd /tmp; export TMPDIR=/tmp
R=/root/repo; RND=$1; C=$2; S=${3:-20}; O=$R/gpurun_out/ev$RND/$C; mkdir -p $O
W=${5:-3}  # c5: 30 - the unimodal regressions reach their steady-state cost (deep stacks) only after ~25 outer iterations
B="python3 $R/bench.py --config $C --steps $S --warmup $W"
$B > $O/bench.json 2> $O/bench.err || { tail -3 $O/bench.err; exit 1; }
echo "bench: $(cut -c1-110 $O/bench.json)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B --regions 1 --no-cpu-baseline --no-api > $O/bench_traced.json 2> $O/trace.err || exit 1
echo "trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- $B --regions 1 --no-cpu-baseline --no-api > /dev/null 2> $O/fetch.err || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- $B --regions 1 --no-cpu-baseline --no-api > /dev/null 2> $O/write.err || exit 1
echo "traffic passes done"
if [ "$4" = "sq" ]; then
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sqA -o a -- $B --regions 1 --no-cpu-baseline --no-api > /dev/null 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY --output-format csv -d $O/sqB -o b -- $B --regions 1 --no-cpu-baseline --no-api > /dev/null 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/sqC -o c -- $B --regions 1 --no-cpu-baseline --no-api > /dev/null 2>&1 || echo "sqC pass failed (counter names)"
  echo "SQ passes done"
fi
# the raw counter CSVs are large: keep only the rows of the library's kernels
python3 - $O <<'PY'
import csv, glob, os, sys
for path in glob.glob(os.path.join(sys.argv[1], "*", "*counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(path)) if "k_" in r["Kernel_Name"]]
    if rows:
        with open(path, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for path in glob.glob(os.path.join(sys.argv[1], "*", "*_kernel_trace.csv")) + glob.glob(os.path.join(sys.argv[1], "*", "*agent_info.csv")):
    if "/trace/" not in path:
        os.remove(path)
PY
du -sh $O | cut -f1
