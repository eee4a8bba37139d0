# kernel-trace stats of the public call with the DEFAULT tolerances (mcl_run: gated kernels + one verdict kernel per iteration)
cd /tmp; export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/dtprof; mkdir -p $O
cat > /tmp/dt.py <<'PY'
import os, sys
sys.path.insert(0, "/root/repo")
import torch, bench
from matcouply_amd import decomposition as dec
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
cfg = dict(bench.CONFIGS[name], name=name)
X, row_ptr, _ = bench.make_shard(cfg, 0, 1, torch.device("cuda", 0))
for _ in range(2):
    _, d = dec.cmf_aoadmm(dec.PackedMatrices(X, row_ptr), cfg["r"], n_iter_max=1500, random_state=0, return_errors=True, **cfg["api_kwargs"])
torch.cuda.synchronize(); print("n_iter", d.n_iter)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o dt -- python3 /tmp/dt.py ${1:-c3} > $O/out.txt 2>&1
python3 - "$O/dt_kernel_stats.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Name"].lstrip("void ").startswith("k_")]
for r in rows[:8]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:10.2f} min {float(r["MinNs"])/1e3:8.2f}')
PY
grep n_iter $O/out.txt
