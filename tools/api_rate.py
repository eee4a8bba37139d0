"""Per-iteration time of the public call at several lengths (is the `api` block of bench.py a steady-state figure?):
python tools/api_rate.py [config] [default]  (GPU box; `default`: the default tolerances, i.e. mcl_run)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from matcouply_amd import decomposition as dec

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
cfg = dict(bench.CONFIGS[name], name=name)
dev = torch.device("cuda", 0)
X, row_ptr, _ = bench.make_shard(cfg, 0, 1, dev)
packed = dec.PackedMatrices(X, row_ptr)
kw = cfg["api_kwargs"]

def call(n, **tols):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dec.cmf_aoadmm(packed, cfg["r"], n_iter_max=n, random_state=0, return_errors=True, **kw, **tols)
    torch.cuda.synchronize(); return time.perf_counter() - t0

call(3, tol=None, absolute_tol=None)
prev = None
TOLS = dict() if (len(sys.argv) > 2 and sys.argv[2] == "default") else dict(tol=None, absolute_tol=None)  # default: mcl_run
print(name, "default tolerances (mcl_run)" if not TOLS else "tol=None (mcl_iterate)", "MCL_RUN_SPINS =", os.environ.get("MCL_RUN_SPINS"))
for n in (100, 400, 1600, 3200):
    t = min(call(n, **TOLS) for _ in range(3))
    msg = f"n={n}: {t:.4f} s"
    if prev:
        msg += f"; marginal {1e6 * (t - prev[1]) / (n - prev[0]):.1f} us/iter = {(n - prev[0]) / (t - prev[1]):.0f} it/s"
    print(msg, flush=True)
    prev = (n, t)
