"""Per-iteration time of the public call at several lengths (is the `api` block of bench.py a steady-state figure?):
python tools/api_rate.py [config]  (GPU box)"""
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
for n in (100, 400, 1600, 3200):
    t = min(call(n, tol=None, absolute_tol=None) for _ in range(3))
    msg = f"n={n}: {t:.4f} s"
    if prev:
        msg += f"; marginal {1e6 * (t - prev[1]) / (n - prev[0]):.1f} us/iter = {(n - prev[0]) / (t - prev[1]):.0f} it/s"
    print(msg, flush=True)
    prev = (n, t)
