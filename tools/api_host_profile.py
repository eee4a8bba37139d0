"""Where the fixed (set-up) part of a `cmf_aoadmm` call goes on the host: cProfile of one call at config 3 / 4.
python tools/api_host_profile.py [config] [n_iter]  (GPU box)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from matcouply_amd import decomposition as dec

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = dict(bench.CONFIGS[name], name=name)
dev = torch.device("cuda", 0)
X, row_ptr, _ = bench.make_shard(cfg, 0, 1, dev)
packed = dec.PackedMatrices(X, row_ptr)
kw = cfg["api_kwargs"]

def call():
    out = dec.cmf_aoadmm(packed, cfg["r"], n_iter_max=n_iter, random_state=0, return_errors=True, tol=None, absolute_tol=None, **kw)
    torch.cuda.synchronize()
    return out

call()
t0 = time.perf_counter(); call(); print(f"{name}: one call of {n_iter} iterations: {time.perf_counter() - t0:.4f} s")
pr = cProfile.Profile(); pr.enable(); call(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
