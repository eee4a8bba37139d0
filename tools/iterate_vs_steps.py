"""mcl_iterate (the fixed-count loop behind cmf_aoadmm) against the step calls bench.py times, SAME engine and buffers:
python tools/iterate_vs_steps.py [config]  (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from matcouply_amd import _engine

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
cfg = dict(bench.CONFIGS[name], name=name)
dev = torch.device("cuda", 0)
X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
ring = torch.zeros((4000, _engine.DIAG_LEN), dtype=torch.float64, device=dev)

def steps(n):
    for i in range(n):
        eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A()
        eng.diagnostics_deferred(include_replicated=True, out=ring[i])
    eng.flush_diagnostics()

def timed(f, n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(n); torch.cuda.synchronize(); return time.perf_counter() - t0

steps(600)  # settle
for rep in range(3):
    a = timed(steps, 1000)
    b = timed(lambda n: eng.iterate(n, diag_ring=ring), 1000)
    print(f"{name}: step calls {1e3 * a:.2f} us/iter, mcl_iterate {1e3 * b:.2f} us/iter", flush=True)
