"""Per-section cycle breakdown of k_sweep (MCL_SWEEP_DBG=32): runs bench-like steps on config c3 and prints the averages."""
import os, sys
os.environ["MCL_SWEEP_DBG"] = os.environ.get("MCL_SWEEP_DBG", "32")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from matcouply_amd import _engine

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
dev = torch.device("cuda:0")
X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
for it in range(5):
    eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A()
torch.cuda.synchronize()
c = eng.internal(_engine.BUF_SWEEP_CYCLES).view(torch.int64).view(-1, 6).cpu().numpy()
import numpy as np
from matcouply_amd import _engine
names = ["tile write + issue", "(1) X C", "(2) inner loop", "(3) stores/diag/transpose", "(4) X^T B + Gram", "-"]
tot = c[:, :5].sum(axis=1)
print("waves", c.shape[0], "cycles per wave: mean", tot.mean(), "min", tot.min(), "max", tot.max())
for i in range(5):
    print(f"{names[i]:28s} mean {c[:, i].mean():10.0f}  ({100 * c[:, i].mean() / tot.mean():5.1f} %)  min {c[:, i].min()}  max {c[:, i].max()}")
wall = c[:, 5].astype(float) / 100e6
print("wall per wave (us): mean", 1e6 * wall.mean(), "-> section-cycle clock", tot.mean() / wall.mean() / 1e9, "GHz (sections cover the block loop only)")
nw = c.shape[0] // max(1, (c.shape[0] // 8 if c.shape[0] % 8 == 0 else 1)) if False else None
per = c[:, :5].sum(axis=1)
for nwv in (8, 4):
    if c.shape[0] % nwv == 0:
        print("by wave index (NW=%d):" % nwv, per.reshape(-1, nwv).mean(axis=0).astype(int))
print("variant", eng.kernel_variant(_engine.PROF_SWEEP))
