"""Section timing of k_pf2_algebra_ns on BASELINE config 4 (GPU box).  Needs an instrumented library:
    MCL_BUILD_DEFS=-DMCL_NS_STAMPS python matcouply_amd/_build.py --force && python tools/ns_stamps.py
(rebuild without the define afterwards).  Prints, over the slabs of the last inner iteration: cycles (s_memtime, 100 MHz
on gfx950) of the statistics prologue, G = Delta S Delta^T, the Newton-Schulz loop, the epilogue, and the steps taken."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from matcouply_amd import _engine

cfg = bench.CONFIGS[os.environ.get("CFG", "c4")]
dev = torch.device("cuda", 0)
X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
for it in range(3):
    eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A()
torch.cuda.synchronize()
buf = eng.internal(_engine.BUF_NS_STAMPS)
st = buf[I_loc:I_loc + 16 * I_loc].view(torch.int64).view(I_loc, 8).cpu().numpy()
d = np.diff(st[:, :5], axis=1).astype(np.float64)
steps = st[:, 5]
names = ["statistics prologue", "G = D S D^T", "Newton-Schulz loop", "epilogue"]
for k, n in enumerate(names):
    print(f"{n:22s} ticks: mean {d[:, k].mean():8.1f} max {d[:, k].max():8.1f}")
print("steps: min %d mean %.1f max %d; loop ticks per step: mean %.1f" % (steps.min(), steps.mean(), steps.max(), (d[:, 2] / np.maximum(steps, 1)).mean()))
tot = (st[:, 4] - st[:, 0]).astype(np.float64)
print("whole kernel per slab ticks: mean %.1f max %.1f; first entry -> last exit: %.1f" % (tot.mean(), tot.max(), st[:, 4].max() - st[:, 0].min()))
J = np.diff(row_ptr)
order = np.argsort(-tot)[:12]
print("slowest slabs: slab J_i | prologue G loop epilogue | steps | ticks/step")
for i in order:
    print(f"  {i:5d} {J[i]:5d} | {d[i,0]:8.0f} {d[i,1]:8.0f} {d[i,2]:8.0f} {d[i,3]:8.0f} | {steps[i]:3d} | {d[i,2]/max(steps[i],1):8.0f}")
fast = np.argsort(tot)[:5]
for i in fast:
    print(f"  fast {i:5d} {J[i]:5d} | {d[i,0]:8.0f} {d[i,1]:8.0f} {d[i,2]:8.0f} {d[i,3]:8.0f} | {steps[i]:3d} | {d[i,2]/max(steps[i],1):8.0f}")
print("corr(steps, ticks/step) = %.2f; corr(J, prologue) = %.2f" % (np.corrcoef(steps, d[:, 2] / np.maximum(steps, 1))[0, 1], np.corrcoef(J, d[:, 0])[0, 1]))
t0 = st[:, 0].min()
print("kernel span: first entry -> last exit %.0f ticks; entries spread %.0f; exits spread %.0f" % (st[:, 4].max() - t0, st[:, 0].max() - t0, st[:, 4].max() - st[:, 4].min()))

# where the waves ran (HW_REG_HW_ID: wave [3:0], SIMD [5:4], pipe [7:6], CU [11:8], SH [12], SE [15:13]; HW_REG_XCC_ID [3:0])
hw, xcc = st[:, 6], st[:, 7] & 0xf
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
uniq, cnt = np.unique(key, return_counts=True)
print("SIMDs used: %d; with 1 / 2 / 3+ Newton-Schulz waves: %d / %d / %d" % (len(uniq), (cnt == 1).sum(), (cnt == 2).sum(), (cnt >= 3).sum()))
cukey = (((xcc * 8 + se) * 2 + sh) * 16 + cu)
cu_u, cu_c = np.unique(cukey, return_counts=True)
print("CUs used: %d; waves per CU histogram: %s" % (len(cu_u), dict(zip(*np.unique(cu_c, return_counts=True)))))
per_simd = dict(zip(uniq, cnt))
shared = np.array([per_simd[k] for k in key])
tps = d[:, 2] / np.maximum(steps, 1)
for n in sorted(set(shared)):
    m = shared == n
    print("  waves on a SIMD with %d such wave(s): %d, loop ticks per step mean %.0f, whole-slab ticks mean %.0f" % (n, m.sum(), tps[m].mean(), tot[m].mean()))
print("per XCD, (SE, SH) -> CUs used:", {int(x): sorted(set(zip(se[xcc == x].tolist(), sh[xcc == x].tolist()))) for x in np.unique(xcc)[:2]})
