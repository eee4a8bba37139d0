"""When and where the waves of the one-pass sweep run (MCL_SWEEP_DBG=64: the instrumented twin of the config-2/3 kernel records
entry / start of the block loop / end on the 100 MHz clock, HW_ID and XCC_ID per wave).  GPU box:
    python tools/sweep_stamps.py [config=c3] [out.npz]
Prints the distribution of the wave durations, the spread of their ends (the kernel lasts as long as its slowest wave), and the
means per XCD / shader engine."""
import os, sys
os.environ["MCL_SWEEP_DBG"] = "64"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from matcouply_amd import _engine

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
dev = torch.device("cuda:0")
X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
for it in range(8):
    eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A()
torch.cuda.synchronize()
c = eng.internal(_engine.BUF_SWEEP_CYCLES).view(torch.int64).view(-1, 6).cpu().numpy()
c = c[c[:, 2] > 0]
if len(c) == 0:
    sys.exit(f"no stamps: the instrumented twin exists for the K = 256 / rank <= 16 sweep only (variant {eng.kernel_variant(_engine.PROF_SWEEP)})")
if len(sys.argv) > 2:
    np.savez_compressed(sys.argv[2], stamps=c)
T0 = c[:, 0].min()
entry, start, end = (c[:, 0] - T0) / 100.0, (c[:, 1] - T0) / 100.0, (c[:, 2] - T0) / 100.0
hw, xcc = c[:, 3], c[:, 4] & 0xf
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
dur = end - start
q = lambda v: "min %.1f p10 %.1f median %.1f mean %.1f p90 %.1f max %.1f" % (v.min(), np.percentile(v, 10), np.median(v), v.mean(), np.percentile(v, 90), v.max())
print(f"{len(c)} waves; variant {eng.kernel_variant(_engine.PROF_SWEEP)}")
print("entry (us after the first):", q(entry))
print("block loop starts:         ", q(start))
print("ends:                      ", q(end))
print("block loop durations:      ", q(dur))
print("kernel span %.1f us; mean end / span = %.3f (what a perfectly balanced launch would gain: %.1f %%)" % (end.max(), end.mean() / end.max(), 100 * (1 - end.mean() / end.max())))
for x in range(8):
    m = xcc == x
    if m.any():
        print("  XCD %d: %4d waves, duration mean %.1f max %.1f, last end %.1f; core clock over the block loop %.3f GHz" % (
            x, m.sum(), dur[m].mean(), dur[m].max(), end[m].max(), (c[m, 5] / (dur[m] * 1e3)).mean()))
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu)
cu_end = np.array([end[key == k].max() for k in np.unique(key)])
cu_dur = np.array([dur[key == k].mean() for k in np.unique(key)])
print("per CU (%d CUs): last end" % len(cu_end), q(cu_end), "| mean duration", q(cu_dur))
sk = ((key * 4) + simd)
uniq, cnt = np.unique(sk, return_counts=True)
print("waves per SIMD histogram:", dict(zip(*np.unique(cnt, return_counts=True))))
order = np.argsort(-dur)[:8]
print("slowest waves: index (xcd, se, cu, simd) duration:", [(int(i), int(xcc[i]), int(se[i]), int(cu[i]), int(simd[i]), round(float(dur[i]), 1)) for i in order])
