"""Where and when the waves of the unimodal regressions run at config 5 (GPU box).  Needs an instrumented library:
    MCL_BUILD_DEFS=-DMCL_UNI_STAMPS python matcouply_amd/_build.py --force && python tools/uni_stamps.py
(rebuild without the define afterwards).  Per column group (= wave) of the LAST launch: start / end on the 100 MHz constant
clock, the time its sweeps ended, HW_ID and XCC_ID.  Prints the distribution of the wave durations, the occupancy of the
SIMDs over the launch (how much of slots x span the waves fill: what a better order or placement could still gain) and the
waves per SIMD the dispatcher produced."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from matcouply_amd import _engine

cfg = dict(bench.CONFIGS[os.environ.get("CFG", "c5")])
dev = torch.device("cuda", 0)
X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
n_it = int(os.environ.get("ITERS", "31"))
for it in range(n_it):
    eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A()
torch.cuda.synchronize()
nw = (I_loc * cfg["r"] + 63) // 64
def grab():
    buf = eng.internal(_engine.BUF_NS_STAMPS)
    return buf[I_loc:I_loc + 16 * I_loc].view(torch.int64).view(I_loc, 8).cpu().numpy()[:nw].copy()
st_prev = grab()
eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A()
torch.cuda.synchronize()
st = grab()
d_prev, d_now = (st_prev[:, 1] - st_prev[:, 0]).astype(float), (st[:, 1] - st[:, 0]).astype(float)
print("durations of the same column groups one outer iteration apart: correlation %.3f, mean |change| %.1f %%" % (
    np.corrcoef(d_prev, d_now)[0, 1], 100 * np.mean(np.abs(d_now - d_prev) / d_prev)))
out = os.environ.get("STAMPS_OUT")
if out:
    np.savez_compressed(out, prev=st_prev, now=st)
t0, t1, hw, xcc, ts = st[:, 0], st[:, 1], st[:, 2], st[:, 3] & 0xf, st[:, 4]
ok = t1 > t0
print(f"{nw} waves, {ok.sum()} with stamps")
dur = (t1 - t0)[ok] / 100.0  # us
sw = (ts - t0)[ok] / 100.0
span = (t1[ok].max() - t0[ok].min()) / 100.0
print("wave duration us: min %.0f p10 %.0f median %.0f mean %.0f p90 %.0f max %.0f; sweeps part mean %.0f (%.0f %%)" % (
    dur.min(), np.percentile(dur, 10), np.median(dur), dur.mean(), np.percentile(dur, 90), dur.max(), sw.mean(), 100 * sw.mean() / dur.mean()))
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)[ok]
uniq = np.unique(key)
print("launch span %.0f us; SIMDs used %d; sum of wave durations / (2 waves x SIMDs x span) = %.3f" % (span, len(uniq), dur.sum() / (2 * len(uniq) * span)))
# per SIMD: the number of waves it ran and the time it was busy with at least one / with two waves
start, end = t0[ok] / 100.0, t1[ok] / 100.0
T0 = start.min()
busy1 = busy2 = 0.0
nper = []
maxc = []
for k in uniq:
    m = key == k
    ev = sorted([(s, 1) for s in start[m]] + [(e, -1) for e in end[m]])
    c, last, mc = 0, T0, 0
    for t, d in ev:
        if c >= 1: busy1 += t - last
        if c >= 2: busy2 += t - last
        c += d; last = t; mc = max(mc, c)
    nper.append(m.sum()); maxc.append(mc)
nper, maxc = np.array(nper), np.array(maxc)
print("waves per SIMD over the launch: min %d mean %.2f max %d; most waves resident at once on a SIMD: histogram %s" % (
    nper.min(), nper.mean(), nper.max(), dict(zip(*np.unique(maxc, return_counts=True)))))
print("SIMD time with >= 1 wave: %.3f of SIMDs x span; with >= 2 waves: %.3f" % (busy1 / (len(uniq) * span), busy2 / (len(uniq) * span)))
order = np.argsort(start)
late = start > T0 + 0.5 * span
print("waves starting in the second half of the launch: %d; last start at %.0f us of %.0f; ends: p50 %.0f p90 %.0f p99 %.0f max %.0f" % (
    late.sum(), start.max() - T0, span, np.percentile(end - T0, 50), np.percentile(end - T0, 90), np.percentile(end - T0, 99), (end - T0).max()))
