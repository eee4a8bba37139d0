"""The PCIe-inclusive rate of the public call: `cmf_aoadmm` handed HOST arrays (the reference's own input kind: a list of NumPy
matrices) against the same call on data that already lives in HBM (`PackedMatrices`).  bench.py's `value` is the resident rate; this
is the number DESIGN.md section 7 quotes beside it.  GPU box:  python tools/pcie_inclusive.py [config=c3] [n_iter=1000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from matcouply_amd import decomposition as dec

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
cfg = dict(bench.CONFIGS[name], name=name)
dev = torch.device("cuda", 0)
X, row_ptr, _ = bench.make_shard(cfg, 0, 1, dev)
torch.cuda.synchronize()
Xh = X.cpu().numpy()
mats32 = [Xh[row_ptr[i]:row_ptr[i + 1]] for i in range(len(row_ptr) - 1)]  # fp32 views of one host array
mats64 = [m.astype(np.float64) for m in mats32]                              # what a NumPy user of the reference holds
nbytes = Xh.nbytes
kw = dict(n_iter_max=n_iter, random_state=0, return_errors=True, tol=None, absolute_tol=None, **cfg["api_kwargs"])


def call(data, label):
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cmf, diag = dec.cmf_aoadmm(data, cfg["r"], **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"{label:44s} {n_iter} iterations in {best:.3f} s = {n_iter / best:8.1f} it/s (fastest of 3)", flush=True)
    return best


print(f"{cfg['desc'] if 'desc' in cfg else name}: X = {nbytes / 1e6:.0f} MB in fp32")
t_res = call(dec.PackedMatrices(X, row_ptr), "resident (PackedMatrices in HBM)")
t_h32 = call(mats32, "host, list of fp32 NumPy matrices")
t_h64 = call(mats64, "host, list of fp64 NumPy matrices")
print(f"hand-over of the host data: fp32 {t_h32 - t_res:.3f} s ({nbytes / 1e9 / max(t_h32 - t_res, 1e-9):.1f} GB/s of fp32 payload), "
      f"fp64 {t_h64 - t_res:.3f} s (conversion to fp32 on the host included)")
