"""GPU: wall time and convergence of the device-side svd initialiser (mcl_svd_init) at bench configurations, against the host
path (copy of X + I LAPACK SVDs) on a sample of the matrices.  python tools/svd_init_probe.py [config ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from matcouply_amd import _engine  # noqa: E402

for name in sys.argv[1:] or ["c2", "c3", "c4"]:
    cfg = dict(bench.CONFIGS[name], name=name)
    dev = torch.device("cuda", 0)
    X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
    r = cfg["r"]
    _engine.svd_init(X[: int(row_ptr[2])], row_ptr[:3], r)  # warm-up (module load)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    B, C, info = _engine.svd_init(X, row_ptr, r)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    it = info.cpu().numpy()
    # host path on the first 8 matrices, scaled
    t0 = time.perf_counter()
    n_s = min(8, I_loc)
    errs = []
    for i in range(n_s):
        m = X[int(row_ptr[i]): int(row_ptr[i + 1])].cpu().numpy().astype(np.float64)
        U = np.linalg.svd(m, full_matrices=False)[0][:, :r]
        got = B[int(row_ptr[i]): int(row_ptr[i + 1])].cpu().numpy().astype(np.float64)
        # subspace distance: || (I - U U^T) got ||
        errs.append(np.linalg.norm(got - U @ (U.T @ got)) / np.linalg.norm(got))
    dt_host = (time.perf_counter() - t0) * I_loc / n_s
    print(f"{name}: device {dt * 1e3:.1f} ms (iterations: median {int(np.median(it[:-1]))}, max {int(it[:-1].max())}, stack {int(it[-1])}, "
          f"not converged {int((it < 0).sum())}); host path ~{dt_host:.1f} s for {I_loc} matrices; subspace error vs LAPACK {max(errs):.1e}", flush=True)
