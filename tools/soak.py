"""Long runs of the public call (chunk boundaries of the device-side stopping rule, ring wrap-around, no drift / hang):
python tools/soak.py  (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from matcouply_amd import decomposition as dec

dev = torch.device("cuda", 0)
for name, n_max, extra in (("c3", 10000, {}), ("c3", 9000, dict(tol=1e-9, absolute_tol=1e-14)), ("c4", 5000, dict(tol=1e-7)),
                           ("c2", 20000, dict(tol=None, absolute_tol=None))):
    cfg = dict(bench.CONFIGS[name], name=name)
    X, row_ptr, _ = bench.make_shard(cfg, 0, 1, dev)
    t0 = time.perf_counter()
    cmf, diag = dec.cmf_aoadmm(dec.PackedMatrices(X, row_ptr), cfg["r"], n_iter_max=n_max, random_state=0, return_errors=True,
                               **cfg["api_kwargs"], **extra)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    re = np.asarray(diag.rec_errors)
    ok = np.isfinite(re).all() and np.isfinite(np.asarray(diag.regularized_loss)).all()
    print(f"{name} {extra or 'default tolerances'}: n_iter {diag.n_iter} of {n_max} in {dt:.2f} s; message {diag.message!r}; "
          f"rec_error {re[0]:.4f} -> {re[-1]:.6f}; lists {len(diag.rec_errors)}/{len(diag.feasibility_gaps)}; finite {ok}", flush=True)
    del X, cmf
