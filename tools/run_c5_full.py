"""BASELINE config 5 at FULL size on ONE MI355X: I=8192, J_i=2048, K=1024, rank 32 (X = 68.7 GB fp32), penalty stack
NN(A) + [PARAFAC2, Unimodality(nn), L2Ball(1, nn)](B) + L1(0.1, nn)(C).  Prints timing and size-independent checks."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from matcouply_amd._engine import DIAG_LEN

I, J, K, r = (int(os.environ.get(k, d)) for k, d in (("C5_I", 8192), ("C5_J", 2048), ("C5_K", 1024), ("C5_R", 32)))
cfg = dict(bench.CONFIGS["c5s"], I=I, J=J, K=K, r=r)
dev = torch.device("cuda", 0)
t0 = time.time()
g = torch.Generator(device=dev); g.manual_seed(0)
C_true = torch.rand((K, r), generator=g, device=dev)
X = torch.empty((I * J, K), dtype=torch.float32, device=dev)
chunk = 64
for i0 in range(0, I, chunk):
    n = min(chunk, I - i0)
    A_t = torch.rand((n, r), generator=g, device=dev) + 0.1
    # unimodal, non-negative B_i*: Gaussian bumps with random centres/widths
    t = torch.linspace(0, 1, J, device=dev)[None, :, None]
    mu = torch.rand((n, 1, r), generator=g, device=dev)
    sig = 0.05 + 0.2 * torch.rand((n, 1, r), generator=g, device=dev)
    B_t = torch.exp(-0.5 * ((t - mu) / sig) ** 2)
    Xc = torch.einsum("ijr,ir,kr->ijk", B_t, A_t, C_true)
    Xc += 0.05 * torch.randn(Xc.shape, generator=g, device=dev)
    X[i0 * J:(i0 + n) * J] = Xc.reshape(n * J, K)
    del Xc, B_t
torch.cuda.synchronize()
row_ptr = np.arange(I + 1, dtype=np.int64) * J
print(f"generated X {X.numel() * 4 / 1e9:.1f} GB in {time.time() - t0:.1f} s", flush=True)
eng = bench.make_engine(cfg, X, row_ptr, I, 0, dev)
print(f"workspace {eng.workspace.numel() / 1e9:.1f} GB; device memory in use {torch.cuda.memory_allocated() / 1e9:.1f} GB", flush=True)
n_it = int(os.environ.get("C5_ITERS", 4))
ring = torch.zeros((n_it + 1, DIAG_LEN), dtype=torch.float64, device=dev)
eng.diagnostics(out=ring[0])
torch.cuda.synchronize()
times = []
for it in range(n_it):
    t1 = time.time()
    eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A(); eng.diagnostics(out=ring[it + 1])
    torch.cuda.synchronize()
    times.append(time.time() - t1)
d = ring.cpu().numpy()
rec = np.sqrt(np.maximum(0, d[:, 5] - 2 * d[:, 3] + d[:, 4]) / d[:, 5])
S_X, S_B = 4.0 * I * J * K, 4.0 * I * J * r
alg = 2 * S_X + (5 + 4 * 3) * S_B
out = dict(config=f"c5 full: I={I} J={J} K={K} r={r}", X_GB=round(S_X / 1e9, 2), s_per_iter=[round(t, 4) for t in times],
           iters_per_s=round(1 / min(times), 3), algorithmic_GB_per_iter=round(alg / 1e9, 1),
           hbm_gbps_algorithmic=round(alg / min(times) / 1e9, 1), rel_rec_error=[round(float(v), 5) for v in rec],
           kernels=[eng.kernel_variant(k) for k in range(3)])
# size-independent checks: constraints hold exactly on the auxiliary variables
auxU, auxL = eng.regs[1][1].aux, eng.regs[1][2].aux
out["unimodal_aux_min"] = float(auxU.min())
col_norm = torch.linalg.norm(auxL.view(I, J, r), dim=1)
out["l2ball_max_col_norm"] = float(col_norm.max())
v = auxU.view(I, J, r)[:64]
dif = torch.sign(v[:, 1:] - v[:, :-1])
# unimodal: once a column starts to decrease it never increases again
dec_seen = torch.cummax((dif < 0).int(), dim=1).values
out["unimodal_violations_first64"] = int(((dif > 0) & (dec_seen == 1)).sum())
P = eng.regs[1][0].aux.view(I, J, r)[:8].double()
out["pf2_orthogonality_err"] = float((P.transpose(1, 2) @ P - torch.eye(r, device=dev, dtype=torch.float64)).abs().max())
out["all_finite"] = bool(torch.isfinite(eng.A).all() and torch.isfinite(eng.B).all() and torch.isfinite(eng.C).all())
print(json.dumps(out), flush=True)
