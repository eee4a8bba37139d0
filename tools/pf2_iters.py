import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from matcouply_amd import _engine
cfg = bench.CONFIGS["c4"]
dev = torch.device("cuda", 0)
X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
r = cfg["r"]
for it in range(3):
    # run the B-phase step by step to look at the LAST inner iteration's NS status before Jacobi overwrites
    eng.B_begin(); eng.B_factor()
    for inner in range(5):
        eng.B_solve()
        eng._check(eng.lib.mcl_B_prox_local(eng._h, 0))
        torch.cuda.synchronize()
        st = eng.internal(_engine.BUF_PF2_STATUS).view(torch.int32).cpu().numpy()
        S = eng.internal(_engine.BUF_PF2_GRAM).view(torch.float64).view(-1, r, r).cpu().numpy()
        D = eng.regs[1][0].aux2.cpu().numpy().astype(np.float64)
        bad = np.where(st > 0)[0]
        msg = f"outer {it} inner {inner}: fallback {len(bad)}"
        if len(bad):
            i = bad[0]
            G = D @ S[i] @ D.T
            ev = np.linalg.eigvalsh(G)
            msg += f" | slab {i}: J={row_ptr[i+1]-row_ptr[i]} eig(G)/tr min {ev[0]/ev.sum():.2e} max {ev[-1]/ev.sum():.2e}; cond(Delta) {np.linalg.cond(D):.2e} cond(S) {np.linalg.cond(S[i]):.2e}"
        print(msg)
        eng.B_prox_finish(0)
        eng.B_prox_local(1); eng.B_prox_finish(1)
    eng.update_C_local(); eng.update_C_finish(); eng.update_A()
