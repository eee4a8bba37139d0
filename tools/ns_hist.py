import os, sys
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import bench
from matcouply_amd import _engine
for name in ("c4", "c5s"):
    cfg = bench.CONFIGS[name]
    dev = torch.device("cuda", 0)
    X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
    eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
    for it in range(3):
        eng.B_begin(); eng.B_factor()
        for inner in range(5):
            eng.B_solve()
            for k in range(len(cfg["regs"][1])):
                eng._check(eng.lib.mcl_B_prox_local(eng._h, k))
                if k == 0:
                    torch.cuda.synchronize()
                    st = eng.internal(_engine.BUF_PF2_STATUS).view(torch.int32).cpu().numpy()
                    its = -st[st <= 0]
                    print(name, f"outer {it} inner {inner}: NS iterations min {its.min()} mean {its.mean():.1f} max {its.max()} fallback {(st>0).sum()}")
                eng.B_prox_finish(k)
        eng.update_C_local(); eng.update_C_finish(); eng.update_A()
    eng.close()
