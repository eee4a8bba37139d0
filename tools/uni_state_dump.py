"""Dump y = B + U of the unimodal penalty (a sample of columns) at an early and a late outer iteration of the config-5
stack, to analyse the pooling behaviour offline: python tools/uni_state_dump.py -> gpurun_out/uni_state.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

cfg = bench.CONFIGS["c5_32nd"]
dev = torch.device("cuda", 0)
X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
out = {}
for it in range(61):
    eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A()
    if it in (2, 10, 20, 40, 60):
        torch.cuda.synchronize()
        reg = eng.regs[1][1]
        y = (eng.B[:8 * 2048] + reg.dual[:8 * 2048]).cpu().numpy()  # 8 slabs x 32 columns
        out[f"y{it}"] = y.astype(np.float32)
np.savez_compressed(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "uni_state.npz"), **out)
print("saved", {k: v.shape for k, v in out.items()})
