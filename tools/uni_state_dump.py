"""Dump y = B + U of the unimodal penalty (a sample of slabs) at chosen outer iterations of a config-5 stack, to analyse
the pooling behaviour offline:  python tools/uni_state_dump.py [config] [n_slabs] [iterations...] -> gpurun_out/uni_state_<config>.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "c5_32nd"
n_s = int(sys.argv[2]) if len(sys.argv) > 2 else 8
its = [int(v) for v in sys.argv[3:]] or [2, 10, 20, 40, 60]
cfg = bench.CONFIGS[name]
J = cfg["J"]
dev = torch.device("cuda", 0)
X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
kuni = [k for k, d in enumerate(cfg["regs"][1]) if d["kind"] == "unimodal"][0]
out = {}
for it in range(max(its) + 1):
    eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A()
    if it in its:
        torch.cuda.synchronize()
        reg = eng.regs[1][kuni]
        out[f"y{it}"] = (eng.B[:n_s * J] + reg.dual[:n_s * J]).cpu().numpy().astype(np.float32)  # n_s slabs x r columns
        print("iteration", it, "dumped", flush=True)
np.savez_compressed(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"uni_state_{name}.npz"), **out)
print("saved", {k: v.shape for k, v in out.items()})
