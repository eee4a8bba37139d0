"""Stack-depth / spill statistics of the unimodal regressions over ALL columns of a config at a given outer iteration
(GPU box; the pooling itself is re-run on the host by tools/uni_depth.c):  python tools/uni_depth.py [config] [iterations...]"""
import ctypes, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np, torch
import bench

subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", os.path.join(root, "tools", "uni_depth.c"), "-o", "/tmp/uni_depth.so"])
lib = ctypes.CDLL("/tmp/uni_depth.so")
lib.uni_depth.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
name = sys.argv[1] if len(sys.argv) > 1 else "c5_8th"
its = [int(v) for v in sys.argv[2:]] or [3, 25]
cfg = bench.CONFIGS[name]
dev = torch.device("cuda", 0)
X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
kuni = [k for k, d in enumerate(cfg["regs"][1]) if d["kind"] == "unimodal"][0]
r, J = cfg["r"], cfg["J"]
for it in range(max(its) + 1):
    eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A()
    if it in its:
        torch.cuda.synchronize()
        n_s = min(I_loc, 256)  # a sample of slabs
        y = (eng.B[: n_s * J] + eng.regs[1][kuni].dual[: n_s * J]).cpu().numpy().astype(np.float32)
        out = np.zeros((n_s, r, 2, 5), dtype=np.int32)
        mps = np.zeros((n_s, r, 2, J), dtype=np.uint8)
        for s in range(n_s):
            blk = np.ascontiguousarray(y[s * J:(s + 1) * J])
            rev = np.ascontiguousarray(blk[::-1])
            for c in range(r):
                lib.uni_depth(blk.ctypes.data + 4 * c, J, r, 8, out[s, c, 0].ctypes.data, mps[s, c, 0].ctypes.data)
                lib.uni_depth(rev.ctypes.data + 4 * c, J, r, 8, out[s, c, 1].ctypes.data, mps[s, c, 1].ctypes.data)
        o = out.reshape(-1, 5)
        print(f"{name} iteration {it}: {o.shape[0]} sweeps | max depth: median {np.median(o[:,0]):.0f} p99 {np.percentile(o[:,0],99):.0f} max {o[:,0].max()} | "
              f"refills per sweep: median {np.median(o[:,1]):.0f} p99 {np.percentile(o[:,1],99):.0f} max {o[:,1].max()} | spills: median {np.median(o[:,2]):.0f} max {o[:,2].max()} | "
              f"max merges in a step: p99 {np.percentile(o[:,4],99):.0f} max {o[:,4].max()}", flush=True)
        mw = mps.reshape(n_s // 2, 2 * r, 2, J).max(1)  # a wave executes the largest merge count of its 64 lanes in every step
        print("   merge iterations a wave executes per element step (max over its lanes): mean %.2f; mean over lanes %.2f" % (mw.mean(), mps.mean()), flush=True)
        w = out.reshape(n_s // 2, 2 * r, 2, 5)  # a wave = 2 slabs x 32 columns, per direction
        print("   per wave (64 lanes): sum of refills: median %d max %d | max depth over lanes: median %d max %d" % (
            np.median(w[..., 1].sum(1)), w[..., 1].sum(1).max(), np.median(w[..., 0].max(1)), w[..., 0].max(1).max()), flush=True)
