"""What the exact-products mode of small problems costs (GPU box): outer iterations of config 3's stack at sizes around the
2^20-element limit, with the mode forced on and off.  python tools/exact_mode_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
for (I, J, K, r) in ((16, 256, 256, 16), (8, 128, 128, 8), (15, 50, 20, 3), (64, 64, 256, 32), (4, 1024, 256, 16)):
    cfg = dict(bench.CONFIGS["c3"], I=I, J=J, K=K, r=r)
    for exact in ("1", "0"):
        os.environ["MCL_EXACT"] = exact
        X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
        eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
        eng.iterate(20); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.iterate(200); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        print(f"I={I} J={J} K={K} r={r} N*K={I*J*K:8d} MCL_EXACT={exact}: {1e6*dt:8.1f} us/iter", flush=True)
        eng.close()
