"""What the exact arithmetic costs BETWEEN 2^20 and 2^24 elements of X (GPU box): outer iterations of three stacks - config 3's
(sweep), config 4's (PARAFAC2 + L2 ball, penalty-free A and C) and L2 ball on B with free A / C - with the mode forced on and off.
python tools/exact_mode_cost_mid.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
for stack in ("c3", "c4"):
    for (I, J, K, r) in ((32, 256, 256, 16), (64, 512, 128, 8), (48, 576, 256, 16), (64, 512, 256, 16), (64, 1024, 256, 32)):
        cfg = dict(bench.CONFIGS[stack], I=I, J=J, K=K, r=r)
        for exact in ("1", "0"):
            os.environ["MCL_EXACT"] = exact
            X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
            eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
            eng.iterate(20); torch.cuda.synchronize()
            t0 = time.perf_counter(); eng.iterate(100); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
            print(f"{stack} I={I} J={J} K={K} r={r} N*K=2^{torch.log2(torch.tensor(float(X.numel()))).item():.1f} MCL_EXACT={exact}: {1e6*dt:8.1f} us/iter", flush=True)
            eng.close()
