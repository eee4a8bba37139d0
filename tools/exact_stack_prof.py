"""One stack at one mid size in the exact arithmetic, for `rocprofv3 --kernel-trace --stats` (where do the fp64 paths spend
their time?).  python tools/exact_stack_prof.py [c3|c4|c5s] [I] [J] [K] [r] [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MCL_EXACT", "1")
import torch, bench
stack = sys.argv[1] if len(sys.argv) > 1 else "c4"
I, J, K, r = (int(v) for v in (sys.argv[2:6] if len(sys.argv) > 5 else (48, 576, 256, 16)))
n_it = int(sys.argv[6]) if len(sys.argv) > 6 else 50
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS[stack], I=I, J=J, K=K, r=r)
X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
eng.iterate(5); torch.cuda.synchronize()
t0 = time.perf_counter(); eng.iterate(n_it); torch.cuda.synchronize()
print(f"{stack} I={I} J={J} K={K} r={r} MCL_EXACT={os.environ['MCL_EXACT']}: {1e6 * (time.perf_counter() - t0) / n_it:.1f} us/iter")
