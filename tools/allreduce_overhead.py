"""What a collective costs a step besides the wire: the sharded loop of bench.py on ONE rank with and without a 1-rank RCCL
all_reduce of [G | R] between the C-phase reduction and its finish (torch.distributed runs collectives on its own stream:
two event dependencies + a launch).  GPU box:  python tools/allreduce_overhead.py [config ...]   (default c3_8th)"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist, bench
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
for name in sys.argv[1:] or ["c3_8th"]:
    cfg = bench.CONFIGS[name]
    X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
    eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
    def run(ar, steps=200):
        for _ in range(20):
            eng.update_B(); gr = eng.update_C_local()
            if ar: dist.all_reduce(gr)
            eng.update_C_finish(); eng.update_A()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            eng.update_B(); gr = eng.update_C_local()
            if ar: dist.all_reduce(gr)
            eng.update_C_finish(); eng.update_A()
        torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / steps
    for rep in range(2):
        print(name, "no all_reduce: %.1f us/step   with 1-rank RCCL all_reduce: %.1f us/step" % (run(False), run(True)), flush=True)
    eng.close()
dist.destroy_process_group()
