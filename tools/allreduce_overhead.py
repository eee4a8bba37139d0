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
from matcouply_amd._rccl import DirectComm
direct = DirectComm.try_create(dist.group.WORLD)
print("direct RCCL communicator:", "ok" if direct is not None else "unavailable (torch.distributed only)", flush=True)
for name in sys.argv[1:] or ["c3_8th"]:
    cfg = bench.CONFIGS[name]
    X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
    eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
    def run(ar, steps=400):
        red = {0: lambda t: None, 1: dist.all_reduce, 2: (direct.all_reduce if direct is not None else dist.all_reduce)}[ar]
        for _ in range(40):
            eng.update_B(); gr = eng.update_C_local(); red(gr); eng.update_C_finish(); eng.update_A()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            eng.update_B(); gr = eng.update_C_local(); red(gr); eng.update_C_finish(); eng.update_A()
        torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / steps
    for rep in range(3):
        a, b, c_ = run(0), run(1), run(2)
        print(name, "no all_reduce: %.1f us/step   torch.distributed (1-rank RCCL): %.1f (+%.1f)   direct on the engine's stream: %.1f (+%.1f)"
              % (a, b, b - a, c_, c_ - a), flush=True)
    eng.close()
dist.destroy_process_group()
