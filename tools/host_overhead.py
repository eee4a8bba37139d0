"""Host enqueue time vs device time of one bench.py step on a shard of config 3 (the per-rank problem of an N-GPU run).

    python tools/host_overhead.py [--config c3_8th] [--steps 300] [--allreduce]

Prints, per step: the host time to ENQUEUE a step (loop without synchronisation), the device time (events around
the whole loop), and - with --allreduce - the same with a single-rank RCCL all-reduce of [G | R] in the loop
(c10d + RCCL launch overhead as a rank of a multi-GPU run pays it; the wire time of a real ring is not in it).
If enqueue >= device, the sharded loop is launch-bound and only a captured graph can shorten it.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3_8th")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--allreduce", action="store_true")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    import bench
    from matcouply_amd._engine import DIAG_LEN

    cfg = bench.CONFIGS[args.config]
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if args.allreduce:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, device)
    eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, device)
    ring = torch.zeros((args.steps + 10, DIAG_LEN), dtype=torch.float64, device=device)

    def step(it, ar):
        eng.update_B()
        gr = eng.update_C_local()
        if ar:
            dist.all_reduce(gr)
        eng.update_C_finish()
        eng.update_A()
        eng.diagnostics(include_replicated=True, out=ring[it])

    for ar in ([False, True] if args.allreduce else [False]):
        for it in range(10):
            step(it, ar)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        for it in range(args.steps):
            step(10 + it, ar)
        t_enq = time.perf_counter() - t0
        e1.record()
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"{args.config} allreduce={ar}: enqueue {1e6 * t_enq / args.steps:.1f} us/step, device "
              f"{1e3 * e0.elapsed_time(e1) / args.steps:.1f} us/step, wall {1e6 * t_all / args.steps:.1f} us/step", flush=True)
    if args.allreduce:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
