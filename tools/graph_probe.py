"""Does a captured HIP graph of the outer iterations run them faster than the stream launches (launch gaps between the four
dependent kernels of a config-3 step)?  GPU box:  python tools/graph_probe.py [config=c3] [steps=20]
The engine is created on a side stream, `steps` iterations of mcl_iterate are captured into one graph, and the same work is timed
as stream launches and as graph replays (HIP events on that stream, 30 repetitions each, median)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = bench.CONFIGS[name]
dev = torch.device("cuda", 0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
    eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
    ring = torch.zeros((steps, 64), dtype=torch.float64, device=dev)
    eng.iterate(50, diag_ring=None)
    s.synchronize()

    def timed(fn, reps=30):
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); fn(); e1.record(s)
            s.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / steps)
        return float(np.median(ts)), float(np.min(ts))

    direct = timed(lambda: eng.iterate(steps))
    print(f"{name}: stream launches  {direct[0]:8.2f} us per iteration (min {direct[1]:.2f})", flush=True)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            eng.iterate(steps)
    except Exception as e:  # noqa: BLE001
        print(f"capture failed: {type(e).__name__}: {str(e)[:400]}", flush=True)
        sys.exit(0)
    s.synchronize()
    graph = timed(lambda: g.replay())
    print(f"{name}: graph replays    {graph[0]:8.2f} us per iteration (min {graph[1]:.2f})", flush=True)
    direct2 = timed(lambda: eng.iterate(steps))
    print(f"{name}: stream launches  {direct2[0]:8.2f} us per iteration (min {direct2[1]:.2f})  (again)", flush=True)
