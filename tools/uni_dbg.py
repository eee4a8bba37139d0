"""Where the time of the pruned unimodal kernel goes (GPU box; needs the library built with -DMCL_UNI_DBG, see below): the prox
on frozen steady-state inputs with parts of its memory traffic switched off (WRONG results - timing only).
    python - <<<'from matcouply_amd import _build; _build.build_library(defs=["-DMCL_UNI_DBG"], out_lib="build_ab/libmatcouply_hip_unidbg.so", build_dir="build_ab", only=["unimodal.hip"])'
    MCL_TEST_LIB=build_ab/libmatcouply_hip_unidbg.so python tools/uni_dbg.py [config=c5] [iterations=30]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matcouply_amd import _engine
if os.environ.get("MCL_TEST_LIB"):
    _engine.LIB_PATH = os.path.abspath(os.environ["MCL_TEST_LIB"])
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "c5"
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = bench.CONFIGS[name]
dev = torch.device("cuda", 0)
X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
kuni = [k for k, d in enumerate(cfg["regs"][1]) if d["kind"] == "unimodal"][0]
reg = eng.regs[1][kuni]
for _ in range(n_it):
    eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A()
torch.cuda.synchronize()
B0, U0 = eng.B.clone(), reg.dual.clone()
eng.B_begin(); eng.B_factor()
os.environ["MCL_UNI_SPLIT"] = "0"
eng.reload_switches()
labels = {0: "everything on", 1: "no record stores", 2: "no error stores", 4: "no spill stores", 8: "no emit", 7: "no record / error / spill stores",
          15: "sweeps only, no stores", 16: "-"}
if hasattr(eng.lib, "mcl_uni_dbg_counters"):
    import ctypes
    eng.lib.mcl_uni_dbg_counters((ctypes.c_ulonglong * 16)())  # clear what the outer iterations counted
labels.update({47: "sweeps only, no stores, no pooling", 79: "sweeps only, no stores, no push", 143: "sweeps only, no stores, no coop refill",
               239: "loads + arithmetic of the steps only", 32: "no pooling", 96: "no pooling, no push"})
labels.update({256: "everything on + section timers", 271: "sweeps only, no stores + section timers", 512: "everything on + batch timers (phase B)"})
pads = [int(v) for v in os.environ.get("UNI_PADS", "0").split(",")]
for pad, dbg in [(p_, d_) for p_ in pads for d_ in ((0, 256, 512) if os.environ.get("UNI_TIMERS") else (0, 1, 2, 4, 8, 7, 15))]:
    os.environ["MCL_UNI_PAD_LDS"] = str(pad)
    if len(pads) > 1:
        print(f" pad {pad} B of LDS", flush=True)
    os.environ["MCL_UNI_DBG"] = str(dbg)
    ts = []
    for rep in range(3):
        eng.B.copy_(B0); reg.dual.copy_(U0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.B_prox_local(kuni); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"  dbg {dbg:2d} {labels[dbg]:40s} " + " ".join(f"{t:8.3f}" for t in ts) + " ms", flush=True)
    if hasattr(eng.lib, "mcl_uni_dbg_counters"):  # event counts of the debug build, per wave and launch
        import ctypes
        out = (ctypes.c_ulonglong * 16)()
        eng.lib.mcl_uni_dbg_counters(out)
        n_waves = (X.shape[0] // cfg["J"]) * cfg["r"] // 64 if isinstance(cfg["J"], int) else 1
        names = ["element steps", "pooling trips", "steps with a spill", "coop refills (prefetched)", "coop refills (blocking load)",
                 "dry refills (prefetched)", "dry refills (blocking load)", "-"]
        print("     per wave and launch: " + ", ".join(f"{nm} {out[i] / (3 * n_waves):.0f}" for i, nm in enumerate(names[:7])), flush=True)
        if out[14]:
            print(f"     phase B, per batch of 8 steps: wait for the batch's loads + conversions {out[14] / max(out[7], 1):.0f} cycles, issue of the next "
                  f"batch's loads {out[11] / max(out[7], 1):.0f}; batches per wave {out[7] / (3 * n_waves):.0f}; kernel per wave {out[13] / (3 * n_waves):.0f}", flush=True)
        elif out[13] and out[8] + out[9] + out[10]:
            steps = max(out[0], 1)
            secs = ["coop refill", "push", "pooling", "post", "between steps"]
            print("     cycles (s_memtime, 100 MHz ticks x ?) per step: " + ", ".join(f"{nm} {out[8 + i] / steps:.1f}" for i, nm in enumerate(secs))
                  + f"; kernel per wave {out[13] / (3 * n_waves):.0f}", flush=True)
