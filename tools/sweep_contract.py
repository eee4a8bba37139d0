"""Times the two X-pass kernels at config-3 shape under the debug/env knobs of contract.hip (GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cfg = bench.CONFIGS[os.environ.get("CFG", "c3")]
dev = torch.device("cuda", 0)
X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A()
torch.cuda.synchronize()

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def prof(fn, slot, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    eng.profile_enable(n)
    for _ in range(n): fn()
    tot, cnt = eng.profile_read(slot)
    eng.profile_enable(0)
    return tot / max(cnt, 1) * 1e3

def xt():
    eng._check(eng.lib.mcl_update_C_local(eng._h))

def xc():
    eng.update_C_finish(); eng.B_begin()

def xca():  # the X C pass as the A-phase issues it (fused per-segment Gram epilogue)
    eng.update_C_finish(); eng.A_begin()

def engine_with(**switches):
    """The wave geometry (MCL_SEG_ROWS, MCL_X?_WAVES) is fixed by mcl_set_problem: one engine per setting, created AFTER the
    switches are in the environment, and the switches are removed again so that they do not leak into the next engine."""
    os.environ.update({k: str(v) for k, v in switches.items()})
    try:
        e = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
    finally:
        for k in switches:
            os.environ.pop(k, None)
    e.update_B(); e.update_C_local(); e.update_C_finish(); e.update_A()
    return e


for seg in os.environ.get("SEGS", "256").split(","):
    for waves in (1024, 2048, 4096):
        eng = engine_with(MCL_SEG_ROWS=seg, MCL_XT_WAVES=waves, MCL_XC_WAVES=waves)
        print(f"seg={seg} waves={waves}: xt {prof(xt, 1):.1f} us   xc+gram {prof(xca, 0):.1f} us  [{eng.kernel_variant(0)}]", flush=True)
        eng.close()
for waves in (512, 1024, 2048):
    eng = engine_with(MCL_XT_WAVES=waves)
    print(f"xt kernel (events) waves={waves}: {prof(xt, 1):.1f} us")
    eng.close()
for norow in ("", "1"):
    for waves in (512, 1024, 2048, 4096):
        eng = engine_with(MCL_XC_WAVES=waves, **({"MCL_XC_NOROW": 1} if norow else {}))
        print(f"xc kernel (events) norow={norow or 0} waves={waves}: {prof(xc, 0):.1f} us  [{eng.kernel_variant(0)}]")
        eng.close()
