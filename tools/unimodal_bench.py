"""Time the unimodal-regression prox (k_slab_unimodal_*) alone at config-5 scale: I slabs of J rows, rank r, on
noise-like and on peak-shaped columns; optionally check that two kernel versions decide identically.

    python tools/unimodal_bench.py [--I 8192] [--J 2048] [--r 32] [--reps 3] [--data noise|peak] [--check]

--check runs the kernel selected by the environment (e.g. MCL_UNIMODAL_V3=1 for the previous default) and the default
one on the same input and compares aux (equal up to last-place roundings of a few levels; a different split decision would show as O(1)).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make(I, J, r, data, device):
    import torch

    from matcouply_amd._engine import PEN_UNIMODAL, HipEngine, NativeReg

    K = 8
    N = I * J
    g = torch.Generator(device=device)
    g.manual_seed(0)
    X = torch.rand((N, K), generator=g, device=device)
    A = torch.rand((I, r), generator=g, device=device) + 0.1
    C = torch.rand((K, r), generator=g, device=device)
    if data == "noise":
        B = torch.rand((N, r), generator=g, device=device)
    else:
        j = torch.arange(J, device=device, dtype=torch.float32)[None, :, None]
        mu = torch.rand((I, 1, r), generator=g, device=device) * J
        sig = (0.05 + 0.2 * torch.rand((I, 1, r), generator=g, device=device)) * J
        B = torch.exp(-0.5 * ((j - mu) / sig) ** 2).reshape(N, r).contiguous()
        B += 0.05 * torch.randn((N, r), generator=g, device=device)
    aux = torch.zeros((N, r), device=device)
    dual = 0.1 * torch.randn((N, r), generator=g, device=device)
    row_ptr = np.arange(I + 1, dtype=np.int64) * J
    eng = HipEngine(X, row_ptr, r, A, B, C, [[], [NativeReg(PEN_UNIMODAL, aux, dual, non_negativity=True)], []])
    return eng, aux


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--I", type=int, default=8192)
    ap.add_argument("--J", type=int, default=2048)
    ap.add_argument("--r", type=int, default=32)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--data", default="noise")
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    import torch

    device = torch.device("cuda", 0)
    eng, aux = make(args.I, args.J, args.r, args.data, device)
    eng.B_begin()
    eng.B_factor()
    Bkeep, Ukeep = eng.B.clone(), eng.regs[1][0].dual.clone()

    def run():
        eng.B.copy_(Bkeep)
        eng.regs[1][0].dual.copy_(Ukeep)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.B_prox_local(0)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    ts = [run() for _ in range(args.reps)]
    tag = ",".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("MCL_UNI") or k == "MCL_NO_UNI_COOP") or "default"
    print(f"unimodal prox + dual [{tag}] I={args.I} J={args.J} r={args.r} data={args.data}: "
          + " ".join(f"{t:.2f}" for t in ts) + " ms", flush=True)
    if args.check:
        ref = aux.clone()
        saved = {k: os.environ.pop(k) for k in list(os.environ) if k.startswith("MCL_UNI") or k == "MCL_NO_UNI_COOP"}
        eng.reload_switches()
        run()
        ndiff = int((ref != aux).sum())
        md = float((ref - aux).abs().max())
        print(f"check vs default kernel (env {saved}): {ndiff} of {aux.numel()} elements differ; max abs diff {md:.3e}",
              flush=True)
        if md > 1e-6:  # a different split would show as an O(1) difference; last-ulp level differences are expected
            sys.exit(1)


if __name__ == "__main__":
    main()
