"""Dump / compare the unimodal prox of two builds on the inputs of tools/unimodal_bench.py:
python tools/uni_dump_cmp.py dump <file> [I] [data]   |   python tools/uni_dump_cmp.py cmp <file_a> <file_b>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if sys.argv[1] == "dump":
    import torch
    import unimodal_bench as ub
    I = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    data = sys.argv[4] if len(sys.argv) > 4 else "peak"
    eng, aux = ub.make(I, 2048, 32, data, torch.device("cuda", 0))
    eng.B_begin(); eng.B_factor()
    U0 = eng.regs[1][0].dual.clone()
    eng.B_prox_local(0)
    torch.cuda.synchronize()
    np.save(sys.argv[2], aux.cpu().numpy())
    if os.environ.get("UNI_VERIFY") == "1":  # every column against the CPU checker
        from oracle import aoadmm_oracle as orc
        Y = (eng.B + U0).cpu().numpy().astype(np.float64)  # the kernel forms the same fp32 sum
        A = aux.cpu().numpy()
        nbad = 0
        for i in range(I):
            ref = orc.unimodal_columns(Y[i * 2048:(i + 1) * 2048], nonneg=True)
            d = np.abs(ref - A[i * 2048:(i + 1) * 2048]).max(axis=0)
            for c in np.nonzero(d > 1e-5)[0]:
                nbad += 1
                print(f"slab {i} col {int(c)}: max diff to the checker {d[c]:.3e}", flush=True)
        print(f"checker: {nbad} of {I * 32} columns differ by more than 1e-5", flush=True)
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    d = np.abs(a - b)
    bad = np.argwhere(d > 0)
    print(f"{len(bad)} of {a.size} elements differ; max abs diff {d.max():.3e}")
    if len(bad):
        cols = sorted({(int(i) // 2048, int(c)) for i, c in bad})
        print(f"{len(cols)} columns (slab, col) affected; first: {cols[:8]}")
        s, c = cols[0]
        rows = [int(i) - s * 2048 for i, cc in bad if int(i) // 2048 == s and int(cc) == c]
        print(f"column {cols[0]}: rows {rows[0]}..{rows[-1]} ({len(rows)} rows); a {a[s*2048+rows[0], c]:.6f} b {b[s*2048+rows[0], c]:.6f}")
