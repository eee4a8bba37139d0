"""-m gpu: BASELINE config 5 at FULL size on one MI355X (I=8192, J_i=2048, K=1024, rank 32: X = 68.7 GB fp32; full penalty
stack NN(A) + [PARAFAC2, Unimodality(nn), L2Ball(1, nn)](B) + L1(0.1, nn)(C)) through the C ABI.  No CPU reference can
run this size in test time, so the checks are the size-independent properties of the domain: bitwise determinism, the
constraints on the auxiliary variables, and the fast reconstruction-error formula against the explicit residual over ALL
of X.  (Oracle parity of the same dimensions and stack at I = 6: test_scale_parity_vs_oracle[c5_dims_stack].)"""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NEED_GB = 140  # X 68.7 + B-sized state 15 + workspace ~45 + transient chunks


def test_full_size_config5_properties():
    import torch
    import bench
    from matcouply_amd._engine import DIAG_LEN

    free, total = torch.cuda.mem_get_info(0)
    if free < NEED_GB * 1e9:
        # a device that HAS the memory (an MI355X: 288 GB) but not free: config 5 would go unexercised without anybody
        # noticing - that is a failure of the run, not a reason to skip; smaller devices skip
        assert total < 200e9, (f"config 5 needs {NEED_GB} GB of free device memory; this {total / 1e9:.0f} GB device has only "
                               f"{free / 1e9:.0f} GB free - something else is holding memory on it")
        pytest.skip(f"needs {NEED_GB} GB of free device memory, the device has {total / 1e9:.0f} GB")
    cfg = bench.CONFIGS["c5"]
    I, J, K, r = cfg["I"], cfg["J"], cfg["K"], cfg["r"]
    dev = torch.device("cuda", 0)
    X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
    assert I_loc == I and X.shape == (I * J, K)
    n_it = 2

    def run():
        eng = bench.make_engine(cfg, X, row_ptr, I, 0, dev)  # seeded: identical initial state on every call
        ring = torch.zeros((n_it, DIAG_LEN), dtype=torch.float64, device=dev)
        for it in range(n_it):
            eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A()
            eng.diagnostics(out=ring[it])
        torch.cuda.synchronize()
        return eng, ring.cpu().numpy()

    eng, d1 = run()
    keep = dict(A=eng.A.clone(), C=eng.C.clone(), B=eng.B.clone(), Delta=eng.regs[1][0].aux2.clone())
    # ---- constraints on the auxiliary variables (exact up to fp32 rounding of the projection itself)
    auxP, auxU, auxL = eng.regs[1][0].aux, eng.regs[1][1].aux, eng.regs[1][2].aux
    assert float(eng.regs[0][0].aux.min()) >= 0 and float(eng.regs[2][0].aux.min()) >= 0
    assert float(auxU.min()) >= 0 and float(auxL.min()) >= 0
    for i0 in range(0, I, 1024):  # unimodality of EVERY column: once it has decreased it never increases again
        v = auxU.view(I, J, r)[i0:i0 + 1024]
        dif = torch.sign(v[:, 1:] - v[:, :-1])
        fell = torch.cummax((dif < 0).to(torch.int8), dim=1).values
        assert int(((dif > 0) & (fell == 1)).sum()) == 0
        del v, dif, fell
    norms = torch.linalg.norm(auxL.view(I, J, r), dim=1)
    assert float(norms.max()) <= 1 + 1e-6, float(norms.max())
    eye = torch.eye(r, device=dev, dtype=torch.float64)
    sample = np.random.RandomState(0).choice(I, 256, replace=False)
    P = auxP.view(I, J, r)[torch.as_tensor(sample, device=dev)].double()
    assert float((P.transpose(1, 2) @ P - eye).abs().max()) < 1e-5
    assert all(bool(torch.isfinite(t).all()) for t in (eng.A, eng.C, eng.regs[1][0].aux2)) and bool(torch.isfinite(eng.B).all())
    # ---- fast error formula (decomposition.py:445-452, no pass over X) == explicit ||X - M|| / ||X|| over all slabs
    num = torch.zeros((), dtype=torch.float64, device=dev)
    for i0 in range(0, I, 64):
        Bc = eng.B.view(I, J, r)[i0:i0 + 64]
        M = torch.einsum("ijr,ir,kr->ijk", Bc, eng.A[i0:i0 + 64], eng.C).reshape(-1, K)
        num += (X[i0 * J:(i0 + 64) * J] - M).double().pow(2).sum()
        del M
    xsq = d1[-1][5]
    fast = np.sqrt(max(0.0, xsq - 2 * d1[-1][3] + d1[-1][4]) / xsq)
    explicit = float(torch.sqrt(num)) / np.sqrt(xsq)
    np.testing.assert_allclose(fast, explicit, rtol=1e-4)
    rec = np.sqrt(np.maximum(0, d1[:, 5] - 2 * d1[:, 3] + d1[:, 4]) / d1[:, 5])
    assert rec[-1] < rec[0] < 1.5, rec
    print(f"config 5 full size: rel. rec. error {rec.tolist()}, explicit {explicit:.6f}, max column norm {float(norms.max()):.7f}")
    # ---- bitwise determinism (fixed summation orders, no float atomics): a second run from the same state
    eng.close()
    del eng, auxP, auxU, auxL, P, norms
    gc.collect()
    torch.cuda.empty_cache()
    eng2, d2 = run()
    assert np.array_equal(d1, d2)
    assert torch.equal(eng2.A, keep["A"]) and torch.equal(eng2.C, keep["C"]) and torch.equal(eng2.regs[1][0].aux2, keep["Delta"])
    assert torch.equal(eng2.B, keep["B"])
    eng2.close()
