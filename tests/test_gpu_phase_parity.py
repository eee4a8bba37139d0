"""-m gpu: one call of each phase through the C ABI vs (a) the golden outputs of the reference and (b) the oracle.

Tolerance: 1e-5 relative (Frobenius) - the bar BASELINE.json's north_star states for the fp32 engine vs the
fp64 NumPy reference."""
import numpy as np
import pytest

from tests.helpers import engine_from_oracle_state, load_npz, manifest_of, rel_err, to_np
from tests.test_oracle_golden import _phase_state

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("kernel_paths")]  # phase goldens: both arithmetic paths
TOL = 1e-5


def _run_phase(name, mode):
    import torch

    arrs = load_npz(f"phase_{name}.npz")
    manifest = manifest_of(arrs)
    worst = 0.0
    for ci, case in enumerate(manifest):
        st = _phase_state(mode, case)
        eng = engine_from_oracle_state(st)
        if mode == 1:
            eng.update_B()
            factor = eng.B
        elif mode == 2:
            eng.update_C_local()
            eng.update_C_finish()
            factor = eng.C
        else:
            eng.update_A()
            factor = eng.A
        torch.cuda.synchronize()
        errs = {"factor": rel_err(to_np(factor), arrs[f"c{ci}_factor"])}
        if mode == 0:
            errs["rhses"] = rel_err(to_np(eng.rhses()), arrs[f"c{ci}_rhses"])
            errs["cross"] = rel_err(to_np(eng.cross_products()), arrs[f"c{ci}_cross_products"])
        for s, d in enumerate(case["regs"]):
            reg = eng.regs[mode][s]
            if d["kind"] == "parafac2":
                errs[f"P{s}"] = rel_err(to_np(reg.aux), arrs[f"c{ci}_aux{s}_P"])
                errs[f"Delta{s}"] = rel_err(to_np(reg.aux2), arrs[f"c{ci}_aux{s}_Delta"])
            else:
                errs[f"aux{s}"] = rel_err(to_np(reg.aux), arrs[f"c{ci}_aux{s}"])
            errs[f"dual{s}"] = rel_err(to_np(reg.dual), arrs[f"c{ci}_dual{s}"])
        bad = {k: v for k, v in errs.items() if not (v < TOL)}
        assert not bad, (name, ci, case, bad)
        worst = max(worst, max(errs.values()))
        eng.close()
    return worst


def test_phase_B_goldens():
    print("worst rel err B:", _run_phase("B", 1))


def test_phase_C_goldens():
    print("worst rel err C:", _run_phase("C", 2))


def test_phase_A_goldens():
    print("worst rel err A:", _run_phase("A", 0))


@pytest.mark.parametrize("name,mode", [("B", 1), ("C", 2), ("A", 0)])
def test_public_admm_update_functions(name, mode):
    """admm_update_A / admm_update_B / admm_update_C with the reference's positional signature against its own outputs"""
    from matcouply_amd import decomposition as dec
    from tests.helpers import split_rows
    from tests.test_gpu_end_to_end import _regs_from_state

    arrs = load_npz(f"phase_{name}.npz")
    n_run = 0
    for ci, case in enumerate(manifest_of(arrs)):
        if case["inner"] <= 0 or (mode == 0 and case["constant"] is False and
                                  any(d["kind"] in ("l2ball", "unimodal") for d in case["regs"])):
            continue  # (matrix penalties on A need a constant feasibility penalty: the reference raises there too)
        st = _phase_state(mode, case)
        rp = st.row_ptr
        mats = split_rows(st.X, rp)
        cmf = (None, (st.A.copy(), split_rows(st.B, rp), st.C.copy()))
        reg = _regs_from_state(st)[mode]
        aux = [(split_rows(z[0], rp), z[1].copy()) if isinstance(z, tuple) else (split_rows(z, rp) if mode == 1 else z.copy())
               for z in st.aux[mode]]
        dual = [split_rows(u, rp) if mode == 1 else u.copy() for u in st.dual[mode]]
        args = (mats, reg, cmf, aux, dual, case["l2"], case["inner"], None, case["scale"])
        if mode == 1:
            out_cmf, aux_o, dual_o = dec.admm_update_B(*args, case["constant"], None)
            factor = np.concatenate(out_cmf[1][1])
        elif mode == 2:
            out_cmf, aux_o, dual_o = dec.admm_update_C(*args, None)
            factor = out_cmf[1][2]
        else:
            out_cmf, aux_o, dual_o, (rhses, cross) = dec.admm_update_A(*args, case["constant"], None)
            factor = out_cmf[1][0]
            assert rel_err(np.stack(rhses), arrs[f"c{ci}_rhses"]) < TOL
            assert rel_err(np.stack(cross), arrs[f"c{ci}_cross_products"]) < TOL
        assert rel_err(factor, arrs[f"c{ci}_factor"]) < TOL, (name, ci, case)
        for s, d in enumerate(case["regs"]):
            if d["kind"] == "parafac2":
                assert rel_err(np.concatenate(aux_o[s][0]), arrs[f"c{ci}_aux{s}_P"]) < TOL
                assert rel_err(aux_o[s][1], arrs[f"c{ci}_aux{s}_Delta"]) < TOL
            else:
                got = np.concatenate(aux_o[s]) if mode == 1 else aux_o[s]
                assert rel_err(got, arrs[f"c{ci}_aux{s}"]) < TOL, (name, ci, case)
        n_run += 1
    assert n_run >= 10
