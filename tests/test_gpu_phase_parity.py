"""-m gpu: one call of each phase through the C ABI vs (a) the golden outputs of the reference and (b) the oracle.

Tolerance: 1e-5 relative (Frobenius) - the bar BASELINE.json's north_star states for the fp32 engine vs the
fp64 NumPy reference."""
import numpy as np
import pytest

from tests.helpers import engine_from_oracle_state, load_npz, manifest_of, rel_err, to_np
from tests.test_oracle_golden import _phase_state

pytestmark = pytest.mark.gpu
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
