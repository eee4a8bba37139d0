"""Reusable pytest kit for authors of ADMM penalties (the contract of the reference's `matcouply.testing`,
/root/reference/src/matcouply/testing/admm_penalty.py:17-561, fixtures.py:8-87): subclass one of the base classes, set
`PenaltyType` (+ `penalty_default_kwargs`) and implement the `get_invariant_*` / `get_non_invariant_*` hooks.
Register the fixtures with ``pytest_plugins = ["matcouply_amd.testing.fixtures"]``."""
from .admm_penalty import (  # noqa: F401
    BaseTestADMMPenalty,
    BaseTestFactorMatricesPenalty,
    BaseTestFactorMatrixPenalty,
    BaseTestRowVectorPenalty,
    MixinTestHardConstraint,
    assert_allclose,
)
