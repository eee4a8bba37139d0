import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")
# the only place that allows a substitute (checker) engine behind cmf_aoadmm: see matcouply_amd/decomposition.py
os.environ["MATCOUPLY_AMD_TEST_ENGINE"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# Small problems (at most 2^20 elements of X) run in the library's exact-products mode (csrc/contract.hip: mcl_exact_mode) -
# and the parity tests run small problems.  Tests whose SUBJECT is a fast kernel (the one-pass sweep, the MFMA contractions,
# the deferral across a sweep) force the size-independent kernels with MCL_EXACT=0; trajectory / phase tests run BOTH ways.
def _with_env(name, value):
    old = os.environ.get(name)
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = value
    return old


@pytest.fixture
def fast_kernels():
    """the kernels of the BASELINE configurations, whatever the problem size (contexts read MCL_EXACT when they are created)"""
    old = _with_env("MCL_EXACT", "0")
    yield
    _with_env("MCL_EXACT", old)


@pytest.fixture(params=["default", "fast-kernels"])
def kernel_paths(request):
    """both arithmetic paths of a small problem: the default (exact products) and the fast kernels large problems take"""
    old = _with_env("MCL_EXACT", "0" if request.param == "fast-kernels" else None)
    yield request.param
    _with_env("MCL_EXACT", old)
