"""MI355X-native AO-ADMM engine for coupled matrix factorization / PARAFAC2 with the API of MatCoupLy.

`matcouply_amd.decomposition.cmf_aoadmm` / `parafac2_aoadmm` and the `matcouply_amd.penalties` class tree mirror
`matcouply.decomposition` / `matcouply.penalties` (MarieRoald/matcouply v0.1.6); the per-mode ADMM updates run in
hand-written HIP kernels (libmatcouply_hip.so, C ABI in include/matcouply_hip.h).  There is no CPU fallback.
"""
__version__ = "0.1.0"

from . import coupled_matrices, data, decomposition, penalties, random  # noqa: F401,E402
from .decomposition import PackedMatrices  # noqa: F401,E402
