"""Random coupled matrix factorizations (mirrors /root/reference/src/matcouply/random.py:9-66)."""
import numpy as np

from ._utils import check_random_state
from .coupled_matrices import CoupledMatrixFactorization


def random_coupled_matrices(shapes, rank, full=False, random_state=None, normalise_factors=True, normalise_B=False,
                            **context):
    """Uniform [0,1) factors in the draw order A, B_0..B_{I-1}, C (random.py:40-43), column-normalised.

    `context`: dtype (NumPy, default float64) as in the reference; additionally `device="cuda"` returns the factors as
    float32 torch tensors on that HIP device (drawn with the same host generator, so the values equal the host result
    rounded to float32), and `full=True` then builds the dense matrices with the native reconstruction kernel."""
    device = context.pop("device", None)
    rns = check_random_state(random_state)
    if not all(shape[1] == shapes[0][1] for shape in shapes):
        raise ValueError("All matrices must have equal number of columns.")
    dtype = context.get("dtype", np.float64)
    A = np.asarray(rns.random_sample((len(shapes), rank)), dtype=dtype)
    B_is = [np.asarray(rns.random_sample((j_i, rank)), dtype=dtype) for j_i, k in shapes]
    K = shapes[0][1]
    C = np.asarray(rns.random_sample((K, rank)), dtype=dtype)
    weights = np.ones(rank, dtype=dtype)
    if normalise_factors or normalise_B:
        B_i_norms = [np.sqrt(np.sum(B_i ** 2, axis=0)) for B_i in B_is]
        B_is = [B_i / B_i_norm for B_i, B_i_norm in zip(B_is, B_i_norms)]
        A = A * np.stack(B_i_norms)
    if normalise_factors:
        A_norm = np.sqrt(np.sum(A ** 2, axis=0))
        A = A / A_norm
        C_norm = np.sqrt(np.sum(C ** 2, axis=0))
        C = C / C_norm
        weights = A_norm * C_norm
    if device is not None:
        import torch

        dev = torch.device(device)
        to = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)
        weights, A, B_is, C = to(weights), to(A), [to(B_i) for B_i in B_is], to(C)
    cmf = CoupledMatrixFactorization((weights, (A, B_is, C)))
    return cmf.to_matrices() if full else cmf
