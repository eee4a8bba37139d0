"""Small host-side helpers (mirrors /root/reference/src/matcouply/_utils.py:6-54 where noted)."""
import numpy as np

try:  # torch is plumbing (device memory, streams, torch.distributed); the package still imports without it
    import torch
except ImportError:  # pragma: no cover
    torch = None


def is_iterable(x):
    """_utils.py:6-12"""
    try:
        iter(x)
    except TypeError:
        return False
    return True


def is_tensor(x):
    return isinstance(x, np.ndarray) or (torch is not None and isinstance(x, torch.Tensor))


def is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


def shape(x):
    return tuple(x.shape)


def check_random_state(seed):
    """None -> NumPy's global RandomState, int -> RandomState(int), RandomState -> itself (tensorly semantics)."""
    if seed is None:
        return np.random.mtrand._rand
    if isinstance(seed, (int, np.integer)):
        return np.random.RandomState(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError("Seed should be None, int or np.random.RandomState")


SVD_NAMES = ("truncated_svd", "numpy_svd", "randomized_svd", "symeig_svd")


def get_svd(svd):
    """_utils.py:15-26.  The engine solves the rank x rank systems with an in-kernel fp64 Gauss-Jordan and the
    PARAFAC2 polar factor with an fp64 Jacobi eigen-solver, so the name only has to be a valid one."""
    if svd in SVD_NAMES:
        def thin_svd(matrix, n_eigenvecs=None, **kwargs):
            U, s, Vh = np.linalg.svd(np.asarray(matrix), full_matrices=False)
            if n_eigenvecs is not None:
                U, s, Vh = U[:, :n_eigenvecs], s[:n_eigenvecs], Vh[:n_eigenvecs]
            return U, s, Vh

        return thin_svd
    raise ValueError(f"Got svd={svd}. However, the possible choices are {list(SVD_NAMES)}")


def get_shapes(matrices):
    """_utils.py:29-30"""
    return [shape(matrix) for matrix in matrices]


def get_padded_tensor_shape(matrices):
    """(I, max J_i, K) of the zero-padded stack; all matrices must share K (_utils.py:33-46)"""
    K = shape(matrices[0])[1]
    J = -float("inf")
    for matrix in matrices:
        J_i, K_i = shape(matrix)
        if K_i != K:
            raise ValueError("All matrices must have the same number of columns")
        J = max(J_i, J)
    return len(matrices), J, K


def create_padded_tensor(matrices):
    """third-order tensor whose frontal slabs are the matrices, zero-padded to the longest one (_utils.py:49-54)"""
    I, J, K = get_padded_tensor_shape(matrices)
    first = matrices[0]
    if is_torch(first):
        import torch

        out = torch.zeros((I, J, K), dtype=first.dtype, device=first.device)
    else:
        out = np.zeros((I, J, K), dtype=np.asarray(first).dtype)
    for i, matrix in enumerate(matrices):
        out[i, : shape(matrix)[0]] = matrix if is_torch(first) else np.asarray(matrix)
    return out


def to_numpy(x):
    if is_torch(x):
        return x.detach().cpu().numpy()
    return np.asarray(x)
