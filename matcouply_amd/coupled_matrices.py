"""Coupled-matrix-factorization container and dense converters.

Mirrors /root/reference/src/matcouply/coupled_matrices.py: `CoupledMatrixFactorization` (:8-240), `_validate_cmf`
(:243-362), `cmf_to_matrix` / `cmf_to_matrices` / `cmf_to_tensor` / `cmf_to_unfolded` / `cmf_to_vec` (:365-799).
Factors may be NumPy arrays or torch tensors (host or MI355X memory)."""
import numpy as np

from ._utils import is_tensor, is_torch, shape


class CoupledMatrixFactorization:
    """(weights, (A, [B_0, ..., B_{I-1}], C)); matrix i is B_i diag(weights * a_i) C^T.  Behaves like a 2-tuple."""

    def __init__(self, cmf_matrices):
        self.shape, self.rank = _validate_cmf(cmf_matrices)
        weights, factors = cmf_matrices
        self.weights = weights
        self.factors = factors

    @classmethod
    def from_CPTensor(cls, cp_tensor, shapes=None):
        """(weights, (A, B, C)) of a third-order CP tensor -> coupled matrix factorization with B_i = B, or the first J_i
        rows of B when `shapes` = [(J_i, K), ...] is given (coupled_matrices.py:101-151)."""
        weights, factors = cp_tensor
        if len(factors) != 3:
            raise ValueError("Must be a third order CP tensor to convert into a coupled matrix factorization")
        A, B, C = factors
        copy = lambda x: None if x is None else (x.clone() if hasattr(x, "clone") else np.array(x))
        if shapes is not None:
            if len(shapes) != A.shape[0]:
                raise ValueError(f"The first mode has length {A.shape[0]}, which is different "
                                 f"than the length indicated by the shapes argument ({len(shapes)})")
            B_is = []
            for J_i, K in shapes:
                if K != C.shape[0]:
                    raise ValueError(f"The third mode has length {C.shape[0]}, which is different "
                                     f"than the length indicated by the shapes argument ({K})")
                if J_i > B.shape[0]:
                    raise ValueError(f"The second mode of the CP tensor mode has length {B.shape[0]}, which "
                                     f"is smaller than the length indicated by the shape ({J_i}) of matrix")
                B_is.append(copy(B)[:J_i, :])
        else:
            B_is = [copy(B) for _ in range(A.shape[0])]
        return cls((copy(weights), [copy(A), B_is, copy(C)]))

    @classmethod
    def from_Parafac2Tensor(cls, parafac2_tensor):
        """(weights, (A, B, C), projections) of a PARAFAC2 tensor -> coupled matrix factorization with B_i = P_i B
        (coupled_matrices.py:153-172)."""
        weights, factors, projection_matrices = parafac2_tensor
        A, B, C = factors
        copy = lambda x: None if x is None else (x.clone() if hasattr(x, "clone") else np.array(x))
        return cls((copy(weights), [copy(A), [P_i @ B for P_i in projection_matrices], copy(C)]))

    def __getitem__(self, item):
        if item == 0:
            return self.weights
        elif item == 1:
            return self.factors
        raise IndexError(
            "You tried to access index {} of a coupled matrix factorization.\n"
            "You can only access index 0 and 1 of a coupled matrix factorization"
            "(corresponding respectively to the weights and factors)".format(item)
        )

    def __iter__(self):
        yield self.weights
        yield self.factors

    def __len__(self):
        return 2

    def __repr__(self):
        return "(weights, factors) : rank-{} CoupledMatrixFactorization of shape {}".format(self.rank, self.shape)

    def to_tensor(self):
        return cmf_to_tensor(self)

    def to_vec(self, pad=True):
        return cmf_to_vec(self, pad=pad)

    def to_unfolded(self, mode, pad=True):
        return cmf_to_unfolded(self, mode, pad=pad)

    def to_matrices(self):
        return cmf_to_matrices(self)

    def to_matrix(self, matrix_idx):
        return cmf_to_matrix(self, matrix_idx)


def _validate_cmf(cmf):
    """Returns (shapes, rank) or raises TypeError / ValueError with the reference's conditions (:296-362)."""
    if isinstance(cmf, CoupledMatrixFactorization):
        return cmf.shape, cmf.rank
    weights, (A, B_is, C) = cmf
    if not (is_tensor(weights) or weights is None):
        raise TypeError("Weights should be a first order tensor of length rank, not {}".format(type(weights)))
    elif weights is not None and len(shape(weights)) != 1:
        raise ValueError("Weights should be a first order tensor. However weights has shape {}".format(shape(weights)))
    if not is_tensor(A):
        raise TypeError(
            "The first factor matrix, A, should be a second order tensor of size (I, rank)), not {}".format(type(A)))
    elif len(shape(A)) != 2:
        raise ValueError(
            "The first factor matrix, A, should be a second order tensor. However A has shape {}".format(shape(A)))
    if not is_tensor(C):
        raise TypeError(
            "The last factor matrix, C, should be a second order tensor of size (K, rank)), not {}".format(type(C)))
    elif len(shape(C)) != 2:
        raise ValueError(
            "The last factor matrix, C, should be a second order tensor. However C has shape {}".format(shape(C)))
    rank = int(shape(A)[1])
    if shape(C)[1] != rank:
        raise ValueError(
            "All the factors of a coupled matrix factorization should have the same number of columns."
            "However, A.shape[1]={} but C.shape[1]={}.".format(rank, shape(C)[1]))
    shapes = []
    for i, B_i in enumerate(B_is):
        if not is_tensor(B_i):
            raise TypeError(
                "The B_is[{}] factor matrix should be second order tensor of size (J_i, rank)), not {}".format(i, type(B_i)))
        elif len(shape(B_i)) != 2:
            raise ValueError(
                "The B_is[{}] factor matrix should be second order tensor. However B_is[{}] has shape {}".format(
                    i, i, shape(B_i)))
        if shape(B_i)[1] != rank:
            raise ValueError(
                "All the factors of a coupled matrix factorization should have the same number of columns."
                "However, A.shape[1]={} but B_is[{}].shape[1]={}.".format(rank, i, shape(B_i)[1]))
        shapes.append((shape(B_i)[0], shape(C)[0]))
    if weights is not None and shape(weights)[0] != rank:
        raise ValueError(
            "Given factors for a rank-{} coupled matrix factorization but len(weights)={}.".format(rank, shape(weights)[0]))
    if shape(A)[0] != len(B_is):
        raise ValueError(
            "The number of rows in A should be the same as the number of B_i matrices"
            "However, tl.shape(A)[0]={}, but len(B_is)={}".format(shape(A)[0], len(B_is)))
    return tuple(shapes), rank


def _on_device(A, B_is, C):
    """factors the native reconstruction kernel can take as they are: float32 torch tensors on a HIP device, outside any
    autograd graph, rank within the kernel's limit.  Everything else (float64 factors, factors that require grad, larger
    ranks) goes through the generic array expression below, which keeps dtype and gradients like the reference's"""
    from . import _engine

    ts = [A, C] + list(B_is)
    if not all(is_torch(t) and t.is_cuda for t in ts):
        return False
    import torch

    return (all(t.dtype == torch.float32 and not t.requires_grad for t in ts)
            and 1 <= int(C.shape[1]) <= _engine.MCL_MAX_RANK)


def _device_matrices(weights, A, B_is, C):
    """all dense matrices of device-resident factors in ONE native launch (csrc/reconstruct.hip), as row views of the
    packed result; float32 like everything the engine computes"""
    import torch
    from . import _engine

    f32 = lambda t: t.detach().to(torch.float32).contiguous()
    row_ptr = np.concatenate([[0], np.cumsum([int(B_i.shape[0]) for B_i in B_is])]).astype(np.int64)
    B = torch.cat([f32(B_i) for B_i in B_is], 0) if len(B_is) else f32(A).new_zeros((0, int(C.shape[1])))
    packed = _engine.cmf_to_packed(f32(A), B, f32(C), row_ptr, weights=None if weights is None else f32(weights))
    return [packed[row_ptr[i]:row_ptr[i + 1]] for i in range(len(B_is))]


def cmf_to_matrix(cmf, matrix_idx, validate=True):
    """Dense matrix i: (B_i * a_i) C^T (:365-423)."""
    if validate:
        cmf = CoupledMatrixFactorization(cmf)
    weights, (A, B_is, C) = cmf
    if _on_device(A, B_is, C):
        i = range(len(B_is))[matrix_idx]
        return _device_matrices(weights, A[i:i + 1], [B_is[i]], C)[0]
    a = A[matrix_idx]
    if weights is not None:
        a = a * weights
    B_i = B_is[matrix_idx]
    Ct = C.T if not is_torch(C) else C.t()
    return (B_i * a) @ Ct


def cmf_to_slice(cmf, slice_idx, validate=True):
    return cmf_to_matrix(cmf, slice_idx, validate=validate)


def cmf_to_matrices(cmf, validate=True):
    """List of all dense matrices (:446-497)."""
    if validate:
        cmf = CoupledMatrixFactorization(cmf)
    weights, (A, B_is, C) = cmf
    if _on_device(A, B_is, C):
        return _device_matrices(weights, A, B_is, C)
    if weights is not None:
        A = A * weights
        weights = None
    decomposition = weights, (A, B_is, C)
    return [cmf_to_matrix(decomposition, i, validate=False) for i in range(len(B_is))]


def cmf_to_slices(cmf, validate=True):
    return cmf_to_matrices(cmf, validate=validate)


def cmf_to_tensor(cmf, validate=True):
    """Zero-padded third-order tensor of the matrices (:520-600)."""
    if validate:
        cmf = CoupledMatrixFactorization(cmf)
    _, (A, B_is, C) = cmf
    matrices = cmf_to_matrices(cmf, validate=False)
    lengths = [B_i.shape[0] for B_i in B_is]
    if is_torch(C):
        tensor = matrices[0].new_zeros((A.shape[0], max(lengths), C.shape[0]))
    else:
        tensor = np.zeros((A.shape[0], max(lengths), C.shape[0]), dtype=matrices[0].dtype)
    for i, (matrix_, length) in enumerate(zip(matrices, lengths)):
        tensor[i, :length] = matrix_
    return tensor


def cmf_to_unfolded(cmf, mode, pad=True, validate=True):
    """:603-720"""
    if pad:
        t = cmf_to_tensor(cmf, validate=validate)
        moved = t.movedim(mode, 0) if is_torch(t) else np.moveaxis(t, mode, 0)
        return moved.reshape(t.shape[mode], -1)
    if mode == 2:
        mats = cmf_to_matrices(cmf, validate=validate)
        cat = __import__("torch").cat(mats, 0).t() if is_torch(mats[0]) else np.concatenate(mats, axis=0).T
        return cat
    raise ValueError(f"Cannot unfold along mode {mode} without padding. ")


def cmf_to_vec(cmf, pad=True, validate=True):
    """:723-799"""
    if pad:
        return cmf_to_tensor(cmf, validate=validate).reshape(-1)
    mats = cmf_to_matrices(cmf, validate=validate)
    if is_torch(mats[0]):
        return __import__("torch").cat([m.reshape(-1) for m in mats])
    return np.concatenate([m.reshape(-1) for m in mats])
