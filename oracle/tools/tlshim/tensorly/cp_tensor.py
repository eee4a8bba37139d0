"""Stand-in for `tensorly.cp_tensor` - only the tuple-like container the reference converts from
(`/root/reference/src/matcouply/coupled_matrices.py`, `from_CPTensor`).  Oracle tooling only."""
import numpy as np


class CPTensor:
    def __init__(self, cp_tensor):
        weights, factors = cp_tensor
        if weights is None:
            weights = np.ones(np.shape(factors[0])[1])
        self.weights = weights
        self.factors = list(factors)

    def __getitem__(self, i):
        return (self.weights, self.factors)[i]

    def __iter__(self):
        yield self.weights
        yield self.factors

    def __len__(self):
        return 2


def cp_to_tensor(cp_tensor):
    weights, factors = CPTensor(cp_tensor)
    out = np.einsum("r,ir,jr,kr->ijk", weights, *factors)
    return out
