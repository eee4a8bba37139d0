"""Stand-in for `tensorly.parafac2_tensor` container (`from_Parafac2Tensor` in the reference's
`coupled_matrices.py`).  Oracle tooling only."""
import numpy as np


class Parafac2Tensor:
    def __init__(self, parafac2_tensor):
        weights, factors, projections = parafac2_tensor
        if weights is None:
            weights = np.ones(np.shape(factors[0])[1])
        self.weights = weights
        self.factors = list(factors)
        self.projections = list(projections)

    def __getitem__(self, i):
        return (self.weights, self.factors, self.projections)[i]

    def __iter__(self):
        yield self.weights
        yield self.factors
        yield self.projections

    def __len__(self):
        return 3
