"""`tensorly.metrics.factors.congruence_coefficient` (used by the reference's tests only):
best-permutation mean |cosine| between the columns of two factor matrices.  Oracle tooling only."""
import numpy as np
from scipy.optimize import linear_sum_assignment


def congruence_coefficient(matrix1, matrix2, absolute_value=True):
    m1 = matrix1 / np.linalg.norm(matrix1, axis=0, keepdims=True)
    m2 = matrix2 / np.linalg.norm(matrix2, axis=0, keepdims=True)
    congruence = m1.T @ m2
    if absolute_value:
        congruence = np.abs(congruence)
    row_ind, col_ind = linear_sum_assignment(-congruence)
    return congruence[row_ind, col_ind].mean(), list(col_ind)
