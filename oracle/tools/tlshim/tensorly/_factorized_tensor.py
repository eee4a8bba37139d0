"""Stand-in for `tensorly._factorized_tensor` (base class of the reference's CMF container,
`/root/reference/src/matcouply/coupled_matrices.py:5,8`).  Oracle tooling only."""


class FactorizedTensor:
    def to_tensor(self):
        raise NotImplementedError

    def to_unfolded(self, mode):
        raise NotImplementedError

    def to_vec(self):
        raise NotImplementedError
