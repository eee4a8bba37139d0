"""`tl.tenalg.svd` is only consulted when `tl.SVD_FUNS` is not a dict (new TensorLy layout,
`/root/reference/src/matcouply/_utils.py:15-20`).  The stand-in follows the old dict layout."""


class _SVDNamespace:
    pass


svd = _SVDNamespace()
