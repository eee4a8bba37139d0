"""Helpers of the penalty test kit."""


def random_length(rng, min=2, mean=5):
    """A random dimension length for test shapes: `min` plus a Poisson draw with mean `mean - min`, so lengths are
    integers scattered around `mean` and never below `min` (contract of the reference's testing/utils.py:5-12)."""
    excess = mean - min
    if excess <= 0:
        raise ValueError("Min must be less than mean.")
    draw = rng.poisson(excess)
    return int(min + round(draw))
