def random_length(rng, min=2, mean=5):
    """dimension length `min + Poisson(mean - min)` (reference testing/utils.py:5-12)"""
    if min >= mean:
        raise ValueError("Min must be less than mean.")
    return min + round(rng.poisson(mean - min))
