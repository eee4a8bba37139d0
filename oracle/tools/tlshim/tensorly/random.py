"""`tensorly.random` names used by the reference's docstrings/tests.  Oracle tooling only."""
import numpy as np


def random_tensor(shape, random_state=None, **context):
    if random_state is None:
        rng = np.random.mtrand._rand
    elif isinstance(random_state, (int, np.integer)):
        rng = np.random.RandomState(random_state)
    else:
        rng = random_state
    return np.array(rng.random_sample(shape), **context)
