"""pytest fixtures for penalty tests (reference testing/fixtures.py:8-87): seeded RandomState, ragged / regular shapes,
random coupled matrix factorizations and matrices."""
import numpy as np
import pytest

from matcouply_amd.random import random_coupled_matrices

from .utils import random_length


@pytest.fixture
def seed(pytestconfig):
    try:
        return pytestconfig.getoption("randomly_seed")
    except ValueError:
        return 1


@pytest.fixture
def rng(seed):
    return np.random.RandomState(seed=seed)


@pytest.fixture
def random_ragged_shapes(rng):
    I, K = random_length(rng), random_length(rng)
    return tuple((random_length(rng), K) for _ in range(I))


@pytest.fixture
def random_regular_shapes(rng):
    I, J, K = random_length(rng), random_length(rng), random_length(rng)
    return tuple((J, K) for _ in range(I))


@pytest.fixture
def random_ragged_cmf(rng, random_ragged_shapes):
    smallest = min(random_ragged_shapes[0][1], min(shape[0] for shape in random_ragged_shapes))
    rank = rng.randint(1, smallest + 1)
    cmf = random_coupled_matrices(random_ragged_shapes, rank, random_state=rng)
    return cmf, random_ragged_shapes, rank


@pytest.fixture
def random_rank5_ragged_cmf(rng):
    I, K = random_length(rng), random_length(rng, min=5, mean=7)
    shapes = tuple((random_length(rng, min=5, mean=7), K) for _ in range(I))
    return random_coupled_matrices(shapes, 5, random_state=rng), shapes, 5


@pytest.fixture
def random_regular_cmf(rng, random_regular_shapes):
    rank = rng.randint(1, min(random_regular_shapes[0]) + 1)
    return random_coupled_matrices(random_regular_shapes, rank, random_state=rng), random_regular_shapes, rank


@pytest.fixture
def random_matrix(rng):
    return rng.random_sample((random_length(rng), random_length(rng)))


@pytest.fixture
def random_matrices(rng):
    shape = (random_length(rng), random_length(rng))
    return [rng.random_sample(shape) for _ in range(random_length(rng))]
