"""`tensorly.testing` names used by the reference's own tests.  Oracle tooling only."""
import numpy as np

assert_array_equal = np.testing.assert_array_equal
assert_array_almost_equal = np.testing.assert_array_almost_equal
assert_allclose = np.testing.assert_allclose
assert_ = np.testing.assert_
