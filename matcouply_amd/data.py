"""Simulated dataset of BASELINE config 1 (mirrors /root/reference/src/matcouply/data.py:28-95).
The downloadable datasets of the reference (bike sharing, semiconductor etch) are I/O and out of scope."""
import numpy as np
import scipy.stats as stats

from ._utils import check_random_state
from .coupled_matrices import CoupledMatrixFactorization


def get_simple_simulated_data(noise_level=0.2, random_state=1):
    """I=15 matrices of 50 x 20, rank 3: A ~ U[0.1, 1.1), B_i = three Gaussian pdfs rolled by i, C = max(N(0,1), 0),
    noise scaled to `noise_level * ||M|| / ||N||`.  Same RandomState draw order as the reference."""
    rank = 3
    I, J, K = 15, 50, 20
    rng = check_random_state(random_state)
    A = rng.uniform(size=(I, rank)) + 0.1
    t = np.linspace(-10, 10, J)
    B_blueprint = np.stack([stats.norm.pdf(t, loc=-5), stats.norm.pdf(t, loc=0), stats.norm.pdf(t, loc=2)], axis=-1)
    B_is = [np.roll(B_blueprint, i, axis=0) for i in range(I)]
    C = rng.standard_normal(size=(K, rank))
    C[C < 0] = 0
    cmf = CoupledMatrixFactorization((None, (A, B_is, C)))
    matrices = cmf.to_matrices()
    noise = [rng.standard_normal(size=M.shape) for M in matrices]
    scale_factor = np.linalg.norm(np.stack(matrices)) / np.linalg.norm(np.stack(noise))
    matrices = [M + noise_level * scale_factor * N for M, N in zip(matrices, noise)]
    return matrices, cmf


def _needs_download(name, where):
    raise NotImplementedError(
        f"matcouply_amd.data.{name}: this dataset of the reference ({where}) is fetched / unpacked from files that are not "
        "part of this build (no network, no bundled archives).  Load the data yourself and pass the list of matrices to "
        "matcouply_amd.decomposition.cmf_aoadmm / parafac2_aoadmm.")


def get_bike_data():
    """Bike-sharing data of three Norwegian cities (data.py:98-150 of the reference): not available in this build."""
    _needs_download("get_bike_data", "data.py:98-150")


def get_semiconductor_etch_raw_data(download_data=True, save_data=True):
    """Semiconductor etch raw data (data.py:153-201 of the reference, downloaded from eigenvector.com): not available."""
    _needs_download("get_semiconductor_etch_raw_data", "data.py:153-201")


def get_semiconductor_etch_machine_data(download_data=True, save_data=True):
    """Semiconductor etch machine data (data.py:204-260 of the reference): not available in this build."""
    _needs_download("get_semiconductor_etch_machine_data", "data.py:204-260")
