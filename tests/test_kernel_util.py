"""``subgrid_kernel`` (lenstronomy ``Util.kernel_util``, third party, restated -- parity unpinned; the reference calls it at
tf/simulator.py:60-70): the product's matrix form (gigalens_amd/kernel_util.py) against the oracle's step-by-step
restatement (oracle/ref_torch.py), and the properties the algorithm guarantees."""
import os

import numpy as np
import pytest

from gigalens_amd import kernel_util as ku
from oracle import ref_torch as ref

PSF = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_assets", "psf.npy"))


def _gauss(n, sigma, skew=0.0):
    ax = np.arange(n) - (n - 1) / 2
    k = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / (2 * sigma ** 2)) * (1 + skew * ax[:, None] / n)
    return k / k.sum()


@pytest.mark.parametrize("kernel", [PSF, _gauss(7, 1.2), _gauss(9, 2.0, skew=0.3)], ids=["reference_psf", "gauss7", "skewed9"])
@pytest.mark.parametrize("ss", [2, 3, 4, 5])
def test_product_matches_the_oracle_restatement(kernel, ss):
    a = ku.subgrid_kernel(kernel, ss, odd=True)
    b = ref.subgrid_kernel(kernel, ss, odd=True)
    n = kernel.shape[0] * ss
    assert a.shape == b.shape == (n - 1 + n % 2,) * 2  # odd=True: an even fine size loses one row / column
    assert np.abs(a - b).max() <= 1e-13
    assert abs(a.sum() - 1) < 1e-12


def test_identity_and_iteration_fixed_point():
    assert ku.subgrid_kernel(PSF, 1, odd=True) is not None and np.array_equal(ku.subgrid_kernel(PSF, 1, odd=True), PSF)
    assert np.array_equal(ref.subgrid_kernel(PSF, 1, odd=True), PSF)
    # the iteration drives "re-bin the fine kernel" to the input kernel (the purpose of the routine)
    fine = ku.subgrid_kernel(PSF / PSF.sum(), 2, odd=True, num_iter=100)
    back = ref._averaging_even_kernel(fine, 2)
    once = ref._averaging_even_kernel(ku.subgrid_kernel(PSF / PSF.sum(), 2, odd=True, num_iter=1), 2)
    assert np.abs(back - PSF / PSF.sum()).max() < 0.2 * np.abs(once - PSF / PSF.sum()).max()
    assert np.abs(back - PSF / PSF.sum()).max() < 2e-3 * PSF.max() / PSF.sum()
    with pytest.warns(RuntimeWarning, match="parity unpinned"):  # odd subgrid_res: restated without a lenstronomy fixture
        fine3 = ku.subgrid_kernel(PSF / PSF.sum(), 3, odd=True)
    back3 = fine3.reshape(13, 3, 13, 3).sum(3).sum(1)
    # the odd branch ends by returning the unmatched residual block-wise: the re-binned kernel IS the input kernel
    k64 = (PSF / PSF.sum()).astype(np.float64)
    k64 /= k64.sum()
    assert np.abs(back3 - k64).max() < 1e-13


def test_symmetry_is_kept():
    k = _gauss(9, 1.7)
    f = ku.subgrid_kernel(k, 2, odd=True)
    assert np.allclose(f, f[::-1, :], atol=1e-15) and np.allclose(f, f.T, atol=1e-15)
    assert np.unravel_index(np.argmax(f), f.shape) == (8, 8)


def test_simulator_config_with_psf_and_supersampling_constructs():
    """``SimulatorConfig(kernel=psf, supersample=2)`` -- the reference's own demo configuration (tf-demo.ipynb cell 6) -- used to
    raise; the host side now derives the 25 x 25 kernel (no GPU needed up to the native model, which is not built here)."""
    from gigalens_amd.kernel_util import subgrid_kernel
    k = subgrid_kernel(PSF.astype(np.float32), 2, odd=True)
    assert k.shape == (25, 25) and k.dtype == np.float64
