"""Supersampled PSF: restatement of lenstronomy's ``Util.kernel_util.subgrid_kernel`` (third party; the reference calls it at
src/gigalens/tf/simulator.py:60-70 with ``odd=True``; README.rst:35 pins lenstronomy 1.9.3, the demo notebook ran 1.9.1).

lenstronomy is not in this image and not under /root/reference, so this is the PUBLISHED algorithm restated, **parity
unpinned**: bilinear interpolation of the kernel onto the finer grid (cell centres on the unit interval, nearest-edge outside
the input centres, as ``scipy.interpolate.interp2d(kind='linear')`` evaluates), normalised to unit sum, then ``num_iter``
rounds of: re-bin the fine kernel to the input pixel scale, add the mismatch to the working
low-resolution kernel, re-interpolate, re-normalise.  With ``odd=True`` an even fine size loses one row / column so the kernel keeps a centre pixel;
re-binning an odd fine kernel at an even ``subgrid_res`` shares the rows / columns that straddle two coarse pixels half and
half (``averaging_even_kernel``).

Written as matrix products (the interpolation and both re-binnings are linear maps applied to rows and columns); the
oracle restates the same routine with explicit loops (oracle/ref_torch.py), and the two are cross-checked in
tests/test_kernel_util.py together with the properties the algorithm guarantees (unit sum, symmetry, re-binning back to the
input kernel).
"""
import warnings

import numpy as np


def _centres(n):
    return (np.arange(n, dtype=np.float64) + 0.5) / n


def _interp_matrix(n_in, n_out):
    """[n_out, n_in] linear interpolation from cell centres of n_in cells to those of n_out cells on [0, 1]."""
    x_in, x_out = _centres(n_in), _centres(n_out)
    W = np.zeros((n_out, n_in))
    for o, x in enumerate(x_out):
        if x <= x_in[0]:
            W[o, 0] = 1.0
        elif x >= x_in[-1]:
            W[o, -1] = 1.0
        else:
            i = int(np.searchsorted(x_in, x, side="right")) - 1
            t = (x - x_in[i]) / (x_in[i + 1] - x_in[i])
            W[o, i], W[o, i + 1] = 1.0 - t, t
    return W


def _rebin_matrix(n_high, subgrid_res):
    """[n_low, n_high] re-binning of a fine axis to the coarse pixel scale, as lenstronomy does it: plain block means for an
    odd ``subgrid_res`` (``util.averaging``: mean, not sum), and for an even one the centred sum of
    ``averaging_even_kernel`` in which every ``subgrid_res``-th fine cell straddles two coarse pixels and gives half to each."""
    if subgrid_res % 2 == 1:
        n_low = n_high // subgrid_res
        A = np.zeros((n_low, n_high))
        for i in range(n_low):
            A[i, i * subgrid_res:(i + 1) * subgrid_res] = 1.0 / subgrid_res
        return A
    n_low = int(round(n_high / subgrid_res + 0.5))
    if n_low % 2 == 0:
        n_low += 1
    n_full = n_low * subgrid_res - 1
    pad = (n_full - n_high) // 2
    A = np.zeros((n_low, n_high))
    for j_full in range(n_full):
        j = j_full - pad
        if not 0 <= j < n_high:
            continue
        blk, off = divmod(j_full, subgrid_res)
        if off < subgrid_res - 1:
            A[blk, j] += 1.0
        else:  # the cell on the border of coarse pixels blk and blk + 1
            A[blk, j] += 0.5
            A[blk + 1, j] += 0.5
    return A


def subgrid_kernel(kernel, subgrid_res, odd=False, num_iter=100):
    """lenstronomy ``kernel_util.subgrid_kernel(kernel, subgrid_res, odd, num_iter)`` restated; identity for ``subgrid_res == 1``."""
    subgrid_res = int(subgrid_res)
    kernel = np.asarray(kernel, dtype=np.float64)
    if subgrid_res == 1:
        return kernel
    nx, ny = kernel.shape
    nx_new, ny_new = nx * subgrid_res, ny * subgrid_res
    if odd:
        nx_new -= 1 - nx_new % 2
        ny_new -= 1 - ny_new % 2
    Wr, Wc = _interp_matrix(nx, nx_new), _interp_matrix(ny, ny_new)  # rows (first axis), columns
    even = subgrid_res % 2 == 0
    Ar, Ac = _rebin_matrix(nx_new, subgrid_res), _rebin_matrix(ny_new, subgrid_res)
    if Ar.shape[0] != nx or Ac.shape[0] != ny:
        raise ValueError(f"a {nx}x{ny} kernel at subgrid_res={subgrid_res}, odd={odd} does not re-bin to its own size "
                         "(lenstronomy has the same restriction: use an odd-sized kernel)")
    norm = lambda k: k / k.sum()
    work = kernel.copy()
    fine = norm(Wr @ work @ Wc.T)
    for _ in range(max(int(num_iter), 1)):
        # even: the re-binning is a sum (unit sum kept); odd: lenstronomy's block MEAN, which it does not re-normalise here
        work = work + (kernel - Ar @ fine @ Ac.T)
        fine = norm(Wr @ work @ Wc.T)
    if even:
        return fine
    # odd subgrid_res: what the iteration has not matched goes back at zeroth order, spread over each coarse pixel's block
    warnings.warn("subgrid_kernel at an odd subgrid_res follows lenstronomy 1.9.x as restated from the published source without "
                  "a lenstronomy-generated fixture: parity unpinned (even subgrid_res is pinned statistically by the "
                  "reference's demo image)", RuntimeWarning, stacklevel=2)
    delta = norm(Ar @ fine @ Ac.T) - norm(kernel)
    return norm(fine - np.kron(delta, np.ones((subgrid_res, subgrid_res))) / subgrid_res ** 2)
