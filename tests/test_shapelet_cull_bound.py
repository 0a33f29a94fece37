"""The table-mode shapelet kernels skip the lens on wave-tiles whose pixels are PROVABLY outside the shapelet table
(csrc/gl_shp.hip.h shp_cull_setup): |beta - c| >= |x - c| - |alpha(x)| with a bound on |alpha| that needs no series.  The skip is
exact only if the bound never fails: here the bound, restated, is checked against the oracle's float64 deflections
(oracle/ref_torch.py: epl.py:19-57, sie.py:13-42, sis.py, shear.py) over random lenses far beyond the priors of the BASELINE
configs -- every pixel the test would cull must map outside |u|, |v| <= 5 (shapelets.py:55-62), and the per-lens bounds themselves
must hold where they are claimed."""
import math

import numpy as np
import pytest
import torch

from oracle import ref_torch as ref

F64 = torch.float64


def _grid(n=72, half=4.7):
    ax = torch.linspace(-half, half, n, dtype=F64)
    Y, X = torch.meshgrid(ax, ax, indexing="ij")
    return X.reshape(-1), Y.reshape(-1), float(torch.sqrt(X ** 2 + Y ** 2).max())


def _epl_consts(theta_E, gamma, e1, e2):
    c = min(math.hypot(e1, e2), 1.0)
    q = (1 - c) / (1 + c)
    b = theta_E / math.sqrt((1 + q * q) / (2 * q)) * math.sqrt((1 + q * q) / 2)
    return q, b, gamma - 1.0  # t = gamma - 1


@pytest.mark.parametrize("seed", range(6))
def test_every_culled_pixel_maps_outside_the_table(seed):
    r = np.random.default_rng(seed)
    x, y, r_max = _grid()
    n_culled = 0
    for _ in range(60):
        theta_E = r.uniform(0.4, 2.0)
        gamma = r.uniform(1.05, 2.95)
        e = r.uniform(0.0, 0.6)
        ang = r.uniform(0, 2 * math.pi)
        e1, e2 = e * math.cos(ang), e * math.sin(ang)
        lx, ly = r.normal(0, 0.3, 2)
        g1, g2 = r.normal(0, 0.08, 2)
        beta_s = r.uniform(0.03, 0.25)
        cx, cy = r.normal(0, 0.3, 2)
        ax_, ay_ = ref.epl_deriv(x, y, theta_E, gamma, e1, e2, lx, ly)
        sx, sy = ref.shear_deriv(x, y, g1, g2)
        bx, by = x - ax_ - sx, y - ay_ - sy
        # ---- the kernel's bound (gl_shp.hip.h shp_cull_setup), in float64 ----
        q, b, t = _epl_consts(theta_E, gamma, e1, e2)
        tm1 = t - 1.0
        assert 0 < q <= 1 and -1 < tm1 < 1
        bq = b / q
        D = r_max + math.hypot(lx, ly)
        grow = max(D / b, 1.0) ** (-tm1) if tm1 < 0 else 1.0
        A_epl = bq * grow
        A = A_epl + math.hypot(g1, g2) * r_max
        T = 1.001 * (7.0710678 * beta_s + A)
        rb2 = 1.002 * bq * bq if tm1 > 0 else 0.0
        d2 = (x - cx) ** 2 + (y - cy) ** 2
        e2l = (x - lx) ** 2 + (y - ly) ** 2
        far = (d2 > T * T) & (e2l >= rb2)
        # the per-lens bound where it is claimed
        a_mag = torch.sqrt(ax_ ** 2 + ay_ ** 2)
        claimed = e2l >= rb2
        assert bool((a_mag[claimed] <= A_epl * (1 + 1e-9)).all()), (theta_E, gamma, e1, e2, float(a_mag[claimed].max()), A_epl)
        if bool(far.any()):
            n_culled += int(far.sum())
            u = (bx[far] - cx) / beta_s
            v = (by[far] - cy) / beta_s
            assert bool((torch.maximum(u.abs(), v.abs()) > 5.0).all())
    assert n_culled > 1000  # the test does cull: the property is not vacuous


def test_sis_and_sie_bounds():
    r = np.random.default_rng(3)
    x, y, _ = _grid(48, 3.0)
    for _ in range(40):
        theta_E = r.uniform(0.3, 2.0)
        lx, ly = r.normal(0, 0.3, 2)
        ax_, ay_ = ref.sis_deriv(x, y, theta_E, lx, ly)
        assert bool((torch.sqrt(ax_ ** 2 + ay_ ** 2) <= theta_E * (1 + 1e-12)).all())
        e = r.uniform(0.02, 0.6)
        ang = r.uniform(0, 2 * math.pi)
        e1, e2 = e * math.cos(ang), e * math.sin(ang)
        ax_, ay_ = ref.sie_deriv(x, y, theta_E, e1, e2, lx, ly)
        c = min(math.hypot(e1, e2), 0.9999)
        q = (1 - c) / (1 + c)
        sq = math.sqrt(1 - q * q)
        bsie = theta_E / math.sqrt((1 + q * q) / (2 * q)) * math.sqrt((1 + q * q) / 2)
        A = bsie / sq  # SIE_A of the kernel's derived block (sie.py:28-41)
        bound = abs(A) * math.hypot(math.pi / 2, math.atanh(sq))
        mag = torch.sqrt(ax_ ** 2 + ay_ ** 2)
        ok = torch.isfinite(mag)
        assert bool((mag[ok] <= bound * (1 + 1e-9)).all()), (theta_E, e1, e2, float(mag[ok].max()), bound)
