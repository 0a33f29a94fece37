"""Pins the oracle (oracle/ref_torch.py) BEFORE it is trusted as the checker.

The reference's own tests compare against lenstronomy at run time (tests/test_profiles.py); lenstronomy
and TensorFlow are not installed here, so the same recipes (10 000 N(0,1) float32 points, the listed
parameter sets, rtol 1e-5 / atol 1e-4) are re-targeted at independent restatements of the published
closed forms (oracle/published.py) plus the one executable known-answer test the reference holds.
"""
import math

import numpy as np
import pytest
import torch

from oracle import published as pub
from oracle import ref_torch as ref

F64 = torch.float64


def _pts(n, seed=0):
    r = np.random.default_rng(seed)
    return r.normal(size=n).astype(np.float32), r.normal(size=n).astype(np.float32)


def _t(a):
    return torch.as_tensor(np.asarray(a, dtype=np.float64))


def test_sersic_known_answer():
    """tests/test_profiles.py:17-26: light(0,1; R=1,n=2,e=0,Ie=5) == 5."""
    a = pub.SERSIC_KAT["args"]
    out = ref.sersic_light(_t([a["x"]]), _t([a["y"]]), a["R_sersic"], a["n_sersic"], a["center_x"], a["center_y"],
                           a["Ie"], a["e1"], a["e2"])
    assert math.isclose(float(out[0]), pub.SERSIC_KAT["expected"])


@pytest.mark.parametrize("theta_E,gamma,e1,e2", [(1.0, 2.0, 0.0, 0.0), (1.2, 2.2, -0.1, 0.1), (0.9, 1.6, 0.25, 0.3)])
def test_epl_vs_published_2f1(theta_E, gamma, e1, e2):
    """tests/test_profiles.py:50-63 recipe; target = Tessore & Metcalf eq. 13 via scipy hyp2f1."""
    x, y = _pts(10000)
    fx, fy = ref.epl_deriv(_t(x), _t(y), theta_E, gamma, e1, e2, 0.0, 0.0, niter_cap=100)
    px, py = pub.epl_deriv_2f1(x, y, theta_E, gamma, e1, e2)
    assert np.allclose(fx.numpy(), px, rtol=1e-5, atol=1e-4)
    assert np.allclose(fy.numpy(), py, rtol=1e-5, atol=1e-4)
    # and much tighter than the reference's own tolerance when |e| is small (series converged)
    if math.hypot(e1, e2) < 0.2:
        assert np.allclose(fx.numpy(), px, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("theta_E,e1,e2", [(1.0, 1e-3, 1e-3), (1.2, 0.1, -0.1)])
def test_sie_vs_published(theta_E, e1, e2):
    """tests/test_profiles.py:82-95 recipe; target = EPL closed form at gamma = 2 (independent code path)."""
    x, y = _pts(10000, 1)
    fx, fy = ref.sie_deriv(_t(x), _t(y), theta_E, e1, e2, 0.0, 0.0)
    px, py = pub.epl_deriv_2f1(x, y, theta_E, 2.0, e1, e2)
    assert np.allclose(fx.numpy(), px, rtol=1e-5, atol=1e-4)
    assert np.allclose(fy.numpy(), py, rtol=1e-5, atol=1e-4)


def test_sie_equals_epl_gamma2_identity():
    x, y = _pts(2000, 2)
    a = ref.sie_deriv(_t(x), _t(y), 1.3, 0.2, -0.15, 0.03, -0.02)
    b = ref.epl_deriv(_t(x), _t(y), 1.3, 2.0, 0.2, -0.15, 0.03, -0.02, niter_cap=200)
    assert np.allclose(a[0].numpy(), b[0].numpy(), rtol=1e-9, atol=1e-11)
    assert np.allclose(a[1].numpy(), b[1].numpy(), rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("theta_E", [1.0, 1.2])
def test_sis_vs_published(theta_E):
    """tests/test_profiles.py:66-79."""
    x, y = _pts(10000, 3)
    fx, fy = ref.sis_deriv(_t(x), _t(y), theta_E, 0.0, 0.0)
    px, py = pub.sis_deriv(x, y, theta_E)
    assert np.allclose(fx.numpy(), px) and np.allclose(fy.numpy(), py)


@pytest.mark.parametrize("g1,g2", [(0.0, 0.0), (0.1, 0.1)])
def test_shear_vs_published(g1, g2):
    """tests/test_profiles.py:98-111."""
    x, y = _pts(10000, 4)
    fx, fy = ref.shear_deriv(_t(x), _t(y), g1, g2)
    px, py = pub.shear_deriv(x, y, g1, g2)
    assert np.allclose(fx.numpy(), px) and np.allclose(fy.numpy(), py)


def test_nfw_vs_mass_integral():
    """No NFW test exists in the reference; pin alpha_r(R) = (2/R) int kappa r dr with the published NFW
    convergence, and alpha_r(Rs) -> alpha_Rs from both sides (rho0 definition, nfw.py:17)."""
    Rs, aRs = 1.7, 0.9
    for R in [0.05, 0.4, 1.2, 1.69, 1.71, 3.0, 9.0]:
        fx, fy = ref.nfw_deriv(_t([R]), _t([0.0]), Rs, aRs, 0.0, 0.0)
        assert math.isclose(float(fx[0]), pub.nfw_alpha_r_numeric(R, Rs, aRs), rel_tol=2e-8)
        assert abs(float(fy[0])) < 1e-15
    for eps in (-1e-6, 1e-6):
        fx, _ = ref.nfw_deriv(_t([Rs * (1 + eps)]), _t([0.0]), Rs, aRs, 0.0, 0.0)
        assert math.isclose(float(fx[0]), aRs, rel_tol=1e-5)
    # the reference's quirk at exactly X == 1: g = 1.0, not 1 - ln 2 (nfw.py:38)
    fx, _ = ref.nfw_deriv(_t([Rs]), _t([0.0]), Rs, aRs, 0.0, 0.0)
    assert math.isclose(float(fx[0]), aRs / (1 - math.log(2)), rel_tol=1e-12)


@pytest.mark.parametrize("interpolate", [True, False])
def test_shapelets_vs_published(interpolate):
    """tests/test_profiles.py:35-47: n_max=5, beta=1, (5,5,1) points, both modes, rtol 1e-5 / atol 1e-4."""
    r = np.random.default_rng(5)
    n_max = 5
    n_layers = (n_max + 1) * (n_max + 2) // 2
    amps = r.normal(size=(n_layers,)).astype(np.float32)
    x, y = r.normal(size=25).astype(np.float32), r.normal(size=25).astype(np.float32)
    out = ref.shapelets_light(_t(x)[:, None], _t(y)[:, None], 0.0, 0.0, 1.0, [_t([a]) for a in amps], n_max, interpolate)
    want = pub.shapelet_set(x, y, amps.astype(np.float64), n_max, 1.0)
    assert np.allclose(out.numpy().ravel(), want, rtol=1e-5, atol=1e-4)


def test_shapelet_basis_orthonormal_and_tables():
    xs = np.linspace(-12, 12, 48001)
    for n in range(0, 11):
        for m in range(n, 11):
            v = np.trapz(ref.phi_n_f64(n, xs) * ref.phi_n_f64(m, xs), xs)
            assert abs(v - (1.0 if n == m else 0.0)) < 1e-9
        assert np.allclose(ref.phi_n_f64(n, xs[::97]), pub.shapelet_phi_n(n, xs[::97]), rtol=1e-10, atol=1e-14)
    tab = ref.shapelet_tables(10)
    assert tab.shape == (11, 6000) and tab.dtype == np.float32


def test_shapelet_index_order_and_names():
    N1, N2 = ref.shapelet_index_order(2)
    assert list(zip(N1, N2)) == [(0, 0), (1, 0), (0, 1), (2, 0), (1, 1), (0, 2)]
    names = ref.shapelet_amp_names(10)
    assert names[0] == "amp00" and names[-1] == "amp65" and names == sorted(names)


def test_grid_matches_closed_form():
    """simulator.py:47-55: x=(col-(Hs-1)/2) d/ss, y=(row-...) d/ss in f64 then cast to f32; row-major."""
    class Cfg:
        delta_pix, num_pix, supersample, kernel, transform_pix2angle, pix_region = 0.065, 8, 2, None, None, None
    wcs, region, img_region, X, Y = ref.build_grid(Cfg)
    Hs = 16
    rows, cols = np.divmod(np.arange(Hs * Hs), Hs)
    assert np.array_equal(region, np.stack([rows, cols], 1))
    assert np.array_equal(X, ((cols - (Hs - 1) / 2) * 0.065 / 2).astype(np.float32))
    assert np.array_equal(Y, ((rows - (Hs - 1) / 2) * 0.065 / 2).astype(np.float32))
    assert math.isclose(ref.conversion_factor(Cfg), 0.065 ** 2, rel_tol=1e-6)
