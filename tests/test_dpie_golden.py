"""The dPIE family and the series accelerator against vectors THE REFERENCE ITSELF produced: ``tests/golden/ref_dpie_series.npz``
holds ``gigalens.series_codegen.profiles.dpie.DPIE().deriv / .hessian`` (series_codegen/profiles/dpie.py:18-105) differentiated
0..5 times in ``r_cut`` by the reference's ``sympy_series`` (sympy_codegen.py:21-29) -- the expressions its generator prints as
``deriv_0..5`` / ``hessian_0..5`` of tf/series/profiles/dpie.py -- evaluated with mpmath at 60 digits
(tests/golden/make_dpie_golden.py; build container only).

CPU half (this file): the oracle's restatement and the product's host-instantiated templates (float64) against the
fixture.  GPU half: tests/test_gpu_dpie.py::test_reference_fixture_*.

Tolerance model.  The Kassiola-Kovner form has removable 0/0 points ("foci") at (0, +-2 sqrt(e) r_w); ANY float64 evaluation
of the k-th r_cut-derivative at distance d from a focus loses about (3/d)^k / d digits (measured on the oracle and on the
jets alike), so a point is compared with  tol = 2e-13 / d * (1 + (3/d)^k) + 1e-11 * 8^k  relative to the order's scale (d capped
at 1; the second term is the ordinary growth of rounding with the order, away from any focus), and is skipped where that
exceeds 3e-2 (orders >= 3 within 1e-3 of a focus: only extended precision is meaningful there).
"""
import math
from ctypes import POINTER, c_double, c_float, c_int

import numpy as np
import pytest
import torch

from oracle import ref_torch as ref

F64 = torch.float64


@pytest.fixture(scope="module")
def fx():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_dpie_series.npz"))
    assert "series_codegen.profiles.dpie.DPIE" in str(g["provenance"])
    return {k: g[k] for k in g.files}


def tolerance(fd, k, floor=0.0, amp=1.0):
    """``amp``: size of the coordinates in units of the focus distance's own scale -- the cancellation is between numbers of
    the size of the focus ordinate 2 sqrt(e) r_w (10 for the r_cut = 10 cases), so the lost digits scale with it."""
    d = np.minimum(fd, 1.0)
    return 2e-13 * amp / d * (1.0 + (3.0 / d) ** k) + 1e-11 * 8.0 ** k + floor


def check(got, want, fd, k, what, floor=0.0, min_checked=0.5, amp=1.0):
    """|got - want| <= tol(point, order) * scale(order); returns the share of points that were comparable."""
    sc = np.abs(want).max()
    tol = tolerance(fd, k, floor, amp)
    ok = tol <= 3e-2
    err = np.abs(got - want) / sc
    bad = ok & ~(err <= tol)
    assert not bad.any(), (what, k, float(err[bad].max()), float(tol[bad].min()), int(bad.sum()))
    assert ok.mean() >= min_checked, (what, k, ok.mean())
    return ok.mean()


def test_fixture_is_selfconsistent(fx):
    """hessian_0 = d deriv_0 / d(x, y) (central differences of nothing -- the fixture has no neighbours -- so: symmetry, and
    the trace identity against the reference's closed-form convergence, piemd.py:140-149)."""
    H = fx["hessian"]
    assert np.array_equal(H[:, :, 1], H[:, :, 2])
    e, rc, rt, x, y = (fx[k] for k in ("e", "r_core", "r_cut", "x", "y"))
    rem2 = x ** 2 / (1 + e) ** 2 + y ** 2 / (1 - e) ** 2
    kappa = rt / (rt - rc) / 2 * (1 / np.sqrt(rem2 + rc ** 2) - 1 / np.sqrt(rem2 + rt ** 2))
    assert np.allclose(0.5 * (H[:, 0, 0] + H[:, 0, 3]), kappa, rtol=1e-12)
    assert len(np.unique(fx["case"])) == 12 and fx["deriv"].shape[1:] == (6, 2)


def test_oracle_series_tower_matches_the_reference(fx):
    """oracle.dpie_series_precompute / _hessian (nested forward-mode JVPs of the restated deflection / closed-form Hessian)
    against the reference's own derivative tower."""
    x, y, e, rc, rt = (torch.as_tensor(fx[k]) for k in ("x", "y", "e", "r_core", "r_cut"))
    z = torch.zeros_like(e)
    f_x, f_y = ref.dpie_series_precompute(5, x, y, 1.0, rc, rt, e, z, 0.0, 0.0)
    h = ref.dpie_series_precompute_hessian(5, x, y, 1.0, rc, rt, e, z, 0.0, 0.0)
    for k in range(6):
        check(f_x[:, k].numpy(), fx["deriv"][:, k, 0], fx["focus_distance"], k, "f_x")
        check(f_y[:, k].numpy(), fx["deriv"][:, k, 1], fx["focus_distance"], k, "f_y")
        for j, col in ((0, 0), (1, 1), (2, 3)):
            check(h[j][:, k].numpy(), fx["hessian"][:, k, col], fx["focus_distance"], k + 1, f"h{j}", min_checked=0.4)


def test_oracle_dpie_deriv_and_hessian_match_the_reference(fx):
    """The free-standing halo (piemd.py:105-138: theta_E scale, radius sort, |e| clamp, rotation by phi) against order 0 of
    the fixture, rotated into a random frame: alpha(R^T p) = R^T alpha'(p), H = R^T H' R."""
    r = np.random.default_rng(3)
    x, y, e, rc, rt = (fx[k] for k in ("x", "y", "e", "r_core", "r_cut"))
    phi = r.uniform(-np.pi / 2, np.pi / 2, x.shape)
    te = r.uniform(0.5, 3.0, x.shape)
    cx, cy = r.normal(0, 1, x.shape), r.normal(0, 1, x.shape)
    c, s = np.cos(phi), np.sin(phi)
    # halo-frame point (x, y) <-> sky point: rotate by +phi, shift by the centre (tf/profiles/mass/piemd.py:109-111)
    xs, ys = c * x - s * y + cx, s * x + c * y + cy
    e1, e2 = e * np.cos(2 * phi), e * np.sin(2 * phi)
    T = lambda a: torch.as_tensor(a, dtype=F64)
    ax, ay = ref.dpie_deriv(T(xs), T(ys), T(te), T(rc), T(rt), T(e1), T(e2), T(cx), T(cy))
    wx = te * (c * fx["deriv"][:, 0, 0] - s * fx["deriv"][:, 0, 1])
    wy = te * (s * fx["deriv"][:, 0, 0] + c * fx["deriv"][:, 0, 1])
    fd = fx["focus_distance"]
    check(ax.numpy(), wx, fd, 0, "alpha_x", floor=1e-13)
    check(ay.numpy(), wy, fd, 0, "alpha_y", floor=1e-13)
    fxx, fxy, fyx, fyy = ref.dpie_hessian(T(xs), T(ys), T(te), T(rc), T(rt), T(e1), T(e2), T(cx), T(cy))
    hxx, hxy, hyy = (fx["hessian"][:, 0, k] for k in (0, 1, 3))
    wxx = te * (c * c * hxx - 2 * c * s * hxy + s * s * hyy)
    wxy = te * (c * s * (hxx - hyy) + (c * c - s * s) * hxy)
    wyy = te * (s * s * hxx + 2 * c * s * hxy + c * c * hyy)
    for got, want, nm in ((fxx, wxx, "f_xx"), (fxy, wxy, "f_xy"), (fyy, wyy, "f_yy")):
        check(got.numpy(), want, fd, 1, nm, floor=1e-13)


def _dp(a):
    return a.ctypes.data_as(POINTER(c_double))


def _halo_table(fx, i):
    """One-galaxy catalogue row of DPIESeries (profiles/mass/dpie_series.py::_series_inputs): the halo IS the catalogue,
    r_cut is the only scaled quantity (unscaled factor 1), theta_E = 1."""
    row = np.array([[1.0, fx["r_core"][i], 1.0, 0.0, 0.0, fx["e"][i], 0.0]], dtype=np.float32)
    assert np.array_equal(row[0, [1, 5]].astype(np.float64), [fx["r_core"][i], fx["e"][i]])  # float32-exact fixture values
    return row, np.array([-1, -1, 0], dtype=np.int32)


def test_product_jets_match_the_reference_tower(fx, hostmath):
    """The product's series precompute -- the dPIE member templates instantiated on truncated Taylor series
    (csrc/gl_series.h, gl_jet.h), here on the host in float64 -- against the reference's tower: C_n n! = f_n."""
    fact = np.array([math.factorial(k) for k in range(6)], dtype=np.float64)
    M = fx["x"].size
    dv, hs = np.zeros((M, 2, 6)), np.zeros((M, 3, 6))
    for i in range(M):
        row, cols = _halo_table(fx, i)
        scales = np.array([fx["r_cut"][i], 1.0, 1.0])
        xi, yi = np.array([fx["x"][i]]), np.array([fx["y"][i]])
        hostmath.hm_series_f64(c_int(7), c_int(1), row.ctypes.data_as(POINTER(c_float)), cols.ctypes.data_as(POINTER(c_int)),
                               _dp(scales), c_int(1), _dp(xi), _dp(yi), _dp(dv[i]))
        hostmath.hm_series_hessian_f64(c_int(7), c_int(1), row.ctypes.data_as(POINTER(c_float)),
                                       cols.ctypes.data_as(POINTER(c_int)), _dp(scales), c_int(1), _dp(xi), _dp(yi), _dp(hs[i]))
    fd = fx["focus_distance"]
    for k in range(6):
        check(dv[:, 0, k] * fact[k], fx["deriv"][:, k, 0], fd, k, "jet f_x")
        check(dv[:, 1, k] * fact[k], fx["deriv"][:, k, 1], fd, k, "jet f_y")
        for j, col in ((0, 0), (1, 1), (2, 3)):
            check(hs[:, j, k] * fact[k], fx["hessian"][:, k, col], fd, k + 1, f"jet h{j}", min_checked=0.4)


def test_product_dpie_templates_match_the_reference(fx, hostmath):
    """The product's free-standing dPIE deflection (csrc/gl_dpie.h, host-instantiated in float64) against deriv_0."""
    M = fx["x"].size
    got = np.zeros((M, 2))
    for i in range(M):
        p = np.array([1.0, fx["r_core"][i], fx["r_cut"][i], 0.0, 0.0, fx["e"][i], 0.0])
        xi, yi, z = np.array([fx["x"][i]]), np.array([fx["y"][i]]), np.zeros(1)
        ax, ay, grad = np.zeros(1), np.zeros(1), np.zeros(8)
        hostmath.hm_mass_f64(c_int(7), c_int(0), _dp(p), c_int(1), _dp(xi), _dp(yi), _dp(z), _dp(z), _dp(ax), _dp(ay), _dp(grad))
        got[i] = ax[0], ay[0]
    check(got[:, 0], fx["deriv"][:, 0, 0], fx["focus_distance"], 0, "gl_dpie alpha_x", floor=1e-13)
    check(got[:, 1], fx["deriv"][:, 0, 1], fx["focus_distance"], 0, "gl_dpie alpha_y", floor=1e-13)
