"""dPIE family and ScalingRelation (tf/profiles/mass/piemd.py, piep.py, scaling_relation.py): oracle pins, then the
kernels' templates (gigalens_amd/csrc/gl_dpie.h, instantiated on the host by tests/hostmath) against the oracle and
torch.autograd of the oracle.  Pure CPU."""
import ctypes
import math
from ctypes import POINTER, c_double, c_float, c_int
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import published
from oracle import ref_torch as ref
from tests.test_hostmath_vjp import _dp, _fp, run_mass

K = dict(DPIS=6, DPIE=7, DPIEP=8)
F64 = torch.float64


# ----------------------------------------------------------------------------------------------------------------
# oracle pins: the reference's own, independently coded, convergence / Hessian closed forms and published limits
# ----------------------------------------------------------------------------------------------------------------
def _pts(n, seed, scale=3.0):
    r = np.random.default_rng(seed)
    return torch.as_tensor(r.normal(size=n) * scale), torch.as_tensor(r.normal(size=n) * scale)


def _autodiff_hessian(fn, x, y, **kw):
    xx, yy = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    fx, fy = fn(xx, yy, **kw)
    a, b = torch.autograd.grad(fx.sum(), [xx, yy], retain_graph=True)
    c, d = torch.autograd.grad(fy.sum(), [xx, yy])
    return a, b, c, d


def test_oracle_dpie_deriv_matches_reference_hessian_and_convergence():
    """piemd.py:121-138 / :140-149 / :257-300 are written from Kassiola & Kovner's dI/dx, dI/dy -- code that shares
    nothing with the complex-log deflection of :201-255; their agreement pins the deflection restatement."""
    x, y = _pts(3000, 0)
    kw = dict(theta_E=1.3, r_core=0.2, r_cut=5.0, e1=0.2, e2=-0.15, center_x=0.1, center_y=-0.2)
    h_auto = _autodiff_hessian(ref.dpie_deriv, x, y, **kw)
    h_ref = ref.dpie_hessian(x, y, **kw)
    for u, v in zip(h_auto, h_ref):
        assert torch.allclose(u, v, rtol=1e-9, atol=1e-11)
    kappa = ref.dpie_convergence(x, y, **kw)
    assert torch.allclose((h_auto[0] + h_auto[3]) / 2, kappa, rtol=1e-9, atol=1e-12)


def test_oracle_dpis_published_and_limits():
    x, y = _pts(3000, 1)
    kw = dict(theta_E=0.9, r_core=0.15, r_cut=3.0, center_x=-0.1, center_y=0.3)
    fx, fy = ref.dpis_deriv(x, y, **kw)
    # Eliasdottir et al. 2007 eq. A20 (restated independently in oracle/published.py)
    px, py = published.dpis_deriv(x.numpy(), y.numpy(), **kw)
    assert np.allclose(fx.numpy(), px, rtol=1e-10) and np.allclose(fy.numpy(), py, rtol=1e-10)
    # the elliptical profile tends to the spherical one as e -> 0 (first-order in e)
    ex, ey = ref.dpie_deriv(x, y, 0.9, 0.15, 3.0, 1e-7, 0.0, -0.1, 0.3)
    assert torch.allclose(ex, fx, atol=2e-6) and torch.allclose(ey, fy, atol=2e-6)
    # dPIEP with e = 0 is the dPIS
    qx, qy = ref.dpiep_deriv(x, y, 0.9, 0.15, 3.0, 0.0, 0.0, -0.1, 0.3)
    assert torch.allclose(qx, fx, rtol=1e-12) and torch.allclose(qy, fy, rtol=1e-12)
    # the analytic Hessian of piemd.py:62-83 carries (rc+rt)/rt on kappa only: shear equals the derivative's,
    # convergence is (rc+rt)/rt times it -- the quirk the position likelihood inherits
    h_auto = _autodiff_hessian(ref.dpis_deriv, x, y, **kw)
    h_ref = ref.dpis_hessian(x, y, **kw)
    assert torch.allclose((h_auto[0] - h_auto[3]) / 2, (h_ref[0] - h_ref[3]) / 2, rtol=1e-9, atol=1e-12)
    assert torch.allclose(h_auto[1], h_ref[1], rtol=1e-9, atol=1e-12)
    assert torch.allclose((h_ref[0] + h_ref[3]) / 2, (h_auto[0] + h_auto[3]) / 2 * (0.15 + 3.0) / 3.0, rtol=1e-9)
    assert torch.allclose(ref.dpis_convergence(x, y, **kw), (h_ref[0] + h_ref[3]) / 2, rtol=1e-12)


def test_oracle_sort_ra_rs_as_written():
    rc, rt = ref._sort_ra_rs(torch.tensor([0.2, 5.0, 1e-6, 1.0]), torch.tensor([5.0, 0.2, 1.0, 1.00005]))
    assert torch.allclose(rc, torch.tensor([0.2, 0.2, 1e-4, 1.0]))
    assert torch.allclose(rt, torch.tensor([5.0, 0.2 + 1e-4, 1.0, 1.00005 + 1e-4]))


# ----------------------------------------------------------------------------------------------------------------
# host instantiation of the kernel templates vs the oracle
# ----------------------------------------------------------------------------------------------------------------
def oracle_mass(name, p, x, y, gx, gy):
    pt = [torch.tensor([v], dtype=F64, requires_grad=True) for v in p]
    X, Y = torch.as_tensor(x)[:, None], torch.as_tensor(y)[:, None]
    if name == "DPIS":
        ax, ay = ref.dpis_deriv(X, Y, *pt)
    elif name == "DPIE":
        ax, ay = ref.dpie_deriv(X, Y, pt[0], pt[1], pt[2], pt[5], pt[6], pt[3], pt[4])
    else:
        ax, ay = ref.dpiep_deriv(X, Y, pt[0], pt[1], pt[2], pt[5], pt[6], pt[3], pt[4])
    L = (ax[:, 0] * torch.as_tensor(gx) + ay[:, 0] * torch.as_tensor(gy)).sum()
    grads = torch.autograd.grad(L, pt)
    return ax[:, 0].detach().numpy(), ay[:, 0].detach().numpy(), np.array([float(g) for g in grads])


CASES = [
    ("DPIS", [1.1, 0.2, 4.0, 0.05, -0.1]),
    ("DPIS", [0.7, 3.0, 0.5, 0.0, 0.0]),        # r_core > r_cut: the sort of piemd.py:52-60
    ("DPIS", [0.7, 1e-5, 2.0, 0.2, 0.1]),       # r_core below r_min
    ("DPIE", [1.3, 0.2, 5.0, 0.1, -0.2, 0.2, -0.15]),
    ("DPIE", [25.0, 8.0, 300.0, 1.0, -2.0, -0.3, 0.25]),   # cluster-scale halo
    ("DPIE", [0.4, 0.6, 0.3, 0.0, 0.0, 0.05, 0.4]),        # swapped radii
    ("DPIEP", [1.3, 0.2, 5.0, 0.1, -0.2, 0.2, -0.15]),
    ("DPIEP", [0.8, 0.05, 1.5, -0.3, 0.2, -0.4, 0.1]),
]


@pytest.mark.parametrize("name,p", CASES)
def test_dpie_fwd_and_vjp_f64(hostmath, name, p):
    r = np.random.default_rng(len(p) + int(p[0] * 10))
    n = 4000
    x, y = r.normal(size=n) * 3.0, r.normal(size=n) * 3.0
    gx, gy = r.normal(size=n), r.normal(size=n)
    ax, ay, grad = run_mass(hostmath, K[name], 0, p, x, y, gx, gy)
    oax, oay, ograd = oracle_mass(name, p, x, y, gx, gy)
    assert np.allclose(ax, oax, rtol=1e-9, atol=1e-11)
    assert np.allclose(ay, oay, rtol=1e-9, atol=1e-11)
    assert np.allclose(grad, ograd, rtol=1e-7, atol=1e-8 * np.abs(ograd).max())


@pytest.mark.parametrize("name,p", [CASES[0], CASES[3], CASES[4], CASES[6]])
def test_dpie_fwd_and_vjp_f32(hostmath, name, p):
    r = np.random.default_rng(5)
    n = 4000
    x, y = r.normal(size=n) * 3.0, r.normal(size=n) * 3.0
    gx, gy = r.normal(size=n), r.normal(size=n)
    ax, ay, grad = run_mass(hostmath, K[name], 0, p, x, y, gx, gy, f32=True)
    oax, oay, ograd = oracle_mass(name, p, x.astype(np.float32).astype(np.float64),
                                  y.astype(np.float32).astype(np.float64), gx.astype(np.float32), gy.astype(np.float32))
    # K&K's I_w = log(znum_w/zden_w) is 0/0 (removable) at the focus (x', y') = (0, 2 sqrt(e) w): within ~0.1" of it
    # fp32 loses ~2 digits in either factor (in the reference's fp32 graph as well); elsewhere the error is ~1e-7.
    sc = np.abs(oax).max()
    assert np.allclose(ax, oax, rtol=2e-5, atol=2e-5 * sc)
    assert np.allclose(ay, oay, rtol=2e-5, atol=2e-5 * sc)
    assert np.median(np.abs(ax - oax)) < 3e-7 * sc
    assert np.allclose(grad, ograd, rtol=2e-3, atol=2e-4 * np.abs(ograd).max())


def test_dpis_nan_at_centre(hostmath):
    """piemd.py:39 divides by r^2: a grid point on the centre is 0/0 = NaN in the reference (zeroed later in the image)."""
    ax, ay, _ = run_mass(hostmath, K["DPIS"], 0, [1.0, 0.1, 2.0, 0.5, -0.5], [0.5, 1.0], [-0.5, 0.0], [1, 1], [1, 1])
    assert np.isnan(ax[0]) and np.isnan(ay[0]) and np.isfinite(ax[1])


# ---- ScalingRelation ---------------------------------------------------------------------------------------------
def make_catalogue(n_gal, seed, elliptical=True):
    r = np.random.default_rng(seed)
    cat = dict(lum=r.lognormal(0.0, 0.6, n_gal).astype(np.float32),
               center_x=r.uniform(-6, 6, n_gal).astype(np.float32),
               center_y=r.uniform(-6, 6, n_gal).astype(np.float32))
    if elliptical:
        cat["e1"] = r.normal(0, 0.15, n_gal).astype(np.float32)
        cat["e2"] = r.normal(0, 0.15, n_gal).astype(np.float32)
    return cat


def oracle_scaled_profile(base_name, cat, scaling_params, power, lum_star=1.0, extra_consts=None):
    base_params = {"dPIS": ['theta_E', 'r_core', 'r_cut', 'center_x', 'center_y'],
                   "dPIE": ['theta_E', 'r_core', 'r_cut', 'center_x', 'center_y', 'e1', 'e2']}[base_name]
    base = SimpleNamespace(name=base_name, params=base_params)
    cat = dict(cat, **(extra_consts or {}))
    return SimpleNamespace(name=f"Scaled-{base_name}", profile=base, params=list(scaling_params),
                           scaling_params=list(scaling_params), lum_star=lum_star, power=power, galaxy_cat=cat,
                           not_scaling_params=[p for p in base_params if p not in scaling_params])


def catalogue_table(prof, n_gal):
    """[G,7] rows in the layout of gl_dpie.h: (L/L*)^power for scaling parameters, catalogue constants otherwise."""
    t = np.zeros((n_gal, 7), dtype=np.float32)
    cat, scaling_params = prof.galaxy_cat, prof.scaling_params
    unscaled = ref.scaled_unscaled_factors(prof)
    for k, name in enumerate(['theta_E', 'r_core', 'r_cut']):
        if name in scaling_params:
            t[:, k] = unscaled[name].numpy()
        else:
            t[:, k] = np.asarray(cat[name], dtype=np.float32)
    t[:, 3], t[:, 4] = cat["center_x"], cat["center_y"]
    if "e1" in cat:
        t[:, 5], t[:, 6] = cat["e1"], cat["e2"]
    cols = np.array([scaling_params.index(n) if n in scaling_params else -1
                     for n in ['theta_E', 'r_core', 'r_cut']], dtype=np.int32)
    return t, cols


@pytest.mark.parametrize("base,scaling", [("dPIE", ['theta_E', 'r_core', 'r_cut']), ("dPIE", ['theta_E', 'r_cut']),
                                          ("dPIS", ['theta_E', 'r_core', 'r_cut']), ("dPIS", ['r_cut', 'theta_E'])])
def test_scaled_fwd_and_scale_gradient_f64(hostmath, base, scaling):
    n_gal, n = 37, 1500
    cat = make_catalogue(n_gal, 3, elliptical=(base == "dPIE"))
    power = {'theta_E': 0.5, 'r_core': 0.5, 'r_cut': 0.4}
    extra = {} if 'r_core' in scaling else {'r_core': np.full(n_gal, 0.03, dtype=np.float32)}
    prof = oracle_scaled_profile(base, cat, scaling, power, lum_star=1.3, extra_consts=extra)
    t, cols = catalogue_table(prof, n_gal)
    r = np.random.default_rng(9)
    x, y = r.uniform(-7, 7, n), r.uniform(-7, 7, n)
    gx, gy = r.normal(size=n), r.normal(size=n)
    true = {'theta_E': 0.35, 'r_core': 0.04, 'r_cut': 2.5}
    scales = np.array([true[k] for k in scaling] + [0.0] * (3 - len(scaling)))
    ax, ay, gs = np.zeros(n), np.zeros(n), np.zeros(3)
    hostmath.hm_scaled_f64(c_int({"dPIS": 6, "dPIE": 7}[base]), c_int(n_gal), _fp(t),
                           cols.ctypes.data_as(POINTER(c_int)), _dp(scales), c_int(n), _dp(x), _dp(y), _dp(gx),
                           _dp(gy), _dp(ax), _dp(ay), _dp(gs))
    st = {k: torch.tensor([true[k]], dtype=F64, requires_grad=True) for k in scaling}
    oax, oay = ref.mass_deriv(prof, torch.as_tensor(x)[:, None], torch.as_tensor(y)[:, None], **st)
    L = (oax[:, 0] * torch.as_tensor(gx) + oay[:, 0] * torch.as_tensor(gy)).sum()
    og = torch.autograd.grad(L, [st[k] for k in scaling])
    assert np.allclose(ax, oax[:, 0].detach().numpy(), rtol=1e-8, atol=1e-10)
    assert np.allclose(ay, oay[:, 0].detach().numpy(), rtol=1e-8, atol=1e-10)
    # the harness folds gradients in canonical (theta_E, r_core, r_cut) order
    canon = [k for k in ['theta_E', 'r_core', 'r_cut'] if k in scaling]
    for i, k in enumerate(['theta_E', 'r_core', 'r_cut']):
        if k in scaling:
            assert np.isclose(gs[i], float(og[scaling.index(k)]), rtol=1e-6), (k, gs, og)
    assert len(canon) == len(scaling)


# ---- series-expansion accelerator ---------------------------------------------------------------------------------
def test_jet_series_coefficients_vs_autodiff_tower(hostmath):
    """Taylor-mode jets through the member templates (gl_series.h) against nested forward-mode derivatives of the
    oracle's deflection -- the quantities the reference's generated deriv_0..deriv_5 evaluate."""
    n_gal, n = 11, 400
    cat = make_catalogue(n_gal, 5)
    power = {'theta_E': 0.5, 'r_core': 0.5, 'r_cut': 0.4}
    prof = oracle_scaled_profile("dPIE", cat, ['theta_E', 'r_core', 'r_cut'], power, lum_star=1.3)
    prof.amplitude_param, prof.series_param = 'theta_E', 'r_cut'
    t, cols = catalogue_table(prof, n_gal)
    r = np.random.default_rng(2)
    x, y = r.uniform(-7, 7, n), r.uniform(-7, 7, n)
    scales = np.array([123.0, 0.04, 2.5])  # theta_E is ignored by the precompute (scaling_series.py:20)
    out = np.zeros((n, 2, 6))
    hostmath.hm_series_f64(c_int(7), c_int(n_gal), _fp(t), cols.ctypes.data_as(POINTER(c_int)), _dp(scales), c_int(n),
                           _dp(x), _dp(y), _dp(out))
    fx, fy = ref.scaled_series_precompute(prof, 5, torch.as_tensor(x)[:, None], torch.as_tensor(y)[:, None],
                                          theta_E=torch.tensor([1.0], dtype=F64), r_core=torch.tensor([0.04], dtype=F64),
                                          r_cut=torch.tensor([2.5], dtype=F64))
    fact = np.array([math.factorial(k) for k in range(6)], dtype=np.float64)
    ox, oy = fx[:, 0].numpy() / fact, fy[:, 0].numpy() / fact  # derivatives -> Taylor coefficients
    for k in range(6):
        sc = np.abs(ox[:, k]).max()
        # rounding grows ~6x per order in either evaluation (orders 0..5: 1e-15 .. 3e-11 absolute)
        assert np.allclose(out[:, 0, k], ox[:, k], rtol=1e-6, atol=1e-7 * sc), k
        assert np.allclose(out[:, 1, k], oy[:, k], rtol=1e-6, atol=1e-7 * sc), k
    # and the series reproduces the exact scaled deflection near r0 (order-5 remainder)
    ex, ey = ref.mass_deriv(prof, torch.as_tensor(x)[:, None], torch.as_tensor(y)[:, None],
                            theta_E=torch.tensor([0.3], dtype=F64), r_core=torch.tensor([0.04], dtype=F64),
                            r_cut=torch.tensor([2.6], dtype=F64))
    sx = 0.3 * (out[:, 0, :] * 0.1 ** np.arange(6)).sum(-1)
    assert np.allclose(sx, ex[:, 0].numpy(), rtol=1e-5, atol=1e-7)


def test_jet_series_hessian_coefficients_vs_closed_form_tower(hostmath):
    """The Hessian half: space duals of r_cut jets through the member templates against the derivative tower of the
    closed-form dPIE Hessian the reference's generated hessian_0..hessian_5 come from
    (series_codegen/profiles/dpie.py:60-105) -- two independent routes to d^n/dr^n of d alpha/d(x, y)."""
    n_gal, n = 9, 300
    cat = make_catalogue(n_gal, 8)
    power = {'theta_E': 0.5, 'r_core': 0.5, 'r_cut': 0.4}
    prof = oracle_scaled_profile("dPIE", cat, ['theta_E', 'r_core', 'r_cut'], power, lum_star=1.3)
    prof.amplitude_param, prof.series_param = 'theta_E', 'r_cut'
    t, cols = catalogue_table(prof, n_gal)
    r = np.random.default_rng(4)
    x, y = r.uniform(-7, 7, n), r.uniform(-7, 7, n)
    scales = np.array([9.0, 0.04, 2.5])
    out = np.zeros((n, 3, 6))
    hostmath.hm_series_hessian_f64(c_int(7), c_int(n_gal), _fp(t), cols.ctypes.data_as(POINTER(c_int)), _dp(scales),
                                   c_int(n), _dp(x), _dp(y), _dp(out))
    f = ref.scaled_series_precompute_hessian(prof, 5, torch.as_tensor(x)[:, None], torch.as_tensor(y)[:, None],
                                             theta_E=torch.tensor([1.0], dtype=F64),
                                             r_core=torch.tensor([0.04], dtype=F64),
                                             r_cut=torch.tensor([2.5], dtype=F64))
    fact = np.array([math.factorial(k) for k in range(6)], dtype=np.float64)
    for j in range(3):
        o = f[j][:, 0].numpy() / fact
        for k in range(6):
            sc = np.abs(o[:, k]).max()
            assert np.allclose(out[:, j, k], o[:, k], rtol=1e-6, atol=1e-7 * sc), (j, k)
    # order 0 is the Hessian of the scaled population itself (oracle: autograd of its deflection)
    fxx, fxy, fyx, fyy = ref.mass_hessian(prof, torch.as_tensor(x)[:, None], torch.as_tensor(y)[:, None],
                                          theta_E=torch.tensor([1.0], dtype=F64), r_core=torch.tensor([0.04], dtype=F64),
                                          r_cut=torch.tensor([2.5], dtype=F64))
    assert np.allclose(out[:, 0, 0], fxx[:, 0].numpy(), rtol=1e-7, atol=1e-10)
    assert np.allclose(out[:, 1, 0], fxy[:, 0].numpy(), rtol=1e-7, atol=1e-10)
    assert np.allclose(out[:, 2, 0], fyy[:, 0].numpy(), rtol=1e-7, atol=1e-10)
