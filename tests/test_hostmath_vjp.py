"""The kernels' per-profile math (gigalens_amd/csrc/gl_profiles.h), instantiated on the host in float64,
against the oracle and torch.autograd of the oracle: forward values, the hand-written VJPs and the
per-sample chain rule back to the reference's raw parameters.  Pure CPU -- no GPU, no product path."""
import ctypes
from ctypes import POINTER, c_double, c_float, c_int, c_uint

import numpy as np
import pytest
import torch

from oracle import ref_torch as ref

K = dict(EPL=1, SIE=2, NFW=3, SHEAR=4, SIS=5, SERSIC=16, SERSIC_ELLIPSE=17, SHAPELETS=18)


def _dp(a):
    return a.ctypes.data_as(POINTER(c_double))


def _fp(a):
    return a.ctypes.data_as(POINTER(c_float))


def run_mass(hm, kind, iparam, p, x, y, gx, gy, f32=False):
    dt = np.float32 if f32 else np.float64
    P = _fp if f32 else _dp
    p, x, y, gx, gy = (np.ascontiguousarray(v, dtype=dt) for v in (p, x, y, gx, gy))
    ax, ay, grad = np.zeros_like(x), np.zeros_like(x), np.zeros(len(p), dtype=dt)
    fn = hm.hm_mass_f32 if f32 else hm.hm_mass_f64
    fn(c_int(kind), c_int(iparam), P(p), c_int(len(x)), P(x), P(y), P(gx), P(gy), P(ax), P(ay), P(grad))
    return ax, ay, grad


def run_light(hm, kind, iparam, flags, p, x, y, gI, f32=False):
    dt = np.float32 if f32 else np.float64
    P = _fp if f32 else _dp
    p, x, y, gI = (np.ascontiguousarray(v, dtype=dt) for v in (p, x, y, gI))
    I, gpx, gpy, grad = np.zeros_like(x), np.zeros_like(x), np.zeros_like(x), np.zeros(len(p), dtype=dt)
    fn = hm.hm_light_f32 if f32 else hm.hm_light_f64
    fn(c_int(kind), c_int(iparam), c_uint(flags), P(p), c_int(len(x)), P(x), P(y), P(gI), P(I), P(grad), P(gpx), P(gpy))
    return I, grad, gpx, gpy


def oracle_mass(name, p, x, y, gx, gy, niter=50):
    pt = [torch.tensor([v], dtype=torch.float64, requires_grad=True) for v in p]
    X, Y = torch.as_tensor(x)[:, None], torch.as_tensor(y)[:, None]
    if name == "EPL":
        ax, ay = ref.epl_deriv(X, Y, *pt, niter_cap=niter)
    elif name == "SIE":
        ax, ay = ref.sie_deriv(X, Y, *pt)
    elif name == "NFW":
        ax, ay = ref.nfw_deriv(X, Y, *pt)
    elif name == "SHEAR":
        ax, ay = ref.shear_deriv(X, Y, *pt)
    else:
        ax, ay = ref.sis_deriv(X, Y, *pt)
    L = (ax[:, 0] * torch.as_tensor(gx) + ay[:, 0] * torch.as_tensor(gy)).sum()
    grads = torch.autograd.grad(L, pt)
    return ax[:, 0].detach().numpy(), ay[:, 0].detach().numpy(), np.array([float(g) for g in grads])


MASS_CASES = [
    ("EPL", [1.2, 2.2, -0.1, 0.1, 0.03, -0.02]),
    ("EPL", [1.0, 1.7, 0.25, 0.3, -0.1, 0.05]),
    ("EPL", [1.5, 2.45, 0.02, -0.01, 0.0, 0.0]),
    ("SIE", [1.2, 0.1, -0.1, 0.02, 0.01]),
    ("SIE", [0.8, 1e-3, 1e-3, 0.0, 0.0]),
    ("NFW", [1.7, 0.9, 0.1, -0.2]),
    ("NFW", [0.6, 1.4, -0.3, 0.25]),
    ("SHEAR", [0.05, -0.03]),
    ("SIS", [1.1, 0.04, -0.06]),
]


@pytest.mark.parametrize("name,p", MASS_CASES)
def test_mass_fwd_and_vjp_f64(hostmath, name, p):
    r = np.random.default_rng(hash(name) % 1000 + len(p))
    n = 4000
    x, y = r.normal(size=n) * 1.5, r.normal(size=n) * 1.5
    gx, gy = r.normal(size=n), r.normal(size=n)
    ax, ay, grad = run_mass(hostmath, K[name], 60, p, x, y, gx, gy)
    oax, oay, ograd = oracle_mass(name, p, x, y, gx, gy, niter=60)
    assert np.allclose(ax, oax, rtol=1e-9, atol=1e-11)
    assert np.allclose(ay, oay, rtol=1e-9, atol=1e-11)
    assert np.allclose(grad, ograd, rtol=1e-7, atol=1e-8 * np.abs(ograd).max())


def test_nfw_through_x_equal_one(hostmath):
    """The series formulation of g(X) must agree with the reference's closed form arbitrarily close to
    X = 1 (where the closed form is 0/0) and reproduce the g(1) = 1.0 quirk exactly at X == 1."""
    Rs, aRs = 2.0, 1.1
    X = np.concatenate([1 + np.array([-0.3, -0.09, -1e-2, -1e-4, -1e-7, 1e-7, 1e-4, 1e-2, 0.09, 0.3, 3.0]), [1.0]])
    x, y = X * Rs, np.zeros_like(X)
    ax, _, _ = run_mass(hostmath, K["NFW"], 0, [Rs, aRs, 0.0, 0.0], x, y, np.ones_like(x), np.zeros_like(x))
    oax, _ = ref.nfw_deriv(torch.as_tensor(x), torch.as_tensor(y), Rs, aRs, 0.0, 0.0)
    assert np.allclose(ax[:-1], oax.numpy()[:-1], rtol=1e-8)  # oracle itself loses digits within 1e-7 of X=1
    assert np.isclose(ax[-1], oax.numpy()[-1], rtol=1e-14)


def oracle_light(name, p, x, y, gI, n_max=0, interpolate=True):
    pt = [torch.tensor([v], dtype=torch.float64, requires_grad=True) for v in p]
    X = torch.as_tensor(x)[:, None].clone().requires_grad_(True)
    Y = torch.as_tensor(y)[:, None].clone().requires_grad_(True)
    if name == "SERSIC":
        I = ref.sersic_light(X, Y, pt[0], pt[1], pt[2], pt[3], pt[4])
    elif name == "SERSIC_ELLIPSE":
        I = ref.sersic_light(X, Y, pt[0], pt[1], pt[4], pt[5], pt[6], pt[2], pt[3])
    else:
        I = ref.shapelets_light(X, Y, pt[1], pt[2], pt[0], pt[3:], n_max, interpolate)
    L = (I[:, 0] * torch.as_tensor(gI)).sum()
    grads = torch.autograd.grad(L, pt + [X, Y])
    return (I[:, 0].detach().numpy(), np.array([float(g) for g in grads[:len(p)]]),
            grads[-2][:, 0].numpy(), grads[-1][:, 0].numpy())


@pytest.mark.parametrize("name,p", [
    ("SERSIC", [0.25, 2.0, 0.1, -0.05, 150.0]),
    ("SERSIC", [0.4, 0.7, -0.2, 0.15, 30.0]),
    ("SERSIC_ELLIPSE", [0.3, 3.2, 0.2, -0.15, 0.05, 0.02, 80.0]),
    ("SERSIC_ELLIPSE", [1.0, 4.0, -0.05, 0.3, 0.0, 0.1, 500.0]),
])
def test_sersic_fwd_and_vjp_f64(hostmath, name, p):
    r = np.random.default_rng(11)
    n = 3000
    x, y = r.normal(size=n) * 0.8, r.normal(size=n) * 0.8
    gI = r.normal(size=n)
    I, grad, gpx, gpy = run_light(hostmath, K[name], 0, 0, p, x, y, gI)
    oI, ograd, ogx, ogy = oracle_light(name, p, x, y, gI)
    assert np.allclose(I, oI, rtol=1e-10, atol=1e-12)
    assert np.allclose(grad, ograd, rtol=1e-8, atol=1e-9 * np.abs(ograd).max())
    assert np.allclose(gpx, ogx, rtol=1e-8, atol=1e-9 * np.abs(ogx).max())
    assert np.allclose(gpy, ogy, rtol=1e-8, atol=1e-9 * np.abs(ogy).max())


@pytest.mark.parametrize("n_max,interpolate", [(10, False), (5, False), (10, True), (3, True)])
def test_shapelets_fwd_and_vjp_f64(hostmath, n_max, interpolate):
    r = np.random.default_rng(n_max)
    L = (n_max + 1) * (n_max + 2) // 2
    p = [0.6, 0.05, -0.03] + list(r.normal(size=L) * 3)
    n = 1500
    x, y = r.normal(size=n) * 1.2, r.normal(size=n) * 1.2
    x[:3] = [4.0, -3.5, 0.2]  # some points outside the table range [-5 beta, 5 beta]
    gI = r.normal(size=n)
    I, grad, gpx, gpy = run_light(hostmath, K["SHAPELETS"], n_max, 1 if interpolate else 0, p, x, y, gI)
    oI, ograd, ogx, ogy = oracle_light("SHAPELETS", p, x, y, gI, n_max, interpolate)
    tol = dict(rtol=1e-9, atol=1e-10)
    assert np.allclose(I, oI, **tol)
    assert np.allclose(grad, ograd, rtol=1e-8, atol=1e-9 * np.abs(ograd).max())
    assert np.allclose(gpx, ogx, rtol=1e-8, atol=1e-9 * np.abs(ogx).max())
    assert np.allclose(gpy, ogy, rtol=1e-8, atol=1e-9 * np.abs(ogy).max())


@pytest.mark.parametrize("has_err", [False, True])
def test_chi2_terms_and_image_cotangent(hostmath, has_err):
    """tf/model.py:92-99 and its adjoint (sigma depends on the model image)."""
    r = np.random.default_rng(3)
    n = 500
    m = np.abs(r.normal(size=n)) * 3 + 0.1
    o = m + r.normal(size=n) * 0.5
    w = (r.uniform(size=n) > 0.2).astype(np.float64)
    err = np.abs(r.normal(size=n)) + 0.3
    bg, t = 0.2, 100.0
    chi2, norm, gm = c_double(), c_double(), np.zeros(n)
    hostmath.hm_chi2_f64(c_int(n), _dp(m), _dp(o), _dp(w), c_int(int(has_err)), _dp(err), c_double(bg * bg),
                         c_double(1 / t), ctypes.byref(chi2), ctypes.byref(norm), _dp(gm))
    mt = torch.tensor(m, requires_grad=True)
    sig = torch.as_tensor(err) if has_err else torch.sqrt(bg ** 2 + mt / t)
    c2 = (((mt - torch.as_tensor(o)) / sig) ** 2 * torch.as_tensor(w)).sum()
    nm = (torch.log(2 * np.pi * sig ** 2) * torch.as_tensor(w)).sum()
    ll = -0.5 * (c2 + nm)
    (g,) = torch.autograd.grad(ll, mt)
    assert np.isclose(chi2.value, float(c2), rtol=1e-12) and np.isclose(norm.value, float(nm), rtol=1e-12)
    assert np.allclose(gm, g.numpy(), rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("name,p", MASS_CASES[:7])
def test_mass_f32_host_close_to_f64(hostmath, name, p):
    """float32 instantiation (host libm stand-ins for the GPU transcendentals) stays within the
    tolerance the GPU tests use; catches formulations that are f32-unstable before spending GPU time."""
    r = np.random.default_rng(5)
    n = 4000
    x, y = r.normal(size=n) * 1.5, r.normal(size=n) * 1.5
    gx, gy = r.normal(size=n), r.normal(size=n)
    ax, ay, grad = run_mass(hostmath, K[name], 50, p, x, y, gx, gy, f32=True)
    x32, y32 = x.astype(np.float32).astype(np.float64), y.astype(np.float32).astype(np.float64)
    oax, oay, ograd = oracle_mass(name, p, x32, y32, gx.astype(np.float32).astype(np.float64),
                                  gy.astype(np.float32).astype(np.float64))
    assert np.allclose(ax, oax, rtol=2e-5, atol=2e-6)
    assert np.allclose(ay, oay, rtol=2e-5, atol=2e-6)
    # SIE at |e| ~ 1e-3 (q -> 1) is ill-conditioned in float32 by construction of the closed form
    # (b/sqrt(1-q^2) * atan(sqrt(1-q^2) ...)): the e1/e2 gradients cancel to ~3 digits in ANY fp32 evaluation.
    loose = name == "SIE" and abs(p[1]) < 0.01
    assert np.allclose(grad, ograd, rtol=1e-2 if loose else 2e-4, atol=(1e-3 if loose else 2e-5) * np.abs(ograd).max())


@pytest.mark.parametrize("name,p", [("EPL", [1.2, 2.2, -0.1, 0.1, 0.03, -0.02]), ("EPL", [0.9, 1.7, 0.25, 0.3, -0.1, 0.05]),
                                    ("SIE", [1.2, 0.1, -0.1, 0.02, 0.01]), ("NFW", [1.7, 0.9, 0.1, -0.2]),
                                    ("NFW", [0.6, 1.4, -0.3, 0.25]), ("SHEAR", [0.05, -0.03]), ("SIS", [1.1, 0.04, -0.06])])
def test_nested_dual_hessian_and_parameter_jet(hostmath, name, p):
    """gl_dual.h instantiated on the profile templates (what the image-position kernels run) == torch autograd of the
    oracle: deflection, Hessian d alpha/d(x,y) (tf/profile.py:9-27) and the parameter derivatives of both."""
    r = np.random.default_rng(len(p))
    for _ in range(6):
        x0, y0 = r.normal(size=2) * 1.2
        out = np.zeros(6 * (1 + len(p)))
        hostmath.hm_lens_jet_f64(c_int(K[name]), c_int(60), _dp(np.asarray(p, dtype=np.float64)), c_double(x0), c_double(y0),
                                 _dp(out))
        pt = [torch.tensor([v], dtype=torch.float64, requires_grad=True) for v in p]
        X = torch.tensor([[x0]], dtype=torch.float64, requires_grad=True)
        Y = torch.tensor([[y0]], dtype=torch.float64, requires_grad=True)
        fn = dict(EPL=lambda: ref.epl_deriv(X, Y, *pt, niter_cap=60), SIE=lambda: ref.sie_deriv(X, Y, *pt),
                  NFW=lambda: ref.nfw_deriv(X, Y, *pt), SHEAR=lambda: ref.shear_deriv(X, Y, *pt), SIS=lambda: ref.sis_deriv(X, Y, *pt))[name]
        fx, fy = fn()
        fxx, fxy = torch.autograd.grad(fx.sum(), [X, Y], create_graph=True)
        fyx, fyy = torch.autograd.grad(fy.sum(), [X, Y], create_graph=True)
        q = [fx.sum(), fy.sum(), fxx.sum(), fxy.sum(), fyx.sum(), fyy.sum()]
        assert np.allclose(out[:6], [float(v) for v in q], rtol=1e-9, atol=1e-11)
        for i, v in enumerate(q):
            g = torch.autograd.grad(v, pt, retain_graph=True, allow_unused=True)
            want = np.array([0.0 if gi is None else float(gi) for gi in g])
            got = out[6 + i::6][: len(p)]
            assert np.allclose(got, want, rtol=1e-7, atol=1e-8 * max(1.0, np.abs(want).max())), (name, i, got, want)
