"""NFW_ELLIPSE, TNFW and CoreSersic (tf/profiles/mass/nfw.py:97-134, tnfw.py, tf/profiles/light/sersic.py:83-131):
oracle pins, then the kernels' templates (gigalens_amd/csrc/gl_extra.h on the host) against the oracle and its
autograd.  Pure CPU."""
import math

import numpy as np
import pytest
import torch

from oracle import ref_torch as ref
from tests.test_hostmath_vjp import run_light, run_mass

F64 = torch.float64
K = dict(NFW_ELLIPSE=11, TNFW=12, CORE_SERSIC=19)


def test_oracle_pins():
    r = np.random.default_rng(0)
    x, y = torch.as_tensor(r.normal(size=2000) * 2), torch.as_tensor(r.normal(size=2000) * 2)
    # NFW_ELLIPSE with e = 0 is the NFW
    a = ref.nfw_ellipse_deriv(x, y, 1.7, 0.9, 0.0, 0.0, 0.1, -0.2)
    b = ref.nfw_deriv(x, y, 1.7, 0.9, 0.1, -0.2)
    assert torch.allclose(a[0], b[0], rtol=1e-12) and torch.allclose(a[1], b[1], rtol=1e-12)
    # TNFW -> NFW as the truncation radius goes to infinity (Baltz, Marshall & Oguri 2009, eq. A.18 -> NFW)
    t = ref.tnfw_deriv(x, y, 1.7, 0.9, 1e6, 0.1, -0.2)
    far = (x - 0.1) ** 2 + (y + 0.2) ** 2 > 1e-4
    assert torch.allclose(t[0][far], b[0][far], rtol=2e-5)
    # TNFW deflection is radial and continuous through X = 1
    Rs = 1.3
    xs = torch.tensor([Rs * (1 - 1e-6), Rs, Rs * (1 + 1e-6)], dtype=F64)
    tx, _ = ref.tnfw_deriv(xs, torch.zeros(3, dtype=F64), Rs, 0.8, 4.0, 0.0, 0.0)
    assert abs(float(tx[0] - tx[1])) < 1e-5 and abs(float(tx[2] - tx[1])) < 1e-5
    # CoreSersic as written: with gamma = 0 it is Ie exp(-bn (R^a + Rb^a)/(Rs^a a n) - 1)
    I = ref.core_sersic_light(x, y, 0.8, 2.0, 0.3, 1.5, 0.0, 0.0, 0.0, 0.0, 0.0, 5.0)
    R = torch.sqrt(x ** 2 + y ** 2)
    bn = 1.9992 * 2.0 - 0.3271
    assert torch.allclose(I, 5.0 * torch.exp(-bn * (R ** 1.5 + 0.3 ** 1.5) / (0.8 ** 1.5 * 1.5 * 2.0) - 1.0), rtol=1e-12)


def oracle_mass(name, p, x, y, gx, gy):
    pt = [torch.tensor([v], dtype=F64, requires_grad=True) for v in p]
    X, Y = torch.as_tensor(x)[:, None], torch.as_tensor(y)[:, None]
    fn = ref.nfw_ellipse_deriv if name == "NFW_ELLIPSE" else ref.tnfw_deriv
    ax, ay = fn(X, Y, *pt)
    L = (ax[:, 0] * torch.as_tensor(gx) + ay[:, 0] * torch.as_tensor(gy)).sum()
    grads = torch.autograd.grad(L, pt)
    return ax[:, 0].detach().numpy(), ay[:, 0].detach().numpy(), np.array([float(g) for g in grads])


@pytest.mark.parametrize("name,p", [("NFW_ELLIPSE", [1.7, 0.9, 0.2, -0.15, 0.1, -0.2]),
                                    ("NFW_ELLIPSE", [0.6, 1.4, -0.05, 0.3, -0.3, 0.25]),
                                    ("TNFW", [1.7, 0.9, 5.0, 0.1, -0.2]),
                                    ("TNFW", [0.6, 1.4, 0.4, -0.3, 0.25])])
def test_mass_fwd_and_vjp_f64(hostmath, name, p):
    r = np.random.default_rng(len(p))
    n = 4000
    x, y = r.normal(size=n) * 1.5, r.normal(size=n) * 1.5
    gx, gy = r.normal(size=n), r.normal(size=n)
    ax, ay, grad = run_mass(hostmath, K[name], 0, p, x, y, gx, gy)
    oax, oay, ograd = oracle_mass(name, p, x, y, gx, gy)
    assert np.allclose(ax, oax, rtol=1e-9, atol=1e-11)
    assert np.allclose(ay, oay, rtol=1e-9, atol=1e-11)
    assert np.allclose(grad, ograd, rtol=1e-7, atol=1e-8 * np.abs(ograd).max())


def test_tnfw_through_x_equal_one(hostmath):
    Rs = 2.0
    X = 1 + np.array([-0.3, -0.09, -1e-2, -1e-4, -1e-7, 0.0, 1e-7, 1e-4, 1e-2, 0.09, 0.3])
    x, y = X * Rs, np.zeros_like(X)
    ax, _, _ = run_mass(hostmath, K["TNFW"], 0, [Rs, 1.1, 7.0, 0.0, 0.0], x, y, np.ones_like(x), np.zeros_like(x))
    oax, _ = ref.tnfw_deriv(torch.as_tensor(x), torch.as_tensor(y), Rs, 1.1, 7.0, 0.0, 0.0)
    assert np.allclose(ax, oax.numpy(), rtol=1e-7)  # the closed form loses digits within 1e-7 of X = 1; the series does not


@pytest.mark.parametrize("p", [[0.8, 2.0, 0.3, 1.5, 0.4, 0.2, -0.15, 0.05, 0.02, 80.0],
                               [1.2, 3.5, 0.1, 2.5, 0.1, -0.05, 0.3, 0.0, 0.1, 10.0]])
def test_core_sersic_fwd_and_vjp_f64(hostmath, p):
    r = np.random.default_rng(3)
    n = 3000
    x, y = r.normal(size=n) * 0.8, r.normal(size=n) * 0.8
    gI = r.normal(size=n)
    I, grad, gpx, gpy = run_light(hostmath, K["CORE_SERSIC"], 0, 0, p, x, y, gI)
    pt = [torch.tensor([v], dtype=F64, requires_grad=True) for v in p]
    X = torch.as_tensor(x)[:, None].clone().requires_grad_(True)
    Y = torch.as_tensor(y)[:, None].clone().requires_grad_(True)
    oI = ref.core_sersic_light(X, Y, *pt)
    L = (oI[:, 0] * torch.as_tensor(gI)).sum()
    gr = torch.autograd.grad(L, pt + [X, Y])
    og = np.array([float(g) for g in gr[:len(p)]])
    assert np.allclose(I, oI[:, 0].detach().numpy(), rtol=1e-10, atol=1e-12)
    assert np.allclose(grad, og, rtol=1e-8, atol=1e-9 * np.abs(og).max())
    assert np.allclose(gpx, gr[-2][:, 0].numpy(), rtol=1e-8, atol=1e-9 * float(gr[-2].abs().max()))
    assert np.allclose(gpy, gr[-1][:, 0].numpy(), rtol=1e-8, atol=1e-9 * float(gr[-1].abs().max()))
