"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Independent float64 restatements of the *published* closed forms the reference's
own tests compare against (``tests/test_profiles.py`` uses lenstronomy at run
time; lenstronomy -- README pins ==1.9.3 -- is not installed here, so its
algorithms are restated from the papers they implement, with scipy special
functions, through code paths that share nothing with ``ref_torch.py``):

* EPL   -- Tessore & Metcalf 2015, eq. 13: closed form with Gauss 2F1
           (lenstronomy ``EPL.derivatives`` evaluates the same expression).
* SIE   -- EPL at gamma=2 (lenstronomy's SIE is the t=1 member of the family).
* SIS   -- alpha = theta_E * (x, y)/r.
* Shear -- alpha = (g1 x + g2 y, g2 x - g1 y).
* NFW   -- alpha_r(R) = (2/R) int_0^R kappa(r) r dr with the NFW convergence
           (Bartelmann 1996), integrated numerically.
* Shapelets -- Refregier 2003: phi_n via scipy's physicists' Hermite polynomials.
* dPIS  -- Eliasdottir et al. 2007, appendix A: projected surface density of the dual pseudo-isothermal
           sphere, deflection by numerical quadrature of it (alpha_r = (2/R) int kappa r dr).
* Sersic -- the reference's single executable known-answer test.
"""
import math

import numpy as np
from scipy import integrate, special


def ellipticity2phi_q(e1, e2, cmax=0.9999):
    phi = np.arctan2(e2, e1) / 2.0
    c = min(math.hypot(e1, e2), cmax)
    return phi, (1 - c) / (1 + c)


def epl_deriv_2f1(x, y, theta_E, gamma, e1, e2, center_x=0.0, center_y=0.0):
    """Tessore & Metcalf (2015) eq. 13:
    alpha(R,phi) = 2b/(1+q) (b/R)^(t-1) e^{i phi} 2F1(1, t/2; 2-t/2; -(1-q)/(1+q) e^{2 i phi}),
    with b = theta_E sqrt(q) (the product-averaged Einstein radius convention)."""
    x = np.asarray(x, dtype=np.float64) - center_x
    y = np.asarray(y, dtype=np.float64) - center_y
    phi_g, q = ellipticity2phi_q(e1, e2, cmax=1.0)
    t = gamma - 1.0
    b = theta_E * math.sqrt(q)
    c, s = math.cos(phi_g), math.sin(phi_g)
    xr, yr = c * x + s * y, -s * x + c * y
    z = q * xr + 1j * yr
    R = np.abs(z)
    e_iphi = z / R
    f = (1 - q) / (1 + q)
    omega = e_iphi * special.hyp2f1(1.0, t / 2.0, 2.0 - t / 2.0, -f * e_iphi ** 2)
    alpha = 2 * b / (1 + q) * (b / R) ** (t - 1) * omega
    ax, ay = alpha.real, alpha.imag
    return c * ax - s * ay, s * ax + c * ay


def sis_deriv(x, y, theta_E, center_x=0.0, center_y=0.0):
    x = np.asarray(x, dtype=np.float64) - center_x
    y = np.asarray(y, dtype=np.float64) - center_y
    r = np.hypot(x, y)
    return theta_E * x / r, theta_E * y / r


def shear_deriv(x, y, gamma1, gamma2):
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    return gamma1 * x + gamma2 * y, gamma2 * x - gamma1 * y


def nfw_kappa(r, Rs, alpha_Rs):
    """Projected NFW convergence, kappa = 2 rho0 Rs F(x) (Bartelmann 1996 / Wright & Brainerd 2000)."""
    rho0 = alpha_Rs / (4.0 * Rs ** 2 * (1.0 + math.log(0.5)))
    x = r / Rs
    if x < 1:
        F = (1 - 2 / math.sqrt(1 - x * x) * math.atanh(math.sqrt((1 - x) / (1 + x)))) / (x * x - 1)
    elif x > 1:
        F = (1 - 2 / math.sqrt(x * x - 1) * math.atan(math.sqrt((x - 1) / (1 + x)))) / (x * x - 1)
    else:
        F = 1.0 / 3.0
    return 2 * rho0 * Rs * F


def nfw_alpha_r_numeric(R, Rs, alpha_Rs):
    """alpha_r(R) = (2/R) int_0^R kappa(r) r dr (axisymmetric lens)."""
    pts = [Rs] if R > Rs else None
    val, _ = integrate.quad(lambda r: nfw_kappa(r, Rs, alpha_Rs) * r, 0.0, R, points=pts, epsabs=1e-13, epsrel=1e-12, limit=200)
    return 2.0 * val / R


def dpis_kappa(r, theta_E, r_core, r_cut):
    """Eliasdottir et al. (2007) appendix A:  Sigma(R) = Sigma_0 a s/(s-a) [1/sqrt(a^2+R^2) - 1/sqrt(s^2+R^2)];
    in the reference's normalisation the central convergence is theta_E/(2 a) (piemd.py:11-12)."""
    a, s = r_core, r_cut
    return theta_E / (2 * a) * a * s / (s - a) * (1 / math.sqrt(a * a + r * r) - 1 / math.sqrt(s * s + r * r))


def dpis_deriv(x, y, theta_E, r_core, r_cut, center_x=0.0, center_y=0.0):
    """alpha_r(R) = (2/R) int_0^R kappa(r) r dr, integrated numerically (independent of the closed form A20)."""
    x = np.asarray(x, dtype=np.float64) - center_x
    y = np.asarray(y, dtype=np.float64) - center_y
    R = np.hypot(x, y)
    ar = np.array([2.0 / r * integrate.quad(lambda u: dpis_kappa(u, theta_E, r_core, r_cut) * u, 0.0, r,
                                            epsabs=1e-14, epsrel=1e-13, limit=200)[0] for r in R.ravel()]).reshape(R.shape)
    return ar * x / R, ar * y / R


def shapelet_phi_n(n, x):
    """Refregier (2003) eq. 1: phi_n(x) = [2^n sqrt(pi) n!]^(-1/2) H_n(x) exp(-x^2/2)."""
    x = np.asarray(x, dtype=np.float64)
    return special.eval_hermite(n, x) * np.exp(-x * x / 2) / math.sqrt(2.0 ** n * math.sqrt(math.pi) * math.factorial(n))


def shapelet_set(x, y, amps, n_max, beta, center_x=0.0, center_y=0.0):
    """lenstronomy ``ShapeletSet.function`` ordering: n1 descending within each shell n1+n2=n."""
    x = (np.asarray(x, dtype=np.float64) - center_x) / beta
    y = (np.asarray(y, dtype=np.float64) - center_y) / beta
    out = np.zeros_like(x)
    i = 0
    for n in range(n_max + 1):
        for n2 in range(n + 1):
            n1 = n - n2
            out = out + amps[i] * shapelet_phi_n(n1, x) * shapelet_phi_n(n2, y)
            i += 1
    return out


# the one executable known-answer test the reference holds for this path
SERSIC_KAT = dict(args=dict(x=0.0, y=1.0, R_sersic=1.0, n_sersic=2.0, center_x=0.0, center_y=0.0,
                            e1=0.0, e2=0.0, Ie=5.0), expected=5.0)  # tests/test_profiles.py:17-26
