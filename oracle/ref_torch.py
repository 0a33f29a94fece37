"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Op-for-op CPU restatement, in torch, of the reference's TF substrate for the
hot path: grid -> ray-shoot -> light render -> NaN->0 -> [PSF, pool] -> chi^2
log-likelihood.  It keeps the reference's *unfused* tensor algebra: every
intermediate is an ``(N_pix, B)`` tensor, the grid is replicated per batch
element, the EPL series is a data-dependent loop with a batch-max trip count,
and gradients come from ``torch.autograd`` (the stand-in for ``tf.GradientTape``).

It is dtype-generic: ``float64`` is the parity oracle, ``float32`` is the
"reference algorithm restated on torch-CPU" baseline that ``bench.py`` times.

All citations are ``path:line`` relative to ``/root/reference/``.
Profiles are dispatched by duck-typing on ``profile.name`` (the reference's
``_name`` strings), so any object carrying ``name`` (+ ``n_max``/``interpolate``/
``niter`` where relevant) works -- including the product's profile classes.
"""
import math
from typing import Dict, List

import numpy as np
import torch

LN2 = math.log(2.0)


def _t(v, like):
    """Broadcast helper: python scalars / tensors -> tensor of like's dtype."""
    if torch.is_tensor(v):
        return v.to(like.dtype)
    return torch.as_tensor(v, dtype=like.dtype)


# --------------------------------------------------------------------------
# grid  (src/gigalens/simulator.py:32-64)
# --------------------------------------------------------------------------
class LensWCS:
    """src/gigalens/simulator.py:32-64, incl. its quirks: ``pix2angle`` applies
    T^T (einsum 'ij,i...->...j', :53) while the origin uses T (:50);
    ``transform_angle2pix`` inverts the un-supersampled T (:37-38)."""

    def __init__(self, n, supersample=1, transform_pix2angle=None, pix_scale=1.0):
        if transform_pix2angle is None:
            transform_pix2angle = np.eye(2) * pix_scale
        transform_pix2angle = np.asarray(transform_pix2angle, dtype=np.float64)
        self.transform_pix2angle = transform_pix2angle / supersample
        self.transform_angle2pix = np.linalg.inv(transform_pix2angle)
        if isinstance(n, int):
            self.n_x, self.n_y = n, n
        else:
            self.n_x, self.n_y = n
        self.supersample = supersample
        low_x = -(self.n_x * supersample - 1) / 2
        low_y = -(self.n_y * supersample - 1) / 2
        self.radec_at_xy_0 = np.squeeze(self.transform_pix2angle @ np.array([[low_x], [low_y]]))

    def pix2angle(self, x, y):
        v = np.stack([np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)])
        T = self.transform_pix2angle
        ra = T[0, 0] * v[0] + T[1, 0] * v[1] + self.radec_at_xy_0[0]
        dec = T[0, 1] * v[0] + T[1, 1] * v[1] + self.radec_at_xy_0[1]
        return ra.astype(np.float32), dec.astype(np.float32)


def build_grid(sim_config):
    """tf/simulator.py:34-51.  Returns (region (N,2) [row,col], img_region (H,W),
    img_X (N,), img_Y (N,)) with the grid as float32 numpy (f64 arithmetic, then cast)."""
    ss = int(sim_config.supersample)
    wcs = LensWCS(n=sim_config.num_pix, supersample=ss,
                  transform_pix2angle=sim_config.transform_pix2angle,
                  pix_scale=sim_config.delta_pix)
    if sim_config.pix_region is None:
        region = np.ones((wcs.n_x * ss, wcs.n_y * ss), dtype=bool)
        img_region = np.ones((wcs.n_x, wcs.n_y))
    else:
        img_region = np.asarray(sim_config.pix_region)
        region = np.repeat(img_region, ss, axis=0).reshape(wcs.n_x * ss, wcs.n_y)
        region = np.repeat(region, ss, axis=1).reshape(wcs.n_x * ss, wcs.n_y * ss)
    region = np.argwhere(region)  # == tf.where: row-major (N,2) [row, col]
    img_X, img_Y = wcs.pix2angle(region[:, 1], region[:, 0])
    return wcs, region, img_region.astype(np.float32), img_X, img_Y


def conversion_factor(sim_config):
    """tf/simulator.py:22-29: det of the UN-supersampled transform."""
    T = (np.eye(2) * sim_config.delta_pix if sim_config.transform_pix2angle is None
         else np.asarray(sim_config.transform_pix2angle, dtype=np.float64))
    return float(np.float32(np.linalg.det(T.astype(np.float32))))


# --------------------------------------------------------------------------
# mass profiles
# --------------------------------------------------------------------------
def _rotate(x, y, phi):
    """tf/profiles/mass/epl.py:59-64 (identical in sie.py:45-49)."""
    c, s = torch.cos(phi), torch.sin(phi)
    return x * c + y * s, -x * s + y * c


def epl_deriv(x, y, theta_E, gamma, e1, e2, center_x, center_y, niter_cap=50):
    """tf/profiles/mass/epl.py:19-57."""
    theta_E, gamma, e1, e2, center_x, center_y = (_t(v, x) for v in (theta_E, gamma, e1, e2, center_x, center_y))
    phi = torch.atan2(e2, e1) / 2
    c = torch.clamp(torch.sqrt(e1 ** 2 + e2 ** 2), 0, 1)
    q = (1 - c) / (1 + c)
    theta_E_conv = theta_E / torch.sqrt((1.0 + q ** 2) / (2.0 * q))
    b = theta_E_conv * torch.sqrt((1 + q ** 2) / 2)
    t = gamma - 1
    x, y = x - center_x, y - center_y
    x, y = _rotate(x, y, phi)
    R = torch.clamp(torch.sqrt((q * x) ** 2 + y ** 2), 1e-10, 1e10)
    angle = torch.atan2(y, q * x)
    f = (1 - q) / (1 + q)
    Cs, Ss = torch.cos(angle), torch.sin(angle)
    Cs2, Ss2 = torch.cos(2 * angle), torch.sin(2 * angle)
    # :37  stop_gradient(log(1e-12)/log(reduce_max(f)) + 2); max over the WHOLE batch
    with torch.no_grad():
        fmax = torch.max(f.detach())
        niter = (torch.log(torch.as_tensor(1e-12, dtype=x.dtype)) / torch.log(fmax) + 2).item()
    last_x, last_y, f_x, f_y = Cs, Ss, Cs, Ss
    n = 1.0
    it = 0
    while n < niter and it < niter_cap:  # :47-54 (maximum_iterations=self.niter)
        prefac_ = -f * (2 * n - (2 - t)) / (2 * n + (2 - t))
        last_x, last_y = prefac_ * (Cs2 * last_x - Ss2 * last_y), prefac_ * (Ss2 * last_x + Cs2 * last_y)
        f_x, f_y = f_x + last_x, f_y + last_y
        n += 1.0
        it += 1
    prefac = (2 * b) / (1 + q) * torch.pow(b / R, t - 1)
    f_x, f_y = f_x * prefac, f_y * prefac
    return _rotate(f_x, f_y, -phi)


def sie_deriv(x, y, theta_E, e1, e2, center_x, center_y):
    """tf/profiles/mass/sie.py:13-42.  s_scale is the LOCAL 0 (sie.py:15), so s==0."""
    theta_E, e1, e2, center_x, center_y = (_t(v, x) for v in (theta_E, e1, e2, center_x, center_y))
    s_scale = 0
    phi = torch.atan2(e2, e1) / 2
    c = torch.clamp(torch.sqrt(e1 ** 2 + e2 ** 2), max=0.9999)
    q = (1 - c) / (1 + c)
    theta_E_conv = theta_E / torch.sqrt((1.0 + q ** 2) / (2.0 * q))
    b = theta_E_conv * torch.sqrt((1 + q ** 2) / 2)
    s = s_scale * torch.sqrt((1 + q ** 2) / (2 * q ** 2))
    x, y = x - center_x, y - center_y
    x, y = _rotate(x, y, phi)
    psi = torch.sqrt(q ** 2 * (s ** 2 + x ** 2) + y ** 2)
    fx = b / torch.sqrt(1.0 - q ** 2) * torch.atan(torch.sqrt(1.0 - q ** 2) * x / (psi + s))
    fy = b / torch.sqrt(1.0 - q ** 2) * torch.atanh(torch.sqrt(1.0 - q ** 2) * y / (psi + q ** 2 * s))
    return _rotate(fx, fy, -phi)


def _nfw_g(x):
    """tf/profiles/mass/nfw.py:33-52: gather/scatter on the x<1 / x>1 index sets;
    entries with x==1 keep the initial 1.0 (:38)."""
    shape = x.shape
    x = x.reshape(-1)
    x = torch.clamp(x, min=1e-6)
    a = torch.ones_like(x)
    m1, m2 = x < 1, x > 1
    x1, x2 = x[m1], x[m2]
    a = a.masked_scatter(m1, torch.log(x1 / 2.0) + 1 / torch.sqrt(1 - x1 ** 2) * torch.acosh(1.0 / x1))
    a = a.masked_scatter(m2, torch.log(x2 / 2.0) + 1 / torch.sqrt(x2 ** 2 - 1) * torch.acos(1.0 / x2))
    return a.reshape(shape)


def nfw_deriv(x, y, Rs, alpha_Rs, center_x, center_y):
    """tf/profiles/mass/nfw.py:15-31."""
    Rs, alpha_Rs, center_x, center_y = (_t(v, x) for v in (Rs, alpha_Rs, center_x, center_y))
    rho0 = alpha_Rs / (4.0 * Rs ** 2 * (1.0 - LN2))
    x, y = x - center_x, y - center_y
    R = torch.sqrt(x ** 2 + y ** 2)
    R = torch.clamp(R, min=1e-7)
    Rs = torch.clamp(Rs, min=1e-7)
    X = R / Rs
    X, _ = torch.broadcast_tensors(X, x)
    gx = _nfw_g(X)
    a = 4 * rho0 * Rs * gx / X ** 2
    return a * x, a * y


def nfw_ellipse_deriv(x, y, Rs, alpha_Rs, e1, e2, center_x, center_y):
    """tf/profiles/mass/nfw.py:108-133: NFW on coordinates stretched by sqrt(1 -+ e), e = |1-q^2|/(1+q^2)."""
    Rs, alpha_Rs, e1, e2, center_x, center_y = (_t(v, x) for v in (Rs, alpha_Rs, e1, e2, center_x, center_y))
    rho0 = alpha_Rs / (4.0 * Rs ** 2 * (1.0 - math.log(2.0)))
    phi = torch.atan2(e2, e1) / 2
    c = torch.clamp(torch.sqrt(e1 ** 2 + e2 ** 2), max=0.9999)
    q = (1 - c) / (1 + c)
    e = torch.abs(1 - q ** 2) / (1 + q ** 2)
    x, y = x - center_x, y - center_y
    x, y = _rotate(x, y, phi)
    x, y = x * torch.sqrt(1 - e), y * torch.sqrt(1 + e)
    R = torch.sqrt(x ** 2 + y ** 2)
    # nfwAlpha (nfw.py:23-31)
    Rc = torch.clamp(R, min=1e-7)
    Rsc = torch.clamp(Rs, min=1e-7)
    X = Rc / Rsc
    a = 4 * rho0 * Rsc * _nfw_g(X) / X ** 2
    fx, fy = a * x * torch.sqrt(1 - e), a * y * torch.sqrt(1 + e)
    return _rotate(fx, fy, -phi)


def _tnfw_F(x):
    """tnfw.py:43-62 (both branches evaluated on safe arguments, selected afterwards; F(1) = 1)."""
    lo = torch.clamp(x, max=1 - 1e-15)
    hi = torch.clamp(x, min=1 + 1e-15)
    f1 = torch.atanh(torch.sqrt(1 - lo ** 2)) / torch.sqrt(1 - lo ** 2)
    f2 = torch.atan(torch.sqrt(hi ** 2 - 1)) / torch.sqrt(hi ** 2 - 1)
    return torch.where(x < 1, f1, torch.where(x > 1, f2, torch.ones_like(x)))


def tnfw_deriv(x, y, Rs, alpha_Rs, r_trunc, center_x, center_y):
    """tf/profiles/mass/tnfw.py:17-41."""
    Rs, alpha_Rs, r_trunc, center_x, center_y = (_t(v, x) for v in (Rs, alpha_Rs, r_trunc, center_x, center_y))
    rho0 = alpha_Rs / (4.0 * Rs ** 2 * (1.0 + math.log(0.5)))
    x, y = x - center_x, y - center_y
    R = torch.sqrt(x ** 2 + y ** 2)
    R = torch.maximum(R, 0.001 * Rs)
    X = R / Rs
    tau = r_trunc / Rs
    L = torch.log(X / (tau + torch.sqrt(tau ** 2 + X ** 2)))
    F = _tnfw_F(X)
    gx = (tau ** 2) / (tau ** 2 + 1) ** 2 * (
        (tau ** 2 + 1 + 2 * (X ** 2 - 1)) * F + tau * math.pi + (tau ** 2 - 1) * torch.log(tau)
        + torch.sqrt(tau ** 2 + X ** 2) * (-math.pi + L * (tau ** 2 - 1) / tau))
    a = 4 * rho0 * Rs * gx / X ** 2
    return a * x, a * y


def shear_deriv(x, y, gamma1, gamma2):
    """tf/profiles/mass/shear.py:14-16."""
    gamma1, gamma2 = _t(gamma1, x), _t(gamma2, x)
    return gamma1 * x + gamma2 * y, gamma2 * x - gamma1 * y


def sis_deriv(x, y, theta_E, center_x, center_y):
    """tf/profiles/mass/sis.py:12-17."""
    theta_E, center_x, center_y = (_t(v, x) for v in (theta_E, center_x, center_y))
    x, y = x - center_x, y - center_y
    R = torch.sqrt(x ** 2 + y ** 2)
    a = torch.where(R == 0, torch.zeros_like(R), theta_E / R)
    return a * x, a * y


# --------------------------------------------------------------------------
# dPIE family (tf/profiles/mass/piemd.py, piep.py) and ScalingRelation (scaling_relation.py)
# --------------------------------------------------------------------------
DPIE_R_MIN = 0.0001  # piemd.py:28,100


def _sort_ra_rs(r_core, r_cut):
    """piemd.py:52-60 / :191-199, statement by statement (note: after the first line r_core <= r_cut, so the
    second ``where`` never fires -- kept as written)."""
    r_core = torch.where(r_core < r_cut, r_core, r_cut)
    r_cut = torch.where(r_core > r_cut, r_core, r_cut)
    r_core = torch.clamp(r_core, min=DPIE_R_MIN)
    r_cut = torch.where(r_cut > r_core + DPIE_R_MIN, r_cut, r_cut + DPIE_R_MIN)
    return r_core, r_cut


def dpis_deriv(x, y, theta_E, r_core, r_cut, center_x, center_y):
    """piemd.py:33-49."""
    theta_E, r_core, r_cut, center_x, center_y = (_t(v, x) for v in (theta_E, r_core, r_cut, center_x, center_y))
    r_core, r_cut = _sort_ra_rs(r_core, r_cut)
    x, y = x - center_x, y - center_y
    r2 = x ** 2 + y ** 2
    scale = theta_E * r_cut / (r_cut - r_core)
    f_a20 = torch.sqrt(r2 + r_core ** 2) - r_core - torch.sqrt(r2 + r_cut ** 2) + r_cut
    alpha_r = scale / r2 * f_a20
    return alpha_r * x, alpha_r * y


def dpis_convergence(x, y, theta_E, r_core, r_cut, center_x=0, center_y=0):
    """piemd.py:85-94 (carries a factor (r_core + r_cut)/r_cut that the deflection does not have -- as written)."""
    theta_E, r_core, r_cut, center_x, center_y = (_t(v, x) for v in (theta_E, r_core, r_cut, center_x, center_y))
    r_core, r_cut = _sort_ra_rs(r_core, r_cut)
    x, y = x - center_x, y - center_y
    r = torch.clamp(torch.sqrt(x ** 2 + y ** 2), min=DPIE_R_MIN)
    scale = theta_E * r_cut / (r_cut - r_core)
    return scale / 2 * (r_core + r_cut) / r_cut * (
        1 / torch.sqrt(r_core ** 2 + r ** 2) - 1 / torch.sqrt(r_cut ** 2 + r ** 2))


def dpis_hessian(x, y, theta_E, r_core, r_cut, center_x, center_y):
    """piemd.py:62-83."""
    theta_E, r_core, r_cut, center_x, center_y = (_t(v, x) for v in (theta_E, r_core, r_cut, center_x, center_y))
    r_core, r_cut = _sort_ra_rs(r_core, r_cut)
    x, y = x - center_x, y - center_y
    r = torch.clamp(torch.sqrt(x ** 2 + y ** 2), min=DPIE_R_MIN)
    scale = theta_E * r_cut / (r_cut - r_core)
    gamma = scale / 2 * (
        2 * (1. / (r_core + torch.sqrt(r_core ** 2 + r ** 2)) - 1. / (r_cut + torch.sqrt(r_cut ** 2 + r ** 2)))
        - (1 / torch.sqrt(r_core ** 2 + r ** 2) - 1 / torch.sqrt(r_cut ** 2 + r ** 2)))
    kappa = scale / 2 * (r_core + r_cut) / r_cut * (
        1 / torch.sqrt(r_core ** 2 + r ** 2) - 1 / torch.sqrt(r_cut ** 2 + r ** 2))
    sin_imphi = -2 * x * y / r ** 2
    cos_imphi = (y ** 2 - x ** 2) / r ** 2
    gamma1 = cos_imphi * gamma
    gamma2 = sin_imphi * gamma
    return kappa + gamma1, gamma2, gamma2, kappa - gamma1


def _dpie_param_conv(e1, e2):
    """piemd.py:183-188."""
    phi = torch.atan2(e2, e1) / 2
    e = torch.clamp(torch.sqrt(e1 ** 2 + e2 ** 2), max=0.9999)
    q = (1 - e) / (1 + e)
    return e, q, phi


def _dpie_complex_deriv_dual(x, y, r_core, r_cut, e, q):
    """piemd.py:201-255 (Kassiola & Kovner 1993 eq. 4.1.2 for the two radii, ratio taken before the log)."""
    sqe = torch.sqrt(e)
    rem2 = x ** 2 / (1. + e) ** 2 + y ** 2 / (1. - e) ** 2
    zci_re = 0
    zci_im = -0.5 * (1. - e ** 2) / sqe
    znum_rc_re = q * x
    znum_rc_im = 2. * sqe * torch.sqrt(r_core ** 2 + rem2) - y / q
    zden_rc_re = x
    zden_rc_im = 2. * r_core * sqe - y
    znum_rcut_im = 2. * sqe * torch.sqrt(r_cut ** 2 + rem2) - y / q
    zden_rcut_im = 2. * r_cut * sqe - y
    aa = (znum_rc_re * zden_rc_re - znum_rc_im * zden_rcut_im)
    bb = (znum_rc_re * zden_rcut_im + znum_rc_im * zden_rc_re)
    cc = (znum_rc_re * zden_rc_re - zden_rc_im * znum_rcut_im)
    dd = (znum_rc_re * zden_rc_im + zden_rc_re * znum_rcut_im)
    norm = (cc ** 2 + dd ** 2)
    aaa = (aa * cc + bb * dd) / norm
    bbb = (bb * cc - aa * dd) / norm
    norm2 = aaa ** 2 + bbb ** 2
    zr_re = torch.log(torch.sqrt(norm2))
    zr_im = torch.atan2(bbb, aaa)
    zres_re = zci_re * zr_re - zci_im * zr_im
    zres_im = zci_im * zr_re + zci_re * zr_im
    return zres_re, zres_im


def dpie_deriv(x, y, theta_E, r_core, r_cut, e1, e2, center_x=0, center_y=0):
    """piemd.py:105-119."""
    theta_E, r_core, r_cut, e1, e2, center_x, center_y = (
        _t(v, x) for v in (theta_E, r_core, r_cut, e1, e2, center_x, center_y))
    e, q, phi = _dpie_param_conv(e1, e2)
    x, y = x - center_x, y - center_y
    x, y = _rotate(x, y, phi)
    r_core, r_cut = _sort_ra_rs(r_core, r_cut)
    scale = theta_E * r_cut / (r_cut - r_core)
    ax, ay = _dpie_complex_deriv_dual(x, y, r_core, r_cut, e, q)
    ax, ay = _rotate(ax, ay, -phi)
    return scale * ax, scale * ay


def _dpie_complex_hessian_single(x, y, r_w, e, q):
    """piemd.py:257-300."""
    sqe = torch.sqrt(e)
    qinv = 1. / q
    cxro = (1. + e) * (1. + e)
    cyro = (1. - e) * (1. - e)
    ci = 0.5 * (1. - e ** 2) / sqe
    wrem = torch.sqrt(r_w ** 2 + x ** 2 / cxro + y ** 2 / cyro)
    den1 = 2. * sqe * wrem - y * qinv
    den1 = q ** 2 * x ** 2 + den1 ** 2
    num2 = 2. * r_w * sqe - y
    den2 = x ** 2 + num2 ** 2
    didxre = ci * (q * (2. * sqe * x ** 2 / cxro / wrem - 2. * sqe * wrem + y * qinv) / den1 + num2 / den2)
    didyre = ci * ((2 * sqe * x * y * q / cyro / wrem - x) / den1 + x / den2)
    didyim = ci * ((2 * sqe * wrem * qinv - y * qinv ** 2 - 4 * e * y / cyro
                    + 2 * sqe * y ** 2 / cyro / wrem * qinv) / den1 - num2 / den2)
    return didxre, didyre, didyim


def _hessian_rotate(f_xx, f_xy, f_yx, f_yy, phi):
    """piemd.py:157-181."""
    cos_2phi = torch.cos(2 * phi)
    sin_2phi = torch.sin(2 * phi)
    a = 1 / 2 * (f_xx + f_yy)
    b = 1 / 2 * (f_xx - f_yy) * cos_2phi
    c = f_xy * sin_2phi
    d = f_xy * cos_2phi
    e = 1 / 2 * (f_xx - f_yy) * sin_2phi
    return a + b + c, d - e, d - e, a - b - c


def dpie_hessian(x, y, theta_E, r_core, r_cut, e1, e2, center_x=0, center_y=0):
    """piemd.py:121-138."""
    theta_E, r_core, r_cut, e1, e2, center_x, center_y = (
        _t(v, x) for v in (theta_E, r_core, r_cut, e1, e2, center_x, center_y))
    e, q, phi = _dpie_param_conv(e1, e2)
    x, y = x - center_x, y - center_y
    x, y = _rotate(x, y, phi)
    r_core, r_cut = _sort_ra_rs(r_core, r_cut)
    scale = theta_E * r_cut / (r_cut - r_core)
    a1, b1, c1 = _dpie_complex_hessian_single(x, y, r_core, e, q)
    a2, b2, c2 = _dpie_complex_hessian_single(x, y, r_cut, e, q)
    f_xx = scale * (a1 - a2)
    f_xy = f_yx = scale * (b1 - b2)
    f_yy = scale * (c1 - c2)
    return _hessian_rotate(f_xx, f_xy, f_yx, f_yy, -phi)


def dpie_convergence(x, y, theta_E, r_core, r_cut, e1, e2, center_x=0, center_y=0):
    """piemd.py:140-149."""
    theta_E, r_core, r_cut, e1, e2, center_x, center_y = (
        _t(v, x) for v in (theta_E, r_core, r_cut, e1, e2, center_x, center_y))
    e, q, phi = _dpie_param_conv(e1, e2)
    x, y = x - center_x, y - center_y
    x, y = _rotate(x, y, phi)
    r_core, r_cut = _sort_ra_rs(r_core, r_cut)
    scale = theta_E * r_cut / (r_cut - r_core)
    rem2 = x ** 2 / (1. + e) ** 2 + y ** 2 / (1. - e) ** 2
    return scale / 2 * (1 / torch.sqrt(rem2 + r_core ** 2) - 1 / torch.sqrt(rem2 + r_cut ** 2))


def dpiep_deriv(x, y, theta_E, Ra, Rs, e1, e2, center_x=0, center_y=0):
    """piep.py:31-55: the spherical dPIS evaluated on coordinates stretched by sqrt(1 -+ e)."""
    theta_E, Ra, Rs, e1, e2, center_x, center_y = (_t(v, x) for v in (theta_E, Ra, Rs, e1, e2, center_x, center_y))
    phi = torch.atan2(e2, e1) / 2
    c = torch.clamp(torch.sqrt(e1 ** 2 + e2 ** 2), max=0.9999)
    q = (1 - c) / (1 + c)
    e = torch.abs(1 - q ** 2) / (1 + q ** 2)
    x, y = x - center_x, y - center_y
    x, y = _rotate(x, y, phi)
    x, y = x * torch.sqrt(1 - e), y * torch.sqrt(1 + e)
    fx, fy = dpis_deriv(x, y, theta_E, Ra, Rs, 0.0, 0.0)
    fx = fx * torch.sqrt(1 - e)
    fy = fy * torch.sqrt(1 + e)
    return _rotate(fx, fy, -phi)


# ---- series-expansion accelerator (tf/series/series_profile.py, dpie_series.py, scaling_series.py) ----------------
def _derivative_tower(fn, r, order):
    """[fn(r), d fn/dr, ..., d^order fn/dr^order], elementwise in the broadcast variable ``r`` (nested forward-mode
    JVPs along the all-ones direction) -- what the sympy-generated deriv_0..deriv_5 of tf/series/profiles/dpie.py
    evaluate (generator: series_codegen/sympy_codegen.py:21-29, ``diff`` of the deflection w.r.t. the series variable)."""
    outs, f = [], fn
    for _ in range(order + 1):
        outs.append(f(r))
        f = (lambda g: (lambda rr: torch.func.jvp(g, (rr,), (torch.ones_like(rr),))[1]))(f)
    return outs


def dpie_series_precompute(order, x, y, theta_E, r_core, r_cut, e1, e2, center_x, center_y):
    """DPIESeries.precompute_deriv (dpie_series.py:19-33): derivatives w.r.t. r_cut of the dPIE deflection per unit
    theta_E (series_codegen/profiles/dpie.py:18-58: ``scale = r_cut/(r_cut - r_core)``, no radius sort, ellipticity
    NOT clamped, dpie_series.py:52-56), rotated back; stacked on a trailing axis of length order + 1."""
    r_core, r_cut, e1, e2, center_x, center_y = (_t(v, x) for v in (r_core, r_cut, e1, e2, center_x, center_y))
    phi = torch.atan2(e2, e1) / 2
    e = torch.sqrt(e1 ** 2 + e2 ** 2)
    q = (1 - e) / (1 + e)
    xs, ys = x - center_x, y - center_y
    xr, yr = _rotate(xs, ys, phi)

    def unit(rc):
        ax, ay = _dpie_complex_deriv_dual(xr, yr, r_core, rc, e, q)
        sc = rc / (rc - r_core)
        return torch.stack(_rotate(sc * ax, sc * ay, -phi))

    tower = _derivative_tower(unit, r_cut * torch.ones_like(xr * r_cut), order)
    f = torch.stack(tower, dim=-1)  # (2, ..., order+1)
    return f[0], f[1]


def _dpie_series_hessian_single(x, y, r_w, e, q):
    """series_codegen/profiles/dpie.py:73-105 ``complex_hessian_single``: Lenstool's closed-form second derivatives of
    one PIEMD term of radius ``r_w`` in the rotated frame (f_xy is ``didyre`` as written)."""
    sqe = torch.sqrt(e)
    qinv = 1.0 / q
    cxro, cyro = (1.0 + e) * (1.0 + e), (1.0 - e) * (1.0 - e)
    ci = 0.5 * (1.0 - e ** 2) / sqe
    wrem = torch.sqrt(r_w ** 2 + x ** 2 / cxro + y ** 2 / cyro)
    den1 = 2.0 * sqe * wrem - y * qinv
    den1 = q ** 2 * x ** 2 + den1 ** 2
    num2 = 2.0 * r_w * sqe - y
    den2 = x ** 2 + num2 ** 2
    didxre = ci * (q * (2.0 * sqe * x ** 2 / cxro / wrem - 2.0 * sqe * wrem + y * qinv) / den1 + num2 / den2)
    didyre = ci * ((2 * sqe * x * y * q / cyro / wrem - x) / den1 + x / den2)
    didyim = ci * ((2 * sqe * wrem * qinv - y * qinv ** 2 - 4 * e * y / cyro
                    + 2 * sqe * y ** 2 / cyro / wrem * qinv) / den1 - num2 / den2)
    return didxre, didyre, didyim


def dpie_series_precompute_hessian(order, x, y, theta_E, r_core, r_cut, e1, e2, center_x, center_y):
    """DPIESeries.precompute_hessian (dpie_series.py:35-49): derivatives w.r.t. r_cut of
    ``r_cut/(r_cut - r_core) * (H(r_core) - H(r_cut))`` (series_codegen/profiles/dpie.py:60-70), each order rotated
    back with ``_hessian_rotate(-phi)`` (dpie_series.py:64-88); trailing axis of length order + 1."""
    r_core, r_cut, e1, e2, center_x, center_y = (_t(v, x) for v in (r_core, r_cut, e1, e2, center_x, center_y))
    phi = torch.atan2(e2, e1) / 2
    e = torch.sqrt(e1 ** 2 + e2 ** 2)
    q = (1 - e) / (1 + e)
    xs, ys = x - center_x, y - center_y
    xr, yr = _rotate(xs, ys, phi)
    c2, s2 = torch.cos(2 * -phi), torch.sin(2 * -phi)

    def unit(rc):
        core = _dpie_series_hessian_single(xr, yr, r_core, e, q)
        cut = _dpie_series_hessian_single(xr, yr, rc, e, q)
        sc = rc / (rc - r_core)
        fxx, fxy, fyy = (sc * (a - b) for a, b in zip(core, cut))
        a, b, c, d, ee = 0.5 * (fxx + fyy), 0.5 * (fxx - fyy) * c2, fxy * s2, fxy * c2, 0.5 * (fxx - fyy) * s2
        return torch.stack((a + b + c, d - ee, a - b - c))

    tower = _derivative_tower(unit, r_cut * torch.ones_like(xr * r_cut), order)
    f = torch.stack(tower, dim=-1)
    return f[0], f[1], f[2]


def scaled_series_precompute_hessian(profile, order, x, y, **scales):
    """ScalingRelationSeries.precompute_hessian (scaling_series.py:37-54), same weights as the deflection series.
    (The reference returns ``f_xx, f_xy, f_xy, f_yy`` into MassSeries.set_hessian's 3-way unpacking,
    series_profile.py:65, which raises; the three distinct fields are restated here.)"""
    scales = dict(scales)
    scales[profile.amplitude_param] = 1.0
    kw = _scaled_galaxy_kwargs(profile, scales, x)
    un = scaled_unscaled_factors(profile)
    n = torch.arange(order + 1, dtype=x.dtype)
    pre = un[profile.amplitude_param].to(x.dtype)[:, None] * un[profile.series_param].to(x.dtype)[:, None] ** n
    f = dpie_series_precompute_hessian(order, x.unsqueeze(-1), y.unsqueeze(-1), **kw)
    return tuple((pre * c).sum(-2) for c in f)


def series_hessian(coefs, order, var, var0, scale):
    """MassSeries.hessian (series_profile.py:83-89) on ``coefs = (f_xx, f_xy, f_yy)``: returns the 4-tuple."""
    n = torch.arange(order + 1, dtype=coefs[0].dtype)
    fact = torch.exp(torch.lgamma(n + 1))
    powers = (_t(var, coefs[0]).unsqueeze(-1) - var0) ** n
    scale = _t(scale, coefs[0])
    fxx, fxy, fyy = (scale * (c * powers / fact).sum(-1) for c in coefs)
    return fxx, fxy, fxy, fyy


def scaled_series_precompute(profile, order, x, y, **scales):
    """ScalingRelationSeries.precompute_deriv (scaling_series.py:19-35): amplitude scale set to 1, every galaxy's
    derivative tower weighted by ``(L/L*)^p_amp * ((L/L*)^p_series)^n`` and summed over the catalogue."""
    scales = dict(scales)
    scales[profile.amplitude_param] = 1.0
    kw = _scaled_galaxy_kwargs(profile, scales, x)
    un = scaled_unscaled_factors(profile)
    n = torch.arange(order + 1, dtype=x.dtype)
    pre = un[profile.amplitude_param].to(x.dtype)[:, None] * un[profile.series_param].to(x.dtype)[:, None] ** n
    fx, fy = dpie_series_precompute(order, x.unsqueeze(-1), y.unsqueeze(-1), **kw)  # (..., G, order+1)
    return (pre * fx).sum(-2), (pre * fy).sum(-2)


def series_deriv(coefs_x, coefs_y, order, var, var0, scale):
    """MassSeries.deriv / _evaluate_series (series_profile.py:76-95): ``scale * sum_n f_n (var - var0)^n / n!``."""
    n = torch.arange(order + 1, dtype=coefs_x.dtype)
    fact = torch.exp(torch.lgamma(n + 1))
    powers = (_t(var, coefs_x).unsqueeze(-1) - var0) ** n
    scale = _t(scale, coefs_x)
    return scale * (coefs_x * powers / fact).sum(-1), scale * (coefs_y * powers / fact).sum(-1)


_SCALED_BASE = {"dPIS": (dpis_deriv, dpis_hessian), "dPIE": (dpie_deriv, dpie_hessian)}


def scaled_unscaled_factors(profile):
    """(L/L*)^power per scaling parameter, in float32 like the reference's constants (scaling_relation.py:27-30,53)."""
    lum = torch.as_tensor(np.asarray(profile.galaxy_cat["lum"], dtype=np.float32))
    lum_star = torch.tensor(float(profile.lum_star), dtype=torch.float32)
    return {k: (lum / lum_star) ** torch.tensor(float(profile.power[k]), dtype=torch.float32)
            for k in profile.scaling_params}


def _scaled_galaxy_kwargs(profile, scales, like):
    """scaling_relation.py:27-59: per-galaxy parameters = (L/L*)^power * scale for the scaling parameters, catalogue
    columns for the rest; a trailing galaxy axis is appended (x, y, b) -> (x, y, b, g), :64."""
    kw = {k: u.to(like.dtype) * _t(scales[k], like).unsqueeze(-1) for k, u in scaled_unscaled_factors(profile).items()}
    for k in profile.not_scaling_params:
        kw[k] = torch.as_tensor(np.asarray(profile.galaxy_cat[k], dtype=np.float32)).to(like.dtype)
    return kw


def scaled_deriv(profile, x, y, **scales):
    """scaling_relation.py:61-70 (chunking only bounds memory; the sum is the same)."""
    base = profile.profile
    kw = _scaled_galaxy_kwargs(profile, scales, x)
    fx, fy = mass_deriv(base, x.unsqueeze(-1), y.unsqueeze(-1), **kw)
    return fx.sum(-1), fy.sum(-1)


def scaled_hessian(profile, x, y, **scales):
    """scaling_relation.py:72-83."""
    base = profile.profile
    kw = _scaled_galaxy_kwargs(profile, scales, x)
    h = mass_hessian(base, x.unsqueeze(-1), y.unsqueeze(-1), **kw)
    return tuple(t.sum(-1) for t in h)


# --------------------------------------------------------------------------
# light profiles
# --------------------------------------------------------------------------
def sersic_distance(x, y, cx, cy, e1=None, e2=None):
    """tf/profiles/light/sersic.py:37-63."""
    cx, cy = _t(cx, x), _t(cy, x)
    e1 = torch.zeros_like(cx) if e1 is None else _t(e1, x)
    e2 = torch.zeros_like(cx) if e2 is None else _t(e2, x)
    phi = torch.atan2(e2, e1) / 2
    c = torch.clamp(torch.sqrt(e1 ** 2 + e2 ** 2), max=0.9999)
    q = (1 - c) / (1 + c)
    dx, dy = x - cx, y - cy
    cos_phi, sin_phi = torch.cos(phi), torch.sin(phi)
    xt1 = (cos_phi * dx + sin_phi * dy) * torch.sqrt(q)
    xt2 = (-sin_phi * dx + cos_phi * dy) / torch.sqrt(q)
    return torch.sqrt(xt1 ** 2 + xt2 ** 2)


def sersic_light(x, y, R_sersic, n_sersic, center_x, center_y, Ie, e1=None, e2=None):
    """tf/profiles/light/sersic.py:29-35 (Sersic) and :74-80 (SersicEllipse)."""
    R_sersic, n_sersic, Ie = _t(R_sersic, x), _t(n_sersic, x), _t(Ie, x)
    R = sersic_distance(x, y, center_x, center_y, e1, e2)
    bn = 1.9992 * n_sersic - 0.3271
    return Ie * torch.exp(-bn * ((R / R_sersic) ** (1 / n_sersic) - 1.0))


def core_sersic_light(x, y, R_sersic, n_sersic, Rb, alpha, gamma, e1, e2, center_x, center_y, Ie):
    """tf/profiles/light/sersic.py:98-131, operator precedence as written: ``R_sersic ** alpha ** 1.0`` is
    ``R_sersic ** alpha`` and the following ``/ (alpha * n_sersic)`` divides."""
    R_sersic, n_sersic, Rb, alpha, gamma, Ie = (_t(v, x) for v in (R_sersic, n_sersic, Rb, alpha, gamma, Ie))
    R = sersic_distance(x, y, center_x, center_y, e1, e2)
    bn = 1.9992 * n_sersic - 0.3271
    return (Ie * (1 + (Rb / R) ** alpha) ** (gamma / alpha)
            * torch.exp(-bn * ((R ** alpha + Rb ** alpha) / R_sersic ** alpha ** 1.0 / (alpha * n_sersic)) - 1.0))


def shapelet_index_order(n_max):
    """tf/profiles/light/shapelets.py:26-46: (n1,n2) = (0,0),(1,0),(0,1),(2,0),(1,1),(0,2),..."""
    n_layers = int((n_max + 1) * (n_max + 2) / 2)
    N1, N2 = [], []
    n1 = n2 = 0
    for _ in range(n_layers):
        N1.append(n1)
        N2.append(n2)
        if n1 == 0:
            n1 = n2 + 1
            n2 = 0
        else:
            n1 -= 1
            n2 += 1
    return N1, N2


def shapelet_amp_names(n_max):
    """tf/profiles/light/shapelets.py:32-36: amp{i:0w}, w=len(str(n_layers))."""
    n_layers = int((n_max + 1) * (n_max + 2) / 2)
    w = len(str(n_layers))
    return [f"amp{str(i).zfill(w)}" for i in range(n_layers)]


def phi_n_f64(n, x):
    """lenstronomy (README pins ==1.9.3) ``Shapelets.phi_n``: H_n(x) e^{-x^2/2} /
    sqrt(2^n sqrt(pi) n!) -- third-party algorithm restated from its published
    definition (Refregier 2003 eq. 1-2); used at shapelets.py:39-40 to build tables."""
    x = np.asarray(x, dtype=np.float64)
    coef = np.zeros(n + 1)
    coef[n] = 1.0
    pref = 1.0 / np.sqrt(2.0 ** n * np.sqrt(np.pi) * math.factorial(n))
    return pref * np.polynomial.hermite.hermval(x, coef) * np.exp(-x ** 2 / 2.0)


def shapelet_tables(n_max, n_nodes=6000):
    """shapelets.py:39-40,50-51: phi_n(linspace(-5,5,6000)) stored as float32, (n_max+1, 6000)."""
    grid = np.linspace(-5.0, 5.0, n_nodes)
    return np.stack([phi_n_f64(n, grid) for n in range(n_max + 1)]).astype(np.float32)


def interp_regular_1d_grid(x, x_ref_min, x_ref_max, y_ref):
    """tensorflow-probability >=0.19 ``tfp.math.interp_regular_1d_grid`` (setup.py:44),
    restated from its published algorithm (linear, fill 0 below/above) -- third-party,
    "parity unpinned".  y_ref: (K, ny); x: any shape -> (K, *x.shape)."""
    ny = y_ref.shape[-1]
    idx_unclipped = (x - x_ref_min) / (x_ref_max - x_ref_min) * (ny - 1)
    idx = torch.clamp(idx_unclipped, 0, ny - 1)
    below = torch.floor(idx)
    above = torch.clamp(below + 1, max=ny - 1)
    below = torch.clamp(above - 1, min=0)
    tt = idx - below
    yb = y_ref[:, below.long().reshape(-1)].reshape(y_ref.shape[0], *x.shape)
    ya = y_ref[:, above.long().reshape(-1)].reshape(y_ref.shape[0], *x.shape)
    y = tt * ya + (1 - tt) * yb
    zero = torch.zeros_like(y)
    y = torch.where(idx_unclipped < 0, zero, y)
    y = torch.where(idx_unclipped > ny - 1, zero, y)
    return y


_TABLE_CACHE: Dict = {}


def shapelets_light(x, y, center_x, center_y, beta, amps, n_max, interpolate=True):
    """tf/profiles/light/shapelets.py:53-85.  ``amps``: list of n_layers tensors (B,)
    in amp-name order (== tf.nest.flatten of the **amp kwargs, keys sorted); ``amps=None`` is the
    ``use_lstsq=True`` branch (:61-62,71-72): the (n_layers, N, B) basis images themselves."""
    center_x, center_y, beta = _t(center_x, x), _t(center_y, x), _t(beta, x)
    N1, N2 = shapelet_index_order(n_max)
    basis_only = amps is None
    A = None if basis_only else torch.stack(
        [_t(a, x) * torch.ones(x.shape[-1], dtype=x.dtype) if _t(a, x).dim() == 0 else _t(a, x)
         for a in amps])  # (n_layers, B)
    u = (x - center_x) / beta
    v = (y - center_y) / beta
    if interpolate:
        key = (n_max, x.dtype)
        if key not in _TABLE_CACHE:
            tab = torch.from_numpy(shapelet_tables(n_max)).to(x.dtype)  # f32 values (shapelets.py:50-51)
            _TABLE_CACHE[key] = tab
        tab = _TABLE_CACHE[key]
        X = interp_regular_1d_grid(u, -5.0, 5.0, tab)  # (n_max+1, N, B)
        Y = interp_regular_1d_grid(v, -5.0, 5.0, tab)
        ret = X[N1] * Y[N2]  # (n_layers, N, B) -- the reference interpolates 66 duplicated rows
        if basis_only:
            return ret
        return torch.einsum('inj,ij->nj', ret, A)
    # direct mode :66-85
    herm = [torch.ones_like(u), 2 * u]
    hermv = [torch.ones_like(v), 2 * v]
    for i in range(2, n_max + 1):
        herm.append(2 * (u * herm[-1] - (i - 1) * herm[-2]))
        hermv.append(2 * (v * hermv[-1] - (i - 1) * hermv[-2]))
    N = torch.arange(0, n_max + 1, dtype=x.dtype)
    pref = 1.0 / torch.sqrt(2 ** N * math.sqrt(math.pi) * torch.exp(torch.lgamma(N + 1)))
    XX = torch.stack(herm[: n_max + 1]) * pref[:, None, None]
    YY = torch.stack(hermv[: n_max + 1]) * pref[:, None, None]
    fac = torch.exp(-(u ** 2 + v ** 2) / 2)
    if basis_only:
        return fac * (XX[N1] * YY[N2])
    return fac * torch.einsum('ij,inj->nj', A, XX[N1] * YY[N2])


# --------------------------------------------------------------------------
# dispatch by the reference's profile names
# --------------------------------------------------------------------------
def mass_deriv(profile, x, y, **kw):
    name = profile.name
    if name == "EPL":
        return epl_deriv(x, y, niter_cap=getattr(profile, "niter", 50), **kw)
    if name == "SIE":
        return sie_deriv(x, y, **kw)
    if name == "NFW":
        return nfw_deriv(x, y, **kw)
    if name == "SHEAR":
        return shear_deriv(x, y, **kw)
    if name == "SIS":
        return sis_deriv(x, y, **kw)
    if name == "NFW_ELLIPSE":
        return nfw_ellipse_deriv(x, y, **kw)
    if name == "TNFW":
        return tnfw_deriv(x, y, **kw)
    if name == "dPIS":
        return dpis_deriv(x, y, **kw)
    if name == "dPIE" and "Ra" in profile.params:  # piep.py:22 reuses the name "dPIE"
        return dpiep_deriv(x, y, **kw)
    if name == "dPIE":
        return dpie_deriv(x, y, **kw)
    if name.startswith("Scaled-SeriesExpansion") or name.startswith("SeriesExpansion"):
        # MassSeries.deriv ignores (x, y) and reads the field precomputed on ITS grid (series_profile.py:76-81)
        cx, cy = profile._oracle_coefs(x.dtype)
        return series_deriv(cx, cy, profile.order, kw[profile.series_param], profile.series_var_0,
                            kw[profile.amplitude_param])
    if name.startswith("Scaled-"):
        return scaled_deriv(profile, x, y, **kw)
    raise NotImplementedError(name)


def mass_hessian(profile, x, y, **kw):
    """``lens.hessian`` as the reference resolves it: the analytic overrides of dPIS / dPIE (piemd.py:62-83,
    :121-138) and of ScalingRelation (sum of the base profile's), autodiff of ``deriv`` otherwise
    (tf/profile.py:9-27; the NFW / Shear / SIS overrides equal that derivative)."""
    name = profile.name
    if name == "dPIS":
        return dpis_hessian(x, y, **kw)
    if name == "dPIE" and "Ra" not in profile.params:
        return dpie_hessian(x, y, **kw)
    if name.startswith("Scaled-"):
        return scaled_hessian(profile, x, y, **kw)
    if not x.requires_grad:
        x = x.clone().requires_grad_(True)
        y = y.clone().requires_grad_(True)
    fx, fy = mass_deriv(profile, x, y, **kw)
    a, b = torch.autograd.grad(fx.sum(), [x, y], create_graph=True)
    cc, d = torch.autograd.grad(fy.sum(), [x, y], create_graph=True)
    return a, b, cc, d


def light_basis(profile, x, y, **kw):
    """``light`` of a ``use_lstsq=True`` profile: (depth, N, B) basis images with unit amplitude
    (sersic.py:30-34 ``Ie = ones``; ``ret[tf.newaxis]``; shapelets.py:61-62,71-72)."""
    name = profile.name
    if name in ("SERSIC", "SERSIC_ELLIPSE"):
        kw = {k: v for k, v in kw.items() if k != "Ie"}
        return sersic_light(x, y, Ie=1.0, **kw)[None]
    if name == "CORE_SERSIC":
        kw = {k: v for k, v in kw.items() if k != "Ie"}
        return core_sersic_light(x, y, Ie=1.0, **kw)[None]
    if name == "SHAPELETS":
        return shapelets_light(x, y, kw["center_x"], kw["center_y"], kw["beta"], None, profile.n_max,
                               getattr(profile, "interpolate", True))
    raise NotImplementedError(name)


def light_eval(profile, x, y, **kw):
    name = profile.name
    if name in ("SERSIC", "SERSIC_ELLIPSE"):
        return sersic_light(x, y, **kw)
    if name == "CORE_SERSIC":
        return core_sersic_light(x, y, **kw)
    if name == "SHAPELETS":
        names = shapelet_amp_names(profile.n_max)
        amps = [kw[k] for k in sorted(k for k in kw if k.startswith("amp"))]
        assert len(amps) == len(names)
        return shapelets_light(x, y, kw["center_x"], kw["center_y"], kw["beta"], amps,
                               profile.n_max, getattr(profile, "interpolate", True))
    raise NotImplementedError(name)


# --------------------------------------------------------------------------
# PSF helper (third party: lenstronomy; restated from its published source, "parity unpinned" for supersample>1)
# --------------------------------------------------------------------------
def _re_size_array(x_in, y_in, values, x_out, y_out):
    """lenstronomy ``image_util.re_size_array``: ``scipy.interpolate.interp2d(x_in, y_in, values, kind='linear')(x_out, y_out)``
    -- bilinear on the regular grid, nearest-edge outside it; ``values[j, i]`` sits at ``(x_in[i], y_in[j])``."""
    tmp = np.stack([np.interp(x_out, x_in, row) for row in values])          # along x for every input row
    return np.stack([np.interp(y_out, y_in, tmp[:, i]) for i in range(tmp.shape[1])], axis=1)


def _averaging_even_kernel(kernel_high_res, subgrid_res):
    """lenstronomy ``kernel_util.averaging_even_kernel``: an odd-sized fine kernel re-binned at an even factor, centred; fine
    cells fully inside a coarse pixel add to it, cells on a border are split in halves (quarters at corners)."""
    n_high_in = len(kernel_high_res)
    n_low = int(round(n_high_in / subgrid_res + 0.5))
    if n_low % 2 == 0:
        n_low += 1
    n_high = int(n_low * subgrid_res - 1)
    if n_high == n_high_in:
        edges = kernel_high_res
    else:
        i0 = int((n_high - n_high_in) / 2)
        edges = np.zeros((n_high, n_high))
        edges[i0:-i0, i0:-i0] = kernel_high_res
    low = np.zeros((n_low, n_low))
    for i in range(subgrid_res - 1):
        for j in range(subgrid_res - 1):
            low += edges[i::subgrid_res, j::subgrid_res]
    i = subgrid_res - 1
    for j in range(subgrid_res - 1):
        low[1:, :] += edges[i::subgrid_res, j::subgrid_res] / 2
        low[:-1, :] += edges[i::subgrid_res, j::subgrid_res] / 2
    j = subgrid_res - 1
    for i in range(subgrid_res - 1):
        low[:, 1:] += edges[i::subgrid_res, j::subgrid_res] / 2
        low[:, :-1] += edges[i::subgrid_res, j::subgrid_res] / 2
    corner = edges[subgrid_res - 1::subgrid_res, subgrid_res - 1::subgrid_res]
    low[1:, 1:] += corner / 4
    low[:-1, 1:] += corner / 4
    low[1:, :-1] += corner / 4
    low[:-1, :-1] += corner / 4
    return low


def subgrid_kernel(kernel, subgrid_res, odd=False, num_iter=100):
    """lenstronomy ``Util.kernel_util.subgrid_kernel`` (used at tf/simulator.py:62-65 with ``odd=True``), THIRD PARTY: not under
    /root/reference and not installed here, so this is the published algorithm (lenstronomy 1.9.x) restated step by step --
    **parity unpinned**.  Interpolate onto the finer grid, normalise, then iterate ``num_iter`` times: re-bin to the input
    pixel scale, correct the working kernel by the mismatch, re-interpolate, normalise.  Odd ``subgrid_res`` (round 3, after the
    advisor's finding): the re-binned proposal of the loop is the plain block MEAN (not re-normalised), and the routine ends by
    subtracting the residual mismatch spread over each coarse pixel's block before the last normalisation."""
    subgrid_res = int(subgrid_res)
    kernel = np.asarray(kernel, dtype=np.float64)
    if subgrid_res == 1:
        return kernel
    nx, ny = kernel.shape
    x_in = np.linspace(1. / nx / 2, 1 - 1. / nx / 2, nx)
    y_in = np.linspace(1. / nx / 2, 1 - 1. / nx / 2, ny)  # lenstronomy writes d_y = 1 / nx as well (square kernels)
    nx_new, ny_new = nx * subgrid_res, ny * subgrid_res
    if odd:
        if nx_new % 2 == 0:
            nx_new -= 1
        if ny_new % 2 == 0:
            ny_new -= 1
    x_out = np.linspace(1. / nx_new / 2., 1 - 1. / nx_new / 2., nx_new)
    y_out = np.linspace(1. / ny_new / 2., 1 - 1. / ny_new / 2., ny_new)
    kernel_input = kernel.copy()
    kernel_subgrid = _re_size_array(x_in, y_in, kernel_input, x_out, y_out)
    kernel_subgrid = kernel_subgrid / kernel_subgrid.sum()
    for _ in range(max(num_iter, 1)):
        if subgrid_res % 2 == 0:
            kernel_pixel = _averaging_even_kernel(kernel_subgrid, subgrid_res)  # a SUM with shared border cells: unit sum kept
        else:  # util.averaging(grid, numGrid, numPix): block MEANS -- the proposal carries 1 / subgrid_res^2, as lenstronomy leaves it
            kernel_pixel = kernel_subgrid.reshape(nx, nx_new // nx, nx, nx_new // nx).mean(3).mean(1)
        delta = kernel - kernel_pixel
        temp_kernel = kernel_input + delta
        kernel_subgrid = _re_size_array(x_in, y_in, temp_kernel, x_out, y_out)
        kernel_subgrid = kernel_subgrid / kernel_subgrid.sum()
        kernel_input = temp_kernel
    if subgrid_res % 2 == 0:
        return kernel_subgrid
    # odd subgrid_res: "whatever has not been matched is added to zeroth order (in squares of the undersampled PSF)"
    kernel_pixel = kernel_subgrid.reshape(nx, nx_new // nx, nx, nx_new // nx).mean(3).mean(1)
    kernel_pixel = kernel_pixel / kernel_pixel.sum()
    delta_kernel = kernel_pixel - kernel / kernel.sum()
    delta_kernel_sub = np.kron(delta_kernel, np.ones((subgrid_res, subgrid_res))) / subgrid_res ** 2
    out = kernel_subgrid - delta_kernel_sub
    return out / out.sum()


# --------------------------------------------------------------------------
# simulator (tf/simulator.py)
# --------------------------------------------------------------------------
class RefSimulator:
    """tf/simulator.py:13-156 (``__init__``, ``beta``, ``simulate``)."""

    def __init__(self, phys_model, sim_config, bs, dtype=torch.float64, supersampled_kernel=None):
        self.phys_model = phys_model
        self.sim_config = sim_config
        self.bs = bs
        self.dtype = dtype
        self.supersample = int(sim_config.supersample)
        self.wcs, self.region, img_region, img_X, img_Y = build_grid(sim_config)
        self.img_region = torch.from_numpy(img_region).to(dtype)
        # :46-51 -- the grid replicated per batch element, (N, bs)
        self.img_X = torch.from_numpy(img_X).to(dtype)[:, None].repeat(1, bs)
        self.img_Y = torch.from_numpy(img_Y).to(dtype)[:, None].repeat(1, bs)
        self.conversion_factor = conversion_factor(sim_config)
        self.flat_kernel = None
        k = supersampled_kernel
        if k is None and sim_config.kernel is not None:
            k = subgrid_kernel(sim_config.kernel, self.supersample, odd=True)
        if k is not None:
            # :62-70 kernel[::-1, ::-1] then tf.nn.conv2d (a cross-correlation) => true convolution
            self.flat_kernel = torch.from_numpy(np.ascontiguousarray(np.asarray(k)[::-1, ::-1]).astype(np.float32)).to(dtype)

    def _consts(self, attr, n):
        c = getattr(self.phys_model, attr, None)
        if c is None:
            return [dict() for _ in range(n)]
        return [{k: torch.as_tensor(np.asarray(v, dtype=np.float32)).to(self.dtype) for k, v in d.items()} for d in c]

    def beta(self, x, y, lens_params: List[Dict]):
        """:72-78."""
        beta_x, beta_y = x, y
        consts = self._consts("lenses_constants", len(self.phys_model.lenses))
        for lens, p, c in zip(self.phys_model.lenses, lens_params, consts):
            f_xi, f_yi = mass_deriv(lens, x, y, **p, **c)
            beta_x, beta_y = beta_x - f_xi, beta_y - f_yi
        return beta_x, beta_y

    def simulate(self, params, no_deflection=False):
        """:109-156."""
        pm = self.phys_model
        lens_params = params.get('lens_mass', [{} for _ in pm.lenses])
        lens_light_params = params.get('lens_light', [{} for _ in pm.lens_light])
        source_light_params = params.get('source_light', [{} for _ in pm.source_light])
        beta_x, beta_y = self.beta(self.img_X, self.img_Y, lens_params)
        if no_deflection:
            beta_x, beta_y = self.img_X, self.img_Y
        Hs, Ws = self.wcs.n_x * self.supersample, self.wcs.n_y * self.supersample
        img = torch.zeros((Hs, Ws, self.bs), dtype=self.dtype)
        rr, cc = torch.from_numpy(self.region[:, 0]), torch.from_numpy(self.region[:, 1])
        for lm, p, c in zip(pm.lens_light, lens_light_params, self._consts("lens_light_constants", len(pm.lens_light))):
            img = img.index_put((rr, cc), light_eval(lm, self.img_X, self.img_Y, **p, **c), accumulate=True)
        for lm, p, c in zip(pm.source_light, source_light_params, self._consts("source_light_constants", len(pm.source_light))):
            img = img.index_put((rr, cc), light_eval(lm, beta_x, beta_y, **p, **c), accumulate=True)
        img = torch.where(torch.isnan(img), torch.zeros_like(img), img)  # :140
        img = img.permute(2, 0, 1)  # :141
        ret = img[:, None]
        if self.flat_kernel is not None:  # :145-147 conv2d SAME, stride 1
            kh, kw = self.flat_kernel.shape
            # TF 'SAME' pads (k-1)//2 before and k//2 after (extra goes to the end)
            ret = torch.nn.functional.pad(ret, ((kw - 1) // 2, kw // 2, (kh - 1) // 2, kh // 2))
            ret = torch.nn.functional.conv2d(ret, self.flat_kernel[None, None])
        if self.supersample != 1:  # :149-155
            ret = torch.nn.functional.avg_pool2d(ret, kernel_size=self.supersample, stride=self.supersample)
        return torch.squeeze(ret) * self.conversion_factor  # :156


def lstsq_simulate(simulator: RefSimulator, params, observed_image, err_map, return_stacked=False,
                   return_coeffs=False, no_deflection=False):
    """tf/simulator.py:158-240: basis stack (depth channels, NaN -> 0, depthwise PSF, pooling), then
    ``coeffs = pinv(X^T X, rcond=1e-6) X^T Y`` with ``X = stack / err_map``, ``Y = obs / err_map``.
    (The reference scatters into a zero-length first axis, :183-193 -- a shape slip; restated as the
    (depth, H, W, bs) stack it evidently builds and :199-205 reshapes.)"""
    sim, pm, dt = simulator, simulator.phys_model, simulator.dtype
    lens_params = params.get('lens_mass', [{} for _ in pm.lenses])
    lens_light_params = params.get('lens_light', [{} for _ in pm.lens_light])
    source_light_params = params.get('source_light', [{} for _ in pm.source_light])
    beta_x, beta_y = sim.beta(sim.img_X, sim.img_Y, lens_params)
    if no_deflection:
        beta_x, beta_y = sim.img_X, sim.img_Y
    Hs, Ws = sim.wcs.n_x * sim.supersample, sim.wcs.n_y * sim.supersample
    rr, cc = torch.from_numpy(sim.region[:, 0]), torch.from_numpy(sim.region[:, 1])
    chans = []
    for lm, p, c in zip(pm.lens_light, lens_light_params, sim._consts("lens_light_constants", len(pm.lens_light))):
        chans.append(light_basis(lm, sim.img_X, sim.img_Y, **p, **c))
    for lm, p, c in zip(pm.source_light, source_light_params, sim._consts("source_light_constants", len(pm.source_light))):
        chans.append(light_basis(lm, beta_x, beta_y, **p, **c))
    flat = torch.cat(chans, dim=0)  # (depth, N, bs)
    depth = flat.shape[0]
    img = torch.zeros((depth, Hs, Ws, sim.bs), dtype=dt)
    img[:, rr, cc, :] = flat
    img = torch.where(torch.isnan(img), torch.zeros_like(img), img)
    ret = img.permute(3, 0, 1, 2)  # (bs, depth, Hs, Ws) -- channels-first for torch's conv
    if sim.flat_kernel is not None:  # depthwise_conv2d SAME (:209-215)
        kh, kw = sim.flat_kernel.shape
        ret = torch.nn.functional.pad(ret, ((kw - 1) // 2, kw // 2, (kh - 1) // 2, kh // 2))
        ret = torch.nn.functional.conv2d(ret, sim.flat_kernel[None, None].repeat(depth, 1, 1, 1), groups=depth)
    if sim.supersample != 1:
        ret = torch.nn.functional.avg_pool2d(ret, kernel_size=sim.supersample, stride=sim.supersample)
    ret = torch.where(torch.isnan(ret), torch.zeros_like(ret), ret)
    if return_stacked:
        return ret.permute(0, 2, 3, 1)  # the reference's (bs, H, W, depth)
    err = torch.as_tensor(np.asarray(err_map, dtype=np.float32)).to(dt)
    obs = torch.as_tensor(np.asarray(observed_image, dtype=np.float32)).to(dt)
    W = 1 / err
    Y = (obs * W).reshape(1, -1, 1)
    X = (ret * W).reshape(sim.bs, depth, -1).permute(0, 2, 1)  # (bs, HW, depth)
    Xt = X.permute(0, 2, 1)
    coeffs = (torch.linalg.pinv(Xt @ X, rcond=1e-6) @ Xt @ Y)[..., 0]
    if return_coeffs:
        return coeffs
    return torch.squeeze((ret * coeffs[:, :, None, None]).sum(dim=1))


def backward_log_prob_terms(simulator: RefSimulator, params, observed_image, background_rms, exp_time):
    """BackwardProbModel (tf/model.py:215-222,264-273) without the prior: err_map from the OBSERVED image,
    Normal log-likelihood of the least-squares image, mean squared normalised residual."""
    dt = simulator.dtype
    obs = torch.as_tensor(np.asarray(observed_image, dtype=np.float32)).to(dt)
    err_map = torch.sqrt(torch.as_tensor(np.float32(background_rms)).to(dt) ** 2
                         + torch.clamp(obs, min=0) / torch.as_tensor(np.float32(exp_time)).to(dt))
    im_sim = lstsq_simulate(simulator, params, observed_image, err_map.to(torch.float32).numpy())
    err32 = err_map.to(torch.float32).to(dt)
    log_like = (-0.5 * ((im_sim - obs) / err32) ** 2 - torch.log(err32) - 0.5 * math.log(2 * math.pi)).sum(dim=(-2, -1))
    return log_like, (((im_sim - obs) / err32) ** 2).mean(dim=(-2, -1))


def stats_pixels(simulator: RefSimulator, params, observed_image, background_rms=None, exp_time=None,
                 error_map=None):
    """tf/model.py:89-101."""
    dt = simulator.dtype
    im_sim = simulator.simulate(params)
    if error_map is not None:
        err_map = torch.as_tensor(np.asarray(error_map, dtype=np.float32)).to(dt)
    else:
        bg = torch.as_tensor(np.float32(background_rms)).to(dt)
        et = torch.as_tensor(np.float32(exp_time)).to(dt)
        err_map = torch.sqrt(bg ** 2 + im_sim / et)
    obs = torch.as_tensor(np.asarray(observed_image, dtype=np.float32)).to(dt)
    reg = simulator.img_region
    chi2 = torch.sum(((im_sim - obs) / err_map) ** 2 * reg, dim=(-2, -1))
    normalization = torch.sum(torch.log(2 * np.pi * err_map ** 2) * reg, dim=(-2, -1))
    log_like = -1 / 2 * (chi2 + normalization)
    red_chi2 = chi2 / torch.count_nonzero(reg).to(dt)
    return log_like, red_chi2


# --------------------------------------------------------------------------
# image-position likelihood (tf/model.py:103-124, tf/simulator.py:72-91, tf/profile.py:9-43)
# --------------------------------------------------------------------------
def lens_hessian_autodiff(simulator: RefSimulator, x, y, lens_params):
    """Sum over lenses of ``lens.hessian`` (tf/simulator.py:80-88): autodiff of ``deriv`` by default
    (tf/profile.py:9-27); the reference overrides it analytically for NFW / Shear / SIS (closed forms equal to that
    derivative away from the clamps, so autodiff serves) and for dPIS / dPIE / ScalingRelation (``mass_hessian``:
    the dPIS override is NOT the derivative of its deflection and is restated as written)."""
    x = x.clone().requires_grad_(True)
    y = y.clone().requires_grad_(True)
    fxx = fxy = fyx = fyy = 0
    consts = simulator._consts("lenses_constants", len(simulator.phys_model.lenses))
    for lens, p, c in zip(simulator.phys_model.lenses, lens_params, consts):
        a, b, cc, d = mass_hessian(lens, x, y, **p, **c)
        fxx, fxy, fyx, fyy = fxx + a, fxy + b, fyx + cc, fyy + d
    return fxx, fxy, fyx, fyy


def stats_positions(simulator: RefSimulator, params, centroids_x, centroids_y, errors_x, errors_y):
    """tf/model.py:103-124 with centroids batched as (n_img, bs) (init_centroids, :187-194)."""
    dt = simulator.dtype
    bs = simulator.bs
    chi2 = 0.0
    log_like = 0.0
    n_position = 0.0
    for cx, cy, cex, cey in zip(centroids_x, centroids_y, errors_x, errors_y):
        cx = torch.as_tensor(np.asarray(cx, dtype=np.float32)).to(dt)[:, None].repeat(1, bs)
        cy = torch.as_tensor(np.asarray(cy, dtype=np.float32)).to(dt)[:, None].repeat(1, bs)
        cex = torch.as_tensor(np.asarray(cex, dtype=np.float32)).to(dt)
        cey = torch.as_tensor(np.asarray(cey, dtype=np.float32)).to(dt)
        n_position += 2.0 * cx.shape[0]
        bx, by = simulator.beta(cx, cy, params['lens_mass'])
        beta = torch.stack([bx, by], dim=0).permute(2, 0, 1)  # batch, xy, images
        bary = beta.mean(dim=2, keepdim=True)
        fxx, fxy, fyx, fyy = lens_hessian_autodiff(simulator, cx, cy, params['lens_mass'])
        mag = (1.0 / ((1 - fxx) * (1 - fyy) - fxy * fyx)).permute(1, 0)  # batch, images
        err = torch.stack([cex / mag, cey / mag], dim=1)  # batch, xy, images
        chi2_i = (((beta - bary) / err) ** 2).sum(dim=(-2, -1))
        norm_i = torch.log(2 * np.pi * err ** 2).sum(dim=(-2, -1))
        log_like = log_like + (-0.5) * (chi2_i + norm_i)
        chi2 = chi2 + chi2_i
    return log_like, chi2 / n_position
