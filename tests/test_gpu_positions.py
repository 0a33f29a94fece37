"""Image-position likelihood (tf/model.py:103-124): HIP kernels (nested forward-mode duals on the profile templates)
against the oracle (torch autograd Hessians, double backward for the parameter gradient)."""
import math

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _setup(kind):
    from gigalens_amd import prior as tfd
    from gigalens_amd import workloads
    from gigalens_amd.model import PhysicalModel
    from gigalens_amd.profiles.light.sersic import Sersic
    from gigalens_amd.profiles.mass.epl import EPL
    from gigalens_amd.profiles.mass.nfw import NFW
    from gigalens_amd.profiles.mass.shear import Shear
    from gigalens_amd.profiles.mass.sie import SIE
    from gigalens_amd.profiles.mass.sis import SIS
    from gigalens_amd.simulator import SimulatorConfig
    J, S = tfd.JointDistributionNamed, tfd.JointDistributionSequential
    if kind == "epl":
        wl = workloads.make("C2", num_pix=24, batch=5)
        return wl.phys_model, wl.prior, wl.sim_config, 5
    sis = J(dict(theta_E=tfd.LogNormal(math.log(0.4), 0.1), center_x=tfd.Normal(0.3, 0.02), center_y=tfd.Normal(-0.2, 0.02)))
    sie = J(dict(theta_E=tfd.LogNormal(math.log(0.9), 0.1), e1=tfd.Normal(0.1, 0.05), e2=tfd.Normal(-0.15, 0.05),
                 center_x=tfd.Normal(0, 0.02), center_y=tfd.Normal(0, 0.02)))
    nfw = J(dict(Rs=tfd.LogNormal(math.log(1.5), 0.2), alpha_Rs=tfd.LogNormal(math.log(0.4), 0.2),
                 center_x=tfd.Normal(-0.2, 0.05), center_y=tfd.Normal(0.1, 0.05)))
    src = J(dict(R_sersic=tfd.LogNormal(math.log(0.2), 0.1), n_sersic=tfd.Uniform(1, 3), center_x=tfd.Normal(0, 0.1),
                 center_y=tfd.Normal(0, 0.1), Ie=tfd.LogNormal(math.log(50.0), 0.3)))
    phys = PhysicalModel([SIE(), NFW(), SIS()], [], [Sersic()])
    prior = J(dict(lens_mass=S([sie, nfw, sis]), source_light=S([src])))
    return phys, prior, SimulatorConfig(delta_pix=0.08, num_pix=24), 4


CX = [np.array([1.05, -0.95, 0.15, -0.2], np.float32), np.array([0.7, -0.6], np.float32)]
CY = [np.array([0.2, -0.1, 1.1, -1.0], np.float32), np.array([-0.75, 0.8], np.float32)]
EX = [np.array([0.01, 0.02, 0.015, 0.01], np.float32), np.array([0.03, 0.02], np.float32)]
EY = [np.array([0.012, 0.02, 0.01, 0.02], np.float32), np.array([0.02, 0.025], np.float32)]


# images far from the critical curve (|x| ~ 1.8 theta_E): moderate magnification, fp32 stays well conditioned
CX_FAR = [np.array([2.1, -1.9, 0.3, -0.4], np.float32), np.array([1.5, -1.4], np.float32)]
CY_FAR = [np.array([0.4, -0.2, 2.0, -2.1], np.float32), np.array([-1.5, 1.6], np.float32)]


@pytest.mark.parametrize("kind,near", [("epl", False), ("epl", True), ("mixed", True)])
def test_stats_positions_vs_oracle(kind, near):
    CX, CY = (globals()["CX"], globals()["CY"]) if near else (CX_FAR, CY_FAR)
    # positions ON the Einstein ring of an EPL make det(A) ~ 0: mu = 1/det amplifies fp32 rounding (conditioning of
    # the quantity itself, in any fp32 evaluation) -- looser tolerance there
    rtol = 2e-3 if (kind == "epl" and near) else 2e-5
    from gigalens_amd import workloads
    from gigalens_amd.model import ForwardProbModel
    from gigalens_amd.simulator import LensSimulator
    from oracle import ref_torch as ref
    phys, prior, cfg, B = _setup(kind)
    sim = LensSimulator(phys, cfg, bs=B)
    wl = workloads.Workload("POS", phys, prior, cfg, B)
    packed = H.sample_packed(wl, sim, seed=6)
    pm = ForwardProbModel(prior, centroids_x=CX, centroids_y=CY, centroids_errors_x=EX, centroids_errors_y=EY,
                          include_pixels=False, include_positions=True)
    assert pm.n_position == 12.0
    p = packed.clone().requires_grad_(True)
    ll, red = pm.stats_positions(sim, p)
    ll.sum().backward()
    rs = ref.RefSimulator(phys, cfg, B, dtype=torch.float64)
    p64 = packed.cpu().double().requires_grad_(True)
    ll_o, red_o = ref.stats_positions(rs, H.struct_from_packed(phys, p64), CX, CY, EX, EY)
    (g_o,) = torch.autograd.grad(ll_o.sum(), p64)
    assert np.allclose(ll.detach().cpu().numpy(), ll_o.detach().numpy(), rtol=rtol)
    assert np.allclose(red.detach().cpu().numpy(), red_o.detach().numpy(), rtol=rtol)
    g, go = p.grad.cpu().numpy(), g_o.numpy()
    scale = np.abs(go).max(axis=1, keepdims=True)
    assert np.all(np.abs(g - go) <= max(5e-4, 5 * rtol) * scale + 1e-6), (np.abs(g - go) / scale).max()
    n_lens_par = sum(len(l.params) for l in phys.lenses)
    assert np.all(g[:, n_lens_par:] == 0)  # the position term does not depend on the light profiles
    # the nested structure works as input too (reference signature: stats_positions(simulator, params))
    x = pm.bij.forward(pm.bij.inverse(prior.sample(B, seed=6)))
    ll2, _ = pm.stats_positions(sim, x)
    assert torch.allclose(ll2, ll.detach(), rtol=1e-3)


def test_log_prob_pixels_plus_positions():
    from gigalens_amd import workloads
    from gigalens_amd.model import ForwardProbModel
    from gigalens_amd.simulator import LensSimulator
    phys, prior, cfg, B = _setup("epl")
    wl = workloads.Workload("POS", phys, prior, cfg, B)
    obs, _, _ = workloads.synthetic_observation(wl, LensSimulator)
    sim = LensSimulator(phys, cfg, bs=B)
    pm = ForwardProbModel(prior, obs.cpu().numpy(), 0.2, 100.0, centroids_x=CX, centroids_y=CY, centroids_errors_x=EX,
                          centroids_errors_y=EY)  # the reference's defaults: pixels AND positions
    z0 = pm.bij.inverse(prior.sample(B, seed=2)).to("cuda")
    lp, red, g = pm.log_prob_and_grad(sim, z0)
    zz = z0.clone().requires_grad_(True)
    lpu, redu = pm.log_prob_unfused(sim, zz)
    lpu.sum().backward()
    assert torch.allclose(lp, lpu.detach(), rtol=2e-5, atol=1e-2) and torch.allclose(red, redu.detach(), rtol=2e-5)
    sc = zz.grad.abs().max(dim=1, keepdim=True).values
    assert ((g - zz.grad).abs() <= 5e-4 * sc + 1e-3).all(), ((g - zz.grad).abs() / sc).max()
    # red_chi2 = (red_pix + red_pos) / 2 and log_like adds up (tf/model.py:150-167)
    pix = ForwardProbModel(prior, obs.cpu().numpy(), 0.2, 100.0, include_positions=False)
    pos = ForwardProbModel(prior, centroids_x=CX, centroids_y=CY, centroids_errors_x=EX, centroids_errors_y=EY,
                           include_pixels=False)
    lp_pix, red_pix, _ = pix.log_prob_and_grad(sim, z0)
    lp_pos, red_pos, _ = pos.log_prob_and_grad(sim, z0)
    prior_term = pix.log_prior(z0)
    assert torch.allclose(lp, lp_pix + lp_pos - prior_term, rtol=2e-5, atol=1e-2)
    assert torch.allclose(red, 0.5 * (red_pix + red_pos), rtol=2e-5)
    assert torch.allclose(pm.log_like(sim, z0), lp - prior_term, rtol=2e-5, atol=1e-2)


def test_magnification_convergence_shear_maps():
    """LensSimulator.magnification / convergence / shear (tf/simulator.py:80-107) on a grid of points."""
    from gigalens_amd import workloads
    from gigalens_amd.simulator import LensSimulator
    from oracle import ref_torch as ref
    phys, prior, cfg, B = _setup("mixed")
    sim = LensSimulator(phys, cfg, bs=B)
    wl = workloads.Workload("POS", phys, prior, cfg, B)
    packed = H.sample_packed(wl, sim, seed=3)
    r = np.random.default_rng(0)
    x = (r.uniform(-2, 2, 300)).astype(np.float32)[:, None].repeat(B, 1)
    y = (r.uniform(-2, 2, 300)).astype(np.float32)[:, None].repeat(B, 1)
    lens_params = H.struct_from_packed(phys, packed)["lens_mass"]
    mu = sim.magnification(x, y, lens_params).cpu().numpy()
    kap = sim.convergence(x, y, lens_params).cpu().numpy()
    g1, g2 = (t.cpu().numpy() for t in sim.shear(x, y, lens_params))
    rs = ref.RefSimulator(phys, cfg, B, dtype=torch.float64)
    p64 = H.struct_from_packed(phys, packed.cpu().double())["lens_mass"]
    fxx, fxy, fyx, fyy = (t.detach().numpy() for t in ref.lens_hessian_autodiff(rs, torch.as_tensor(x).double(),
                                                                               torch.as_tensor(y).double(), p64))
    mu_o = 1.0 / ((1 - fxx) * (1 - fyy) - fxy * fyx)
    ok = np.abs(mu_o) < 50  # away from the critical curves (1/det amplifies fp32 rounding there)
    assert ok.mean() > 0.8
    assert np.allclose(mu[ok], mu_o[ok], rtol=2e-3)
    assert np.allclose(kap, 0.5 * (fxx + fyy), rtol=1e-4, atol=1e-5)
    assert np.allclose(g1, 0.5 * (fxx - fyy), rtol=1e-4, atol=1e-5)
    assert np.allclose(g2, fxy, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("cls,kw", [("EPL", dict(theta_E=1.2, gamma=2.2, e1=-0.1, e2=0.1, center_x=0.05, center_y=0.0)),
                                    ("NFW", dict(Rs=1.5, alpha_Rs=0.8, center_x=0.1, center_y=-0.1)),
                                    ("DPIS", dict(theta_E=1.1, r_core=0.2, r_cut=4.0, center_x=0.05, center_y=-0.1)),
                                    ("DPIE", dict(theta_E=1.3, r_core=0.2, r_cut=5.0, center_x=0.1, center_y=-0.2, e1=0.2, e2=-0.15))])
def test_profile_hessian_convergence_shear(cls, kw):
    """MassProfile.hessian / convergence / shear at plugin level (tf/profile.py:9-43) vs the oracle's resolution of
    ``lens.hessian`` (autodiff; analytic dPIS / dPIE overrides as written)."""
    from gigalens_amd.profiles.mass import epl, nfw, piemd
    from oracle import ref_torch as ref
    prof = {"EPL": epl.EPL, "NFW": nfw.NFW, "DPIS": piemd.DPIS, "DPIE": piemd.DPIE}[cls]()
    r = np.random.default_rng(1)
    x, y = (r.uniform(-3, 3, 2000)).astype(np.float32), (r.uniform(-3, 3, 2000)).astype(np.float32)
    h = [t.cpu().numpy() for t in prof.hessian(x, y, **kw)]
    ho = [t.detach().numpy() for t in ref.mass_hessian(prof, torch.as_tensor(x).double(), torch.as_tensor(y).double(), **kw)]
    sc = max(np.abs(t).max() for t in ho)
    rr = np.hypot(x - kw["center_x"], y - kw["center_y"])
    far = rr > 0.05  # the Hessian diverges like 1/r^2 towards a cuspy centre: compare where fp32 resolves it
    for a, b in zip(h, ho):
        assert np.allclose(a[far], b[far], rtol=2e-4, atol=2e-5 * sc)
    kap = prof.convergence(x, y, **kw).cpu().numpy()
    g1, g2 = (t.cpu().numpy() for t in prof.shear(x, y, **kw))
    assert np.allclose(kap[far], 0.5 * (ho[0] + ho[3])[far], rtol=2e-4, atol=2e-5 * sc)
    assert np.allclose(g1[far], 0.5 * (ho[0] - ho[3])[far], rtol=2e-4, atol=2e-5 * sc)
    assert np.allclose(g2[far], ho[1][far], rtol=2e-4, atol=2e-5 * sc)
