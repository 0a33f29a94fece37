"""GPU parity of NFW_ELLIPSE, TNFW and CoreSersic (gl_extra.h): plugin level, inside the fused pixel likelihood,
the image-position likelihood and the linear solve -- HIP path vs the oracle."""
import math

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_gpu_parity import GRAD_RTOL, IMG_RTOL, LL_RTOL, gl  # noqa: F401

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _pts(n, seed=0, scale=1.5):
    r = np.random.default_rng(seed)
    return (r.normal(size=n) * scale).astype(np.float32), (r.normal(size=n) * scale).astype(np.float32)


def test_plugin_level_recipes(gl):
    """The reference's profile-test recipe (tests/test_profiles.py:50-58: rtol 1e-5 / atol 1e-4) on the three families."""
    from gigalens_amd.profiles.light.sersic import CoreSersic
    from gigalens_amd.profiles.mass.nfw import NFW_ELLIPSE
    from gigalens_amd.profiles.mass.tnfw import TNFW
    from oracle import ref_torch as ref
    x, y = _pts(10000)
    X, Y = torch.as_tensor(x, dtype=F64), torch.as_tensor(y, dtype=F64)
    for prof, kw in ((NFW_ELLIPSE(), dict(Rs=1.7, alpha_Rs=0.9, e1=0.2, e2=-0.15, center_x=0.1, center_y=-0.2)),
                     (TNFW(), dict(Rs=1.7, alpha_Rs=0.9, r_trunc=5.0, center_x=0.1, center_y=-0.2))):
        fx, fy = prof.deriv(x=x, y=y, **kw)
        ox, oy = ref.mass_deriv(prof, X, Y, **kw)
        assert np.allclose(fx.cpu().numpy(), ox.numpy(), rtol=1e-5, atol=1e-4)
        assert np.allclose(fy.cpu().numpy(), oy.numpy(), rtol=1e-5, atol=1e-4)
        h = [t.cpu().numpy() for t in prof.hessian(x, y, **kw)]
        ho = [t.detach().numpy() for t in ref.mass_hessian(prof, X, Y, **kw)]
        far = np.hypot(x - 0.1, y + 0.2) > 0.05  # the cuspy centre: second derivatives diverge like 1/r
        for a, b in zip(h, ho):
            assert np.allclose(a[far], b[far], rtol=5e-4, atol=5e-5 * np.abs(ho[0]).max())
    cs = CoreSersic()
    kw = dict(R_sersic=0.8, n_sersic=2.0, Rb=0.3, alpha=1.5, gamma=0.4, e1=0.2, e2=-0.15, center_x=0.05, center_y=0.02, Ie=80.0)
    I = cs.light(x=x, y=y, **kw).cpu().numpy()
    oI = ref.light_eval(cs, X, Y, **kw).numpy()
    assert np.allclose(I, oI, rtol=2e-5, atol=1e-5 * oI.max())


def _model(num_pix, batch, lstsq=False):
    from gigalens_amd import prior as tfd
    from gigalens_amd import workloads
    from gigalens_amd.model import PhysicalModel
    from gigalens_amd.profiles.light.sersic import CoreSersic, Sersic
    from gigalens_amd.profiles.mass.nfw import NFW_ELLIPSE
    from gigalens_amd.profiles.mass.tnfw import TNFW
    from gigalens_amd.simulator import SimulatorConfig
    J, S = tfd.JointDistributionNamed, tfd.JointDistributionSequential
    nfe = J(dict(Rs=tfd.LogNormal(math.log(1.5), 0.1), alpha_Rs=tfd.LogNormal(math.log(0.9), 0.1), e1=tfd.Normal(0.15, 0.05),
                 e2=tfd.Normal(-0.1, 0.05), center_x=tfd.Normal(0, 0.03), center_y=tfd.Normal(0, 0.03)))
    tn = J(dict(Rs=tfd.LogNormal(math.log(0.4), 0.1), alpha_Rs=tfd.LogNormal(math.log(0.15), 0.1),
                r_trunc=tfd.LogNormal(math.log(1.2), 0.2), center_x=tfd.Normal(0.6, 0.03), center_y=tfd.Normal(-0.4, 0.03)))
    core = dict(R_sersic=tfd.LogNormal(math.log(0.25), 0.1), n_sersic=tfd.Uniform(1, 3), Rb=tfd.LogNormal(math.log(0.08), 0.2),
                alpha=tfd.Uniform(1.0, 3.0), gamma=tfd.Uniform(0.05, 0.5), e1=tfd.Normal(0, 0.1), e2=tfd.Normal(0, 0.1),
                center_x=tfd.Normal(0, 0.1), center_y=tfd.Normal(0, 0.1))
    ser = dict(R_sersic=tfd.LogNormal(math.log(0.15), 0.1), n_sersic=tfd.Uniform(1, 3), center_x=tfd.Normal(0.2, 0.1),
               center_y=tfd.Normal(-0.1, 0.1))
    if not lstsq:
        core["Ie"] = tfd.LogNormal(math.log(60.0), 0.3)
        ser["Ie"] = tfd.LogNormal(math.log(40.0), 0.3)
    phys = PhysicalModel([NFW_ELLIPSE(), TNFW()], [], [CoreSersic(use_lstsq=lstsq), Sersic(use_lstsq=lstsq)])
    prior = J(dict(lens_mass=S([nfe, tn]), source_light=S([J(core), J(ser)])))
    return workloads.Workload("XTR", phys, prior, SimulatorConfig(delta_pix=0.08, num_pix=num_pix), batch)


def test_simulate_loglike_grad_vs_oracle(gl):
    wl = _model(36, 4)
    obs, _, _ = gl.workloads.synthetic_observation(wl, gl.LensSimulator)
    sim = gl.LensSimulator(wl.phys_model, wl.sim_config, bs=wl.batch)
    packed = H.sample_packed(wl, sim, seed=11)
    obs_np = obs.cpu().numpy()
    ll_o, red_o, g_o, img_o = H.oracle_loglike_and_grad(wl, packed.cpu().double(), obs_np, None, wl.batch)
    img = sim.simulate(packed).cpu().numpy().reshape(img_o.shape)
    assert np.abs(img - img_o).max() <= 2 * IMG_RTOL * np.abs(img_o).max() + 1e-7
    pm = gl.ForwardProbModel(wl.prior, obs_np, wl.background_rms, wl.exp_time, include_positions=False)
    p = packed.clone().requires_grad_(True)
    ll, red = pm._pixel_stats_packed(sim, p)
    ll.sum().backward()
    assert np.all(np.abs(ll.detach().cpu().numpy() - ll_o) <= 5 * LL_RTOL * np.maximum(np.abs(ll_o), red_o * 36 * 36))
    g = p.grad.cpu().numpy()
    scale = np.maximum(np.abs(g_o).max(axis=1, keepdims=True), 1e-3 * np.abs(g_o).max())
    bad = np.abs(g - g_o) > GRAD_RTOL * np.maximum(np.abs(g_o), 1e-2 * scale) + 1e-6
    assert not bad.any(), (np.argwhere(bad)[:5], g[bad][:5], g_o[bad][:5])
    # fused unconstrained-space entry
    z = pm.bij.inverse(wl.prior.sample(wl.batch, seed=4)).to(sim.device)
    z1, z2 = z.clone().requires_grad_(True), z.clone().requires_grad_(True)
    lp1, _ = pm.log_prob(sim, z1)
    lp2, _ = pm.log_prob_unfused(sim, z2)
    lp1.sum().backward()
    lp2.sum().backward()
    assert torch.allclose(lp1, lp2, rtol=2e-5)
    assert torch.all((z1.grad - z2.grad).abs() <= 2e-4 * z2.grad.abs().max(dim=1, keepdim=True).values + 1e-5)


def test_positions_and_lstsq(gl):
    from oracle import ref_torch as ref
    wl = _model(24, 3)
    sim = gl.LensSimulator(wl.phys_model, wl.sim_config, bs=wl.batch)
    packed = H.sample_packed(wl, sim, seed=6)
    cx = [np.array([1.6, -1.5, 0.3, -0.4], np.float32)]
    cy = [np.array([0.4, -0.2, 1.7, -1.6], np.float32)]
    ex = [np.array([0.01, 0.02, 0.015, 0.01], np.float32)]
    pm = gl.ForwardProbModel(wl.prior, centroids_x=cx, centroids_y=cy, centroids_errors_x=ex, centroids_errors_y=ex,
                             include_pixels=False, include_positions=True)
    p = packed.clone().requires_grad_(True)
    ll, red = pm.stats_positions(sim, p)
    ll.sum().backward()
    rs = ref.RefSimulator(wl.phys_model, wl.sim_config, wl.batch, dtype=F64)
    p64 = packed.cpu().double().requires_grad_(True)
    ll_o, red_o = ref.stats_positions(rs, H.struct_from_packed(wl.phys_model, p64), cx, cy, ex, ex)
    (g_o,) = torch.autograd.grad(ll_o.sum(), p64)
    assert np.allclose(ll.detach().cpu().numpy(), ll_o.detach().numpy(), rtol=2e-4)
    g, go = p.grad.cpu().numpy(), g_o.numpy()
    scale = np.abs(go).max(axis=1, keepdims=True)
    assert np.all(np.abs(g - go) <= 2e-3 * np.maximum(np.abs(go), 1e-2 * scale) + 1e-6)
    # CoreSersic as a least-squares component
    wl2 = _model(30, 3, lstsq=True)
    sim2 = gl.LensSimulator(wl2.phys_model, wl2.sim_config, bs=3)
    x = wl2.prior.sample(3, seed=5)
    obs, _, _ = gl.workloads.synthetic_observation(_model(30, 1), gl.LensSimulator)
    obs = obs.cpu().numpy()
    err = np.sqrt(0.04 + np.clip(obs, 0, None) / 100.0).astype(np.float32)
    rs2 = ref.RefSimulator(wl2.phys_model, wl2.sim_config, 3, dtype=F64)
    x64 = {g_: [{k: v.double() for k, v in d.items()} for d in lst] for g_, lst in x.items()}
    img_o = ref.lstsq_simulate(rs2, x64, obs, err)
    img = sim2.lstsq_simulate(x, obs, err)
    assert np.abs(img.cpu().numpy() - img_o.numpy()).max() <= 2e-4 * np.abs(img_o.numpy()).max()
