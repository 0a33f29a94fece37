"""The open plugin boundary on the GPU (csrc/gl_user.hip): the reference lets a user subclass MassProfile / LightProfile and write
deriv / light (src/gigalens/profile.py:58-82); here such a class carries a `hip_body`, compiled at run time, and serves the same
plugin-level calls.  Parity: a user-written SIS equals the built-in kind; derivatives (forward-mode duals in the kernel,
contracted by torch.autograd) equal a float64 torch restatement."""
import numpy as np
import pytest
import torch

from tests.test_gpu_parity import gl  # noqa: F401
from tests.test_user_profile_compile import SIS_BODY

pytestmark = pytest.mark.gpu


def _profiles():
    from gigalens_amd.profile import LightProfile, MassProfile

    class UserSIS(MassProfile):
        _name, _params = "USER_SIS", ["theta_E", "center_x", "center_y"]
        hip_body = SIS_BODY

    class UserGauss(LightProfile):
        """An elliptical Gaussian with a branch and a transcendental of its own: I = amp exp(-q2 / 2) / (1 + |x - cx|)."""
        _name, _params, _amp = "USER_GAUSS", ["sigma", "q", "center_x", "center_y"], "amp"
        hip_body = """
        template <class R> __device__ R light(R x, R y, const R* p) {
          R dx = x - p[2], dy = y - p[3];
          R q2 = (dx * dx * p[1] + dy * dy / p[1]) / (p[0] * p[0]);
          R damp = dx < 0.f ? 1.f - dx : 1.f + dx;    // |dx| with a branch on the value
          return p[4] * exp(-0.5f * q2) / damp;
        }
        """
    return UserSIS, UserGauss


def test_user_written_sis_equals_the_built_in(gl):
    from gigalens_amd.profiles.mass.sis import SIS
    UserSIS, _ = _profiles()
    r = np.random.default_rng(0)
    B = 5
    x = torch.tensor(r.uniform(-2, 2, (7, 9, 1)), dtype=torch.float32, device="cuda")
    y = torch.tensor(r.uniform(-2, 2, (7, 9, 1)), dtype=torch.float32, device="cuda")
    kw = dict(theta_E=torch.tensor(r.uniform(0.5, 1.5, B), dtype=torch.float32, device="cuda"),
              center_x=torch.tensor(r.normal(0, 0.1, B), dtype=torch.float32, device="cuda"),
              center_y=torch.tensor(r.normal(0, 0.1, B), dtype=torch.float32, device="cuda"))
    ax, ay = UserSIS().deriv(x, y, **kw)
    bx, by = SIS().deriv(x, y, **kw)
    assert ax.shape == bx.shape == (7, 9, B)
    assert torch.allclose(ax, bx, rtol=2e-6, atol=1e-7) and torch.allclose(ay, by, rtol=2e-6, atol=1e-7)
    # hessian / convergence / shear (tf/profile.py:9-42) come from the same duals
    for hu, hb in zip(UserSIS().hessian(x, y, **kw), SIS().hessian(x, y, **kw)):
        assert torch.allclose(hu, hb, rtol=2e-5, atol=2e-6)
    assert torch.allclose(UserSIS().convergence(x, y, **kw), SIS().convergence(x, y, **kw), rtol=2e-5, atol=2e-6)


def test_gradients_of_user_bodies_match_float64_autograd(gl):
    UserSIS, UserGauss = _profiles()
    r = np.random.default_rng(1)
    B = 4
    mk = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True)
    x, y = mk(r.uniform(-2, 2, (11, 1))), mk(r.uniform(-2, 2, (11, 1)))
    th, cx, cy = mk(r.uniform(0.5, 1.5, B)), mk(r.normal(0, 0.1, B)), mk(r.normal(0, 0.1, B))
    ax, ay = UserSIS().deriv(x, y, theta_E=th, center_x=cx, center_y=cy)
    w0, w1 = torch.tensor(r.normal(size=(11, B)), device="cuda", dtype=torch.float32), torch.tensor(r.normal(size=(11, B)), device="cuda", dtype=torch.float32)
    (ax * w0 + ay * w1).sum().backward()
    d = lambda t: t.detach().double().cpu().requires_grad_(True)
    x6, y6, th6, cx6, cy6 = d(x), d(y), d(th), d(cx), d(cy)
    dx, dy = x6 - cx6, y6 - cy6
    rr = torch.sqrt(dx * dx + dy * dy)
    (th6 * dx / rr * w0.double().cpu() + th6 * dy / rr * w1.double().cpu()).sum().backward()
    for a, b in ((x, x6), (y, y6), (th, th6), (cx, cx6), (cy, cy6)):
        assert a.grad is not None and a.grad.shape == a.shape
        assert np.allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=2e-4, atol=2e-5 * float(b.grad.abs().max())), (a.grad, b.grad)
    # light profile: amplitude is the last parameter (profile.py:24-60)
    sg, q, amp = mk(r.uniform(0.5, 1.0, B)), mk(r.uniform(0.6, 1.0, B)), mk(r.uniform(1.0, 3.0, B))
    cx2, cy2 = mk(r.normal(0, 0.1, B)), mk(r.normal(0, 0.1, B))
    x2, y2 = mk(r.uniform(-2, 2, (13, 1))), mk(r.uniform(-2, 2, (13, 1)))
    I = UserGauss().light(x2, y2, sigma=sg, q=q, center_x=cx2, center_y=cy2, amp=amp)
    assert I.shape == (13, B)
    w = torch.tensor(r.normal(size=(13, B)), device="cuda", dtype=torch.float32)
    (I * w).sum().backward()
    X, Y, S, Q, A, CX, CY = d(x2), d(y2), d(sg), d(q), d(amp), d(cx2), d(cy2)
    ddx, ddy = X - CX, Y - CY
    I6 = A * torch.exp(-0.5 * (ddx * ddx * Q + ddy * ddy / Q) / (S * S)) / (1 + ddx.abs())
    assert np.allclose(I.detach().cpu().numpy(), I6.detach().numpy(), rtol=2e-5, atol=1e-6)
    (I6 * w.double().cpu()).sum().backward()
    for a, b in ((x2, X), (y2, Y), (sg, S), (q, Q), (amp, A), (cx2, CX), (cy2, CY)):
        assert np.allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=3e-4, atol=3e-5 * float(b.grad.abs().max())), (a.grad, b.grad)


def test_compile_errors_inside_a_model_come_back_verbatim(gl):
    from gigalens_amd import _native
    from gigalens_amd.model import PhysicalModel
    from gigalens_amd.profiles.light.sersic import Sersic
    from gigalens_amd.simulator import LensSimulator, SimulatorConfig
    UserSIS, _ = _profiles()

    class Broken(UserSIS):
        hip_body = "template <class R> __device__ void deriv(R x, R y, const R* p, R& fx, R& fy) { fx = nope; }"
    with pytest.raises(_native.NativeLibraryError, match="does not compile"):
        Broken().deriv(torch.zeros(3, device="cuda"), torch.zeros(3, device="cuda"), theta_E=1.0, center_x=0.0, center_y=0.0)
    with pytest.raises(_native.NativeLibraryError, match="does not compile"):
        LensSimulator(PhysicalModel([Broken()], [], [Sersic()]), SimulatorConfig(delta_pix=0.1, num_pix=8), bs=1)


def test_scaling_relation_over_any_base_profile(gl):
    """scaling_relation.py:8-19 wraps ANY MassProfile.  The fused kernels serve the dPIE family; every other population goes
    through the generic plugin-level sum (chunks of galaxies folded into the base profile's batch axis).  Checked three ways: on a
    dPIE population against the fused kernel, on an NFW population against a sum written out by hand, on a user-written SIS
    population (differentiable) against float64 autograd."""
    from gigalens_amd.profiles.mass.nfw import NFW
    from gigalens_amd.profiles.mass.piemd import DPIS
    from gigalens_amd.profiles.mass.scaling_relation import ScalingRelation
    UserSIS, _ = _profiles()
    r = np.random.default_rng(4)
    G, B = 7, 3
    cat = dict(lum=r.uniform(0.3, 2.0, G).astype(np.float32), center_x=r.normal(0, 0.8, G).astype(np.float32),
               center_y=r.normal(0, 0.8, G).astype(np.float32))
    x = torch.tensor(r.uniform(-2, 2, (5, 6, 1)), dtype=torch.float32, device="cuda")
    y = torch.tensor(r.uniform(-2, 2, (5, 6, 1)), dtype=torch.float32, device="cuda")
    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    # (1) dPIS population: fused kernel vs the generic path (forced), chunked
    names = DPIS().params
    sr = ScalingRelation(DPIS(), names[:3], 1.0, {names[0]: 0.5, names[1]: 0.5, names[2]: 0.5}, cat, chunk_size=3)
    scales = {names[0]: t(r.uniform(0.5, 1.0, B)), names[1]: t(r.uniform(0.02, 0.05, B)), names[2]: t(r.uniform(1.0, 2.0, B))}
    fx, fy = sr.deriv(x, y, **scales)
    assert not sr._generic
    gx, gy = sr._sum_over_galaxies(sr.profile.deriv, 2, x, y, scales)
    assert torch.allclose(fx, gx, rtol=2e-5, atol=2e-6) and torch.allclose(fy, gy, rtol=2e-5, atol=2e-6)
    # (2) NFW population scaled in Rs: against the members summed by hand
    cat_n = dict(cat, alpha_Rs=r.uniform(0.5, 1.0, G).astype(np.float32))
    pop = ScalingRelation(NFW(), ["Rs"], 1.0, {"Rs": 0.4}, cat_n, chunk_size=4)
    assert pop._generic
    Rs = t(r.uniform(0.5, 1.0, B))
    ax, ay = pop.deriv(x, y, Rs=Rs)
    sx, sy = torch.zeros_like(ax), torch.zeros_like(ay)
    for g in range(G):
        mx, my = NFW().deriv(x, y, Rs=Rs * float(cat_n["lum"][g] ** 0.4), alpha_Rs=float(cat_n["alpha_Rs"][g]),
                             center_x=float(cat_n["center_x"][g]), center_y=float(cat_n["center_y"][g]))
        sx, sy = sx + mx, sy + my
    assert torch.allclose(ax, sx, rtol=2e-5, atol=2e-6) and torch.allclose(ay, sy, rtol=2e-5, atol=2e-6)
    # (3) a population of user-written SIS members, differentiable in the scale
    popu = ScalingRelation(UserSIS(), ["theta_E"], 1.0, {"theta_E": 0.5}, cat)
    th = torch.tensor(r.uniform(0.5, 1.0, B), dtype=torch.float32, device="cuda", requires_grad=True)
    ux, uy = popu.deriv(x, y, theta_E=th)
    (ux.sum() + 2 * uy.sum()).backward()
    th6 = th.detach().double().cpu().requires_grad_(True)
    x6, y6 = x.double().cpu(), y.double().cpu()
    tot = 0
    for g in range(G):
        dx, dy = x6 - float(cat["center_x"][g]), y6 - float(cat["center_y"][g])
        rr = torch.sqrt(dx * dx + dy * dy)
        tE = th6 * float(np.float32(cat["lum"][g]) ** np.float32(0.5))
        tot = tot + (tE * dx / rr).sum() + 2 * (tE * dy / rr).sum()
    tot.backward()
    assert np.allclose(th.grad.cpu().numpy(), th6.grad.numpy(), rtol=2e-4)


SERSIC_BODY = """
template <class R> __device__ R light(R x, R y, const R* p) {
  // p = R_sersic, n_sersic, center_x, center_y, Ie   (src/gigalens/tf/profiles/light/sersic.py:23-66, spherical)
  R dx = x - p[2], dy = y - p[3];
  R r = sqrt(dx * dx + dy * dy);
  R bn = 1.9992f * p[1] - 0.3271f;
  return p[4] * exp(-bn * (pow(r / p[0], 1.f / p[1]) - 1.f));
}
"""


def test_user_written_profiles_inside_a_model_equal_the_built_in_kinds(gl):
    """The second half of the boundary: a PhysicalModel may hold user-written profiles; LensSimulator then compiles the
    interpreter kernel of the likelihood path with their bodies (gl_model_create_user).  A model of a user-written SIS lens and a
    user-written Sersic source must reproduce the model of the built-in kinds: image, image VJP, log-likelihood and its gradient."""
    from gigalens_amd.model import PhysicalModel
    from gigalens_amd.profile import LightProfile
    from gigalens_amd.profiles.light.sersic import Sersic
    from gigalens_amd.profiles.mass.shear import Shear
    from gigalens_amd.profiles.mass.sis import SIS
    from gigalens_amd.simulator import LensSimulator, SimulatorConfig
    UserSIS, _ = _profiles()

    class UserSersic(LightProfile):
        _name, _params, _amp = "USER_SERSIC", ["R_sersic", "n_sersic", "center_x", "center_y"], "Ie"
        hip_body = SERSIC_BODY

    cfg = SimulatorConfig(delta_pix=0.08, num_pix=40)
    B = 6
    sim_u = LensSimulator(PhysicalModel([UserSIS(), Shear()], [], [UserSersic()]), cfg, bs=B)
    sim_b = LensSimulator(PhysicalModel([SIS(), Shear()], [], [Sersic()]), cfg, bs=B)
    r = np.random.default_rng(5)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    params = {"lens_mass": [dict(theta_E=t(r.uniform(0.8, 1.2, B)), center_x=t(r.normal(0, 0.05, B)), center_y=t(r.normal(0, 0.05, B))),
                            dict(gamma1=t(r.normal(0, 0.03, B)), gamma2=t(r.normal(0, 0.03, B)))],
              "source_light": [dict(R_sersic=t(r.uniform(0.2, 0.4, B)), n_sersic=t(r.uniform(1.0, 3.0, B)),
                                    center_x=t(r.normal(0.05, 0.1, B)), center_y=t(r.normal(0, 0.1, B)), Ie=t(r.uniform(20, 60, B)))]}
    img_u, img_b = sim_u.simulate(params), sim_b.simulate(params)
    assert img_u.shape == img_b.shape == (B, 40, 40)
    top = float(img_b.abs().max())
    assert torch.allclose(img_u, img_b, rtol=1e-4, atol=2e-5 * top), float((img_u - img_b).abs().max()) / top
    pu, pb = sim_u.pack(params), sim_b.pack(params)
    assert torch.equal(pu, pb)
    mu, mb = sim_u._model, sim_b._model
    obs = img_b[0] + 0.5 * t(r.normal(size=(40, 40)))
    ll_u, c_u, g_u = mu.loglike(pu, obs, None, None, 0.5, 100.0, True)
    ll_b, c_b, g_b = mb.loglike(pb, obs, None, None, 0.5, 100.0, True)
    assert torch.allclose(ll_u, ll_b, rtol=2e-5) and torch.allclose(c_u, c_b, rtol=2e-5)
    scale = g_b.abs().amax(dim=0, keepdim=True)
    assert torch.all((g_u - g_b).abs() <= 2e-3 * scale + 1e-6), ((g_u - g_b).abs() / scale).max()
    w = t(r.normal(size=(B, 40, 40)))
    v_u, v_b = mu.simulate_bwd(pu, w), mb.simulate_bwd(pb, w)
    vs = v_b.abs().amax(dim=0, keepdim=True)
    assert torch.all((v_u - v_b).abs() <= 2e-3 * vs + 1e-6)
    # lens maps (tf/simulator.py:72-107): the fused kernel compiled with the bodies on the nested duals of the point kernels
    maps_u, maps_b = mu.lens_maps(pu, None, None), mb.lens_maps(pb, None, None)
    for a, b2 in zip(maps_u, maps_b):
        assert torch.allclose(a, b2, rtol=2e-4, atol=2e-5)
    xs, ys = sim_b.img_X[:200, None], sim_b.img_Y[:200, None]
    for fn in ("magnification", "convergence"):
        a, b2 = getattr(sim_u, fn)(xs, ys, params["lens_mass"]), getattr(sim_b, fn)(xs, ys, params["lens_mass"])
        assert a.shape == b2.shape == (200, B)
        assert torch.allclose(a, b2, rtol=2e-4, atol=2e-5 * float(b2.abs().median()) + 1e-6), fn
    ga, gb = sim_u.shear(xs, ys, params["lens_mass"]), sim_b.shear(xs, ys, params["lens_mass"])
    assert torch.allclose(ga[0], gb[0], rtol=2e-4, atol=1e-5) and torch.allclose(ga[1], gb[1], rtol=2e-4, atol=1e-5)


def test_log_prob_and_map_on_a_model_with_user_written_profiles(gl):
    """The fused unconstrained-space entry (bijectors + kernels + prior, tf/model.py:76-162) and a short MAP run on a model whose
    lens and source are user-written: same log-prob and gradient as the built-in twin, and the optimiser improves the fit."""
    import math
    from gigalens_amd import prior as tfd
    from gigalens_amd.inference import Adam, ModellingSequence
    from gigalens_amd.model import ForwardProbModel, PhysicalModel
    from gigalens_amd.profile import LightProfile
    from gigalens_amd.profiles.light.sersic import Sersic
    from gigalens_amd.profiles.mass.sis import SIS
    from gigalens_amd.simulator import LensSimulator, SimulatorConfig
    UserSIS, _ = _profiles()

    class UserSersic(LightProfile):
        _name, _params, _amp = "USER_SERSIC", ["R_sersic", "n_sersic", "center_x", "center_y"], "Ie"
        hip_body = SERSIC_BODY

    J, S = tfd.JointDistributionNamed, tfd.JointDistributionSequential
    prior = J(dict(lens_mass=S([J(dict(theta_E=tfd.LogNormal(math.log(1.0), 0.1), center_x=tfd.Normal(0, 0.03), center_y=tfd.Normal(0, 0.03)))]),
                   source_light=S([J(dict(R_sersic=tfd.LogNormal(math.log(0.3), 0.1), n_sersic=tfd.Uniform(1, 3), center_x=tfd.Normal(0, 0.1),
                                          center_y=tfd.Normal(0, 0.1), Ie=tfd.LogNormal(math.log(40.0), 0.2)))])))
    cfg = SimulatorConfig(delta_pix=0.08, num_pix=32)
    B = 8
    phys_u, phys_b = PhysicalModel([UserSIS()], [], [UserSersic()]), PhysicalModel([SIS()], [], [Sersic()])
    sim_u, sim_b = LensSimulator(phys_u, cfg, bs=B), LensSimulator(phys_b, cfg, bs=B)
    truth = prior.sample(1, seed=3)
    obs = sim_b.simulate({g: [{k: v.expand(B) for k, v in d.items()} for d in lst] for g, lst in truth.items()})[0]
    obs = (obs + 0.3 * torch.randn_like(obs)).cpu().numpy()
    pm = ForwardProbModel(prior, obs, 0.3, 100.0, include_positions=False)
    z = pm.bij.inverse(prior.sample(B, seed=4)).to("cuda")
    zu, zb = z.clone().requires_grad_(True), z.clone().requires_grad_(True)
    lp_u, _ = pm.log_prob(sim_u, zu)
    lp_b, _ = pm.log_prob(sim_b, zb)
    lp_u.sum().backward()
    lp_b.sum().backward()
    assert torch.allclose(lp_u, lp_b, rtol=2e-5)
    sc = zb.grad.abs().amax(dim=0, keepdim=True)
    assert torch.all((zu.grad - zb.grad).abs() <= 2e-3 * sc + 1e-5)
    seq = ModellingSequence(phys_u, pm, cfg)
    sol = seq.MAP(Adam(2e-2), None, n_samples=16, num_steps=40, seed=1)
    lp_end, _ = pm.log_prob(LensSimulator(phys_u, cfg, bs=16), sol)
    lp_start, _ = pm.log_prob(LensSimulator(phys_u, cfg, bs=16), pm.bij.inverse(prior.sample(16, seed=1)).to("cuda"))
    assert float(lp_end.max()) > float(lp_start.max())


def test_user_written_lens_light_beside_built_in_kinds(gl):
    """A user-written profile mixes with built-in components: EPL + Shear lens (the wavefront-per-sample front end and the
    cost-ordered dispatch run), a user-written Sersic as LENS LIGHT, a built-in Sersic source, one body shared by two components --
    against the all-built-in twin."""
    from gigalens_amd.model import PhysicalModel
    from gigalens_amd.profile import LightProfile
    from gigalens_amd.profiles.light.sersic import Sersic
    from gigalens_amd.profiles.mass.epl import EPL
    from gigalens_amd.profiles.mass.shear import Shear
    from gigalens_amd.simulator import LensSimulator, SimulatorConfig

    class UserSersic(LightProfile):
        _name, _params, _amp = "USER_SERSIC", ["R_sersic", "n_sersic", "center_x", "center_y"], "Ie"
        hip_body = SERSIC_BODY

    cfg = SimulatorConfig(delta_pix=0.08, num_pix=36)
    B = 5
    sim_u = LensSimulator(PhysicalModel([EPL(), Shear()], [UserSersic()], [Sersic(), UserSersic()]), cfg, bs=B)
    sim_b = LensSimulator(PhysicalModel([EPL(), Shear()], [Sersic()], [Sersic(), Sersic()]), cfg, bs=B)
    r = np.random.default_rng(7)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    ser = lambda rs, ie: dict(R_sersic=t(r.uniform(0.8, 1.2, B) * rs), n_sersic=t(r.uniform(1.0, 3.0, B)), center_x=t(r.normal(0, 0.05, B)),
                              center_y=t(r.normal(0, 0.05, B)), Ie=t(r.uniform(0.8, 1.2, B) * ie))
    params = {"lens_mass": [dict(theta_E=t(r.uniform(0.9, 1.2, B)), gamma=t(r.uniform(1.8, 2.2, B)), e1=t(r.normal(0.1, 0.05, B)),
                                 e2=t(r.normal(-0.05, 0.05, B)), center_x=t(r.normal(0, 0.03, B)), center_y=t(r.normal(0, 0.03, B))),
                            dict(gamma1=t(r.normal(0, 0.03, B)), gamma2=t(r.normal(0, 0.03, B)))],
              "lens_light": [ser(0.8, 30.0)], "source_light": [ser(0.25, 40.0), ser(0.15, 20.0)]}
    img_u, img_b = sim_u.simulate(params), sim_b.simulate(params)
    top = float(img_b.abs().max())
    assert torch.allclose(img_u, img_b, rtol=1e-4, atol=3e-5 * top), float((img_u - img_b).abs().max()) / top
    pu = sim_u.pack(params)
    assert torch.equal(pu, sim_b.pack(params))
    obs = img_b[0] + 0.5 * t(r.normal(size=(36, 36)))
    ll_u, _, g_u = sim_u._model.loglike(pu, obs, None, None, 0.5, 100.0, True)
    ll_b, _, g_b = sim_b._model.loglike(pu, obs, None, None, 0.5, 100.0, True)
    assert torch.allclose(ll_u, ll_b, rtol=3e-5)
    scale = g_b.abs().amax(dim=0, keepdim=True)
    assert torch.all((g_u - g_b).abs() <= 3e-3 * scale + 1e-6), ((g_u - g_b).abs() / scale).max()


def test_models_with_the_same_bodies_share_one_compile_and_need_no_source_tree(gl, tmp_path, monkeypatch):
    """gl_model_create_user compiles the interpreter with the model's bodies through hiprtc (seconds).  A modelling sequence builds a
    LensSimulator per stage and per batch size: models with the same program text must share ONE compile per process
    (csrc/gl_user.hip: code-object cache keyed on the text), and the kernel headers the compile includes come from inside the
    library (embedded by __graft_entry__.build()), not from a source checkout -- the compile below runs with the working
    directory somewhere else and no GIGALENS_HIP_CSRC.  Ref: src/gigalens/profile.py:58-82 (the extension point),
    src/gigalens/tf/inference.py:17-302 (one simulator per stage)."""
    from gigalens_amd import _native
    from gigalens_amd.model import PhysicalModel
    from gigalens_amd.profile import MassProfile
    from gigalens_amd.profiles.light.sersic import Sersic
    from gigalens_amd.simulator import LensSimulator, SimulatorConfig

    class UserSIS2(MassProfile):  # a body no other test compiles (its text differs by a comment): the first model must compile
        _name, _params = "USER_SIS2", ["theta_E", "center_x", "center_y"]
        hip_body = "// cache test\n" + SIS_BODY

    monkeypatch.delenv("GIGALENS_HIP_CSRC", raising=False)
    monkeypatch.chdir(tmp_path)
    L = _native.lib()
    n0 = L.gl_user_model_compile_count()
    cfg = SimulatorConfig(delta_pix=0.08, num_pix=24)
    sims = [LensSimulator(PhysicalModel([UserSIS2()], [], [Sersic()]), cfg, bs=b) for b in (3, 3, 7)]
    assert L.gl_user_model_compile_count() == n0 + 1
    r = np.random.default_rng(2)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    imgs = []
    for s in sims[:2]:
        params = {"lens_mass": [dict(theta_E=t([1.0, 1.1, 0.9]), center_x=t([0.0, 0.02, -0.03]), center_y=t([0.01, 0.0, 0.02]))],
                  "source_light": [dict(R_sersic=t([0.3, 0.25, 0.35]), n_sersic=t([1.5, 2.0, 1.0]), center_x=t([0.05, 0.0, -0.05]),
                                        center_y=t([0.0, 0.03, 0.02]), Ie=t([30.0, 40.0, 50.0]))]}
        imgs.append(s.simulate(params))
    assert torch.isfinite(imgs[0]).all() and torch.equal(imgs[0], imgs[1])
    assert "run-time compiled" in sims[0]._model.last_main_kernel()


def test_linear_amplitude_solve_with_a_user_written_light(gl):
    """lstsq_simulate / BackwardProbModel (tf/simulator.py:158-240, tf/model.py:197-273) on a model whose source is a USER-written
    light with ``use_lstsq=True``: its unit-amplitude basis image comes from the run-time compiled basis-stack kernel
    (gl_main_kernel<IMG_BASIS> with the body inside; the amplitude column is the body's last parameter, gl_component::reserved),
    the normal matrix / solve / envelope gradient are the built-in ones.  Must equal the same model of built-in kinds."""
    from gigalens_amd.model import BackwardProbModel, PhysicalModel
    from gigalens_amd.profile import LightProfile
    from gigalens_amd.profiles.light.sersic import Sersic
    from gigalens_amd.profiles.mass.shear import Shear
    from gigalens_amd.profiles.mass.sis import SIS
    from gigalens_amd.simulator import LensSimulator, SimulatorConfig
    UserSIS, _ = _profiles()

    class UserSersic(LightProfile):
        _name, _params, _amp = "USER_SERSIC", ["R_sersic", "n_sersic", "center_x", "center_y"], "Ie"
        hip_body = SERSIC_BODY

    cfg = SimulatorConfig(delta_pix=0.08, num_pix=40)
    B = 5
    sim_u = LensSimulator(PhysicalModel([UserSIS(), Shear()], [], [UserSersic(use_lstsq=True)]), cfg, bs=B)
    sim_b = LensSimulator(PhysicalModel([SIS(), Shear()], [], [Sersic(use_lstsq=True)]), cfg, bs=B)
    r = np.random.default_rng(7)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    params = {"lens_mass": [dict(theta_E=t(r.uniform(0.8, 1.2, B)), center_x=t(r.normal(0, 0.05, B)), center_y=t(r.normal(0, 0.05, B))),
                            dict(gamma1=t(r.normal(0, 0.03, B)), gamma2=t(r.normal(0, 0.03, B)))],
              "source_light": [dict(R_sersic=t(r.uniform(0.2, 0.4, B)), n_sersic=t(r.uniform(1.0, 3.0, B)),
                                    center_x=t(r.normal(0.05, 0.1, B)), center_y=t(r.normal(0, 0.1, B)))]}
    truth = {"lens_mass": params["lens_mass"], "source_light": [dict(params["source_light"][0], Ie=t(np.full(B, 40.0)))]}
    full = LensSimulator(PhysicalModel([SIS(), Shear()], [], [Sersic()]), cfg, bs=B)
    obs = (full.simulate(truth)[0] + 0.3 * t(r.normal(size=(40, 40)))).contiguous()
    err = torch.full_like(obs, 0.3)
    st_u, st_b = sim_u.lstsq_simulate(params, obs, err, return_stacked=True), sim_b.lstsq_simulate(params, obs, err, return_stacked=True)
    assert st_u.shape == st_b.shape
    assert torch.allclose(st_u, st_b, rtol=1e-4, atol=2e-5 * float(st_b.abs().max()))
    c_u, c_b = sim_u.lstsq_simulate(params, obs, err, return_coeffs=True), sim_b.lstsq_simulate(params, obs, err, return_coeffs=True)
    assert c_u.shape == c_b.shape == (B, 1) and torch.allclose(c_u, c_b, rtol=2e-4)
    # sample 0 holds the truth's nonlinear parameters: the solve recovers its amplitude (the coefficient multiplies the rendered
    # basis image, which carries det(T) = delta_pix^2: tf/simulator.py:156)
    assert abs(float(c_b[0, 0]) / (40.0 * 0.08 ** 2) - 1.0) < 0.1
    im_u, im_b = sim_u.lstsq_simulate(params, obs, err), sim_b.lstsq_simulate(params, obs, err)
    assert torch.allclose(im_u, im_b, rtol=2e-4, atol=2e-5 * float(im_b.abs().max()))


@pytest.mark.parametrize("base", ["NFW", "SIS", "SIE", "USER_SIS"])
def test_population_over_any_base_profile_inside_a_model(gl, base):
    """scaling_relation.py:8-19 wraps ANY MassProfile, and a PhysicalModel may hold the population.  Outside the dPIE family the
    population becomes one run-time compiled lens: the member loop around the base profile's body (its own `hip_body`, or the
    restated body of a built-in kind), catalogue as constants, gradient with respect to the scales from duals.  Against the
    oracle's sum over members in float64 (oracle/ref_torch.py scaled_deriv): image, image VJP, log-likelihood and gradient; a
    population of user-written members must equal the population of the built-in kind."""
    from oracle import ref_torch as ref
    from tests import helpers as H
    from gigalens_amd.model import PhysicalModel
    from gigalens_amd.profiles.light.sersic import Sersic
    from gigalens_amd.profiles.mass.epl import EPL
    from gigalens_amd.profiles.mass.nfw import NFW
    from gigalens_amd.profiles.mass.sis import SIS
    from gigalens_amd.profiles.mass.scaling_relation import ScalingRelation
    from gigalens_amd.simulator import LensSimulator, SimulatorConfig
    UserSIS, _ = _profiles()
    r = np.random.default_rng(11)
    G, B, n = 9, 4, 36
    cat = dict(lum=r.uniform(0.3, 2.0, G).astype(np.float32), center_x=r.uniform(-1.3, 1.3, G).astype(np.float32),
               center_y=r.uniform(-1.3, 1.3, G).astype(np.float32))
    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    if base == "NFW":
        pop = ScalingRelation(NFW(), ["Rs", "alpha_Rs"], 1.2, {"Rs": 0.4, "alpha_Rs": 0.6}, cat)
        scales = dict(Rs=t(r.uniform(0.3, 0.6, B)), alpha_Rs=t(r.uniform(0.05, 0.12, B)))
        twin = None
    elif base == "SIE":
        from gigalens_amd.profiles.mass.sie import SIE
        cat = dict(cat, e1=r.normal(0, 0.15, G).astype(np.float32), e2=r.normal(0, 0.15, G).astype(np.float32))
        pop = ScalingRelation(SIE(), ["theta_E"], 1.2, {"theta_E": 0.5}, cat)
        scales = dict(theta_E=t(r.uniform(0.04, 0.08, B)))
        twin = None
    else:
        pop = ScalingRelation(UserSIS() if base == "USER_SIS" else SIS(), ["theta_E"], 1.2, {"theta_E": 0.5}, cat)
        scales = dict(theta_E=t(r.uniform(0.04, 0.08, B)))
        twin = ScalingRelation(SIS(), ["theta_E"], 1.2, {"theta_E": 0.5}, cat)
    assert pop._generic and pop.hip_body
    cfg = SimulatorConfig(delta_pix=0.08, num_pix=n)
    phys = PhysicalModel([EPL(), pop], [], [Sersic()])
    sim = LensSimulator(phys, cfg, bs=B)
    params = {"lens_mass": [dict(theta_E=t(r.uniform(0.9, 1.1, B)), gamma=t(r.uniform(1.9, 2.1, B)), e1=t(r.normal(0.05, 0.03, B)),
                                 e2=t(r.normal(-0.03, 0.03, B)), center_x=t(r.normal(0, 0.03, B)), center_y=t(r.normal(0, 0.03, B))), scales],
              "source_light": [dict(R_sersic=t(r.uniform(0.2, 0.3, B)), n_sersic=t(r.uniform(1.5, 2.5, B)), center_x=t(r.normal(0.05, 0.05, B)),
                                    center_y=t(r.normal(0, 0.05, B)), Ie=t(r.uniform(20, 40, B)))]}
    packed = sim.pack(params)
    # the oracle knows the population by the reference's name of its base profile
    phys_o = phys if twin is None else PhysicalModel([EPL(), twin], [], [Sersic()])
    rs = ref.RefSimulator(phys_o, cfg, B, dtype=torch.float64)
    p64 = packed.cpu().double().requires_grad_(True)
    img_o = rs.simulate(H.struct_from_packed(phys_o, p64))
    img = sim.simulate(packed)
    top = float(img_o.detach().abs().max())
    assert np.allclose(img.cpu().numpy(), img_o.detach().numpy(), rtol=1e-4, atol=3e-5 * top), float((img.cpu() - img_o.detach()).abs().max()) / top
    # image VJP
    w = torch.tensor(r.normal(size=(B, n, n)), dtype=torch.float32)
    (g_o,) = torch.autograd.grad((img_o * w.double()).sum(), p64, retain_graph=True)
    v = sim._model.simulate_bwd(packed, w.cuda())
    S = g_o.abs().amax(dim=0, keepdim=True).numpy()
    assert np.all(np.abs(v.cpu().numpy() - g_o.numpy()) <= 2e-3 * S + 1e-6), (np.abs(v.cpu().numpy() - g_o.numpy()) / S).max()
    # fused log-likelihood and gradient
    obs = (img_o[0].detach() + 0.5 * torch.tensor(r.normal(size=(n, n)))).float()
    ll, chi, g = sim._model.loglike(packed, obs.cuda(), None, None, 0.5, 100.0, True)
    ll_o, _ = ref.stats_pixels(rs, H.struct_from_packed(phys_o, p64), obs.numpy(), 0.5, 100.0)
    (gl_o,) = torch.autograd.grad(ll_o.sum(), p64)
    assert np.allclose(ll.cpu().numpy(), ll_o.detach().numpy(), rtol=2e-5)
    S = gl_o.abs().amax(dim=0, keepdim=True).numpy()
    assert np.all(np.abs(g.cpu().numpy() - gl_o.numpy()) <= 2e-3 * S + 1e-6), (np.abs(g.cpu().numpy() - gl_o.numpy()) / S).max()
    # the scales' columns carry a real gradient
    lo = len(EPL().params)
    assert np.all(np.abs(gl_o.numpy()[:, lo:lo + len(scales)]).max(axis=0) > 0)
    # lens maps of the population through the plugin-level sum (tf/simulator.py:72-107)
    xs, ys = sim.img_X[:50, None], sim.img_Y[:50, None]
    kap = sim.convergence(xs, ys, params["lens_mass"])
    assert kap.shape == (50, B) and bool(torch.isfinite(kap).all())


def test_image_position_likelihood_with_user_written_lenses(gl):
    """tf/model.py:103-124 on a model whose lens is user-written: beta and the magnification at the image positions need the body's
    Hessian and, for the gradient, its mixed second derivatives -- the four position kernels are compiled at run time with the body
    on the nested duals (Dual<float, 2>, Dual<Dual<float, 1>, 2>).  Same numbers as the built-in twin; a population of user-written
    members (the generated member loop) against the oracle."""
    import math
    from oracle import ref_torch as ref
    from tests import helpers as H
    from gigalens_amd import prior as tfd
    from gigalens_amd.model import ForwardProbModel, PhysicalModel
    from gigalens_amd.profiles.light.sersic import Sersic
    from gigalens_amd.profiles.mass.scaling_relation import ScalingRelation
    from gigalens_amd.profiles.mass.shear import Shear
    from gigalens_amd.profiles.mass.sis import SIS
    from gigalens_amd.simulator import LensSimulator, SimulatorConfig
    UserSIS, _ = _profiles()
    J, S = tfd.JointDistributionNamed, tfd.JointDistributionSequential
    r = np.random.default_rng(21)
    cat = dict(lum=r.uniform(0.5, 1.5, 5).astype(np.float32), center_x=r.uniform(-1.5, 1.5, 5).astype(np.float32),
               center_y=r.uniform(-1.5, 1.5, 5).astype(np.float32))
    pop_u = ScalingRelation(UserSIS(), ["theta_E"], 1.0, {"theta_E": 0.5}, cat)
    pop_b = ScalingRelation(SIS(), ["theta_E"], 1.0, {"theta_E": 0.5}, cat)
    prior = J(dict(lens_mass=S([J(dict(theta_E=tfd.LogNormal(math.log(1.2), 0.1), center_x=tfd.Normal(0, 0.05), center_y=tfd.Normal(0, 0.05))),
                                J(dict(gamma1=tfd.Normal(0, 0.05), gamma2=tfd.Normal(0, 0.05))),
                                J(dict(theta_E=tfd.LogNormal(math.log(0.05), 0.2)))]),
                   source_light=S([J(dict(R_sersic=tfd.LogNormal(math.log(0.3), 0.1), n_sersic=tfd.Uniform(1, 3), center_x=tfd.Normal(0, 0.1),
                                          center_y=tfd.Normal(0, 0.1), Ie=tfd.LogNormal(math.log(40.0), 0.2)))])))
    cfg = SimulatorConfig(delta_pix=0.08, num_pix=16)
    B = 5
    phys_u = PhysicalModel([UserSIS(), Shear(), pop_u], [], [Sersic()])
    phys_b = PhysicalModel([SIS(), Shear(), pop_b], [], [Sersic()])
    sim_u, sim_b = LensSimulator(phys_u, cfg, bs=B), LensSimulator(phys_b, cfg, bs=B)
    cx = [np.array([1.25, -1.15, 0.2, -0.3], np.float32), np.array([0.9, -0.95], np.float32)]
    cy = [np.array([0.3, -0.25, 1.2, -1.22], np.float32), np.array([-0.9, 0.95], np.float32)]
    ex = [np.full(4, 0.01, np.float32), np.full(2, 0.02, np.float32)]
    pm = ForwardProbModel(prior, centroids_x=cx, centroids_y=cy, centroids_errors_x=ex, centroids_errors_y=ex,
                          include_pixels=False, include_positions=True)
    packed = sim_u.pack(prior.sample(B, seed=2))
    pu, pb = packed.clone().requires_grad_(True), packed.clone().requires_grad_(True)
    ll_u, red_u = pm.stats_positions(sim_u, pu)
    ll_b, red_b = pm.stats_positions(sim_b, pb)
    ll_u.sum().backward()
    ll_b.sum().backward()
    assert torch.allclose(ll_u, ll_b, rtol=1e-4) and torch.allclose(red_u, red_b, rtol=1e-4)
    sc = pb.grad.abs().amax(dim=1, keepdim=True)
    assert torch.all((pu.grad - pb.grad).abs() <= 2e-3 * torch.maximum(pb.grad.abs(), 1e-2 * sc) + 1e-6)
    # ... and both against the oracle (float64, autograd through the member sum)
    rs = ref.RefSimulator(phys_b, cfg, B, dtype=torch.float64)
    p64 = packed.cpu().double().requires_grad_(True)
    ll_o, red_o = ref.stats_positions(rs, H.struct_from_packed(phys_b, p64), cx, cy, ex, ex)
    (g_o,) = torch.autograd.grad(ll_o.sum(), p64)
    assert np.allclose(ll_u.detach().cpu().numpy(), ll_o.detach().numpy(), rtol=1e-4)
    g, go = pu.grad.cpu().numpy(), g_o.numpy()
    scale = np.abs(go).max(axis=1, keepdims=True)
    assert np.all(np.abs(g - go) <= 2e-3 * np.maximum(np.abs(go), 1e-2 * scale) + 1e-6), (np.abs(g - go) / scale).max()
    # the fused log-prob with both likelihood terms (tf/model.py:126-162) off the same model
    obs = (sim_b.simulate(packed)[0] + 0.3 * torch.randn(16, 16, device="cuda")).cpu().numpy()
    pm2 = ForwardProbModel(prior, obs, 0.3, 100.0, centroids_x=cx, centroids_y=cy, centroids_errors_x=ex, centroids_errors_y=ex,
                           include_pixels=True, include_positions=True)
    z = pm2.bij.inverse(prior.sample(B, seed=2)).to("cuda")
    zu, zb = z.clone().requires_grad_(True), z.clone().requires_grad_(True)
    lp_u, _ = pm2.log_prob(sim_u, zu)
    lp_b, _ = pm2.log_prob(sim_b, zb)
    lp_u.sum().backward()
    lp_b.sum().backward()
    assert torch.allclose(lp_u, lp_b, rtol=1e-4)
    sc = zb.grad.abs().amax(dim=1, keepdim=True)
    assert torch.all((zu.grad - zb.grad).abs() <= 2e-3 * torch.maximum(zb.grad.abs(), 1e-2 * sc) + 1e-5)
