"""The reference's pipeline smoke tests (tests/tf/test_model.py:29-72: test_map, test_vi, test_hmc) against
this repo's ModellingSequence, on the GPU: lr = 0 leaves parameters unchanged, lr > 0 moves them, HMC returns
num_results samples; plus: MAP actually descends on a synthetic observation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from gigalens_amd import workloads
    from gigalens_amd.inference import ModellingSequence
    from gigalens_amd.model import ForwardProbModel
    from gigalens_amd.simulator import LensSimulator
    wl = workloads.make("C2", num_pix=20, batch=2)
    obs, _, truth = workloads.synthetic_observation(wl, LensSimulator)
    pm = ForwardProbModel(wl.prior, obs.cpu().numpy(), wl.background_rms, wl.exp_time, include_positions=False)
    return wl, pm, ModellingSequence(wl.phys_model, pm, wl.sim_config)


def test_map(setup):
    from gigalens_amd.inference import Adam
    from gigalens_amd.prior import nest_flatten
    wl, pm, seq = setup
    start = pm.prior.sample(2, seed=0)
    ret = seq.MAP(Adam(0.0), start, n_samples=2, num_steps=5, seed=0)
    end = pm.bij.forward(ret)
    for a, b in zip(nest_flatten(start), nest_flatten(end)):
        assert np.allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-6)
    ret = seq.MAP(Adam(1e-3), start, n_samples=2, num_steps=5, seed=0)
    end = pm.bij.forward(ret)
    assert not all(np.allclose(a.cpu().numpy(), b.cpu().numpy()) for a, b in zip(nest_flatten(start), nest_flatten(end)))


def test_map_descends(setup):
    from gigalens_amd.inference import Adam
    wl, pm, seq = setup
    hist = []
    seq.MAP(Adam(5e-2), None, n_samples=64, num_steps=60, seed=1, progress=lambda s, red: hist.append(float(red.mean())))
    assert hist[-1] < 0.5 * hist[0] and hist[-1] < 1.5  # mean reduced chi^2 falls to the noise floor (~1)


def test_vi_and_hmc(setup):
    from gigalens_amd.inference import Adam
    wl, pm, seq = setup
    start = pm.bij.inverse(pm.prior.sample(2, seed=0))[0]
    (mean, L), losses = seq.SVI(Adam(0.0), start, n_vi=5, num_steps=5)
    assert torch.allclose(mean.cpu(), start.cpu())
    (mean, L), losses = seq.SVI(Adam(1e-3), start, n_vi=5, num_steps=5)
    assert not torch.allclose(mean.cpu(), start.cpu()) and len(losses) == 5 and np.all(np.isfinite(losses))
    samples, stats = seq.HMC((mean, L), n_hmc=3, init_eps=0.3, init_l=3, max_leapfrog_steps=5, num_burnin_steps=3,
                             num_results=5)
    assert len(samples) == 5 and samples.shape == (5, 3, 13) and torch.isfinite(samples).all()
    # mean-field surrogate (full_rank=False) and the 'simple' step-size adaptation (tf/inference.py:47-48,159-164)
    (mean_d, L_d), losses = seq.SVI(Adam(1e-3), start, n_vi=5, num_steps=4, full_rank=False)
    assert torch.allclose(L_d, torch.diag(torch.diagonal(L_d))) and len(losses) == 4 and np.all(np.isfinite(losses))
    samples, stats = seq.HMC((mean_d, L_d), n_hmc=2, num_burnin_steps=3, num_results=2, adapt_mode="simple")
    assert samples.shape == (2, 2, 13)
    with pytest.raises(ValueError):
        seq.HMC((mean_d, L_d), adapt_mode="nope")


def test_smc(setup):
    """ModellingSequence.SMC (tf/inference.py:184-288): the temperature ladder reaches 1, the particles end at the
    noise floor, the evidence is finite and two independent ensembles agree on it to Monte-Carlo accuracy."""
    wl, pm, seq = setup
    samples, info = seq.SMC(num_particles=96, num_ensembles=2, num_leapfrog_steps=4, post_sampling_steps=0,
                            max_sampling_per_stage=3, target="pixels", auxiliar="none", seed=4, max_stage=200)
    assert samples.shape == (96, 2, 13) and torch.isfinite(samples).all()
    betas = np.array([s["beta"] for s in info["stages"]])
    assert np.all(np.diff(betas, axis=0) >= 0) and np.allclose(betas[-1], 1.0)
    from gigalens_amd.simulator import LensSimulator
    sim = LensSimulator(wl.phys_model, wl.sim_config, bs=192)
    _, red = pm.log_prob(sim, samples.permute(1, 0, 2).reshape(192, 13))
    assert float(red.median()) < 1.6
    lz = info["log_evidence"].numpy()
    assert np.all(np.isfinite(lz)) and abs(lz[0] - lz[1]) < 0.05 * abs(lz).max() + 8.0
    chain, _ = seq.SMC(num_particles=32, num_leapfrog_steps=3, post_sampling_steps=4, max_sampling_per_stage=2,
                       target="pixels", auxiliar="none", seed=5, max_stage=200)
    assert chain.shape == (4, 32, 13) and torch.isfinite(chain).all()


def test_map_graph_matches_stepwise(setup):
    """The HIP-graph replay of one MAP step (native launch sequence + Adam) walks the same trajectory as launching every
    step, and leaves the optimiser's step counter where the loop would."""
    from gigalens_amd.inference import Adam
    wl, pm, seq = setup
    start = pm.prior.sample(16, seed=3)
    o1, o2 = Adam(2e-2), Adam(2e-2)
    r1 = seq.MAP(o1, start, n_samples=16, num_steps=40, seed=0, graph=False)
    red1 = seq.last_red_chi2.clone()
    r2 = seq.MAP(o2, start, n_samples=16, num_steps=40, seed=0, graph=True)
    red2 = seq.last_red_chi2.clone()
    assert o1.t == o2.t == 40
    assert torch.allclose(r1, r2, rtol=2e-4, atol=2e-5)
    assert torch.allclose(red1, red2, rtol=1e-3)
    # a learning-rate schedule cannot be captured: falls back to stepwise launches
    o3 = Adam(lambda t: 2e-2)
    r3 = seq.MAP(o3, start, n_samples=16, num_steps=40, seed=0, graph=True)
    assert torch.allclose(r1, r3, rtol=1e-6, atol=1e-7)


def test_fused_adam_matches_the_formula():
    """gl_adam_update (one launch) against the textbook update in float64, host step count and device counter."""
    from gigalens_amd.inference import Adam
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(700, 13, generator=g)
    grads = [torch.randn(700, 13, generator=g) * (10.0 ** torch.randint(-3, 3, (1,), generator=g)) for _ in range(6)]
    lr, b1, b2, eps, scale = 3e-2, 0.9, 0.999, 1e-7, -0.25
    x, m, v = x0.double().clone(), torch.zeros_like(x0).double(), torch.zeros_like(x0).double()
    for t, gr in enumerate(grads, 1):
        gg = gr.double() * scale
        m = b1 * m + (1 - b1) * gg
        v = b2 * v + (1 - b2) * gg * gg
        x = x - lr * (m / (1 - b1 ** t)) / (torch.sqrt(v / (1 - b2 ** t)) + eps)
    for captured in (False, True):
        opt = Adam(lr, b1, b2, eps)
        xd = x0.cuda().clone()
        for gr in grads:
            (opt.step_captured if captured else opt.step)(xd, gr.cuda(), scale)
        if captured:
            opt.sync_from_device()
        assert opt.t == len(grads)
        assert torch.allclose(xd.cpu().double(), x, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("full_rank,d", [(True, 13), (False, 13), (True, 1), (False, 1), (True, 40), (True, 132), (False, 132)])
def test_native_svi_surrogate_matches_the_torch_path(full_rank, d):
    """gl_svi_sample / gl_svi_grad (the two launches around the forward+gradient call) against the torch formulation of
    the same step on identical draws: ELBO, d/dmu and d/d(packed scale), Exp diagonal and diag_shift included."""
    from gigalens_amd import inference as inf
    dev = "cuda"
    n = 777
    g = torch.Generator().manual_seed(1)
    mu = (torch.randn(d, generator=g) * 0.3).to(dev)
    scale = torch.tril(torch.randn(d, d, generator=g) * 0.05, diagonal=-1) + torch.diag(torch.rand(d, generator=g) * 0.2 + 0.05)
    lp = (inf.tril_pack(scale) if full_rank else torch.log(torch.diagonal(scale))).to(dev)
    a = torch.linspace(0.5, 2.0, d, device=dev)

    def log_p(z):
        return -0.5 * ((z - 0.3) ** 2 * a).sum(-1) - 0.1 * torch.sin(z).sum(-1) + 0.05 * z[..., 0] * z[..., -1]

    def vg(z):
        zz = z.clone().requires_grad_(True)
        v = log_p(zz)
        (gr,) = torch.autograd.grad(v.sum(), zz)
        return v.detach(), gr

    gen1 = torch.Generator(device=dev).manual_seed(9)
    gen2 = torch.Generator(device=dev).manual_seed(9)
    got = inf.svi_step(mu, lp, None, n, gen1, value_and_grad_fn=vg, full_rank=full_rank)   # native surrogate kernels
    want = inf.svi_step(mu, lp, log_p, n, gen2, full_rank=full_rank)                        # torch formulation
    assert got[1].shape == (d,) and got[2].shape == lp.shape
    assert got[2].numel() == (d * (d + 1) // 2 if full_rank else d)  # d = 132 full rank: the 8 911-float buffer of config 5
    for x, y in zip(got, want):
        assert torch.allclose(x, y, rtol=2e-4, atol=2e-5 * float(y.abs().max() + 1))


@pytest.mark.parametrize("d", [13, 64, 132, 300])
def test_hmc_kernels_match_the_torch_leapfrog(d):
    """gl_hmc_kick_drift / gl_hmc_accept against the torch formulation of the same leapfrog pieces (tf/inference.py:95-182:
    momentum precision = the surrogate covariance), non-finite proposals rejected; d = 132 is the cluster model of
    BASELINE config 5."""
    from gigalens_amd import _native
    g0 = torch.Generator().manual_seed(4)
    n = 300
    L = (torch.tril(torch.randn(d, d, generator=g0) * (0.1 / (d / 13) ** 0.5)) + torch.diag(torch.rand(d, generator=g0) + 0.3)).cuda()
    Sigma = (L @ L.T).contiguous()
    z, p, gr = (torch.randn(n, d, generator=g0).cuda() for _ in range(3))
    eps, kick = 0.07, 0.035
    p_ref = p + kick * gr
    z_ref = z + eps * (p_ref @ Sigma)
    p_out, z_out = torch.empty_like(p), torch.empty_like(z)
    _native.hmc_kick_drift(p, gr, kick, z, Sigma, eps, p_out, z_out)
    assert torch.allclose(p_out, p_ref, rtol=1e-6, atol=1e-6) and torch.allclose(z_out, z_ref, rtol=1e-5, atol=2e-6)
    pi, zi = p.clone(), z.clone()  # in place
    _native.hmc_kick_drift(pi, gr, kick, zi, Sigma, eps, pi, zi)
    assert torch.equal(pi, p_out) and torch.equal(zi, z_out)
    # Metropolis step
    lp = torch.randn(n, generator=g0).cuda()
    lpn = (lp.cpu() + 0.5 * torch.randn(n, generator=g0)).cuda()
    lpn[5] = float("nan")
    lpn[6] = float("-inf")
    zn, gn, p0, pn = (torch.randn(n, d, generator=g0).cuda() for _ in range(4))
    u = torch.rand(n, generator=g0).cuda()
    p1 = pn + kick * gn
    ke0, ke1 = 0.5 * ((p0 @ L) ** 2).sum(-1), 0.5 * ((p1 @ L) ** 2).sum(-1)
    log_acc = (lpn - ke1) - (lp - ke0)
    log_acc = torch.where(torch.isfinite(log_acc), log_acc, torch.full_like(log_acc, -float("inf")))
    acc = torch.log(u) < log_acc
    margin = (torch.log(u) - log_acc).abs() > 1e-4 * (1 + float(ke1.abs().max()))  # decisions at rounding distance may differ
    zs, gs, lps, accp = z.clone(), gr.clone(), lp.clone(), torch.empty(n, device="cuda")
    _native.hmc_accept(zs, gs, lps, zn, gn, lpn, p0, pn, kick, L.contiguous(), u, accp)
    moved = (zs == zn).all(-1)
    assert torch.equal(moved[margin], acc[margin]) and not moved[5] and not moved[6]
    assert torch.allclose(accp, torch.exp(torch.clamp(log_acc, max=0.0)), rtol=2e-4 * (1 + d / 13), atol=1e-5)
    sel = moved[:, None]
    assert torch.equal(zs, torch.where(sel, zn, z)) and torch.equal(gs, torch.where(sel, gn, gr))
    assert torch.equal(lps[~torch.isnan(lpn)], torch.where(moved, lpn, lp)[~torch.isnan(lpn)])


def test_config5_per_rank_shard():
    """BASELINE config 5 on one rank: the C4 cluster model (8 NFW + 20 Sersic) at 256 x 256 px, 256 particles, full-rank
    Gaussian surrogate of dimension 132.  The native SVI step (gl_svi_sample -> fused forward+gradient -> gl_svi_grad) forms
    the 8 911-float [ELBO, dmu, dL] buffer of the one all-reduce (jax/inference.py:113-128); it is checked against the torch
    formulation of the same step differentiated by autograd through ``log_prob`` on identical draws.  Then the HMC driver
    takes transitions at d = 132 with the native leapfrog kernels."""
    from gigalens_amd import inference as inf
    from gigalens_amd import workloads
    from gigalens_amd.model import ForwardProbModel
    from gigalens_amd.simulator import LensSimulator
    wl = workloads.make("C5")
    assert wl.batch == 256 and wl.sim_config.num_pix == 256
    obs, _, truth = workloads.synthetic_observation(wl, LensSimulator)
    pm = ForwardProbModel(wl.prior, obs.cpu().numpy(), wl.background_rms, wl.exp_time, include_positions=False)
    sim = LensSimulator(wl.phys_model, wl.sim_config, bs=wl.batch)
    n, dev = wl.batch, pm.device
    mu = pm.bij.inverse(truth)[0].to(dev).contiguous()
    d = mu.numel()
    assert d == 132
    g = torch.Generator().manual_seed(5)
    scale = torch.tril(torch.randn(d, d, generator=g) * 2e-4, diagonal=-1) + torch.diag(torch.rand(d, generator=g) * 2e-3 + 5e-4)
    lp = inf.tril_pack(scale).to(dev)
    assert 1 + d + lp.numel() == 8911
    eps = torch.randn(n, d, generator=g).to(dev)

    def vg(z):
        v, _, gr = pm.log_prob_and_grad(sim, z)
        return v, gr

    got = inf.svi_step(mu, lp, None, n, value_and_grad_fn=vg, full_rank=True, eps=eps)
    want = inf.svi_step(mu, lp, lambda z: pm.log_prob(sim, z)[0], n, full_rank=True, eps=eps)
    assert got[2].shape == (8778,) and torch.isfinite(got[0]) and torch.isfinite(got[1]).all() and torch.isfinite(got[2]).all()
    assert torch.allclose(got[0], want[0], rtol=1e-5)
    for a, b in zip(got[1:], want[1:]):
        assert float((a - b).abs().max()) <= 5e-4 * float(b.abs().max())
    # HMC at d = 132: native kick/drift + Metropolis kernels, two-leapfrog transitions from the surrogate
    seq = inf.ModellingSequence(wl.phys_model, pm, wl.sim_config)
    L = inf.tril_unpack(lp, d)
    samples, stats = seq.HMC((mu, L), n_hmc=256, init_eps=0.01, init_l=2, max_leapfrog_steps=4, num_burnin_steps=2,
                             num_results=2, seed=7)
    assert samples.shape == (2, 256, 132) and torch.isfinite(samples).all()
    assert all(0.0 <= a <= 1.0 for a in stats["accept"]) and stats["accept"][0] > 0.05
