"""Host logic of the inference drivers (no GPU): the written-out reparameterisation gradient of ``svi_step`` against
autograd through the surrogate's sampling path -- what tf/inference.py:85-91 obtains from ``tf.GradientTape``."""
import math

import pytest
import torch

from gigalens_amd import inference as inf


def _toy_log_prob(z):
    a = torch.linspace(0.5, 2.0, z.shape[-1], dtype=z.dtype)
    return -0.5 * ((z - 0.3) ** 2 * a).sum(-1) - 0.1 * torch.sin(z).sum(-1) + 0.05 * (z[..., 0] * z[..., -1])


def _taped(mu, lp, n, gen, full_rank):
    d = mu.numel()
    mu_ = mu.clone().requires_grad_(True)
    lp_ = lp.clone().requires_grad_(True)
    L = inf.tril_unpack(lp_, d) if full_rank else torch.diag(torch.exp(lp_))
    eps = torch.randn((n, d), generator=gen, dtype=mu.dtype)
    z = mu_ + eps @ L.T
    log_q = -0.5 * (eps * eps).sum(-1) - torch.log(torch.diagonal(L)).sum() - 0.5 * d * math.log(2 * math.pi)
    elbo = (log_q - _toy_log_prob(z)).mean()
    g_mu, g_lp = torch.autograd.grad(elbo, (mu_, lp_))
    return elbo.detach(), g_mu, g_lp


@pytest.mark.parametrize("full_rank", [True, False])
@pytest.mark.parametrize("use_vg", [False, True])
def test_svi_step_gradient_equals_the_taped_one(full_rank, use_vg):
    torch.manual_seed(0)
    d, n = 6, 64
    mu = torch.randn(d, dtype=torch.float64) * 0.2
    scale = torch.tril(torch.randn(d, d, dtype=torch.float64) * 0.1) + torch.diag(torch.rand(d, dtype=torch.float64) + 0.2)
    lp = inf.tril_pack(scale) if full_rank else torch.log(torch.diagonal(scale))
    want = _taped(mu, lp, n, torch.Generator().manual_seed(5), full_rank)

    def vg(z):
        zz = z.clone().requires_grad_(True)
        v = _toy_log_prob(zz)
        (g,) = torch.autograd.grad(v.sum(), zz)
        return v.detach(), g

    got = inf.svi_step(mu, lp, None if use_vg else _toy_log_prob, n, torch.Generator().manual_seed(5),
                       value_and_grad_fn=vg if use_vg else None)
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=1e-10, atol=1e-12)


def test_adam_scale_argument_on_cpu():
    """CPU tensors take the torch formula (the gloo tests of the sharding logic); ``scale`` multiplies the gradient."""
    x1, x2 = torch.ones(5), torch.ones(5)
    g = torch.tensor([0.1, -0.2, 0.3, 0.0, 1.0])
    o1, o2 = inf.Adam(1e-2), inf.Adam(1e-2)
    for _ in range(3):
        o1.step(x1, g * -0.5)
        o2.step(x2, g, -0.5)
    assert torch.allclose(x1, x2) and o1.t == o2.t == 3


# ---- HMC driver: trajectory-length adaptation (tfe.mcmc.GradientBasedTrajectoryLengthAdaptation, tf/inference.py:150-154) ----
class _ToyModel:
    """A closed-form log-density behind the ForwardProbModel surface the drivers use (the lens likelihood needs a GPU)."""
    device = torch.device("cpu")
    include_pixels, include_positions, n_position = True, False, 0.0

    def __init__(self, sigma):
        self.sigma = torch.as_tensor(sigma, dtype=torch.float32)

    def init_centroids(self, bs):
        return None

    def log_prob_and_grad(self, simulator, z):
        lp = -0.5 * ((z / self.sigma) ** 2).sum(-1)
        return lp, torch.zeros_like(lp), -z / self.sigma ** 2


def _hmc(monkeypatch, sigma, **kw):
    monkeypatch.setattr(inf, "LensSimulator", lambda *a, **k: None)
    d = len(sigma)
    seq = inf.ModellingSequence(None, _ToyModel(sigma), None)
    return seq.HMC((torch.zeros(d), torch.diag(torch.as_tensor(sigma, dtype=torch.float32))), **kw)


def test_halton_sequence():
    assert [inf._halton2(i) for i in range(1, 8)] == [0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875]


def test_hmc_trajectory_length_adapts_to_the_known_optimum(monkeypatch):
    """Standard normal in the preconditioned coordinates: ChEES per unit mass is d sin^2(t), so with lengths h T, h uniform on
    (0, 1), the criterion is maximal where tan 2T = 2T, T = 2.2467 -- in the limit of small steps; the integration time is a
    whole number of steps, ceil(h T / eps) eps, so the adapted T approaches that value from below as eps shrinks (measured:
    1.47 at eps = 0.92, 1.90 at 0.61, 2.00 at 0.36, 2.17 at 0.18).  Starting from T = eps * L = 0.2 the adaptation must get
    there (the step-size adaptation runs beside it), keep the acceptance at the target, and the chain must sample the target."""
    sigma = [0.5, 1.0, 2.0, 4.0, 1.0, 1.0, 3.0, 0.7, 1.5, 1.0]
    samples, st = _hmc(monkeypatch, sigma, init_eps=0.1, init_l=2, n_hmc=256, num_burnin_steps=600, num_results=200,
                       max_leapfrog_steps=200, seed=11, target_accept=0.97)
    T = st["max_trajectory_length"]
    assert 1.8 < T < 2.4, T
    assert 0.93 < sum(st["accept"][-200:]) / 200 < 0.995
    assert samples.shape == (200, 256, 10)
    sd = samples.reshape(-1, 10).std(0)
    assert torch.allclose(sd, torch.tensor(sigma), rtol=0.1)
    # the jitter spreads the number of leapfrog steps between 1 and ceil(T / eps)
    late = st["num_leapfrog_steps"][-200:]
    assert min(late) >= 1 and max(late) <= math.ceil(T / st["step_size"]) and len(set(late)) > 3


def test_hmc_honours_max_leapfrog_steps(monkeypatch):
    sigma = [1.0] * 6
    samples, st = _hmc(monkeypatch, sigma, init_eps=0.05, init_l=9, n_hmc=64, num_burnin_steps=60, num_results=10,
                       max_leapfrog_steps=4, seed=2)
    assert max(st["num_leapfrog_steps"]) <= 4 and min(st["num_leapfrog_steps"]) >= 1
    assert st["max_trajectory_length"] <= 4 * st["step_size"] * 1.5  # kept near the cap eps * max_leapfrog_steps
    # adaptation stops after int(0.8 * burn-in) transitions (tf/inference.py:140): the step size is frozen afterwards
    s2, st2 = _hmc(monkeypatch, sigma, init_eps=0.05, init_l=2, n_hmc=64, num_burnin_steps=0, num_results=5, seed=2)
    assert st2["step_size"] == pytest.approx(0.05) and set(st2["num_leapfrog_steps"]) <= {1, 2}
