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
