"""Host logic: prior tree, tf.nest flatten order, default bijectors (CPU, no GPU needed).

Mirrors the reference's tests/tf/test_model.py:10-26 (test_bij, test_prior) and pins the restated TFP
maths against torch.distributions / scipy (TFP itself is not installed -- "parity unpinned" for TFP's
choice of default bijector per distribution, recorded in DESIGN.md)."""
import math

import numpy as np
import pytest
import torch
from scipy import stats

from gigalens_amd import prior as tfd
from gigalens_amd import workloads


def default_prior():
    """tests/conftest.py:20-73 with the new-API grouping used on this branch."""
    lens = tfd.JointDistributionSequential([
        tfd.JointDistributionNamed(dict(theta_E=tfd.LogNormal(math.log(1.25), 0.25), gamma=tfd.TruncatedNormal(2, 0.25, 1, 3),
                                        e1=tfd.Normal(0, 0.1), e2=tfd.Normal(0, 0.1), center_x=tfd.Normal(0, 0.05),
                                        center_y=tfd.Normal(0, 0.05))),
        tfd.JointDistributionNamed(dict(gamma1=tfd.Normal(0, 0.05), gamma2=tfd.Normal(0, 0.05)))])
    ll = tfd.JointDistributionSequential([tfd.JointDistributionNamed(dict(
        R_sersic=tfd.LogNormal(math.log(1.0), 0.15), n_sersic=tfd.Uniform(2, 6), e1=tfd.TruncatedNormal(0, 0.1, -0.3, 0.3),
        e2=tfd.TruncatedNormal(0, 0.1, -0.3, 0.3), center_x=tfd.Normal(0, 0.05), center_y=tfd.Normal(0, 0.05),
        Ie=tfd.LogNormal(math.log(500.0), 0.3)))])
    src = tfd.JointDistributionSequential([tfd.JointDistributionNamed(dict(
        R_sersic=tfd.LogNormal(math.log(0.25), 0.15), n_sersic=tfd.Uniform(0.5, 4), e1=tfd.TruncatedNormal(0, 0.15, -0.5, 0.5),
        e2=tfd.TruncatedNormal(0, 0.15, -0.5, 0.5), center_x=tfd.Normal(0, 0.25), center_y=tfd.Normal(0, 0.25),
        Ie=tfd.LogNormal(math.log(150.0), 0.5)))])
    return tfd.JointDistributionNamed(dict(lens_mass=lens, lens_light=ll, source_light=src))


def test_bij_round_trip_and_fldj_shape():
    from gigalens_amd.model import ForwardProbModel
    prior = default_prior()
    model = ForwardProbModel(prior, np.ones((20, 20)), 1, 1, include_positions=False)
    sample = prior.sample(5, seed=0)
    z = model.bij.inverse(sample)
    assert z.shape == (5, 22)
    back = model.bij.forward(z)
    for a, b in zip(tfd.nest_flatten(back), tfd.nest_flatten(sample)):
        assert np.allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-6)  # tests/tf/test_model.py:14-17
    det = model.unconstraining_bij.forward_log_det_jacobian(model.pack_bij.forward(z))
    assert det.numel() == 5  # tests/tf/test_model.py:19-26


def test_nest_flatten_order():
    """tf.nest.flatten: dict keys sorted (ASCII), list items in order -- SURVEY.md Appendix B."""
    wl = workloads.make("C2")
    paths = tfd.nest_paths(wl.prior.sample(seed=0))
    names = [p[-1] for p in paths]
    assert names == ["center_x", "center_y", "e1", "e2", "gamma", "theta_E", "gamma1", "gamma2",
                     "Ie", "R_sersic", "center_x", "center_y", "n_sersic"]
    assert [p[0] for p in paths] == ["lens_mass"] * 8 + ["source_light"] * 5
    d = tfd.nest_paths({"source_light": 1, "lens_mass": 2, "lens_light": 3})
    assert [p[0] for p in d] == ["lens_light", "lens_mass", "source_light"]


def test_densities_and_bijectors_vs_torch_and_scipy():
    leaves = [tfd.Normal(0.3, 0.7), tfd.LogNormal(math.log(1.25), 0.25), tfd.Uniform(0.5, 4.0),
              tfd.TruncatedNormal(2.0, 0.25, 1.0, 3.0)]
    flat = tfd.FlatPrior(leaves)
    g = torch.Generator().manual_seed(0)
    z = torch.randn((64, 4), generator=g)
    x = flat.forward(z)
    assert torch.allclose(flat.inverse(x), z, rtol=1e-4, atol=1e-5)
    lp = flat.log_prob_columns(x).double().numpy()
    xd = x.double().numpy()
    assert np.allclose(lp[:, 0], stats.norm(0.3, 0.7).logpdf(xd[:, 0]), rtol=1e-5, atol=1e-6)
    assert np.allclose(lp[:, 1], stats.lognorm(s=0.25, scale=1.25).logpdf(xd[:, 1]), rtol=1e-5, atol=1e-5)
    assert np.allclose(lp[:, 2], stats.uniform(0.5, 3.5).logpdf(xd[:, 2]), rtol=1e-6)
    assert np.allclose(lp[:, 3], stats.truncnorm(-4, 4, loc=2.0, scale=0.25).logpdf(xd[:, 3]), rtol=1e-5, atol=1e-5)
    # log|dx/dz| vs torch's transforms
    T = torch.distributions.transforms
    zz = z.double()
    fl = flat.fldj_columns(z).double()
    assert torch.allclose(fl[:, 0], torch.zeros(64, dtype=torch.float64))
    assert torch.allclose(fl[:, 1], T.ExpTransform().log_abs_det_jacobian(zz[:, 1], zz[:, 1].exp()), atol=1e-6)
    for col, (lo, hi) in ((2, (0.5, 4.0)), (3, (1.0, 3.0))):
        tr = T.ComposeTransform([T.SigmoidTransform(), T.AffineTransform(lo, hi - lo)])
        assert torch.allclose(fl[:, col], tr.log_abs_det_jacobian(zz[:, col], tr(zz[:, col])), atol=1e-5)
    # samples land in the support and have the right moments
    s = flat.sample(20000, seed=1)
    assert (s[:, 2] >= 0.5).all() and (s[:, 2] <= 4.0).all() and (s[:, 3] >= 1.0).all() and (s[:, 3] <= 3.0).all()
    assert abs(float(s[:, 0].mean()) - 0.3) < 0.02 and abs(float(torch.log(s[:, 1]).std()) - 0.25) < 0.01


def test_joint_log_prob_is_sum_of_leaves():
    wl = workloads.make("C1")
    x = wl.prior.sample(7, seed=3)
    lp = wl.prior.log_prob(x)
    flat = wl.prior.flat()
    cols = torch.stack(tfd.nest_flatten(x), dim=-1)
    assert lp.shape == (7,) and torch.allclose(lp, flat.log_prob(cols))
