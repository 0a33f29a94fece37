"""Priors and unconstraining bijectors for ``ForwardProbModel`` (host-side orchestration in torch).

The reference builds its prior from TFP ``JointDistributionNamed`` / ``JointDistributionSequential``
trees of scalar distributions and uses ``experimental_default_event_space_bijector`` to map the
unconstrained vector ``z`` to physical parameters (src/gigalens/tf/model.py:76-87,164-166).  This
module provides the same tree vocabulary (same class names, same constructor arguments) over torch,
plus a *flat* vectorised view -- one ``(B, d)`` tensor whose column k is the k-th leaf in
``tf.nest.flatten`` order (dict keys sorted, list items in order) -- so that the whole prior + bijector
evaluation is a handful of elementwise ops on ``(B, d)``.

TFP default event-space bijectors restated (TFP is not installed here -- "parity unpinned", see
DESIGN.md): Normal -> Identity; LogNormal -> Exp; Uniform(a,b) / TruncatedNormal(.,.,a,b) -> Sigmoid(a,b);
HalfNormal -> Softplus is NOT provided (unused by the reference's priors).
"""
import math
from typing import Dict, List, Sequence

import torch

_ID, _EXP, _SIG = 0, 1, 2
_NORMAL, _LOGNORMAL, _UNIFORM, _TRUNCNORMAL = 0, 1, 2, 3


class Distribution:
    """Scalar leaf distribution."""

    kind = -1
    bij = _ID

    def _p(self):  # (a, b, lo, hi)
        raise NotImplementedError

    def sample(self, sample_shape=(), seed=None, generator=None, device=None):
        flat = FlatPrior([self], device=device)
        n = int(torch.Size(_shape(sample_shape)).numel())
        out = flat.sample(n, seed=seed, generator=generator)[:, 0]
        return out.reshape(_shape(sample_shape))

    def log_prob(self, x):
        x = torch.as_tensor(x, dtype=torch.float32)
        flat = FlatPrior([self], device=x.device)
        return flat.log_prob_columns(x.reshape(-1, 1))[:, 0].reshape(x.shape)


def _shape(s):
    if s is None:
        return ()
    if isinstance(s, int):
        return (s,)
    return tuple(s)


class Normal(Distribution):
    kind, bij = _NORMAL, _ID

    def __init__(self, loc, scale):
        self.loc, self.scale = float(loc), float(scale)

    def _p(self):
        return (self.loc, self.scale, 0.0, 1.0)


class LogNormal(Distribution):
    kind, bij = _LOGNORMAL, _EXP

    def __init__(self, loc, scale):
        self.loc, self.scale = float(loc), float(scale)

    def _p(self):
        return (self.loc, self.scale, 0.0, 1.0)


class Uniform(Distribution):
    kind, bij = _UNIFORM, _SIG

    def __init__(self, low=0.0, high=1.0):
        self.low, self.high = float(low), float(high)

    def _p(self):
        return (0.0, 1.0, self.low, self.high)


class TruncatedNormal(Distribution):
    kind, bij = _TRUNCNORMAL, _SIG

    def __init__(self, loc, scale, low, high):
        self.loc, self.scale, self.low, self.high = float(loc), float(scale), float(low), float(high)

    def _p(self):
        return (self.loc, self.scale, self.low, self.high)


# ---- nested structure helpers (tf.nest semantics) ------------------------------------------------
def nest_flatten(struct):
    """tf.nest.flatten: dicts by sorted key, sequences in order."""
    if isinstance(struct, dict):
        out = []
        for k in sorted(struct):
            out += nest_flatten(struct[k])
        return out
    if isinstance(struct, (list, tuple)):
        out = []
        for v in struct:
            out += nest_flatten(v)
        return out
    return [struct]


def nest_pack(template, flat: List):
    """tf.nest.pack_sequence_as."""
    it = iter(flat)

    def rec(t):
        if isinstance(t, dict):
            vals = {k: rec(t[k]) for k in sorted(t)}
            return {k: vals[k] for k in t}
        if isinstance(t, (list, tuple)):
            return [rec(v) for v in t]
        return next(it)

    return rec(template)


def nest_paths(struct, prefix=()):
    if isinstance(struct, dict):
        out = []
        for k in sorted(struct):
            out += nest_paths(struct[k], prefix + (k,))
        return out
    if isinstance(struct, (list, tuple)):
        out = []
        for i, v in enumerate(struct):
            out += nest_paths(v, prefix + (i,))
        return out
    return [prefix]


class _Joint(Distribution):
    """Common behaviour of the two joint containers."""

    def _model(self):
        raise NotImplementedError

    def _tree(self):
        def rec(node):
            if isinstance(node, _Joint):
                return rec(node._model())
            if isinstance(node, dict):
                return {k: rec(v) for k, v in node.items()}
            if isinstance(node, (list, tuple)):
                return [rec(v) for v in node]
            return node

        return rec(self._model())

    def flat(self, device=None):
        return FlatPrior(nest_flatten(self._tree()), device=device, template=self._tree())

    def sample(self, sample_shape=(), seed=None, generator=None, device=None):
        shape = _shape(sample_shape)
        n = int(torch.Size(shape).numel())
        f = self.flat(device)
        x = f.sample(n, seed=seed, generator=generator)
        leaves = [x[:, k].reshape(shape) for k in range(f.d)]
        return nest_pack(self._tree(), leaves)

    def log_prob(self, value):
        leaves = nest_flatten(value)
        dev = next((v.device for v in leaves if torch.is_tensor(v)), None)
        f = self.flat(dev)
        cols = torch.stack(torch.broadcast_tensors(*[torch.as_tensor(v, dtype=torch.float32, device=f.device)
                                                     for v in leaves]), dim=-1)
        return f.log_prob_columns(cols.reshape(-1, f.d)).sum(-1).reshape(cols.shape[:-1])

    def experimental_default_event_space_bijector(self, device=None):
        return JointBijector(self.flat(device))


class JointDistributionNamed(_Joint):
    def __init__(self, model: Dict):
        self.model = dict(model)

    def _model(self):
        return self.model


class JointDistributionSequential(_Joint):
    def __init__(self, model: Sequence):
        self.model = list(model)

    def _model(self):
        return self.model


# ---- flat, vectorised view -------------------------------------------------------------------------
class FlatPrior:
    """Column k <-> k-th leaf in nest-flatten order.  All maths elementwise on ``(B, d)``."""

    def __init__(self, leaves: List[Distribution], device=None, template=None):
        self.leaves = leaves
        self.d = len(leaves)
        self.template = template
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        p = torch.tensor([l._p() for l in leaves], dtype=torch.float64).reshape(self.d, 4)
        self.a = p[:, 0].to(torch.float32).to(self.device)
        self.b = p[:, 1].to(torch.float32).to(self.device)
        self.lo = p[:, 2].to(torch.float32).to(self.device)
        self.hi = p[:, 3].to(torch.float32).to(self.device)
        self.kind = torch.tensor([l.kind for l in leaves], dtype=torch.int64, device=self.device)
        self.bij = torch.tensor([l.bij for l in leaves], dtype=torch.int64, device=self.device)
        # truncated-normal normaliser log(Phi(beta) - Phi(alpha)) in float64
        z = torch.zeros(self.d, dtype=torch.float64)
        for k, l in enumerate(leaves):
            if l.kind == _TRUNCNORMAL:
                al, be = (l.low - l.loc) / l.scale, (l.high - l.loc) / l.scale
                z[k] = math.log(0.5 * (math.erf(be / math.sqrt(2)) - math.erf(al / math.sqrt(2))))
        self.logz = z.to(torch.float32).to(self.device)

    def to(self, device):
        return FlatPrior(self.leaves, device=device, template=self.template)

    # -- bijector z -> x -------------------------------------------------------------------------
    def forward(self, z):
        sig = self.lo + (self.hi - self.lo) * torch.sigmoid(z)
        # mask BEFORE exp: an identity column with |z| ~ 500 (shapelet amplitudes) would otherwise give
        # exp(z) = inf in the unselected branch and 0 * inf = NaN in its backward
        ze = torch.where(self.bij == _EXP, z, torch.zeros_like(z))
        return torch.where(self.bij == _ID, z, torch.where(self.bij == _EXP, torch.exp(ze), sig))

    def inverse(self, x):
        u = ((x - self.lo) / (self.hi - self.lo)).clamp(1e-12, 1 - 1e-7)
        logit = torch.log(u) - torch.log1p(-u)
        xs = torch.where(self.bij == _EXP, x, torch.ones_like(x))
        return torch.where(self.bij == _ID, x, torch.where(self.bij == _EXP, torch.log(xs), logit))

    def fldj_columns(self, z):
        """log |dx/dz| per column (TFP: Identity 0; Exp z; Sigmoid(lo,hi) log(hi-lo) - softplus(-z) - softplus(z))."""
        sp = torch.nn.functional.softplus
        sig = torch.log(self.hi - self.lo) - sp(-z) - sp(z)
        return torch.where(self.bij == _ID, torch.zeros_like(z), torch.where(self.bij == _EXP, z, sig))

    # -- densities ---------------------------------------------------------------------------------
    def log_prob_columns(self, x):
        half_log_2pi = 0.5 * math.log(2 * math.pi)
        xs = torch.where(self.kind == _LOGNORMAL, x, torch.ones_like(x))
        logx = torch.log(xs)
        t = torch.where(self.kind == _LOGNORMAL, logx, x)
        zed = (t - self.a) / self.b
        gauss = -0.5 * zed * zed - torch.log(self.b) - half_log_2pi
        lognormal = gauss - logx
        inside = (x >= self.lo) & (x <= self.hi)
        neg_inf = torch.full_like(x, -float("inf"))
        uniform = torch.where(inside, -torch.log(self.hi - self.lo).expand_as(x), neg_inf)
        trunc = torch.where(inside, gauss - self.logz, neg_inf)
        return torch.where(self.kind == _NORMAL, gauss,
                           torch.where(self.kind == _LOGNORMAL, lognormal,
                                       torch.where(self.kind == _UNIFORM, uniform, trunc)))

    def log_prob(self, x):
        return self.log_prob_columns(x).sum(-1)

    # -- sampling ------------------------------------------------------------------------------------
    def sample(self, n, seed=None, generator=None):
        g = generator
        if g is None and seed is not None:
            g = torch.Generator(device="cpu")
            g.manual_seed(int(seed))
        eps = torch.randn((n, self.d), generator=g, dtype=torch.float32).to(self.device)
        u = torch.rand((n, self.d), generator=g, dtype=torch.float32).to(self.device)
        normal = self.a + self.b * eps
        lognormal = torch.exp(normal)
        uniform = self.lo + (self.hi - self.lo) * u
        # inverse-CDF sampling of the truncated normal (float64 for the tails)
        a64, b64 = self.a.double(), self.b.double()
        nd = torch.distributions.Normal(0.0, 1.0)
        ca = nd.cdf(((self.lo.double() - a64) / b64).clamp(-40, 40))
        cb = nd.cdf(((self.hi.double() - a64) / b64).clamp(-40, 40))
        pt = (ca + u.double() * (cb - ca)).clamp(1e-15, 1 - 1e-15)
        trunc = (a64 + b64 * nd.icdf(pt)).float()
        trunc = torch.minimum(torch.maximum(trunc, self.lo), self.hi)
        return torch.where(self.kind == _NORMAL, normal,
                           torch.where(self.kind == _LOGNORMAL, lognormal,
                                       torch.where(self.kind == _UNIFORM, uniform, trunc)))


class JointBijector:
    """``prior.experimental_default_event_space_bijector()`` on nested structures of ``(B,)`` leaves."""

    def __init__(self, flat: FlatPrior):
        self.flat = flat

    def _cols(self, struct):
        leaves = [torch.as_tensor(v, dtype=torch.float32, device=self.flat.device) for v in nest_flatten(struct)]
        return torch.stack(torch.broadcast_tensors(*leaves), dim=-1)

    def _struct(self, cols):
        return nest_pack(self.flat.template, [cols[..., k] for k in range(self.flat.d)])

    def forward(self, z_struct):
        return self._struct(self.flat.forward(self._cols(z_struct)))

    def inverse(self, x_struct):
        return self._struct(self.flat.inverse(self._cols(x_struct)))

    def forward_log_det_jacobian(self, z_struct):
        return self.flat.fldj_columns(self._cols(z_struct)).sum(-1)
