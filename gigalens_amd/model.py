"""``PhysicalModel`` and ``ForwardProbModel`` with the reference's surface
(src/gigalens/model.py:7-73, src/gigalens/tf/model.py:12-194,276-306).

``ForwardProbModel.log_prob(simulator, z)`` returns ``(log_prob, red_chi2)`` exactly like the reference;
the pixel log-likelihood and its gradient w.r.t. every profile parameter come from ONE fused HIP launch
sequence (prep -> main -> finalize, see csrc/gl_kernels.hip.h); prior densities and bijectors are a few
elementwise torch ops on ``(B, d)``.
"""
from typing import Dict, List

import numpy as np
import torch

from gigalens_amd import _native
from gigalens_amd import prior as _prior

_GROUPS = ("lens_mass", "lens_light", "source_light")


class _Packing:
    """Map between the reference's nested parameter structure and the native ``[B, P]`` rows
    (component-major: lenses, lens light, source light; inside a component the reference's
    ``params`` order).  Fixed parameters come from the ``*_constants`` dicts (model.py:29-44)."""

    def __init__(self, phys_model):
        self.slots = []  # (group, index, name, const_value_or_None)
        groups = ((phys_model.lenses, phys_model.lenses_constants),
                  (phys_model.lens_light, phys_model.lens_light_constants),
                  (phys_model.source_light, phys_model.source_light_constants))
        self.linear = []  # packed columns of amplitudes solved by least squares (use_lstsq profiles)
        for gname, (profiles, consts) in zip(_GROUPS, groups):
            for i, (prof, c) in enumerate(zip(profiles, consts)):
                for name in prof._native_params():
                    if name not in prof.params:  # linear amplitude: unit placeholder until lstsq fills it
                        self.linear.append(len(self.slots))
                        self.slots.append((gname, i, name, np.float32(1.0)))
                    else:
                        self.slots.append((gname, i, name, c.get(name)))
        self.P = len(self.slots)

    def pack(self, params: Dict[str, List[Dict]], bs: int, device):
        cols = []
        for gname, i, name, const in self.slots:
            grp = params.get(gname)
            v = grp[i].get(name) if grp is not None and i < len(grp) else None
            if v is None:
                v = const
            if v is None:
                raise KeyError(f"parameter {gname}[{i}]['{name}'] is neither given nor a model constant")
            v = torch.as_tensor(v, dtype=torch.float32, device=device)
            cols.append(v.reshape(-1).expand(bs) if v.numel() != bs else v.reshape(bs))
        if not cols:
            return torch.zeros((bs, 0), dtype=torch.float32, device=device)
        return torch.stack(cols, dim=1)


class PhysicalModelBase:
    """src/gigalens/model.py:7-44."""

    def __init__(self, lenses, lens_light, source_light, lenses_constants: List[Dict] = None,
                 lens_light_constants: List[Dict] = None, source_light_constants: List[Dict] = None):
        self.lenses = lenses
        self.lens_light = lens_light
        self.source_light = source_light
        if lenses_constants is None:
            lenses_constants = [dict() for _ in range(len(lenses))]
        if lens_light_constants is None:
            lens_light_constants = [dict() for _ in range(len(lens_light))]
        if source_light_constants is None:
            source_light_constants = [dict() for _ in range(len(source_light))]
        self.lenses_constants = lenses_constants
        self.lens_light_constants = lens_light_constants
        self.source_light_constants = source_light_constants


class PhysicalModel(PhysicalModelBase):
    """src/gigalens/tf/model.py:276-306: constants are cast to float32."""

    def __init__(self, lenses, lens_light, source_light, lenses_constants: List[Dict] = None,
                 lens_light_constants: List[Dict] = None, source_light_constants: List[Dict] = None):
        super().__init__(lenses, lens_light, source_light, lenses_constants, lens_light_constants,
                         source_light_constants)
        cast = lambda ds: [{k: np.asarray(v, dtype=np.float32) for k, v in d.items()} for d in ds]
        self.lenses_constants = cast(self.lenses_constants)
        self.lens_light_constants = cast(self.lens_light_constants)
        self.source_light_constants = cast(self.source_light_constants)

    def _packing(self):
        return _Packing(self)


class ProbabilisticModel:
    """src/gigalens/model.py:47-73."""

    def __init__(self, prior, bij=None, *args):
        self.prior = prior
        self.bij = bij

    def log_prob(self, simulator, z):
        raise NotImplementedError


class _LogLikeFn(torch.autograd.Function):
    """autograd glue around gl_loglike_fwd_bwd: the gradient w.r.t. the packed parameters is produced
    by the same fused pass as the value and kept for ``backward``."""

    @staticmethod
    def forward(ctx, packed, model, obs, err, mask, bg_rms, exp_time):
        want = packed.requires_grad
        ll, chi2, grad = model.loglike(packed.detach(), obs, err, mask, bg_rms, exp_time, want)
        ctx.has_grad = want
        if want:
            ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(chi2)
        return ll, chi2

    @staticmethod
    def backward(ctx, g_ll, g_chi2):
        (grad,) = ctx.saved_tensors
        return g_ll[:, None] * grad, None, None, None, None, None, None


class _LogProbFn(torch.autograd.Function):
    """autograd glue around gl_logprob_fwd_bwd (bijector + kernels + prior in one native launch sequence)."""

    @staticmethod
    def forward(ctx, z, model, obs, err, mask, bg_rms, exp_time, n_eff, terms):
        want = z.requires_grad
        lp, ll, chi2, grad = model.logprob(z.detach(), obs, err, mask, bg_rms, exp_time, want, n_eff, terms)
        if want:
            ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(ll, chi2)
        return lp, ll, chi2

    @staticmethod
    def backward(ctx, g_lp, g_ll, g_chi2):
        (grad,) = ctx.saved_tensors
        return g_lp[:, None] * grad, None, None, None, None, None, None, None, None


class _PositionsFn(torch.autograd.Function):
    """autograd glue around gl_positions_fwd_bwd (image-position likelihood)."""

    @staticmethod
    def forward(ctx, packed, model):
        want = packed.requires_grad
        ll, chi2, grad = model.positions(packed.detach(), want)
        if want:
            ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(chi2)
        return ll, chi2

    @staticmethod
    def backward(ctx, g_ll, g_chi2):
        (grad,) = ctx.saved_tensors
        return g_ll[:, None] * grad, None


class _PackBijector:
    """``pack_bij`` (tf/model.py:78-85): ``(B, d)`` <-> nested structure, column k = k-th nest leaf."""

    def __init__(self, template):
        self.template = template
        self.d = len(_prior.nest_flatten(template))

    def forward(self, z):
        return _prior.nest_pack(self.template, [z[..., k] for k in range(self.d)])

    def inverse(self, struct):
        leaves = [torch.as_tensor(v, dtype=torch.float32) for v in _prior.nest_flatten(struct)]
        dev = next((v.device for v in leaves if v.is_cuda), leaves[0].device)
        return torch.stack(torch.broadcast_tensors(*[v.to(dev) for v in leaves]), dim=-1)


class _ChainBijector:
    """``bij = Chain([unconstraining_bij, pack_bij])`` (tf/model.py:87)."""

    def __init__(self, unconstraining, pack):
        self.unconstraining, self.pack = unconstraining, pack

    def forward(self, z):
        return self.unconstraining.forward(self.pack.forward(z))

    def inverse(self, x_struct):
        return self.pack.inverse(self.unconstraining.inverse(x_struct))


class ForwardProbModel(ProbabilisticModel):
    """Drop-in for ``gigalens.tf.model.ForwardProbModel`` (tf/model.py:12-194): pixel likelihood and
    image-position likelihood.  With the reference's default ``include_positions=True`` and no centroids the
    reference itself fails (it iterates ``None``, tf/model.py:69-70), so that combination raises ``TypeError``."""

    def __init__(self, prior, observed_image=None, background_rms=None, exp_time=None, error_map=None,
                 centroids_x=None, centroids_y=None, centroids_errors_x=None, centroids_errors_y=None,
                 include_pixels=True, include_positions=True):
        super().__init__(prior)
        self.include_pixels = include_pixels
        self.include_positions = include_positions
        # host logic (prior, bijectors) also runs without a GPU; the likelihood itself never does
        self.device = _native.device() if torch.cuda.is_available() else torch.device("cpu")
        self.observed_image = None
        self.error_map = None
        self.background_rms = None
        self.exp_time = None
        if self.include_pixels:
            self.observed_image = torch.as_tensor(np.asarray(observed_image, dtype=np.float32), device=self.device).contiguous()
            if error_map is not None:
                self.error_map = torch.as_tensor(np.asarray(error_map, dtype=np.float32), device=self.device).contiguous()
            else:
                self.background_rms = float(np.float32(background_rms))
                self.exp_time = float(np.float32(exp_time))
        self.centroids_x = self.centroids_y = self.centroids_errors_x = self.centroids_errors_y = None
        self.n_position = 0.0
        if self.include_positions:
            if centroids_x is None:
                raise TypeError("include_positions=True needs centroids_x/centroids_y (the reference iterates them, "
                                "tf/model.py:69-70); pass include_positions=False for a pixel-only model")
            f32 = lambda L: [np.atleast_1d(np.asarray(v, dtype=np.float32)) for v in L]
            self.centroids_x, self.centroids_y = f32(centroids_x), f32(centroids_y)
            self.centroids_errors_x = [np.broadcast_to(e, x.shape).copy() for e, x in zip(f32(centroids_errors_x), self.centroids_x)]
            self.centroids_errors_y = [np.broadcast_to(e, x.shape).copy() for e, x in zip(f32(centroids_errors_y), self.centroids_y)]
            self.n_position = 2.0 * float(sum(x.size for x in self.centroids_x))  # tf/model.py:74
        self._flat = prior.flat(self.device)
        example = prior.sample(seed=0)
        self.pack_bij = _PackBijector(example)
        self.unconstraining_bij = _prior.JointBijector(self._flat)
        self.bij = _ChainBijector(self.unconstraining_bij, self.pack_bij)
        self._paths = _prior.nest_paths(example)
        self._perm_cache = {}

    # ---- z columns -> native packed rows, without materialising the nested structure ----------------
    def _perm(self, simulator):
        key = id(simulator._layout)
        hit = self._perm_cache.get(key)
        if hit is None:
            index = {}
            for k, path in enumerate(self._paths):
                if len(path) != 3 or path[0] not in _GROUPS:
                    raise ValueError(f"prior leaf {path} does not follow {{group: [ {{name: dist}} ]}}")
                index[path] = k
            d = len(self._paths)
            cols, consts = [], []
            for gname, i, name, const in simulator._layout.slots:
                k = index.get((gname, i, name))
                if k is not None:
                    cols.append(k)
                elif const is not None:
                    cols.append(d + len(consts))
                    consts.append(float(np.asarray(const, dtype=np.float32).reshape(-1)[0]))
                else:
                    raise KeyError(f"{gname}[{i}]['{name}'] has neither a prior nor a constant")
            hit = (torch.tensor(cols, dtype=torch.int64, device=self.device),
                   torch.tensor(consts, dtype=torch.float32, device=self.device))
            self._perm_cache = {key: hit}
        return hit

    def _bind_prior(self, simulator):
        """Hand the prior / bijector column table to the simulator's native model (once per pairing)."""
        model = simulator._model
        if getattr(model, "_prior_owner", None) is not self:
            slot_of = {(g, i, n): p for p, (g, i, n, _) in enumerate(simulator._layout.slots)}
            columns = []
            for k, (path, leaf) in enumerate(zip(self._paths, self._flat.leaves)):
                if tuple(path) not in slot_of:
                    raise KeyError(f"prior leaf {path} is not a parameter of the physical model")
                a, b, lo, hi = leaf._p()
                columns.append((slot_of[tuple(path)], leaf.bij, leaf.kind, a, b, lo, hi, float(self._flat.logz[k])))
            const_row = np.zeros(simulator._layout.P, dtype=np.float32)
            driven = {c[0] for c in columns}
            for p, (g, i, n, const) in enumerate(simulator._layout.slots):
                if p not in driven:
                    if const is None:
                        raise KeyError(f"{g}[{i}]['{n}'] has neither a prior nor a constant")
                    const_row[p] = float(np.asarray(const, dtype=np.float32).reshape(-1)[0])
            model.set_prior(columns, const_row)
            model._prior_owner = self
        return model

    def _fused_ok(self, simulator):
        return self.include_pixels or self.include_positions

    def _bind_positions(self, simulator):
        model = simulator._model
        if getattr(model, "_positions_owner", None) is not self:
            model.set_positions(self.centroids_x, self.centroids_y, self.centroids_errors_x, self.centroids_errors_y)
            model._positions_owner = self
        return model

    def _terms(self):
        return (1 if self.include_pixels else 0) | (2 if self.include_positions else 0)

    def stats_positions(self, simulator, params):
        """tf/model.py:103-124: ``(log_like, red_chi2)`` of the image-position term."""
        packed = params if torch.is_tensor(params) else simulator.pack(params)
        ll, chi2 = _PositionsFn.apply(packed, self._bind_positions(simulator))
        return ll, chi2 / self.n_position

    def _packed_from_x(self, simulator, x):
        cols, consts = self._perm(simulator)
        if consts.numel():
            x = torch.cat([x, consts.expand(x.shape[0], -1)], dim=1)
        return x.index_select(1, cols)

    def _pixel_stats_packed(self, simulator, packed):
        ll, chi2 = _LogLikeFn.apply(packed, simulator._model, self.observed_image, self.error_map,
                                    simulator.img_region if simulator.sim_config.pix_region is not None else None,
                                    self.background_rms or 0.0, self.exp_time or 1.0)
        return ll, chi2 / self._n_eff(simulator)  # tf/model.py:100

    def stats_pixels(self, simulator, params):
        """tf/model.py:89-101: ``params`` is the nested constrained structure."""
        return self._pixel_stats_packed(simulator, simulator.pack(params))

    def log_prob(self, simulator, z):
        """tf/model.py:126-167: ``z`` is ``(bs, d)`` unconstrained; returns ``(log_prob, red_chi2)``."""
        z = torch.as_tensor(z, dtype=torch.float32, device=self.device)
        if self._fused_ok(simulator):
            # bijector -> prep -> fused render/chi2/VJP -> finalize + prior, all inside the native library
            model = self._bind_prior(simulator)
            if self.include_positions:
                self._bind_positions(simulator)
            lp, _, red_chi2 = _LogProbFn.apply(z, model, self.observed_image, self.error_map, self._mask(simulator),
                                               self.background_rms or 0.0, self.exp_time or 1.0, self._n_eff(simulator),
                                               self._terms())
            return lp, red_chi2
        return self.log_prob_unfused(simulator, z)

    def _mask(self, simulator):
        return simulator.img_region if simulator.sim_config.pix_region is not None else None

    def term_log_prob_and_grad(self, simulator, z, term):
        """``(log_prior + log_like_term, log_like_term, gradient of the former w.r.t. z)`` for ONE likelihood term
        (``"pixels"`` or ``"positions"``) -- the pieces the tempered SMC target is assembled from
        (tf/inference.py:213-238,292-303)."""
        z = torch.as_tensor(z, dtype=torch.float32, device=self.device)
        model = self._bind_prior(simulator)
        bit = {"pixels": 1, "positions": 2}[term]
        if bit == 2:
            self._bind_positions(simulator)
        lp, ll, _, grad = model.logprob(z.detach(), self.observed_image, self.error_map, self._mask(simulator),
                                        self.background_rms or 0.0, self.exp_time or 1.0, True,
                                        self._n_eff(simulator) if bit == 1 else 1.0, bit)
        return lp, ll, grad

    def log_prob_and_grad(self, simulator, z, graph=False):
        """``(log_prob, red_chi2, d log_prob / d z)`` from ONE native launch sequence and no autograd graph --
        what one MAP / HMC-leapfrog step of the reference computes with ``tf.GradientTape`` (tf/inference.py:33-39).

        ``graph=True`` (small problems, where the three to seven short launches of a step are host-issue bound -- BASELINE
        configs[0], one 64 x 64 sample: 7 us of kernels in a 21 us step): the launch sequence is captured once per
        (simulator, shape of ``z``) in a HIP graph and replayed.  The contract is that of torch's CUDA graphs: the three returned
        tensors are the graph's STATIC outputs, overwritten by the next call with ``graph=True`` on the same simulator; ``z`` is
        copied into the graph's static input unless it already IS that tensor (``graph_input(simulator, z)`` hands it out, for loops
        that update ``z`` in place)."""
        z = torch.as_tensor(z, dtype=torch.float32, device=self.device)
        if graph and z.is_cuda and self._fused_ok(simulator):
            return self._log_prob_and_grad_graph(simulator, z)
        if self._fused_ok(simulator):
            model = self._bind_prior(simulator)
            if self.include_positions:
                self._bind_positions(simulator)
            lp, _, red, grad = model.logprob(z.detach(), self.observed_image, self.error_map, self._mask(simulator),
                                             self.background_rms or 0.0, self.exp_time or 1.0, True,
                                             self._n_eff(simulator), self._terms())
            return lp, red, grad
        zz = z.detach().requires_grad_(True)
        lp, red = self.log_prob_unfused(simulator, zz)
        (g,) = torch.autograd.grad(lp.sum(), zz)
        return lp.detach(), red.detach(), g

    def _log_prob_and_grad_graph(self, simulator, z):
        key = (id(self), tuple(z.shape))
        cache = simulator.__dict__.setdefault("_lp_graphs", {})
        ent = cache.get(key)
        if ent is None:
            z_static = z.detach().clone().contiguous()
            side = torch.cuda.Stream(device=z.device)
            side.wait_stream(torch.cuda.current_stream(z.device))
            with torch.cuda.stream(side):  # warm-up off the default stream: binds the prior, sizes the workspaces
                for _ in range(2):
                    self.log_prob_and_grad(simulator, z_static)
            torch.cuda.current_stream(z.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs = self.log_prob_and_grad(simulator, z_static)
            ent = cache[key] = (g, z_static, outs, self)  # (self: keeps id(self) from being reused while the entry lives)
        g, z_static, outs, _ = ent
        if z.data_ptr() != z_static.data_ptr():
            z_static.copy_(z.detach())
        g.replay()
        return outs

    def graph_input(self, simulator, z):
        """The static input tensor of ``log_prob_and_grad(..., graph=True)`` for this shape of ``z`` (created, and the graph
        captured, on first use), initialised with ``z``: update it in place and pass it back to skip the copy."""
        z = torch.as_tensor(z, dtype=torch.float32, device=self.device)
        self._log_prob_and_grad_graph(simulator, z)
        return simulator._lp_graphs[(id(self), tuple(z.shape))][1]

    def _n_eff(self, simulator):
        n = getattr(simulator, "_n_eff", None)
        if n is None:
            n = simulator._n_eff = float(torch.count_nonzero(simulator.img_region))
        return n

    def log_prob_unfused(self, simulator, z):
        """Same quantity with the bijector and prior evaluated by torch ops around the likelihood kernels
        (kept as the cross-check of the fused native path and for configurations it does not cover)."""
        z = torch.as_tensor(z, dtype=torch.float32, device=self.device)
        x = self._flat.forward(z)
        log_like = torch.zeros(z.shape[0], dtype=torch.float32, device=self.device)
        red_chi2 = torch.zeros_like(log_like)
        n_chi = 0
        packed = self._packed_from_x(simulator, x)
        if self.include_pixels:
            ll, rc = self._pixel_stats_packed(simulator, packed)
            log_like = log_like + ll
            red_chi2 = red_chi2 + rc
            n_chi += 1
        if self.include_positions:
            ll, rc = self.stats_positions(simulator, packed)
            log_like = log_like + ll
            red_chi2 = red_chi2 + rc
            n_chi += 1
        red_chi2 = red_chi2 / max(n_chi, 1)
        log_prior = self._flat.log_prob(x) + self._flat.fldj_columns(z).sum(-1)
        return log_like + log_prior, red_chi2

    def log_like(self, simulator, z):
        """tf/model.py:169-180."""
        z = torch.as_tensor(z, dtype=torch.float32, device=self.device)
        x = self._flat.forward(z)
        ll = torch.zeros(z.shape[0], dtype=torch.float32, device=self.device)
        packed = self._packed_from_x(simulator, x)
        if self.include_pixels:
            ll = ll + self._pixel_stats_packed(simulator, packed)[0]
        if self.include_positions:
            ll = ll + self.stats_positions(simulator, packed)[0]
        return ll

    def log_prior(self, z):
        """tf/model.py:182-185."""
        z = torch.as_tensor(z, dtype=torch.float32, device=self.device)
        return self._flat.log_prob(self._flat.forward(z)) + self._flat.fldj_columns(z).sum(-1)

    def init_centroids(self, bs):
        """tf/model.py:187-194 (no-op without the position branch)."""
        return None


class BackwardProbModel(ProbabilisticModel):
    """Drop-in for ``gigalens.tf.model.BackwardProbModel`` (tf/model.py:197-273): the noise map comes from the
    OBSERVED image, and the linear light amplitudes are solved by least squares inside the likelihood.

    The reference differentiates through ``tf.linalg.pinv``; here the gradient uses the envelope property of the
    solve -- the coefficients minimise exactly the chi^2 that the (fixed-noise) log-likelihood is made of, so
    ``d log_like / d theta`` equals the partial derivative at fixed coefficients, which the fused forward+gradient
    kernels evaluate with the solved amplitudes written into their columns (equal to the reference's total derivative
    wherever the normal matrix has full rank above the ``rcond`` cut)."""

    def __init__(self, prior, observed_image, background_rms, exp_time):
        super().__init__(prior)
        self.device = _native.device() if torch.cuda.is_available() else torch.device("cpu")
        obs = np.asarray(observed_image, dtype=np.float32)
        err = np.sqrt(np.float32(background_rms) ** 2 + np.clip(obs, 0, np.inf) / np.float32(exp_time)).astype(np.float32)
        self.observed_image = torch.as_tensor(obs, device=self.device).contiguous()
        self.err_map = torch.as_tensor(err, device=self.device).contiguous()
        self._flat = prior.flat(self.device)
        example = prior.sample(seed=0)
        self.pack_bij = _PackBijector(example)
        self.unconstraining_bij = _prior.JointBijector(self._flat)
        self.bij = _ChainBijector(self.unconstraining_bij, self.pack_bij)

    def log_prob(self, simulator, z):
        """tf/model.py:242-273: ``(log_like + log_prior, mean squared normalised residual)``."""
        z = torch.as_tensor(z, dtype=torch.float32, device=self.device)
        x = self._flat.forward(z)
        log_prior = self._flat.log_prob(x) + self._flat.fldj_columns(z).sum(-1)
        packed = simulator.pack(self.pack_bij.forward(x))
        coeffs = simulator._model.lstsq(packed.detach(), self.observed_image, self.err_map, 7, want="coeffs")[0]
        lin = getattr(simulator, "_linear_cols_dev", None)  # uploaded once per simulator, not per call
        if lin is None or lin.device != packed.device:
            lin = simulator._linear_cols_dev = torch.tensor(simulator._layout.linear, dtype=torch.int64, device=packed.device)
        full = packed.index_copy(1, lin, coeffs / simulator.conversion_factor)  # amplitude = coeff / det(T)
        ll, chi2 = _LogLikeFn.apply(full, simulator._model, self.observed_image, self.err_map, None, 0.0, 1.0)
        return ll + log_prior, chi2 / float(self.observed_image.numel())

    # -- what ModellingSequence.MAP / SVI / HMC need from a probabilistic model (shapelets-demo.ipynb runs them on it) --
    include_pixels, include_positions, n_position = True, False, 0.0

    def log_prob_and_grad(self, simulator, z):
        zz = torch.as_tensor(z, dtype=torch.float32, device=self.device).detach().requires_grad_(True)
        lp, red = self.log_prob(simulator, zz)
        (g,) = torch.autograd.grad(lp.sum(), zz)
        return lp.detach(), red.detach(), g

    def log_prior(self, z):
        z = torch.as_tensor(z, dtype=torch.float32, device=self.device)
        return self._flat.log_prob(self._flat.forward(z)) + self._flat.fldj_columns(z).sum(-1)

    def init_centroids(self, bs):
        return None
