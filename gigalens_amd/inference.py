"""Callers of the hot path: the MAP -> SVI -> HMC modelling sequence (reference:
src/gigalens/inference.py:10-139, src/gigalens/tf/inference.py:17-182, multi-device semantics from
src/gigalens/jax/inference.py:32-208), as thin torch drivers around ``ForwardProbModel.log_prob``.

These drivers are host orchestration, not kernels: each optimiser / leapfrog step costs one fused native
forward+gradient call.  Data parallelism follows the JAX substrate: the sample batch is sharded over ranks
(one process per GPU); MAP and HMC need no collective, SVI all-reduces ONE fused buffer
``[ELBO, dELBO/dmu (d), dELBO/dL_packed (d(d+1)/2)]`` per step (``lax.pmean``, jax/inference.py:123-128).

The TFP kernels the reference composes (PreconditionedHamiltonianMonteCarlo, GradientBasedTrajectoryLengthAdaptation,
Dual-averaging / Simple step-size adaptation, sample_sequential_monte_carlo) are third party and not installed here; their
published algorithms are restated: ChEES trajectory-length adaptation (Hoffman, Radul & Sountsov 2021) with Halton-jittered
trajectory lengths, dual averaging (Hoffman & Gelman 2014), adaptive tempered SMC.  Stochastic drivers cannot be pinned
bit for bit against TFP (different random streams); they are checked on targets with known answers.
"""
import math
from typing import Callable, Optional, Tuple

import torch

from gigalens_amd import _native
from gigalens_amd import dist as gdist
from gigalens_amd.simulator import LensSimulator


class Adam:
    """Plain Adam on one tensor (the reference passes a Keras / optax Adam; lr == 0 must be a no-op)."""

    def __init__(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.m = self.v = None
        self.t = 0

    def state_tensors(self):
        """The moment estimates (empty before the first step): what a multi-rank driver re-synchronises with the parameters."""
        return [t for t in (self.m, self.v) if t is not None]

    def _native_ok(self, x, grad):
        return (x.is_cuda and grad.is_cuda and x.dtype == torch.float32 and grad.dtype == torch.float32
                and x.is_contiguous() and grad.is_contiguous() and x.shape == grad.shape)

    def step(self, x: torch.Tensor, grad: torch.Tensor, scale: float = 1.0):
        """``x -= lr * mhat / (sqrt(vhat) + eps)`` for the gradient ``scale * grad``.  On the GPU the whole update is one
        native launch (gl_adam_update); CPU tensors (the gloo tests of the sharding logic) take the same formula in torch."""
        if self.m is None:
            self.m, self.v = torch.zeros_like(x), torch.zeros_like(x)
        self.t += 1
        lr = self.lr(self.t) if callable(self.lr) else self.lr
        if self._native_ok(x, grad):
            _native.adam_update(x, grad, self.m, self.v, scale, lr, self.b1, self.b2, self.eps, self.t)
            return x
        grad = grad * scale if scale != 1.0 else grad
        self.m.mul_(self.b1).add_(grad, alpha=1 - self.b1)
        self.v.mul_(self.b2).addcmul_(grad, grad, value=1 - self.b2)
        mhat = self.m / (1 - self.b1 ** self.t)
        vhat = self.v / (1 - self.b2 ** self.t)
        x.sub_(lr * mhat / (vhat.sqrt() + self.eps))
        return x

    # -- the same update with the step counter on the device, so that one step can be captured in a HIP graph ------
    @property
    def capturable(self):
        return not callable(self.lr)

    def step_captured(self, x: torch.Tensor, grad: torch.Tensor, scale: float = 1.0):
        """:meth:`step` with ``t`` in a device counter the kernel advances itself (gl_adam_update's ``t_dev``), so the
        launch can be replayed from a graph.  ``sync_from_device`` brings ``t`` back afterwards."""
        if self.m is None:
            self.m, self.v = torch.zeros_like(x), torch.zeros_like(x)
        if getattr(self, "_t_dev", None) is None or self._t_dev.device != x.device:
            self._t_dev = torch.zeros(2, dtype=torch.float64, device=x.device)
            self._t_dev[0] = float(self.t)
        _native.adam_update(x, grad, self.m, self.v, scale, self.lr, self.b1, self.b2, self.eps, 0, self._t_dev)
        return x

    def sync_from_device(self):
        if getattr(self, "_t_dev", None) is not None:
            self.t = int(round(float(self._t_dev[0])))
            self._t_dev = None


# ---- full-rank Gaussian surrogate ------------------------------------------------------------------
def tril_unpack(packed: torch.Tensor, d: int, diag_shift: float = 1e-6) -> torch.Tensor:
    """FillScaleTriL(diag_bijector=Exp, diag_shift=1e-6) (tf/inference.py:69-72).  Packing order here is
    ``torch.tril_indices`` (row-major lower triangle); TFP's fill_triangular uses a different but equivalent
    ordering -- only the coordinate labels of the variational parameters differ."""
    idx = _tril_idx(d, packed.device)
    L = torch.zeros((d, d), dtype=packed.dtype, device=packed.device)
    L = L.index_put((idx[0], idx[1]), packed)
    diag = torch.diagonal(L)
    return L - torch.diag(diag) + torch.diag(torch.exp(diag) + diag_shift)


def tril_pack(L: torch.Tensor, diag_shift: float = 1e-6) -> torch.Tensor:
    d = L.shape[0]
    idx = torch.tril_indices(d, d, device=L.device)
    M = L.clone()
    M[range(d), range(d)] = torch.log(torch.diagonal(L) - diag_shift)
    return M[idx[0], idx[1]]


_TRIL_IDX = {}


def _warn_remainder(stage, what, n, world):
    """The JAX substrate gives every device ``n // dev_cnt`` samples and says nothing about the rest (jax/inference.py:33-38,
    93-98,159-165); same split here, but the count that is actually run is stated."""
    if world > 1 and n % world:
        import warnings
        warnings.warn(f"{stage}: {what}={n} is not a multiple of the {world} ranks; each rank runs {n // world}, "
                      f"{n - world * (n // world)} fewer in total (like the reference's n // dev_cnt)", RuntimeWarning, stacklevel=3)


def _tril_idx(d, device):
    key = (d, str(device))
    if key not in _TRIL_IDX:
        _TRIL_IDX[key] = torch.tril_indices(d, d, device=device)
    return _TRIL_IDX[key]


def svi_step(mu, l_packed, log_prob_fn, n_local, generator=None, value_and_grad_fn=None, full_rank=None, eps=None
             ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """:func:`svi_step_buffer` split into ``(loss, grad_mu, grad_l_packed)`` (views of the one fused buffer)."""
    buf = svi_step_buffer(mu, l_packed, log_prob_fn, n_local, generator, value_and_grad_fn, full_rank, eps)
    d = mu.numel()
    return buf[0], buf[1:1 + d], buf[1 + d:]


class NormalPool:
    """Standard-normal draws ``(n, d)`` for the steps of a loop, generated ``block`` steps per launch: a step-sized
    ``torch.randn`` is a kernel launch of its own (~5 us on the critical path of a 0.1 ms SVI step)."""

    def __init__(self, generator: Optional[torch.Generator], n: int, d: int, dtype=torch.float32, device=None, block: int = 32):
        self.gen, self.shape, self.dtype, self.block = generator, (int(block), int(n), int(d)), dtype, int(block)
        self.device = device if device is not None else (generator.device if generator is not None else "cpu")
        self.i, self.buf = self.block, None

    def next(self) -> torch.Tensor:
        if self.i == self.block:
            gdev = self.gen.device if self.gen is not None else self.device
            self.buf = torch.randn(self.shape, generator=self.gen, dtype=self.dtype, device=gdev).to(self.device)
            self.i = 0
        self.i += 1
        return self.buf[self.i - 1]


def svi_step_buffer(mu: torch.Tensor, l_packed: torch.Tensor,
                    log_prob_fn: Optional[Callable[[torch.Tensor], torch.Tensor]],
                    n_local: int, generator: Optional[torch.Generator] = None,
                    value_and_grad_fn: Optional[Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor]]] = None,
                    full_rank: Optional[bool] = None, eps: Optional[torch.Tensor] = None, reduce: bool = True) -> torch.Tensor:
    """One ELBO evaluation on this rank's particle shard, all-reduced over ranks (``reduce=False``: this shard's buffer as it
    stands, no collective -- what the multi-rank tests average by hand): the fused buffer
    ``[loss, grad_mu (d), grad_l_packed]`` -- ``buf[1:]`` is the gradient of ``cat([mu, l_packed])`` as it stands.

    The buffer is identical on every rank: the mean over ranks of the per-rank
    means (== the mean over all particles, shards being equal-sized).  ``l_packed`` of length ``d`` is the mean-field
    surrogate (``MultivariateNormalDiag`` with ``Exp`` on the scales, tf/inference.py:75-83), of length
    ``d (d + 1) / 2`` the full-rank one.

    The reparameterisation gradient is written out instead of taped: with ``z = mu + L eps`` and
    ``G = d log p / d z`` (one native forward+gradient call through ``value_and_grad_fn``; ``log_prob_fn`` is
    differentiated with autograd when only that is given),
    ``d ELBO / d mu = -mean G``,  ``d ELBO / d L = -mean G eps^T`` on the lower triangle, and the ``Exp`` diagonal
    contributes ``dL_ii / dp_ii = exp(p_ii)`` plus ``-exp(p_ii) / L_ii`` from ``-log det L`` in ``log q``.

    ``full_rank`` says which surrogate ``l_packed`` parameterises; ``None`` infers it from the length, which is
    ambiguous only at ``d == 1`` (read as mean field there, like ``SVI`` stores it).  ``eps``: the standard-normal draws
    ``(n_local, d)`` to use instead of drawing them (tests)."""
    d = mu.numel()
    mu, l_packed = mu.detach(), l_packed.detach()
    if full_rank is None:
        full_rank = l_packed.numel() != d
    if l_packed.numel() != (d * (d + 1) // 2 if full_rank else d):
        raise ValueError(f"l_packed has {l_packed.numel()} entries, a {'full-rank' if full_rank else 'mean-field'} "
                         f"surrogate of dimension {d} needs {d * (d + 1) // 2 if full_rank else d}")
    diag_mode = not full_rank
    if eps is None:
        eps = torch.randn((n_local, d), generator=generator, dtype=mu.dtype,
                          device=generator.device if generator is not None else mu.device).to(mu.device)
    if value_and_grad_fn is not None and mu.is_cuda and mu.dtype == torch.float32:
        # on the GPU the surrogate is two native launches around the forward+gradient call (gl_svi_sample / gl_svi_grad)
        mu_c, lp_c, eps = mu.contiguous(), l_packed.contiguous(), eps.contiguous()
        z = _native.svi_sample(mu_c, lp_c, eps, not diag_mode)
        lp, G = value_and_grad_fn(z)
        buf = _native.svi_grad(lp_c, eps, lp.contiguous(), G.contiguous(), not diag_mode)
        return gdist.allreduce_mean_(buf) if reduce else buf
    if diag_mode:
        sdiag = torch.exp(l_packed)
        z = mu + eps * sdiag
        log_det = l_packed.sum()
    else:
        L = tril_unpack(l_packed, d)
        z = mu + eps @ L.T
        ldiag = torch.diagonal(L)
        log_det = torch.log(ldiag).sum()
    if value_and_grad_fn is not None:
        lp, G = value_and_grad_fn(z)
    else:
        zz = z.clone().requires_grad_(True)
        lp = log_prob_fn(zz)
        (G,) = torch.autograd.grad(lp.sum(), zz)
        lp = lp.detach()
    # log q(z) of MultivariateNormalTriL: -1/2 |eps|^2 - sum log diag(L) - d/2 log 2pi
    log_q = -0.5 * (eps * eps).sum(-1) - log_det - 0.5 * d * math.log(2 * math.pi)
    elbo = (log_q - lp).mean()  # jax/inference.py:113-119
    g_mu = -G.mean(0)
    if diag_mode:
        g_lp = -(G * eps).mean(0) * sdiag - 1.0
    else:
        idx = _tril_idx(d, mu.device)
        gl = (-(G.T @ eps) / n_local)[idx[0], idx[1]]
        on_diag = idx[0] == idx[1]
        e = torch.exp(l_packed)  # only its diagonal entries are used: L_ii = exp(p_ii) + shift
        g_lp = torch.where(on_diag, gl * e - e / ldiag[idx[0]], gl)
    buf = torch.cat([elbo.reshape(1), g_mu, g_lp])  # ONE fused buffer -> ONE collective
    return gdist.allreduce_mean_(buf) if reduce else buf


def _halton2(i: int) -> float:
    """i-th point of the base-2 Halton (van der Corput) sequence in (0, 1): the trajectory-length jitter."""
    f, r = 0.5, 0.0
    while i > 0:
        r += f * (i & 1)
        i >>= 1
        f *= 0.5
    return r


class _TrajectoryLength:
    """State of the ChEES trajectory-length adaptation (see ``ModellingSequence.HMC``): the maximum trajectory length ``T``,
    the running mean of the squared gradient and the iterate average of ``log T``."""

    def __init__(self, initial, rate=0.025):
        self.T = float(initial)
        self.T_avg = float(initial)
        self.rate = float(rate)
        self.sq = 0.0
        self.step = 0

    def current(self, adapting):
        return self.T if adapting or self.step == 0 else self.T_avg

    def update(self, x, x_new, v_new, accept_prob, jitter, eps, max_leapfrog_steps):
        """One ascent step on ChEES.  ``x`` / ``x_new``: states before the transition / proposed, ``(n, d)``; ``v_new``: final
        velocity ``d x_new / d t``; ``accept_prob (n,)``; the jittered length is ``jitter * T`` so ``d / dT = jitter * d / dt``."""
        a = accept_prob.to(x.dtype)
        xc = x - x.mean(0, keepdim=True)
        xn = x_new - (a[:, None] * x_new).sum(0, keepdim=True) / (a.sum() + 1e-20)
        diff = (xn * xn).sum(-1) - (xc * xc).sum(-1)
        g = jitter * diff * (xn * v_new).sum(-1)  # d/dT of 1/4 diff^2, per chain
        g = torch.where((a > 1e-4) & torch.isfinite(g), g, torch.zeros_like(g))
        grad = float((g * a).sum() / (a.sum() + 1e-20))
        self.step += 1
        self.sq = 0.95 * self.sq + 0.05 * grad * grad
        sq_hat = self.sq / (1.0 - 0.95 ** self.step)
        log_update = min(max(self.rate * grad / math.sqrt(sq_hat + 1e-20), -0.35), 0.35)
        T = self.T * math.exp(log_update)
        self.T = min(T, eps * max_leapfrog_steps)  # never ask for more than max_leapfrog_steps steps
        w = self.step ** -0.5
        self.T_avg = math.exp(w * math.log(self.T) + (1.0 - w) * math.log(1e-10 + self.T_avg))


class ModellingSequence:
    """Drop-in for ``gigalens.tf.inference.ModellingSequence`` (MAP / SVI / HMC / SMC)."""

    def __init__(self, phys_model, prob_model, sim_config):
        self.phys_model = phys_model
        self.prob_model = prob_model
        self.sim_config = sim_config

    def _event_size(self, lens_sim):
        pm, n = self.prob_model, 0.0  # tf/inference.py:27-31
        if pm.include_pixels:
            n += float(torch.count_nonzero(lens_sim.img_region))
        if pm.include_positions:
            n += pm.n_position
        return n

    def MAP(self, optimizer: Adam, start=None, n_samples=500, num_steps=350, seed=0, progress=None, graph=None):
        """tf/inference.py:18-45.  ``n_samples`` is the GLOBAL count; each rank optimises its own shard and the
        solutions are gathered at the end (jax/inference.py:62-68).

        ``graph``: one optimisation step -- the native launch sequence of ``log_prob_and_grad`` plus the fused Adam
        update, 5-6 short kernels -- is captured once in a HIP graph and replayed.  Measured on one MI355X: the stepwise
        loop is host-issue bound at 30 us per step for SIE+Sersic 64x64 at 1-64 samples, the replay runs 25 us; at
        60x60 x 500 samples the graph is already 9 % slower than stream launches (53 vs 49 us) and 3 % slower at
        128x128 x 1024.  ``None`` (default) therefore picks the graph below 3e5 pixel-samples per step; ``True`` /
        ``False`` force it.  Learning-rate schedules and the autograd path always launch step by step."""
        rank, world = (torch.distributed.get_rank(), gdist.world_size()) if gdist.world_size() > 1 else (0, 1)
        _warn_remainder("MAP", "n_samples", n_samples, world)
        lo, hi = gdist.shard_bounds(n_samples, rank, world)
        n_local = hi - lo
        pm = self.prob_model
        if start is None:
            start = pm.prior.sample(n_samples, seed=seed)
        trial = pm.bij.inverse(start)[lo:hi].to(pm.device).contiguous().clone()
        pm.init_centroids(bs=n_local)
        lens_sim = LensSimulator(self.phys_model, self.sim_config, bs=n_local)
        event_size = self._event_size(lens_sim)
        red = None
        denom = event_size * n_local  # agg_loss = mean(-log_prob / event_size)  (tf/inference.py:36)
        if graph is None:
            graph = lens_sim.img_X.numel() * n_local <= 300_000
        use_graph = (graph and trial.is_cuda and num_steps > 8 and getattr(optimizer, "capturable", False)
                     and getattr(pm, "_fused_ok", lambda s: False)(lens_sim))
        step0 = 0
        if use_graph:
            side = torch.cuda.Stream(device=trial.device)
            side.wait_stream(torch.cuda.current_stream(trial.device))
            with torch.cuda.stream(side):  # warm-up off the default stream: binds the prior, sizes the workspaces
                for step0 in range(3):
                    _, red, g = pm.log_prob_and_grad(lens_sim, trial)
                    optimizer.step_captured(trial, g, -1.0 / denom)
                    if progress is not None:
                        progress(step0, red)
                step0 = 3
            torch.cuda.current_stream(trial.device).wait_stream(side)
            hip_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(hip_graph):
                _, red_static, g = pm.log_prob_and_grad(lens_sim, trial)
                optimizer.step_captured(trial, g, -1.0 / denom)
            # capture records, it does not run: every replay below is one real step
            for step in range(step0, num_steps):
                hip_graph.replay()
                if progress is not None:
                    progress(step, red_static)
            red = red_static.clone()
            optimizer.sync_from_device()
            del hip_graph
        else:
            for step in range(num_steps):
                log_prob, red, g = pm.log_prob_and_grad(lens_sim, trial)
                optimizer.step(trial, g, -1.0 / denom)
                if progress is not None:
                    progress(step, red)
        self.last_red_chi2 = red
        return gdist.gather_rows(trial)

    def SVI(self, optimizer: Adam, start_mean, n_vi=250, init_scales=1e-3, num_steps=500, seed=2, full_rank=True,
            progress=None, sync_every=200):
        """tf/inference.py:47-93 (full-rank or mean-field surrogate), sharded like jax/inference.py:91-144.

        Every rank applies the same all-reduced gradient to the same parameters, so the surrogates stay identical as long as
        the collective returns the same bits on every rank (RCCL's ring / tree all-reduce does; checked bitwise in the
        two-rank tests).  As a guard that costs one 35 KB broadcast per ``sync_every`` steps, rank 0's parameters and Adam
        moments overwrite the others' (SURVEY 8e); ``sync_every=0`` turns it off."""
        rank, world = (torch.distributed.get_rank(), gdist.world_size()) if gdist.world_size() > 1 else (0, 1)
        _warn_remainder("SVI", "n_vi", n_vi, world)
        n_local = max(1, n_vi // world)
        pm = self.prob_model
        lens_sim = LensSimulator(self.phys_model, self.sim_config, bs=n_local)
        pm.init_centroids(bs=n_local)
        mu = torch.as_tensor(start_mean, dtype=torch.float32, device=pm.device).reshape(-1).clone()
        d = mu.numel()
        scale = (torch.eye(d, device=pm.device) * float(init_scales) if not torch.is_tensor(init_scales)
                 else init_scales.to(pm.device))
        lp = tril_pack(scale) if full_rank else torch.log(torch.diagonal(scale))
        gen = gdist.rank_generator(seed, rank, device=pm.device if torch.device(pm.device).type == "cuda" else "cpu")
        params = torch.cat([mu, lp])
        losses = []

        def value_and_grad(z):
            lp_, _, g_ = pm.log_prob_and_grad(lens_sim, z)
            return lp_, g_

        pool = NormalPool(gen, n_local, d, dtype=params.dtype, device=params.device)
        for step in range(num_steps):
            buf = svi_step_buffer(params[:d], params[d:], None, n_local, gen, value_and_grad_fn=value_and_grad,
                                  full_rank=full_rank, eps=pool.next())
            loss = buf[0]
            optimizer.step(params, buf[1:])  # the fused buffer's tail IS the gradient of cat([mu, l_packed])
            if world > 1 and sync_every and (step + 1) % sync_every == 0:
                gdist.broadcast_(params)
                for t in optimizer.state_tensors():
                    gdist.broadcast_(t)
            losses.append(loss)  # stays on the device: no host round trip per step
            if progress is not None:
                progress(step, loss)
        losses = torch.stack(losses).tolist() if losses else []
        self.q_mean = params[:d].clone()
        self.q_scale_tril = tril_unpack(params[d:], d) if full_rank else torch.diag(torch.exp(params[d:]))
        return (self.q_mean, self.q_scale_tril), losses

    def HMC(self, q_z, init_eps=0.3, init_l=3, n_hmc=50, num_burnin_steps=250, num_results=750,
            max_leapfrog_steps=30, adapt_rate=0.05, adapt_mode="dual", seed=3, target_accept=0.75,
            trajectory_adaptation_rate=0.025):
        """tf/inference.py:95-182: preconditioned HMC (momentum precision = SVI covariance) wrapped, like the reference's
        kernel stack, in gradient-based trajectory-length adaptation and a step-size adaptation, both active for the first
        ``int(0.8 * num_burnin_steps)`` transitions (:140,150-162).  Chains are sharded over ranks with no collective
        (jax/inference.py:157-208); samples are gathered along the chain axis at the end.

        Trajectory length (``tfe.mcmc.GradientBasedTrajectoryLengthAdaptation`` with its default ChEES criterion; Hoffman,
        Radul & Sountsov 2021): every transition integrates for ``h_t * T`` with ``h_t`` the base-2 Halton sequence in (0, 1)
        and ``T`` the maximum trajectory length, i.e. ``ceil(h_t T / eps)`` leapfrog steps clipped to
        ``[1, max_leapfrog_steps]``; ``T`` starts at ``init_eps * init_l`` and climbs the gradient of
        ``ChEES = 1/4 E[(|x' - E x'|^2 - |x - E x|^2)^2]`` estimated over the chains (acceptance-weighted, the final
        velocity as ``d x' / d T``) with an RMSProp-normalised step on ``log T`` clipped to +-0.35, kept below
        ``eps * max_leapfrog_steps``; after the adaptation the iterate average of ``log T`` is used.  With several ranks
        each rank adapts on its own chains (the JAX driver adapts per device too)."""
        if adapt_mode not in ("dual", "simple"):
            raise ValueError(f"Invalid adaptation mode {adapt_mode}, the options are 'simple' and 'dual'")  # :163-164
        rank, world = (torch.distributed.get_rank(), gdist.world_size()) if gdist.world_size() > 1 else (0, 1)
        _warn_remainder("HMC", "n_hmc", n_hmc, world)
        n_local = max(1, n_hmc // world)
        pm = self.prob_model
        lens_sim = LensSimulator(self.phys_model, self.sim_config, bs=n_local)
        pm.init_centroids(bs=n_local)
        mean, L = q_z
        mean, L = mean.to(pm.device), L.to(pm.device)
        d = mean.numel()
        on_gpu = torch.device(pm.device).type == "cuda"
        gen = gdist.rank_generator(seed, rank, device=pm.device if on_gpu else "cpu")  # no host round trip per draw
        rnd = lambda *s: torch.randn(*s, generator=gen, device=gen.device).to(pm.device)
        z = mean + rnd(n_local, d) @ L.T
        # momentum ~ N(0, Sigma^-1)  <=>  p = L^-T xi ; kinetic energy 1/2 p^T Sigma p = 1/2 |L^T p|^2
        # d x d, once per run: inverted on the host (a first rocSOLVER call costs ~0.3 s of library start-up)
        Linv_T = torch.linalg.inv(L.detach().cpu().double()).T.to(device=pm.device, dtype=L.dtype)
        Sigma = L @ L.T

        def value_and_grad(zz):
            lp, _, g = pm.log_prob_and_grad(lens_sim, zz)
            return lp, g

        lp, g = value_and_grad(z)
        max_leapfrog_steps = max(int(max_leapfrog_steps), 1)
        num_adaptation_steps = int(num_burnin_steps * 0.8)  # :140
        log_eps, log_eps_bar, h_bar, mu_da = math.log(init_eps), 0.0, 0.0, math.log(10 * init_eps)
        traj = _TrajectoryLength(init_eps * min(max(int(init_l), 1), max_leapfrog_steps), trajectory_adaptation_rate)
        samples, accept_hist, leap_hist = [], [], []
        native = on_gpu and z.dtype == torch.float32
        if native:  # one launch per kick+drift, one per Metropolis step (gl_hmc_kick_drift / gl_hmc_accept)
            z, g, lp = z.contiguous(), g.contiguous().clone(), lp.contiguous().clone()
            L_c, Sigma_c = L.contiguous(), Sigma.contiguous()
            zn, pn = torch.empty_like(z), torch.empty_like(z)
            acc_buf = torch.empty(n_local, dtype=torch.float32, device=z.device)
        for it in range(num_burnin_steps + num_results):
            eps = math.exp(log_eps)
            adapting = it < num_adaptation_steps
            jitter = _halton2(it + 1)
            n_leap = int(min(max(math.ceil(jitter * traj.current(adapting) / eps), 1), max_leapfrog_steps))
            leap_hist.append(n_leap)
            p0 = rnd(n_local, d) @ Linv_T.T
            z_prev = z.clone() if adapting else None
            if native:
                _native.hmc_kick_drift(p0, g, 0.5 * eps, z, Sigma_c, eps, pn, zn)
                for i in range(n_leap):
                    lpn, gn = value_and_grad(zn)
                    if i < n_leap - 1:
                        _native.hmc_kick_drift(pn, gn, eps, zn, Sigma_c, eps, pn, zn)
                u = torch.rand(n_local, generator=gen, device=gen.device)
                if adapting:
                    z_prop, v_prop = zn.clone(), (pn + 0.5 * eps * gn) @ Sigma
                _native.hmc_accept(z, g, lp, zn, gn.contiguous(), lpn.contiguous(), p0, pn, 0.5 * eps, L_c, u, acc_buf)
                acc_prob = acc_buf
            else:
                zn, pn, gn, lpn = z, p0, g, lp
                pn = pn + 0.5 * eps * gn
                for i in range(n_leap):
                    zn = zn + eps * (pn @ Sigma)
                    lpn, gn = value_and_grad(zn)
                    pn = pn + (eps if i < n_leap - 1 else 0.5 * eps) * gn
                ke0 = 0.5 * ((p0 @ L) ** 2).sum(-1)
                ke1 = 0.5 * ((pn @ L) ** 2).sum(-1)
                log_acc = (lpn - ke1) - (lp - ke0)
                log_acc = torch.where(torch.isfinite(log_acc), log_acc, torch.full_like(log_acc, -float("inf")))
                acc = torch.log(torch.rand(n_local, generator=gen, device=gen.device).to(pm.device)) < log_acc
                z_prop, v_prop = zn, pn @ Sigma
                z = torch.where(acc[:, None], zn, z)
                g = torch.where(acc[:, None], gn, g)
                lp = torch.where(acc, lpn, lp)
                acc_prob = torch.exp(torch.clamp(log_acc, max=0.0))
            a_dev = acc_prob.mean()
            accept_hist.append(a_dev)  # read back once at the end; only the adaptation below needs it on the host
            if adapting:
                traj.update(z_prev, z_prop, v_prop, acc_prob, jitter, eps, max_leapfrog_steps)
                a = float(a_dev)
                if adapt_mode == "simple":
                    # tfp.mcmc.SimpleStepSizeAdaptation (tf/inference.py:159-162): multiplicative nudge towards the target
                    log_eps += math.log1p(adapt_rate) if a > target_accept else -math.log1p(adapt_rate)
                else:  # Nesterov dual averaging (Hoffman & Gelman 2014, alg. 5)
                    m = it + 1
                    h_bar = (1 - 1 / (m + 10)) * h_bar + (target_accept - a) / (m + 10)
                    log_eps = mu_da - math.sqrt(m) / adapt_rate * h_bar
                    eta = m ** -0.75
                    log_eps_bar = eta * log_eps + (1 - eta) * log_eps_bar
                    if it == num_adaptation_steps - 1:
                        log_eps = log_eps_bar
            if it >= num_burnin_steps:
                samples.append(z.clone())
        out = torch.stack(samples)  # (num_results, n_local, d)
        if world > 1:
            out = gdist.gather_rows(out.permute(1, 0, 2).contiguous()).permute(1, 0, 2)
        accept_hist = torch.stack(accept_hist).tolist() if accept_hist else []
        return out, {"accept": accept_hist, "step_size": math.exp(log_eps), "num_leapfrog_steps": leap_hist,
                     "max_trajectory_length": traj.current(False)}

    def SMC(self, start=None, num_particles=1000, num_ensembles=1, num_leapfrog_steps=10, post_sampling_steps=100,
            ess_threshold_ratio=0.5, max_sampling_per_stage=8, target="pixels", auxiliar="positions", seed=1,
            max_stage=100):
        """tf/inference.py:184-288: adaptive tempered sequential Monte Carlo with HMC mutation.

        Particles start from the prior (or from ``start``) and move through the targets
        ``prior + aux + beta (like - aux)`` (``make_tempered_target_log_prob_fn_with_auxiliar``, :292-303), ``beta``
        going 0 -> 1.  Each stage restates what ``tfp.experimental.mcmc.sample_sequential_monte_carlo`` does with the
        reference's arguments: next ``beta`` by bisection so that the effective sample size of the incremental
        weights is ``0.8 N`` (the reference hard-codes 0.8, :243; ``ess_threshold_ratio`` is accepted and unused
        there too), systematic resampling, then 1..``max_sampling_per_stage`` HMC transitions whose per-particle
        step-size scalings follow ``simple_heuristic_tuning(optimal_accept=0.651)``.  ``num_ensembles`` independent
        populations run side by side.  Returns ``(samples, info)``; with ``post_sampling_steps > 0`` the samples are
        the HMC chain continued at ``beta = 1`` (:270-281), shape ``(steps, particles * ensembles, d)``.
        The stochastic driver cannot be pinned bit-for-bit against TFP (different random streams): tests check the
        evidence and posterior moments on a conjugate toy problem and against HMC.
        """
        del ess_threshold_ratio  # as in the reference (:243 passes 0.8)
        pm = self.prob_model
        N, E = int(num_particles), int(num_ensembles)
        n_all = N * E
        gen = gdist.rank_generator(seed, 0, device="cpu")
        randn = lambda *s: torch.randn(*s, generator=gen).to(pm.device)
        rand = lambda *s: torch.rand(*s, generator=gen).to(pm.device)
        if start is None:
            z = pm.bij.inverse(pm.prior.sample(n_all, seed=seed)).to(pm.device)
        else:
            flat = torch.as_tensor(start, dtype=torch.float32).reshape(-1, start.shape[-1])
            pick = torch.randint(0, flat.shape[0], (n_all,), generator=gen)
            z = flat[pick].to(pm.device)
        d = z.shape[-1]
        lens_sim = LensSimulator(self.phys_model, self.sim_config, bs=n_all)
        pm.init_centroids(bs=n_all)

        def term(zz, name):
            if name == "none":  # (:215) no auxiliary / no likelihood: prior only
                lp = pm.log_prior(zz)
                zz_ = zz.detach().requires_grad_(True)
                (g,) = torch.autograd.grad(pm.log_prior(zz_).sum(), zz_)
                return lp, torch.zeros_like(lp), g
            return pm.term_log_prob_and_grad(lens_sim, zz, name)

        def evaluate(zz):
            """(prior + like, like, grad), (prior + aux, aux, grad)"""
            return term(zz, target), term(zz, auxiliar)

        def tempered(ev, beta):
            (lpl, ll, gl), (lpa, la, ga) = ev
            b = beta.repeat_interleave(N) if beta.numel() > 1 else beta  # beta per ensemble, particle-major layout below
            return (1 - b) * lpa + b * lpl, (1 - b)[:, None] * ga + b[:, None] * gl

        # layout: row = e * N + i  (ensemble-major) so that per-ensemble statistics are contiguous
        ev = evaluate(z)
        beta = torch.zeros(E, device=pm.device)
        log_scal = torch.full((n_all,), math.log(1.0 / d ** 0.25), device=pm.device)  # HMC default d^(-1/4) scaling
        log_Z = torch.zeros(E, device=pm.device)
        n_mut = 1
        stages = []
        for stage in range(max_stage):
            if bool((beta >= 1.0).all()):
                break
            dl = (ev[0][1] - ev[1][1]).reshape(E, N)  # like - aux
            dl = torch.where(torch.isfinite(dl), dl, torch.full_like(dl, -float("inf")))
            # next inverse temperature: largest step whose incremental weights keep ESS >= 0.8 N (bisection)
            lo, hi = beta.clone(), torch.ones_like(beta)
            def ess(bn):
                lw = (bn - beta)[:, None] * dl
                lw = lw - lw.max(dim=1, keepdim=True).values
                w = torch.exp(lw)
                return w.sum(1) ** 2 / (w * w).sum(1)
            ok = ess(hi) >= 0.8 * N
            for _ in range(40):
                mid = 0.5 * (lo + hi)
                good = ess(mid) >= 0.8 * N
                lo = torch.where(good, mid, lo)
                hi = torch.where(good, hi, mid)
            new_beta = torch.where(ok, torch.ones_like(beta), lo)
            new_beta = torch.maximum(new_beta, beta + 1e-6).clamp(max=1.0)
            lw = (new_beta - beta)[:, None] * dl
            m = lw.max(dim=1, keepdim=True).values
            log_Z = log_Z + (m[:, 0] + torch.log(torch.exp(lw - m).mean(1)))  # marginal-likelihood increment
            w = torch.softmax(lw, dim=1)
            # systematic resampling per ensemble
            u = (rand(E, 1) + torch.arange(N, device=pm.device)[None, :]) / N
            idx = torch.searchsorted(torch.cumsum(w, 1).contiguous(), u.contiguous()).clamp(max=N - 1)
            rows = (idx + (torch.arange(E, device=pm.device) * N)[:, None]).reshape(-1)
            z = z[rows]
            log_scal = log_scal[rows]
            ev = tuple(tuple(t[rows] for t in tr) for tr in ev)
            beta = new_beta
            # mutation: n_mut HMC transitions on the tempered target, step = scaling * population std
            std = z.reshape(E, N, d).std(dim=1).clamp_min(1e-8).repeat_interleave(N, dim=0)
            acc_sum = torch.zeros(n_all, device=pm.device)
            for _ in range(n_mut):
                eps = torch.exp(log_scal)[:, None] * std
                lp0, g0 = tempered(ev, beta)
                p0 = randn(n_all, d)
                zn, pn = z, p0 + 0.5 * eps * g0
                evn = ev
                for i in range(num_leapfrog_steps):
                    zn = zn + eps * pn
                    evn = evaluate(zn)
                    lpn, gn = tempered(evn, beta)
                    pn = pn + (eps if i < num_leapfrog_steps - 1 else 0.5 * eps) * gn
                log_acc = (lpn - 0.5 * (pn * pn).sum(-1)) - (lp0 - 0.5 * (p0 * p0).sum(-1))
                log_acc = torch.where(torch.isfinite(log_acc), log_acc, torch.full_like(log_acc, -float("inf")))
                acc = torch.log(rand(n_all)) < log_acc
                z = torch.where(acc[:, None], zn, z)
                ev = tuple(tuple(torch.where(acc if a.dim() == 1 else acc[:, None], a, b) for a, b in zip(tn, to))
                           for tn, to in zip(evn, ev))
                acc_sum = acc_sum + torch.exp(log_acc.clamp(max=0.0))
            acc_rate = (acc_sum / n_mut).reshape(E, N)
            # simple_heuristic_tuning: per-ensemble average acceptance drives the scalings towards 0.651 and sets the
            # number of transitions of the next stage (enough for a 99 % chance that every particle moved)
            avg = acc_rate.mean(1).clamp(1e-6, 1 - 1e-6)
            avg_log_scal = log_scal.reshape(E, N).mean(1)
            new_ls = 0.5 * (avg_log_scal[:, None] + (log_scal.reshape(E, N) + torch.log(acc_rate.clamp_min(1e-6)) - math.log(0.651)))
            log_scal = new_ls.reshape(-1)
            n_mut = int(min(max(math.ceil(math.log(0.01) / math.log1p(-float(avg.min()))), 1), max_sampling_per_stage))
            stages.append(dict(beta=beta.cpu().tolist(), accept=float(avg.mean()), n_mut=n_mut))
        samples = z.reshape(E, N, d).permute(1, 0, 2).contiguous()  # (particles, ensembles, d) like the reference
        info = dict(stages=stages, log_evidence=log_Z.cpu(), log_scalings=log_scal.reshape(E, N).cpu())
        if post_sampling_steps <= 0:
            return samples, info
        std = z.reshape(E, N, d).std(dim=1).clamp_min(1e-8).repeat_interleave(N, dim=0)
        eps = torch.exp(log_scal)[:, None] * std
        one = torch.ones(1, device=pm.device)
        chain = []
        for _ in range(post_sampling_steps):
            lp0, g0 = tempered(ev, one)
            p0 = randn(n_all, d)
            zn, pn = z, p0 + 0.5 * eps * g0
            for i in range(num_leapfrog_steps):
                zn = zn + eps * pn
                evn = evaluate(zn)
                lpn, gn = tempered(evn, one)
                pn = pn + (eps if i < num_leapfrog_steps - 1 else 0.5 * eps) * gn
            log_acc = (lpn - 0.5 * (pn * pn).sum(-1)) - (lp0 - 0.5 * (p0 * p0).sum(-1))
            log_acc = torch.where(torch.isfinite(log_acc), log_acc, torch.full_like(log_acc, -float("inf")))
            acc = torch.log(rand(n_all)) < log_acc
            z = torch.where(acc[:, None], zn, z)
            ev = tuple(tuple(torch.where(acc if a.dim() == 1 else acc[:, None], a, b) for a, b in zip(tn, to))
                       for tn, to in zip(evn, ev))
            chain.append(z.clone())
        return torch.stack(chain), info
