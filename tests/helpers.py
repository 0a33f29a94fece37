"""Shared helpers for the parity tests (oracle <-> product plumbing)."""
import numpy as np
import torch

GROUPS = ("lens_mass", "lens_light", "source_light")


def struct_from_packed(phys, packed):
    """[B,P] component-major packed rows -> the reference's nested {'lens_mass': [{name: (B,)}], ...}."""
    k = 0
    out = {}
    for g, profs in zip(GROUPS, (phys.lenses, phys.lens_light, phys.source_light)):
        lst = []
        for p in profs:
            d = {}
            names = p._native_params() if hasattr(p, "_native_params") else p.params
            for n in names:  # native column order; columns that are not sampled parameters (lstsq amplitudes) are skipped
                if n in p.params:
                    d[n] = packed[:, k]
                k += 1
            lst.append(d)
        out[g] = lst
    assert k == packed.shape[1]
    return out


def sample_packed(wl, sim, seed):
    """Draw B parameter sets from the workload prior and pack them (float32, on sim.device)."""
    x = wl.prior.sample(sim.bs, seed=seed)
    return sim.pack(x)


def oracle_loglike_and_grad(wl, packed64, obs, err, bs, dtype=torch.float64, grid_shift=None, supersampled_kernel=None):
    """``grid_shift = (dx, dy)`` (broadcastable to the oracle's ``(N, bs)`` grid) moves the points the oracle evaluates the model
    at: what float32_conditioning_bound perturbs."""
    from oracle import ref_torch as ref
    rs = ref.RefSimulator(wl.phys_model, wl.sim_config, bs, dtype=dtype, supersampled_kernel=supersampled_kernel)
    if grid_shift is not None:
        rs.img_X = rs.img_X + torch.as_tensor(grid_shift[0], dtype=dtype)
        rs.img_Y = rs.img_Y + torch.as_tensor(grid_shift[1], dtype=dtype)
    p = packed64.clone().to(dtype).requires_grad_(True)
    params = struct_from_packed(wl.phys_model, p)
    ll, red = ref.stats_pixels(rs, params, obs, wl.background_rms, wl.exp_time, error_map=err)
    (g,) = torch.autograd.grad(ll.sum(), p)
    img = rs.simulate(struct_from_packed(wl.phys_model, packed64.to(dtype)))
    return ll.detach().numpy(), red.detach().numpy(), g.numpy(), img.detach().numpy()


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def oracle_loglike_chunked(wl, packed64, obs, err, n, step=16):
    """Float64 oracle log-likelihood of the first ``n`` rows, forward only, ``step`` rows at a time (full-size pixel grids)."""
    from oracle import ref_torch as ref
    out = []
    with torch.no_grad():
        for i0 in range(0, n, step):
            m = min(step, n - i0)
            rs = ref.RefSimulator(wl.phys_model, wl.sim_config, m, dtype=torch.float64)
            params = struct_from_packed(wl.phys_model, packed64[i0:i0 + m])
            ll, _ = ref.stats_pixels(rs, params, obs, wl.background_rms, wl.exp_time, error_map=err)
            out.append(ll.numpy())
    return np.concatenate(out)


def grad_col_err(g, g_o):
    """Gradient error per element relative to the scale of its own parameter COLUMN, ``S_k = max_b |g_o[b, k]|``: the gate of the
    pixel-likelihood parity tests (a column whose values are orders of magnitude below the row's largest entry is held to
    its own scale, not the row's).  Columns that vanish identically on this batch are held to 1e-3 of the batch-wide
    maximum (table-mode shapelets: a source that misses the field has an all-zero row)."""
    g, g_o = np.asarray(g, dtype=np.float64), np.asarray(g_o, dtype=np.float64)
    finite = np.where(np.isfinite(g_o), np.abs(g_o), 0.0)
    S = np.maximum(finite.max(axis=0, keepdims=True), 1e-3 * finite.max())
    return np.abs(g - g_o) / np.maximum(S, 1e-300)


def float32_conditioning_bound(wl, packed64, obs, err, bs, g_o, S, bg=None, supersampled_kernel=None, seed=0):
    """How far float32 ROUNDING OF THE RAY-SHOOT ALONE moves each element of the float64 gradient ``g_o`` -- the yardstick for
    gradient elements whose error exceeds the plain tolerance.

    Any float32 implementation forms beta = x - alpha with an error of about one unit in the last place of the coordinates
    (2.4e-7 arcsec for |x| < 4; measured: 2.5e-7 rms for the HIP pair kernels and for the reference's algorithm in float32 alike,
    tools/dev/epl_beta_check.hip), and the gradient w.r.t. a source or lens centre of a sample whose source maps next to a pixel has
    a condition number of 1e3-1e4 in such offsets (tools/dev/psf_grad_probe.py: a uniform 1e-7 offset of beta moves
    d loglike / d center_y of one row by 1.1e-3 of its column scale; R^(1/n - 2) near the centre of a Sersic profile).  The bound
    is measured ON THE ORACLE: its grid is displaced by d = ulp(max |x|) -- uniformly along (+,+) and (+,-), and by two draws of
    pixel noise of that rms -- and the largest change of each gradient element, relative to the column scale ``S``, is returned
    (shape of ``g_o``).  An implementation whose error stays within a small multiple of it loses nothing that float32 evaluation
    of the reference's own algorithm would keep."""
    from oracle import ref_torch as ref
    rs = ref.RefSimulator(wl.phys_model, wl.sim_config, 1, dtype=torch.float64)
    xmax = float(max(rs.img_X.abs().max(), rs.img_Y.abs().max()))
    d = float(np.spacing(np.float32(xmax)))
    rng = np.random.default_rng(seed)
    n_pts = rs.img_X.shape[0]
    shifts = [(d, d), (d, -d), (d * rng.normal(size=(n_pts, 1)), d * rng.normal(size=(n_pts, 1))),
              (d * rng.normal(size=(n_pts, 1)), d * rng.normal(size=(n_pts, 1)))]
    out = np.zeros_like(np.asarray(g_o, dtype=np.float64))
    for sh in shifts:
        kw = dict(grid_shift=sh, supersampled_kernel=supersampled_kernel)
        if bg is not None:
            g_p = _oracle_grad_with(wl, packed64, obs, err, bs, bg, **kw)
        else:
            _, _, g_p, _ = oracle_loglike_and_grad(wl, packed64, obs, err, bs, **kw)
        out = np.maximum(out, np.abs(g_p - g_o) / np.maximum(S, 1e-300))
    return out


def _oracle_grad_with(wl, packed64, obs, err, bs, bg, grid_shift=None, supersampled_kernel=None):
    """Gradient of the oracle's pixel log-likelihood with an explicit (background_rms, exp_time) = ``bg`` (tests that do not take
    them from the workload)."""
    from oracle import ref_torch as ref
    rs = ref.RefSimulator(wl.phys_model, wl.sim_config, bs, dtype=torch.float64, supersampled_kernel=supersampled_kernel)
    if grid_shift is not None:
        rs.img_X = rs.img_X + torch.as_tensor(grid_shift[0], dtype=torch.float64)
        rs.img_Y = rs.img_Y + torch.as_tensor(grid_shift[1], dtype=torch.float64)
    p = packed64.clone().double().requires_grad_(True)
    ll, _ = ref.stats_pixels(rs, struct_from_packed(wl.phys_model, p), obs, bg[0], bg[1], error_map=err)
    (g,) = torch.autograd.grad(ll.sum(), p)
    return g.numpy()


def grad_gate(g, g_o, rtol, cond_bound_fn, k_cond=4.0):
    """The gradient gate of the pixel-likelihood parity tests: every element within ``rtol`` of its column's scale (taken from
    the ORACLE rows), or -- for the elements beyond it -- within ``k_cond`` x the float32 conditioning bound of that element
    (``cond_bound_fn(S)``, computed only when needed).  Returns (ok, report)."""
    g, g_o = np.asarray(g, dtype=np.float64), np.asarray(g_o, dtype=np.float64)
    finite = np.where(np.isfinite(g_o), np.abs(g_o), 0.0)
    S = np.maximum(finite.max(axis=0, keepdims=True), 1e-3 * finite.max())
    e = np.abs(g - g_o) / S
    bad = ~(e <= rtol)
    if not bad.any():
        return True, dict(worst=float(e.max()), conditioned=0)
    bound = cond_bound_fn(S)
    still = bad & ~(e <= rtol + k_cond * bound)
    rep = dict(worst=float(e.max()), conditioned=int(bad.sum()), worst_over_bound=float((e[bad] / np.maximum(bound[bad], 1e-30)).max()),
               unexplained=[(int(i), int(j), float(e[i, j]), float(bound[i, j])) for i, j in np.argwhere(still)[:8]])
    return not still.any(), rep
