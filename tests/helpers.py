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


def oracle_loglike_and_grad(wl, packed64, obs, err, bs, dtype=torch.float64):
    from oracle import ref_torch as ref
    rs = ref.RefSimulator(wl.phys_model, wl.sim_config, bs, dtype=dtype)
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
