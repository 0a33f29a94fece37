"""Worst-case accuracy of the fused forward+gradient call's PARAMETER GRADIENT at full config size against the float64
oracle (torch.autograd through the restated reference graph), per parameter column.

For every BASELINE config the first `n_check` samples of the batch are differentiated by the oracle at the config's full
pixel grid; the error of the float32 HIP gradient is reported per column k as
    max_b |g[b,k] - g_o[b,k]| / max(|g_o[b,k]|, floor * S_k),     S_k = max_b |g_o[b,k]|  (column scale),
i.e. relative to the element itself unless the element is tiny within its own column.  This is the gate form of
tests/test_gpu_parity.py (GRAD_RTOL, GRAD_FLOOR).  Output: one JSON line per config with the worst column and all columns.

    python tools/dev/grad_accuracy_scan.py [--configs C1,C2,C3,C3D,C4] [--n 8] [--floor 1e-2]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gigalens_amd import workloads  # noqa: E402
from gigalens_amd.model import ForwardProbModel  # noqa: E402
from gigalens_amd.simulator import LensSimulator  # noqa: E402
import helpers as H  # noqa: E402


def column_names(phys):
    out = []
    for grp, profs in zip(("L", "LL", "S"), (phys.lenses, phys.lens_light, phys.source_light)):
        for i, p in enumerate(profs):
            names = p._native_params() if hasattr(p, "_native_params") else p.params
            out += [f"{grp}{i}.{n}" for n in names]
    return out


def scan(name, n_check, floor, **kw):
    wl = workloads.make(name, **kw)
    obs, err, _ = workloads.synthetic_observation(wl, LensSimulator)
    sim = LensSimulator(wl.phys_model, wl.sim_config, bs=wl.batch)
    packed = H.sample_packed(wl, sim, seed=5)
    err_np = None if err is None else err.cpu().numpy()
    pm = ForwardProbModel(wl.prior, obs.cpu().numpy(), wl.background_rms, wl.exp_time, error_map=err_np, include_positions=False)
    p = packed.clone().requires_grad_(True)
    ll, _ = pm._pixel_stats_packed(sim, p)
    ll.sum().backward()
    n_check = min(n_check, wl.batch)
    g = p.grad[:n_check].double().cpu().numpy()
    step = 4 if sim._model.N > 16384 else 8
    g_o = []
    for i0 in range(0, n_check, step):
        m = min(step, n_check - i0)
        wl_o = workloads.make(name, **{**kw, "batch": m})
        _, _, go, _ = H.oracle_loglike_and_grad(wl_o, packed[i0:i0 + m].double().cpu(), obs.cpu().numpy(), err_np, m)
        g_o.append(go)
    g_o = np.concatenate(g_o)
    # the reference algorithm itself in float32 (the oracle's op-for-op restatement of the TF graph, torch autograd): how far
    # float32 evaluation alone moves the gradient from the float64 truth -- the yardstick for the HIP path's own distance
    g_32 = []
    for i0 in range(0, n_check, step):
        m = min(step, n_check - i0)
        wl_o = workloads.make(name, **{**kw, "batch": m})
        _, _, g32, _ = H.oracle_loglike_and_grad(wl_o, packed[i0:i0 + m].float().cpu(), obs.cpu().numpy(), err_np, m, dtype=torch.float32)
        g_32.append(g32.astype(np.float64))
    g_32 = np.concatenate(g_32)
    S = np.abs(g_o).max(axis=0, keepdims=True)
    rel = np.abs(g - g_o) / np.maximum(np.abs(g_o), floor * S + 1e-300)
    per_col = rel.max(axis=0)
    rel32 = np.abs(g_32 - g_o) / np.maximum(np.abs(g_o), floor * S + 1e-300)
    per_col32 = rel32.max(axis=0)
    cs_hip = (np.abs(g - g_o) / np.maximum(S, 1e-300)).max(axis=0)    # error relative to the COLUMN scale S_k
    cs_f32 = (np.abs(g_32 - g_o) / np.maximum(S, 1e-300)).max(axis=0)
    rowmax = np.abs(g_o).max(axis=1, keepdims=True)
    old_form = (np.abs(g - g_o) / rowmax).max()
    names = column_names(wl.phys_model)
    worst = int(per_col.argmax())
    out = {"config": name, "kw": {k: v for k, v in kw.items()}, "batch": wl.batch, "pixels": sim._model.N, "n_checked": n_check,
           "floor": floor, "worst_rel": float(per_col[worst]), "worst_column": names[worst] if worst < len(names) else worst,
           "p50_column_rel": float(np.median(per_col)), "err_over_row_max": float(old_form),
           "col_scale_rel_worst": float(cs_hip.max()), "col_scale_rel_worst_column": names[int(cs_hip.argmax())],
           "f32_oracle_col_scale_rel_worst": float(cs_f32.max()),
           "f32_oracle_col_scale_rel_in_that_column": float(cs_f32[int(cs_hip.argmax())]),
           "f32_oracle_worst_rel": float(per_col32.max()), "f32_oracle_worst_column": names[int(per_col32.argmax())],
           "f32_oracle_rel_in_hip_worst_column": float(per_col32[worst]),
           "hip_over_f32_oracle_worst_ratio": float((per_col / np.maximum(per_col32, 1e-12)).max()),
           "columns": {names[k] if k < len(names) else str(k): [float(f"{per_col[k]:.3e}"), float(f"{per_col32[k]:.3e}")]
                       for k in range(len(per_col))}}
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C1,C2,C3,C3direct,C3D,C4")
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--floor", type=float, default=1e-2)
    a = ap.parse_args()
    if a.configs == "cases":  # the reduced-size cases of tests/test_gpu_parity.py::CASES (what the test gate has to hold on)
        sys.path.insert(0, ROOT)
        from tests.test_gpu_parity import CASES
        for name, kw in CASES:
            scan(name, 64, a.floor, **kw)
        sys.exit(0)
    for c in a.configs.split(","):
        if c == "C1":
            scan("C1", a.n, a.floor, batch=64)
        elif c == "C3direct":
            scan("C3", a.n, a.floor, interpolate=False)
        elif c in ("C3", "C3D"):
            scan(c, a.n, a.floor, interpolate=True)
        else:
            scan(c, a.n, a.floor)
