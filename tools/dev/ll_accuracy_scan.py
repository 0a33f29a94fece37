"""Worst-case accuracy of the fused log-likelihood at FULL config size against the float64 oracle over many samples
(the oracle runs forward-only in chunks): tells apart fp32 rounding from formulation weaknesses (how the NFW closed form's
2e-6 loss in 0.6 < X < 0.95 was found)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gigalens_amd import workloads
from gigalens_amd.model import ForwardProbModel
from gigalens_amd.simulator import LensSimulator
from oracle import ref_torch as ref
import helpers as H

def scan(name, n_check, **kw):
    wl = workloads.make(name, **kw)
    obs, err, _ = workloads.synthetic_observation(wl, LensSimulator)
    sim = LensSimulator(wl.phys_model, wl.sim_config, bs=wl.batch)
    packed = H.sample_packed(wl, sim, seed=5)
    pm = ForwardProbModel(wl.prior, obs.cpu().numpy(), wl.background_rms, wl.exp_time,
                          error_map=None if err is None else err.cpu().numpy(), include_positions=False)
    p = packed.clone().requires_grad_(True)
    ll, _ = pm._pixel_stats_packed(sim, p)
    ll.sum().backward()
    ll = ll.detach().double().cpu().numpy()
    rel = []
    step = 16
    with torch.no_grad():
        for i0 in range(0, n_check, step):
            rs = ref.RefSimulator(wl.phys_model, wl.sim_config, step, dtype=torch.float64)
            params = H.struct_from_packed(wl.phys_model, packed[i0:i0 + step].double().cpu())
            ll_o, _ = ref.stats_pixels(rs, params, obs.cpu().numpy(), wl.background_rms, wl.exp_time,
                                       error_map=None if err is None else err.cpu().numpy())
            rel.append(np.abs(ll[i0:i0 + step] - ll_o.numpy()) / np.abs(ll_o.numpy()))
    rel = np.concatenate(rel)
    print(f"{name} {kw}: {n_check} samples, fused ll vs float64 oracle: max rel {rel.max():.2e} (sample {rel.argmax()}), "
          f"p99 {np.quantile(rel, 0.99):.2e}, median {np.median(rel):.2e}", flush=True)

scan("C2", 256)
scan("C3", 128, interpolate=True)
scan("C3", 128, interpolate=False)
scan("C4", 64)
