#!/usr/bin/env python3
"""Secondary measurements (not the headline bench): forward+gradient sims/s of every BASELINE config and of the
reference's tf-demo set-up, through the product API (ForwardProbModel.log_prob_and_grad), HIP-event timed."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gigalens_amd import workloads  # noqa: E402
from gigalens_amd.model import ForwardProbModel, PhysicalModel  # noqa: E402
from gigalens_amd.simulator import LensSimulator, SimulatorConfig  # noqa: E402


def _timed(pm, sim, z, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        pm.log_prob_and_grad(sim, z)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def time_step(pm, sim, z, warm=5, seconds=0.25):
    """Steady-state time per step: a short probe sizes the run to ~``seconds`` of sustained load (the chip settles at its
    working clock only after tens of milliseconds; 30-step bursts read up to 15 % slow), like bench.py's 1000 steps."""
    for _ in range(warm):
        pm.log_prob_and_grad(sim, z)
    probe = _timed(pm, sim, z, 20)
    iters = int(min(max(seconds * 1e3 / probe, 30), 3000))
    _timed(pm, sim, z, max(iters // 4, 10))  # ramp
    return _timed(pm, sim, z, iters)


def run(name, wl, supersampled_kernel=None, obs=None):
    if obs is None:
        obs, err, _ = workloads.synthetic_observation(wl, lambda p, c, bs: LensSimulator(p, c, bs, supersampled_kernel=supersampled_kernel))
    else:
        err = None
    pm = ForwardProbModel(wl.prior, np.asarray(obs.cpu() if torch.is_tensor(obs) else obs), wl.background_rms, wl.exp_time,
                          error_map=None if err is None else err.cpu().numpy(), include_positions=False)
    sim = LensSimulator(wl.phys_model, wl.sim_config, bs=wl.batch, supersampled_kernel=supersampled_kernel)
    z = pm.bij.inverse(wl.prior.sample(wl.batch, seed=0)).to("cuda").contiguous()
    ms = time_step(pm, sim, z)
    out = dict(config=name, batch=wl.batch, pixels=sim._model.N, params=sim._model.P, ms_per_step=round(ms, 4),
               sims_per_s=round(wl.batch / (ms * 1e-3), 1))
    print(json.dumps(out), flush=True)
    return out


def main():
    res = [run("C1 SIE+Sersic 64x64 B=1", workloads.make("C1")),
           run("C1 SIE+Sersic 64x64 B=1024", workloads.make("C1", batch=1024)),
           run("C2 EPL+Shear|Sersic 128x128 B=1024", workloads.make("C2")),
           run("C3 shapelets n_max=10 (table) 128x128 B=1024", workloads.make("C3", interpolate=True)),
           run("C3 shapelets n_max=10 (direct) 128x128 B=1024", workloads.make("C3", interpolate=False)),
           run("C3D shapelets-demo model (lens light + shapelets n_max=8, table) 128x128 B=1024", workloads.make("C3D", n_max=8)),
           run("C4 8 NFW + 20 Sersic 256x256 B=512", workloads.make("C4")),
           run("C6 dPIE halo + 200 scaled dPIE galaxies + 20 Sersic 256x256 B=128", workloads.make("C6")),
           run("C6S same, galaxies through the order-3 series expansion, B=128", workloads.make("C6S")),
           run("C6S same, B=512", workloads.make("C6S", batch=512))]
    # the reference's tf-demo MAP set-up: EPL+Shear | SersicEllipse | SersicEllipse,
    # 60x60 px, supersample 2, 13x13 PSF (here block-replicated to the supersampled grid), 500 samples per step
    from tests.test_prior_host import default_prior
    from gigalens_amd.profiles.light.sersic import SersicEllipse
    from gigalens_amd.profiles.mass.epl import EPL
    from gigalens_amd.profiles.mass.shear import Shear
    psf = np.load(os.path.join(ROOT, "tests", "golden", "reference_assets", "psf.npy")).astype(np.float32)
    obs = np.load(os.path.join(ROOT, "tests", "golden", "reference_assets", "demo.npy")).astype(np.float32)
    phys = PhysicalModel([EPL(50), Shear()], [SersicEllipse()], [SersicEllipse()])
    for ss, k in ((2, np.kron(psf, np.ones((2, 2), np.float32) / 4)), (1, psf)):
        wl = workloads.Workload("DEMO", phys, default_prior(), SimulatorConfig(delta_pix=0.065, num_pix=60, supersample=ss), 500)
        res.append(run(f"tf-demo MAP step: 60x60, supersample {ss}, PSF {k.shape[0]}x{k.shape[1]}, B=500", wl,
                       supersampled_kernel=k, obs=obs))
    return res


if __name__ == "__main__":
    main()
