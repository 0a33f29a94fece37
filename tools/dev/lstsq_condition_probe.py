"""Condition numbers of the normal matrices X^T W^2 X of the linear-amplitude solve over a batch of prior samples (float64
eigenvalues of the float32 basis stack): how often tf.linalg.pinv's rcond = 1e-6 cut (tf/simulator.py:238) is active at all.

    python tools/dev/lstsq_condition_probe.py [--workload C3L] [--batch 64] [--direct]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gigalens_amd import workloads  # noqa: E402
from gigalens_amd.simulator import LensSimulator  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C3L")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--direct", action="store_true")
a = ap.parse_args()
wl = workloads.make(a.workload, batch=a.batch, interpolate=not a.direct)
sim = LensSimulator(wl.phys_model, wl.sim_config, bs=a.batch)
c2 = workloads.make("C2", num_pix=wl.sim_config.num_pix, batch=1)   # the observation tools/prof_kernel.py --mode lstsq solves against
obs, _, _ = workloads.synthetic_observation(c2, LensSimulator)
err = torch.sqrt(wl.background_rms ** 2 + obs.clamp_min(0) / wl.exp_time).contiguous()
x = wl.prior.sample(a.batch, seed=11)
st = sim.lstsq_simulate(x, obs, err, return_stacked=True).double()          # (B, H, W, D)
X = (st / err.double()[None, :, :, None]).reshape(a.batch, -1, st.shape[-1])
N = X.transpose(1, 2) @ X
ev = torch.linalg.eigvalsh(N).cpu().numpy()
cond = ev[:, -1] / np.maximum(ev[:, 0], 1e-300)
cut = (ev < 1e-6 * ev[:, -1:]).sum(axis=1)
print(f"{a.workload} direct={a.direct} B={a.batch} D={st.shape[-1]}: cond min {cond.min():.3g} median {np.median(cond):.3g} "
      f"max {cond.max():.3g}; samples with eigenvalues under the 1e-6 cut: {(cut > 0).sum()} (most cut in one sample: {cut.max()})")
print("cond percentiles 10/50/90/99:", np.percentile(cond, [10, 50, 90, 99]))
