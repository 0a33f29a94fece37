#!/usr/bin/env python3
"""Step time of one workload through the product API under whatever GIGALENS_HIP_* knobs the environment carries (bench.py
refuses those): experiments only.   python3 tools/dev/step_time.py C2 [batch] [seconds]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as bc  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
from gigalens_amd import workloads  # noqa: E402
from gigalens_amd.model import ForwardProbModel  # noqa: E402
from gigalens_amd.simulator import LensSimulator  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else None
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
wl = workloads.make(name, batch=batch) if batch else workloads.make(name)
obs, err, _ = workloads.synthetic_observation(wl, LensSimulator)
pm = ForwardProbModel(wl.prior, np.asarray(obs.cpu()), wl.background_rms, wl.exp_time, include_positions=False)
sim = LensSimulator(wl.phys_model, wl.sim_config, bs=wl.batch)
z = pm.bij.inverse(wl.prior.sample(wl.batch, seed=0)).to("cuda").contiguous()
ts = [bc.time_step(pm, sim, z, seconds=seconds) for _ in range(3)]
lp, _, g = pm.log_prob_and_grad(sim, z)
knobs = {k: v for k, v in os.environ.items() if k.startswith("GIGALENS_HIP_") and k != "GIGALENS_HIP_LIB"}
print(json.dumps(dict(workload=name, batch=wl.batch, knobs=knobs, ms=[round(t, 5) for t in ts],
                      lp_sum=float(lp.double().sum()), g_abs=float(g.double().abs().sum()))), flush=True)
