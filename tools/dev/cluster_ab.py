#!/usr/bin/env python3
"""A/B of the cluster kernels on one workload: GIGALENS_HIP_CLUSTER=1 (pixel split) vs 2 (components over waves): kernel time,
step time, and the difference of log-prob / gradient.   python3 tools/dev/cluster_ab.py [C4|C5] [variants]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs as bc
    import numpy as np
    import torch
    from gigalens_amd import workloads
    from gigalens_amd.model import ForwardProbModel
    from gigalens_amd.simulator import LensSimulator
    wl = workloads.make(sys.argv[2])
    obs, err, _ = workloads.synthetic_observation(wl, LensSimulator)
    pm = ForwardProbModel(wl.prior, np.asarray(obs.cpu()), wl.background_rms, wl.exp_time, include_positions=False)
    sim = LensSimulator(wl.phys_model, wl.sim_config, bs=wl.batch)
    z = pm.bij.inverse(wl.prior.sample(wl.batch, seed=0)).to("cuda").contiguous()
    ts = [bc.time_step(pm, sim, z, seconds=0.4) for _ in range(3)]
    m = sim._model
    m.set_timing(1)
    ks = []
    for _ in range(12):
        pm.log_prob_and_grad(sim, z)
        ks.append(m.last_main_ms())
    lp, _, g = pm.log_prob_and_grad(sim, z)
    torch.save(dict(lp=lp.cpu(), g=g.cpu()), sys.argv[3])
    print(json.dumps(dict(kernel=m.last_main_kernel()[:60], step_ms=[round(t, 4) for t in ts], kernel_ms=round(sorted(ks[2:])[len(ks[2:]) // 2], 4))), flush=True)
    sys.exit(0)

import torch  # noqa: E402
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "2"]
outs = {}
for v in variants:
    flag, _, lib = v.partition("@")  # "2@gigalens_amd/lib/exp/x.so": the flag with an experiment build of the library
    env = dict(os.environ, GIGALENS_HIP_CLUSTER=flag)
    if lib:
        env["GIGALENS_HIP_LIB"] = os.path.join(ROOT, lib)
    f = f"/tmp/cluster_ab_{abs(hash(v))}.pt"
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, f], env=env, capture_output=True, text=True)
    print(v, r.stdout.strip(), r.stderr.strip()[-400:], flush=True)
    if r.returncode == 0:
        outs[v] = torch.load(f)
ks = list(outs)
for v in ks[1:]:
    a, b = outs[ks[0]], outs[v]
    S = a["g"].abs().amax(dim=0, keepdim=True).clamp_min(1e-30)
    print(f"{ks[0]} vs {v}: lp rel {float(((a['lp'] - b['lp']).abs() / a['lp'].abs()).max()):.3e}  grad col-rel {float(((a['g'] - b['g']).abs() / S).max()):.3e}")
