#!/usr/bin/env python3
"""Wall time per ModellingSequence.SVI step and per HMC transition (drivers included), for comparison with the native
forward+gradient call they wrap."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gigalens_amd import workloads
from gigalens_amd.inference import Adam, ModellingSequence
from gigalens_amd.model import ForwardProbModel
from gigalens_amd.simulator import LensSimulator
for name, kw, n in (("C2", dict(num_pix=60, batch=250), 250), ("C2", dict(), 1024)):
    wl = workloads.make(name, **kw)
    obs, _, _ = workloads.synthetic_observation(wl, LensSimulator)
    pm = ForwardProbModel(wl.prior, obs.cpu().numpy(), wl.background_rms, wl.exp_time, include_positions=False)
    seq = ModellingSequence(wl.phys_model, pm, wl.sim_config)
    start = pm.bij.inverse(pm.prior.sample(2, seed=0))[0]
    for full in (True, False):
        # warm-up long enough to take in the one-off host costs of a run (first refill of the pre-generated noise pool, allocator
        # growth for this batch size: tens of ms that 200 timed steps would otherwise carry)
        seq.SVI(Adam(1e-3), start, n_vi=n, num_steps=300, full_rank=full)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        (mean, L), losses = seq.SVI(Adam(1e-3), start, n_vi=n, num_steps=200, full_rank=full)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"SVI {name} {kw} n_vi={n} full_rank={full}: {dt/200*1e3:.3f} ms per step", flush=True)
    (mean, L), _ = seq.SVI(Adam(1e-3), start, n_vi=n, num_steps=20)
    nh = min(n, 256)
    # warm-up: the first HMC call of a process pays one-off host start-up (torch.linalg initialisation, workspace sizing) -- in
    # round 2 that start-up was divided by 60 transitions and reported as a step time
    seq.HMC((mean, L), n_hmc=nh, init_eps=0.1, init_l=5, max_leapfrog_steps=5, num_burnin_steps=4, num_results=4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    samples, stats = seq.HMC((mean, L), n_hmc=nh, init_eps=0.1, init_l=5, max_leapfrog_steps=5, num_burnin_steps=20, num_results=40)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"HMC {name} {kw} n_hmc={nh}: {dt/60*1e3:.3f} ms per transition of 5 leapfrog steps", flush=True)
