#!/usr/bin/env python3
"""Steady-state time of one ModellingSequence.MAP step (native launch sequence + Adam), stream launches vs HIP-graph
replay, HIP-event and host-clock timed between steps 20 and 519."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gigalens_amd import workloads
from gigalens_amd.inference import Adam, ModellingSequence
from gigalens_amd.model import ForwardProbModel
from gigalens_amd.simulator import LensSimulator
for name, kw, n in (("C1", dict(batch=1), 1), ("C1", dict(batch=64), 64), ("C2", dict(num_pix=60, batch=500), 500), ("C2", dict(), 1024)):
    wl = workloads.make(name, **kw)
    obs, _, _ = workloads.synthetic_observation(wl, LensSimulator)
    pm = ForwardProbModel(wl.prior, obs.cpu().numpy(), wl.background_rms, wl.exp_time, include_positions=False)
    seq = ModellingSequence(wl.phys_model, pm, wl.sim_config)
    for g in (False, True):
        ev = {}
        wall = {}
        def prog(step, red):
            if step in (20, 519):
                ev[step] = torch.cuda.Event(enable_timing=True)
                ev[step].record()
                wall[step] = time.perf_counter()
        seq.MAP(Adam(1e-2), None, n_samples=n, num_steps=520, seed=1, graph=g, progress=prog)
        torch.cuda.synchronize()
        print(f"{name} {kw} graph={g}: {ev[20].elapsed_time(ev[519])/499:.4f} ms per MAP step (GPU), "
              f"{(wall[519]-wall[20])/499*1e3:.4f} ms (host issue)", flush=True)
