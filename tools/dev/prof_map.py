"""ModellingSequence.MAP on C2 (1024 samples, 128x128) for a rocprofv3 kernel trace: what a MAP step launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gigalens_amd import workloads
from gigalens_amd.inference import Adam, ModellingSequence
from gigalens_amd.model import ForwardProbModel
from gigalens_amd.simulator import LensSimulator
wl = workloads.make("C2")
obs, _, _ = workloads.synthetic_observation(wl, LensSimulator)
pm = ForwardProbModel(wl.prior, obs.cpu().numpy(), wl.background_rms, wl.exp_time, include_positions=False)
seq = ModellingSequence(wl.phys_model, pm, wl.sim_config)
seq.MAP(Adam(1e-2), None, n_samples=1024, num_steps=int(sys.argv[1]) if len(sys.argv) > 1 else 300, seed=1, graph=False)
torch.cuda.synchronize()
