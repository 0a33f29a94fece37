"""The reference's tf-demo set-up (60x60 px, supersample 2, PSF through subgrid_kernel, 500 samples): repeated
log_prob_and_grad calls for a rocprofv3 kernel trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gigalens_amd import workloads
from gigalens_amd.model import ForwardProbModel, PhysicalModel
from gigalens_amd.simulator import LensSimulator, SimulatorConfig
from gigalens_amd.profiles.light.sersic import SersicEllipse
from gigalens_amd.profiles.mass.epl import EPL
from gigalens_amd.profiles.mass.shear import Shear
from tests.test_prior_host import default_prior
psf = np.load(os.path.join(ROOT, "tests", "golden", "reference_assets", "psf.npy")).astype(np.float32)
obs = np.load(os.path.join(ROOT, "tests", "golden", "reference_assets", "demo.npy")).astype(np.float32)
phys = PhysicalModel([EPL(50), Shear()], [SersicEllipse()], [SersicEllipse()])
ss = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = SimulatorConfig(delta_pix=0.065, num_pix=60, supersample=ss, kernel=psf)
B = 500
prior = default_prior()
pm = ForwardProbModel(prior, obs, 0.2, 100.0, include_positions=False)
sim = LensSimulator(phys, cfg, bs=B)
z = pm.bij.inverse(prior.sample(B, seed=0)).to("cuda").contiguous()
for _ in range(20):
    pm.log_prob_and_grad(sim, z)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    pm.log_prob_and_grad(sim, z)
e1.record(); torch.cuda.synchronize()
print("ms per step", e0.elapsed_time(e1) / 200)
