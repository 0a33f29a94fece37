import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import bench_configs as bc
from gigalens_amd import workloads
from gigalens_amd.model import PhysicalModel
from gigalens_amd.simulator import SimulatorConfig
from tests.test_prior_host import default_prior
from gigalens_amd.profiles.light.sersic import SersicEllipse
from gigalens_amd.profiles.mass.epl import EPL
from gigalens_amd.profiles.mass.shear import Shear
psf = np.load(os.path.join(ROOT, "tests", "golden", "reference_assets", "psf.npy")).astype(np.float32)
obs = np.load(os.path.join(ROOT, "tests", "golden", "reference_assets", "demo.npy")).astype(np.float32)
phys = PhysicalModel([EPL(50), Shear()], [SersicEllipse()], [SersicEllipse()])
k = np.kron(psf, np.ones((2, 2), np.float32) / 4)
wl = workloads.Workload("DEMO", phys, default_prior(), SimulatorConfig(delta_pix=0.065, num_pix=60, supersample=2), 500)
bc.run("demo ss2", wl, supersampled_kernel=k, obs=obs)
