import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gigalens_amd.inference import Adam, ModellingSequence
from gigalens_amd.model import ForwardProbModel
from tests.test_prior_host import default_prior
from tests.test_reference_demo import _setup
obs, psf, phys, cfg = _setup(supersample=2)
prior = default_prior()
pm = ForwardProbModel(prior, obs, background_rms=0.2, exp_time=100, include_positions=False)
seq = ModellingSequence(phys, pm, cfg)
poly = lambda i, s, e, p=1.0: (lambda t: (i - e) * (1 - min(t, s) / s) ** p + e)
MAP = seq.MAP(Adam(poly(1e-2, 300, 2e-3)), n_samples=500, num_steps=300, seed=0)
from gigalens_amd.simulator import LensSimulator
lps, red = pm.log_prob(LensSimulator(phys, cfg, bs=500), MAP)
best = MAP[int(torch.argmax(lps))]
q_z, _ = seq.SVI(Adam(poly(0.0, 500, 4e-3, 2)), best, n_vi=500, num_steps=1000)
for n_hmc in (50, 256):
    torch.cuda.synchronize(); t0 = time.time()
    samples, stats = seq.HMC(q_z, n_hmc=n_hmc, init_eps=0.3, init_l=3, max_leapfrog_steps=300, num_burnin_steps=250, num_results=750)
    torch.cuda.synchronize(); dt = time.time() - t0
    nl = stats["num_leapfrog_steps"]
    nl = np.asarray(nl.cpu() if torch.is_tensor(nl) else nl, dtype=float)
    print(f"n_hmc={n_hmc}: {dt:.2f} s for 1000 transitions; leapfrog steps per transition mean {nl.mean():.1f} max {nl.max():.0f}; "
          f"{dt / nl.sum() * 1e6 if nl.size > 1 else float('nan'):.0f} us per leapfrog step (all-in); accept {np.mean(np.asarray(stats['accept'].cpu() if torch.is_tensor(stats['accept']) else stats['accept'], dtype=float)):.2f}")
