"""The reference's tf-demo.ipynb pipeline (cells 12-19) on the reference's own demo image: MAP -> SVI -> HMC with the notebook's
hyper-parameters; prints what the GPU test asserts."""
import os, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gigalens_amd.inference import Adam, ModellingSequence
from gigalens_amd.model import ForwardProbModel
from gigalens_amd.simulator import LensSimulator
from gigalens_amd import prior as tfd
from tests.test_prior_host import default_prior
from tests.test_reference_demo import _setup, TRUTH

def poly(initial, steps, end, power=1.0):
    return lambda t: (initial - end) * (1 - min(t, steps) / steps) ** power + end

obs, psf, phys, cfg = _setup(supersample=2)
prior = default_prior()
pm = ForwardProbModel(prior, obs, background_rms=0.2, exp_time=100, include_positions=False)
seq = ModellingSequence(phys, pm, cfg)
t0 = time.time()
MAP = seq.MAP(Adam(poly(1e-2, 300, 1e-2 / 5)), n_samples=500, num_steps=300, seed=0)
torch.cuda.synchronize(); t1 = time.time()
sim = LensSimulator(phys, cfg, bs=500)
lps, red = pm.log_prob(sim, MAP)
i = int(torch.argmax(lps))
best = MAP[i]
print("MAP %.2fs: best log_prob %.1f red_chi2 %.4f; samples with red < 1.1: %d" % (t1 - t0, float(lps[i]), float(red[i]), int((red < 1.1).sum())))
xb = pm.bij.forward(best[None])
names, tru, got = [], [], []
for grp in ("lens_mass", "lens_light", "source_light"):
    for k, comp in enumerate(TRUTH[grp]):
        for n, v in comp.items():
            names.append(f"{grp}[{k}].{n}"); tru.append(v); got.append(float(xb[grp][k][n][0]))
for n, a, b in zip(names, tru, got): print("   %-28s truth %9.4f  MAP %9.4f" % (n, a, b))
t2 = time.time()
q_z, losses = seq.SVI(Adam(poly(0.0, 500, 4e-3, 2)), best, n_vi=500, num_steps=1000)
torch.cuda.synchronize(); t3 = time.time()
print("SVI %.2fs: loss first 50 %.1f last 50 %.1f" % (t3 - t2, np.mean(losses[:50]), np.mean(losses[-50:])))
samples, stats = seq.HMC(q_z, n_hmc=50, init_eps=0.3, init_l=3, max_leapfrog_steps=300, num_burnin_steps=250, num_results=750)
torch.cuda.synchronize(); t4 = time.time()
s = samples.double().cpu().numpy()  # [750, 50, d]
n, m = s.shape[0], s.shape[1]
W = s.var(axis=0, ddof=1).mean(axis=0); Bv = n * s.mean(axis=0).var(axis=0, ddof=1)
rhat = np.sqrt(((n - 1) / n * W + Bv / n) / W)
print("HMC %.2fs: Rhat max %.4f; stats keys %s" % (t4 - t3, rhat.max(), list(stats.keys()) if isinstance(stats, dict) else type(stats)))
xs = pm.bij.forward(samples.reshape(-1, samples.shape[-1]))
k = 0
for grp in ("lens_mass", "lens_light", "source_light"):
    for c, comp in enumerate(TRUTH[grp]):
        for nme, v in comp.items():
            col = xs[grp][c][nme].double().cpu().numpy()
            print("   %-28s truth %9.4f  post %9.4f +- %.4f  (%.1f sigma)" % (f"{grp}[{c}].{nme}", v, col.mean(), col.std(), (col.mean() - v) / col.std()))
