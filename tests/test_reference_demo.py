"""End-to-end pin against data the REFERENCE ITSELF generated (tests/golden/reference_assets): the demo image
of tf-demo.ipynb.  Simulating the notebook's truth parameters (EPL+Shear, SersicEllipse lens light and source,
13x13 PSF) must explain that image statistically: reduced chi^2 = 1 up to the unknown noise realisation.
Any error in the grid convention, a deflection, a light profile, the PSF handling or the det(T) scale moves
chi^2 far from 1 (dropping the PSF alone gives 1.59)."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.join(os.path.dirname(__file__), "golden", "reference_assets")
TRUTH = {  # tf-demo.ipynb cell 5
    "lens_mass": [{"theta_E": 1.1, "gamma": 2.0, "e1": 0.1, "e2": 0.1, "center_x": 0.1, "center_y": 0.0},
                  {"gamma1": -0.01, "gamma2": 0.03}],
    "lens_light": [{"R_sersic": 0.8, "n_sersic": 2.5, "e1": 0.09534746574143645, "e2": 0.14849487967198177,
                    "center_x": 0.1, "center_y": 0.0, "Ie": 499.3695906504067}],
    "source_light": [{"R_sersic": 0.25, "n_sersic": 1.5, "e1": 0.0, "e2": 0.0, "center_x": 0.09566681002252231,
                      "center_y": -0.0639623054267272, "Ie": 149.58828877085668}],
}


def _setup(supersample=1):
    from gigalens_amd.model import PhysicalModel
    from gigalens_amd.profiles.light.sersic import SersicEllipse
    from gigalens_amd.profiles.mass.epl import EPL
    from gigalens_amd.profiles.mass.shear import Shear
    from gigalens_amd.simulator import SimulatorConfig
    obs = np.load(os.path.join(HERE, "demo.npy"))
    psf = np.load(os.path.join(HERE, "psf.npy")).astype(np.float32)
    phys = PhysicalModel([EPL(50), Shear()], [SersicEllipse()], [SersicEllipse()])
    cfg = SimulatorConfig(delta_pix=0.065, num_pix=60, supersample=supersample, kernel=psf)
    return obs, psf, phys, cfg


def test_oracle_explains_reference_demo_image():
    from oracle import ref_torch as ref
    obs, psf, phys, cfg = _setup()
    rs = ref.RefSimulator(phys, cfg, 1, dtype=torch.float64)
    tt = {k: [{n: torch.tensor([v], dtype=torch.float64) for n, v in d.items()} for d in lst] for k, lst in TRUTH.items()}
    ll, red = ref.stats_pixels(rs, tt, obs, 0.2, 100.0)
    assert 0.95 < float(red) < 1.05, float(red)          # measured: 0.9989
    assert abs(float(rs.simulate(tt).sum()) / obs.sum() - 1) < 2e-3
    # sensitivity: without the PSF the same parameters do NOT explain the image
    cfg.kernel = None
    rs0 = ref.RefSimulator(phys, cfg, 1, dtype=torch.float64)
    assert float(ref.stats_pixels(rs0, tt, obs, 0.2, 100.0)[1]) > 1.4


def test_oracle_explains_reference_demo_image_at_supersample_2():
    """The notebook's own configuration (tf-demo.ipynb cell 6: ``SimulatorConfig(delta_pix=0.065, num_pix=60, supersample=2,
    kernel=psf)``), which needs lenstronomy's ``subgrid_kernel`` (restated, parity unpinned): the truth parameters still
    explain the reference's image (measured 0.981; the notebook's cell-9 print of this number was stripped from the
    committed outputs).  A naive supersampled PSF (each PSF pixel split into 2 x 2) does not: 1.33."""
    from oracle import ref_torch as ref
    obs, psf, phys, cfg = _setup(supersample=2)
    tt = {k: [{n: torch.tensor([v], dtype=torch.float64) for n, v in d.items()} for d in lst] for k, lst in TRUTH.items()}
    rs = ref.RefSimulator(phys, cfg, 1, dtype=torch.float64)
    assert rs.flat_kernel.shape == (25, 25)
    red = float(ref.stats_pixels(rs, tt, obs, 0.2, 100.0)[1])
    assert 0.95 < red < 1.05, red
    naive = ref.RefSimulator(phys, cfg, 1, dtype=torch.float64, supersampled_kernel=np.kron(psf, np.ones((2, 2)) / 4))
    assert float(ref.stats_pixels(naive, tt, obs, 0.2, 100.0)[1]) > 1.2


@pytest.mark.gpu
@pytest.mark.parametrize("supersample", [1, 2, 3])
def test_hip_explains_reference_demo_image(supersample):
    from gigalens_amd import prior as tfd
    from gigalens_amd.model import ForwardProbModel
    from gigalens_amd.simulator import LensSimulator
    from oracle import ref_torch as ref
    obs, psf, phys, cfg = _setup(supersample)
    sim = LensSimulator(phys, cfg, bs=1)
    prior = tfd.JointDistributionNamed(dict(lens_mass=tfd.JointDistributionSequential(
        [tfd.JointDistributionNamed(dict(theta_E=tfd.Normal(0, 1)))])))
    pm = ForwardProbModel(prior, obs, 0.2, 100.0, include_positions=False)
    ll, red = pm.stats_pixels(sim, TRUTH)
    assert 0.94 < float(red) < 1.06, float(red)  # measured: 0.999 / 0.981 / 1.036 at supersample 1 / 2 / 3
    rs = ref.RefSimulator(phys, cfg, 1, dtype=torch.float64)
    tt = {k: [{n: torch.tensor([np.float32(v)], dtype=torch.float64) for n, v in d.items()} for d in lst] for k, lst in TRUTH.items()}
    ll_o, red_o = ref.stats_pixels(rs, tt, obs, 0.2, 100.0)
    assert np.isclose(float(ll), float(ll_o), rtol=1e-5) and np.isclose(float(red), float(red_o), rtol=1e-5)
    img = sim.simulate(TRUTH)
    assert img.shape == (60, 60)  # bs == 1 squeezes like tf.squeeze (tf/simulator.py:156)


@pytest.mark.gpu
def test_notebook_pipeline_recovers_the_demo_truth():
    """tf-demo.ipynb cells 12-19 end to end on the reference's own demo image, with the notebook's hyper-parameters:
    MAP (500 samples, 300 steps, lr 1e-2 -> 2e-3) -> best sample by log_prob -> SVI (500 particles, 1000 steps, lr 0 -> 4e-3)
    -> HMC (50 chains, eps 0.3, 3 leapfrog steps to start, <= 300, 250 burn-in, 750 results).  Asserted: the best MAP sample
    explains the image (reduced chi^2 ~ 1), the ELBO improves, the chains mix (R-hat < 1.05; the notebook prints 1.000-1.005)
    and every one of the 22 posterior means lies within 4 posterior standard deviations of the parameters the image was
    simulated with (measured: within 1.6; profiles/r2_demo_pipeline.log -- MAP 0.55 s, SVI 0.29 s, HMC 1.06 s on one MI355X)."""
    from gigalens_amd.inference import Adam, ModellingSequence
    from gigalens_amd.model import ForwardProbModel
    from gigalens_amd.simulator import LensSimulator
    from tests.test_prior_host import default_prior

    def poly(initial, steps, end, power=1.0):  # tf.keras.optimizers.schedules.PolynomialDecay
        return lambda t: (initial - end) * (1 - min(t, steps) / steps) ** power + end

    obs, psf, phys, cfg = _setup(supersample=2)
    prior = default_prior()
    pm = ForwardProbModel(prior, obs, background_rms=0.2, exp_time=100, include_positions=False)
    seq = ModellingSequence(phys, pm, cfg)
    MAP = seq.MAP(Adam(poly(1e-2, 300, 1e-2 / 5)), n_samples=500, num_steps=300, seed=0)
    lps, red = pm.log_prob(LensSimulator(phys, cfg, bs=500), MAP)
    best = MAP[int(torch.argmax(lps))]
    assert 0.9 < float(red[int(torch.argmax(lps))]) < 1.05
    q_z, losses = seq.SVI(Adam(poly(0.0, 500, 4e-3, 2)), best, n_vi=500, num_steps=1000)
    assert np.all(np.isfinite(losses)) and np.mean(losses[-50:]) < np.mean(losses[:50])
    samples, stats = seq.HMC(q_z, n_hmc=50, init_eps=0.3, init_l=3, max_leapfrog_steps=300, num_burnin_steps=250,
                             num_results=750)
    s = samples.double().cpu().numpy()
    n = s.shape[0]
    W, Bv = s.var(axis=0, ddof=1).mean(axis=0), n * s.mean(axis=0).var(axis=0, ddof=1)
    rhat = np.sqrt(((n - 1) / n * W + Bv / n) / W)
    assert rhat.max() < 1.05, rhat
    xs = pm.bij.forward(samples.reshape(-1, samples.shape[-1]))
    for grp in ("lens_mass", "lens_light", "source_light"):
        for c, comp in enumerate(TRUTH[grp]):
            for name, v in comp.items():
                col = xs[grp][c][name].double().cpu().numpy()
                assert abs(col.mean() - v) < 4.0 * col.std(), (grp, c, name, v, col.mean(), col.std())
