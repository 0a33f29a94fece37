"""Synthetic workloads C1-C4 of BASELINE.json / SURVEY.md 8(d): physical model, prior, camera and batch.

All inputs are synthetic (there is no network for data): priors borrowed from the reference's
``tests/conftest.py:26-66`` and ``shapelets-demo.ipynb`` cell 4; the "observed" image of a workload is
the forward simulation of one fixed truth draw plus Gaussian noise with the model's own sigma.
"""
import math
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from gigalens_amd import prior as tfd
from gigalens_amd.model import PhysicalModel
from gigalens_amd.profiles.light.sersic import Sersic, SersicEllipse
from gigalens_amd.profiles.light.shapelets import Shapelets
from gigalens_amd.profiles.mass.epl import EPL
from gigalens_amd.profiles.mass.dpie_series import DPIESubhaloSeries
from gigalens_amd.profiles.mass.dpie_subhalo import DPIESubhalo
from gigalens_amd.profiles.mass.nfw import NFW
from gigalens_amd.profiles.mass.piemd import DPIE
from gigalens_amd.profiles.mass.shear import Shear
from gigalens_amd.profiles.mass.sie import SIE
from gigalens_amd.simulator import SimulatorConfig


@dataclass
class Workload:
    name: str
    phys_model: PhysicalModel
    prior: tfd.JointDistributionNamed
    sim_config: SimulatorConfig
    batch: int
    background_rms: float = 0.2
    exp_time: float = 100.0
    use_error_map: bool = False
    description: str = ""


def _epl_prior():
    return tfd.JointDistributionNamed(dict(
        theta_E=tfd.LogNormal(math.log(1.25), 0.25), gamma=tfd.TruncatedNormal(2, 0.25, 1, 3),
        e1=tfd.Normal(0, 0.1), e2=tfd.Normal(0, 0.1), center_x=tfd.Normal(0, 0.05), center_y=tfd.Normal(0, 0.05)))


def _shear_prior():
    return tfd.JointDistributionNamed(dict(gamma1=tfd.Normal(0, 0.05), gamma2=tfd.Normal(0, 0.05)))


def _sersic_src_prior(center_sigma=0.25, uniform_center=None):
    if uniform_center is None:
        cx, cy = tfd.Normal(0, center_sigma), tfd.Normal(0, center_sigma)
    else:
        cx, cy = tfd.Uniform(-uniform_center, uniform_center), tfd.Uniform(-uniform_center, uniform_center)
    return tfd.JointDistributionNamed(dict(
        R_sersic=tfd.LogNormal(math.log(0.25), 0.15), n_sersic=tfd.Uniform(0.5, 4), center_x=cx, center_y=cy,
        Ie=tfd.LogNormal(math.log(150.0), 0.5)))


def make(name: str, num_pix: Optional[int] = None, batch: Optional[int] = None, interpolate: bool = True,
         n_max: int = 10, n_halos: int = 8, n_sources: int = 20, n_galaxies: int = 200) -> Workload:
    name = name.upper()
    if name == "C1":  # SIE + Sersic source, 64x64, B=1
        phys = PhysicalModel([SIE()], [], [Sersic()])
        prior = tfd.JointDistributionNamed(dict(
            lens_mass=tfd.JointDistributionSequential([tfd.JointDistributionNamed(dict(
                theta_E=tfd.LogNormal(math.log(1.25), 0.25), e1=tfd.Normal(0, 0.1), e2=tfd.Normal(0, 0.1),
                center_x=tfd.Normal(0, 0.05), center_y=tfd.Normal(0, 0.05)))]),
            source_light=tfd.JointDistributionSequential([_sersic_src_prior()])))
        return Workload("C1", phys, prior, SimulatorConfig(delta_pix=0.065, num_pix=num_pix or 64), batch or 1,
                        description="SIE lens + Sersic source")
    if name == "C2":  # EPL + Shear + Sersic source, 128x128, B=1024
        phys = PhysicalModel([EPL(), Shear()], [], [Sersic()])
        prior = tfd.JointDistributionNamed(dict(
            lens_mass=tfd.JointDistributionSequential([_epl_prior(), _shear_prior()]),
            source_light=tfd.JointDistributionSequential([_sersic_src_prior()])))
        return Workload("C2", phys, prior, SimulatorConfig(delta_pix=0.065, num_pix=num_pix or 128), batch or 1024,
                        description="EPL+shear lens, Sersic source")
    if name == "C3":  # EPL + Shear + Shapelets(n_max) source, error_map
        shp = Shapelets(n_max=n_max, interpolate=interpolate)
        phys = PhysicalModel([EPL(), Shear()], [], [shp])
        amps = {nm: tfd.Normal(0, 500.0 / math.sqrt(i + 1)) for i, nm in enumerate(shp._amp_names)}
        src = tfd.JointDistributionNamed(dict(beta=tfd.LogNormal(math.log(0.1), 0.15), center_x=tfd.Normal(0, 0.01),
                                              center_y=tfd.Normal(0, 0.01), **amps))
        prior = tfd.JointDistributionNamed(dict(
            lens_mass=tfd.JointDistributionSequential([_epl_prior(), _shear_prior()]),
            source_light=tfd.JointDistributionSequential([src])))
        return Workload("C3", phys, prior, SimulatorConfig(delta_pix=0.065, num_pix=num_pix or 128), batch or 1024,
                        use_error_map=True, description=f"EPL+shear lens, Shapelets n_max={n_max} source "
                        f"({'table' if interpolate else 'direct'} mode)")
    if name == "C5":  # BASELINE configs[4]: the C4 model, 2048 SVI particles over 8 GPUs = 256 per rank, full-rank q, d = 132
        wl = make("C4", num_pix=num_pix, batch=batch or 256, n_halos=n_halos, n_sources=n_sources)
        wl.name = "C5"
        wl.description += " (per-rank shard of the 2048-particle full-rank SVI)"
        return wl
    if name == "C4":  # cluster: 8 NFW + 20 Sersic sources, 256x256, B=512
        phys = PhysicalModel([NFW() for _ in range(n_halos)], [], [Sersic() for _ in range(n_sources)])
        halo = lambda: tfd.JointDistributionNamed(dict(
            Rs=tfd.LogNormal(math.log(5.0), 0.3), alpha_Rs=tfd.LogNormal(math.log(1.0), 0.3),
            center_x=tfd.Uniform(-6, 6), center_y=tfd.Uniform(-6, 6)))
        prior = tfd.JointDistributionNamed(dict(
            lens_mass=tfd.JointDistributionSequential([halo() for _ in range(n_halos)]),
            source_light=tfd.JointDistributionSequential([_sersic_src_prior(uniform_center=4.0) for _ in range(n_sources)])))
        return Workload("C4", phys, prior, SimulatorConfig(delta_pix=0.065, num_pix=num_pix or 256), batch or 512,
                        description=f"cluster: {n_halos} NFW halos + {n_sources} Sersic sources")
    if name == "C3D":  # the reference's shapelets-demo.ipynb model: EPL+shear, SersicEllipse lens light, Shapelets source
        phys = PhysicalModel([EPL(), Shear()], [SersicEllipse()], [Shapelets(n_max, interpolate=interpolate)])
        L = (n_max + 1) * (n_max + 2) // 2
        names = phys.source_light[0]._amp_names
        src = dict(beta=tfd.LogNormal(math.log(0.1), 0.15), center_x=tfd.Normal(0, 0.01), center_y=tfd.Normal(0, 0.01))
        src.update({nm: tfd.Normal(0, 500.0 / math.sqrt(i + 1)) for i, nm in enumerate(names)})
        ll = tfd.JointDistributionNamed(dict(
            R_sersic=tfd.LogNormal(math.log(1.0), 0.15), n_sersic=tfd.Uniform(2, 6), e1=tfd.TruncatedNormal(0, 0.1, -0.3, 0.3),
            e2=tfd.TruncatedNormal(0, 0.1, -0.3, 0.3), center_x=tfd.Normal(0, 0.05), center_y=tfd.Normal(0, 0.05),
            Ie=tfd.LogNormal(math.log(500.0), 0.3)))
        prior = tfd.JointDistributionNamed(dict(
            lens_mass=tfd.JointDistributionSequential([_epl_prior(), _shear_prior()]),
            lens_light=tfd.JointDistributionSequential([ll]),
            source_light=tfd.JointDistributionSequential([tfd.JointDistributionNamed(src)])))
        assert L == len(names)
        return Workload("C3D", phys, prior, SimulatorConfig(delta_pix=0.065, num_pix=num_pix or 128), batch or 1024,
                        use_error_map=True, description=f"shapelets-demo model: EPL+shear, SersicEllipse lens light, "
                        f"Shapelets n_max={n_max} source")
    if name == "C3L":  # C3 with the shapelet amplitudes solved by least squares (shapelets-demo cell 7)
        phys = PhysicalModel([EPL(), Shear()], [], [Shapelets(n_max, use_lstsq=True, interpolate=interpolate)])
        src = tfd.JointDistributionNamed(dict(beta=tfd.LogNormal(math.log(0.1), 0.15), center_x=tfd.Normal(0, 0.01),
                                              center_y=tfd.Normal(0, 0.01)))
        prior = tfd.JointDistributionNamed(dict(
            lens_mass=tfd.JointDistributionSequential([_epl_prior(), _shear_prior()]),
            source_light=tfd.JointDistributionSequential([src])))
        return Workload("C3L", phys, prior, SimulatorConfig(delta_pix=0.065, num_pix=num_pix or 128), batch or 1024,
                        description=f"EPL+shear lens, Shapelets n_max={n_max} source with least-squares amplitudes")
    if name == "C6S":  # C6 with the member population behind the series-expansion accelerator (order 3 in r_cut)
        cat = galaxy_catalogue(n_galaxies, half_width=0.5 * 0.065 * (num_pix or 256))
        members = DPIESubhaloSeries(lum_star=1.0, galaxy_catalogue=cat, order=3)
        members.set_constants(dict(theta_E=0.3, r_core=0.02, r_cut=2.0))
        phys = PhysicalModel([DPIE(), members], [], [Sersic() for _ in range(n_sources)])
        halo = tfd.JointDistributionNamed(dict(
            theta_E=tfd.LogNormal(math.log(12.0), 0.1), r_core=tfd.LogNormal(math.log(3.0), 0.2),
            r_cut=tfd.LogNormal(math.log(150.0), 0.1), center_x=tfd.Normal(0, 0.3), center_y=tfd.Normal(0, 0.3),
            e1=tfd.Normal(0.15, 0.05), e2=tfd.Normal(-0.1, 0.05)))
        mem_prior = tfd.JointDistributionNamed(dict(theta_E=tfd.LogNormal(math.log(0.3), 0.2),
                                                    r_cut=tfd.LogNormal(math.log(2.0), 0.1)))
        prior = tfd.JointDistributionNamed(dict(
            lens_mass=tfd.JointDistributionSequential([halo, mem_prior]),
            source_light=tfd.JointDistributionSequential([_sersic_src_prior(uniform_center=3.0) for _ in range(n_sources)])))
        return Workload("C6S", phys, prior, SimulatorConfig(delta_pix=0.065, num_pix=num_pix or 256), batch or 128,
                        description=f"cluster: dPIE halo + series expansion of {n_galaxies} scaled dPIE galaxies + "
                        f"{n_sources} Sersic sources")
    if name == "C6":  # cluster with member galaxies: dPIE halo + DPIESubhalo catalogue + Sersic sources
        cat = galaxy_catalogue(n_galaxies, half_width=0.5 * 0.065 * (num_pix or 256))
        phys = PhysicalModel([DPIE(), DPIESubhalo(lum_star=1.0, galaxy_catalogue=cat)], [],
                             [Sersic() for _ in range(n_sources)])
        halo = tfd.JointDistributionNamed(dict(
            theta_E=tfd.LogNormal(math.log(12.0), 0.1), r_core=tfd.LogNormal(math.log(3.0), 0.2),
            r_cut=tfd.LogNormal(math.log(150.0), 0.1), center_x=tfd.Normal(0, 0.3), center_y=tfd.Normal(0, 0.3),
            e1=tfd.Normal(0.15, 0.05), e2=tfd.Normal(-0.1, 0.05)))
        members = tfd.JointDistributionNamed(dict(
            theta_E=tfd.LogNormal(math.log(0.3), 0.2), r_core=tfd.LogNormal(math.log(0.02), 0.2),
            r_cut=tfd.LogNormal(math.log(2.0), 0.3)))
        prior = tfd.JointDistributionNamed(dict(
            lens_mass=tfd.JointDistributionSequential([halo, members]),
            source_light=tfd.JointDistributionSequential([_sersic_src_prior(uniform_center=3.0) for _ in range(n_sources)])))
        return Workload("C6", phys, prior, SimulatorConfig(delta_pix=0.065, num_pix=num_pix or 256), batch or 128,
                        description=f"cluster: dPIE halo + {n_galaxies} scaled dPIE member galaxies + {n_sources} "
                        "Sersic sources")
    raise ValueError(f"unknown workload {name}")


def galaxy_catalogue(n_galaxies: int, half_width: float, seed: int = 11):
    """Synthetic cluster-member catalogue (luminosities, positions, ellipticities) for ScalingRelation workloads."""
    r = np.random.default_rng(seed)
    return dict(lum=r.lognormal(-0.3, 0.6, n_galaxies).astype(np.float32),
                center_x=r.uniform(-half_width, half_width, n_galaxies).astype(np.float32),
                center_y=r.uniform(-half_width, half_width, n_galaxies).astype(np.float32),
                e1=np.clip(r.normal(0, 0.15, n_galaxies), -0.5, 0.5).astype(np.float32),
                e2=np.clip(r.normal(0, 0.15, n_galaxies), -0.5, 0.5).astype(np.float32))


def synthetic_observation(wl: Workload, simulator_cls, seed_truth=1, seed_noise=2):
    """Observed image = simulate(truth) + N(0, sigma_model); returns (obs, err_map_or_None, truth)."""
    truth = wl.prior.sample(1, seed=seed_truth)
    sim1 = simulator_cls(wl.phys_model, wl.sim_config, bs=1)
    img = sim1.simulate(truth).reshape(wl.sim_config.num_pix, wl.sim_config.num_pix)
    g = torch.Generator(device="cpu")
    g.manual_seed(seed_noise)
    noise = torch.randn(img.shape, generator=g, dtype=torch.float32).to(img.device)
    if wl.use_error_map:
        sigma = torch.full_like(img, float(0.05 * img.abs().max().clamp_min(1e-3) + wl.background_rms))
        return (img + sigma * noise).contiguous(), sigma.contiguous(), truth
    sigma = torch.sqrt(wl.background_rms ** 2 + img.clamp_min(0) / wl.exp_time)
    return (img + sigma * noise).contiguous(), None, truth
