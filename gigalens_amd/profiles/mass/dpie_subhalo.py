from typing import Dict, List

from gigalens_amd.profiles.mass.piemd import DPIE
from gigalens_amd.profiles.mass.scaling_relation import ScalingRelation


class DPIESubhalo(ScalingRelation):
    """Cluster-member population of dPIE halos (reference: src/gigalens/tf/profiles/mass/dpie_subhalo.py:6-21)."""

    def __init__(self, lum_star: float, galaxy_catalogue: Dict[str, List], scaling_params_power=None, **kwargs):
        if scaling_params_power is None:
            scaling_params_power = {"theta_E": 0.5, "r_core": 0.5, "r_cut": 0.5}
        super().__init__(profile=DPIE(), scaling_params=["theta_E", "r_core", "r_cut"], lum_star=lum_star,
                         scaling_params_power=scaling_params_power, galaxy_catalogue=galaxy_catalogue, **kwargs)
