"""Series-expansion accelerator of the (scaled) dPIE (reference: src/gigalens/tf/series/series_profile.py:9-95,
tf/profiles/mass/dpie_series.py, scaling_series.py, dpie_subhalo_series.py).

The deflection per unit amplitude is expanded in the cut radius around ``r_cut0`` on a fixed grid; afterwards a lens
evaluation is a per-pixel polynomial, independent of the number of galaxies.  The reference produces the expansion
terms with sympy-generated code (tf/series/profiles/dpie.py); here they are Taylor coefficients obtained by running the
ordinary dPIE kernels on truncated-series arithmetic (gigalens_amd/csrc/gl_jet.h, gl_series.h).
"""
from typing import Dict, List

import numpy as np
import torch

from gigalens_amd import _native
from gigalens_amd.profile import MassProfile
from gigalens_amd.profiles.mass.piemd import DPIE
from gigalens_amd.profiles.mass.scaling_relation import ScalingRelation


class MassSeries(MassProfile):
    """series_profile.py:9-95: ``set_grid`` / ``set_constants`` / ``set_deriv`` then ``deriv`` on that grid."""

    _series_param: str
    _amplitude_param: str
    _name = "SeriesExpansion"
    _constants: List[str] = []
    _kind = 10

    def __init__(self, grid=(None, None), params=None, order=3, **kwargs):
        self._init_series(grid, params, order)
        MassProfile.__init__(self, **kwargs)

    def _init_series(self, grid=(None, None), params=None, order=3):
        self.series_param = self._series_param
        self.amplitude_param = self._amplitude_param
        self._series_var_0 = None
        self.constants = list(self._constants)
        if not 0 <= int(order) <= 5:
            raise ValueError("order must be in 0..5 (the reference ships deriv_0..deriv_5)")
        self._order = int(order)
        self._constants_dict = {}
        if params is not None:
            self.set_constants(params)
        self._x, self._y = grid
        self._coefs = None  # device [2, order + 1, n_points] Taylor coefficients of the deflection
        self._hcoefs = None  # device [3, order + 1, n_points] Taylor coefficients of f_xx, f_xy, f_yy

    order = property(lambda self: self._order)
    series_var_0 = property(lambda self: self._series_var_0)
    x = property(lambda self: self._x)
    y = property(lambda self: self._y)
    constants_dict = property(lambda self: self._constants_dict)

    def set_constants(self, params):
        self._series_var_0 = float(np.asarray(params[self.series_param], dtype=np.float32).reshape(-1)[0])
        self._constants_dict = dict(params)
        self._coefs = self._hcoefs = None

    def set_grid(self, x, y):
        self._x, self._y = x, y
        self._coefs = self._hcoefs = None

    def set_deriv(self):
        """series_profile.py:61-62: precompute the expansion on the grid (one native launch, gl_series_precompute)."""
        if self._x is None or self._series_var_0 is None:
            raise ValueError("set_grid(x, y) and set_constants(params) must be called before set_deriv()")
        self._coefs = _native.series_precompute(self)

    def set_hessian(self):
        """series_profile.py:64-65: precompute the expansion of the Hessian on the grid (gl_series_precompute_hessian).
        The reference's ScalingRelationSeries hands a 4-tuple to this 3-way unpacking (scaling_series.py:54) and so
        raises for catalogues; here the catalogue's Hessian series is built like its deflection series."""
        if self._x is None or self._series_var_0 is None:
            raise ValueError("set_grid(x, y) and set_constants(params) must be called before set_hessian()")
        self._hcoefs = _native.series_precompute(self, hessian=True)

    def hessian(self, x, y, **kwargs):
        """series_profile.py:83-89: ``(f_xx, f_xy, f_xy, f_yy)`` on the set grid (``(x, y)`` are NOT used)."""
        if self._hcoefs is None:
            self.set_hessian()
        return _native.series_hessian_eval(self, kwargs[self.amplitude_param], kwargs[self.series_param])

    def deriv(self, x, y, **kwargs):
        """series_profile.py:76-81: like the reference, ``(x, y)`` are NOT used -- the field lives on the set grid."""
        if self._coefs is None:
            self.set_deriv()
        return _native.series_eval(self, kwargs[self.amplitude_param], kwargs[self.series_param])

    # -- native descriptors ------------------------------------------------------------------------------------------
    def _component(self):
        return (self._kind, self._order, 0)

    def _native_params(self):
        return [self.amplitude_param, self.series_param]

    def _series_inputs(self):
        """(base_kind, scale_col[3], table [G,7], scales) for gl_series_precompute."""
        raise NotImplementedError


class DPIESeries(MassSeries):
    """dpie_series.py:6-33: one dPIE halo expanded in its own r_cut."""

    _params = ["r_cut", "theta_E"]
    _constants = ["r_core", "center_x", "center_y", "e1", "e2"]
    _series_param = "r_cut"
    _amplitude_param = "theta_E"
    _name = "SeriesExpansion-dPIE"

    def __init__(self, order=3):
        super().__init__(order=order)

    def _series_inputs(self):
        c = {k: float(np.asarray(v, dtype=np.float32).reshape(-1)[0]) for k, v in self._constants_dict.items()}
        row = np.array([[1.0, c["r_core"], 1.0, c["center_x"], c["center_y"], c["e1"], c["e2"]]], dtype=np.float32)
        return DPIE._kind, [-1, -1, 0], row, [self._series_var_0]


class ScalingRelationSeries(MassSeries, ScalingRelation):
    """scaling_series.py:8-35: the catalogue's summed expansion, weights ``(L/L*)^p_amp ((L/L*)^p_series)^n``."""

    def __init__(self, profile: MassSeries, order=3, **kwargs):
        self._series_param = profile.series_param
        self._amplitude_param = profile.amplitude_param
        ScalingRelation.__init__(self, profile=DPIE(), **kwargs)
        self._init_series(order=order)
        self._name = self.name = f"Scaled-{profile.name}"
        self.profile_series = profile
        self.params = [self.amplitude_param, self.series_param]
        self.scaling_constants = [p for p in self.scaling_params if p in self.constants]

    def _series_inputs(self):
        base_kind, cols, table = self._catalogue()
        c = self._constants_dict
        scales = []
        for name in self.scaling_params:
            scales.append(1.0 if name == self.amplitude_param else
                          float(np.asarray(c[name], dtype=np.float32).reshape(-1)[0]))
        return base_kind, cols, table, scales

    def deriv(self, x, y, **kwargs):
        return MassSeries.deriv(self, x, y, **kwargs)

    def hessian(self, x, y, **kwargs):
        return MassSeries.hessian(self, x, y, **kwargs)

    def convergence(self, x, y, **kwargs):  # scaling_series.py:56-60: the generic ones, from the series Hessian
        return MassProfile.convergence(self, x, y, **kwargs)

    def shear(self, x, y, **kwargs):
        return MassProfile.shear(self, x, y, **kwargs)

    def _component(self):
        return MassSeries._component(self)


class DPIESubhaloSeries(ScalingRelationSeries):
    """dpie_subhalo_series.py:6-28."""

    _constants = ["r_core", "center_x", "center_y", "e1", "e2"]
    _name = "Scaled-SeriesExpansion-dPIE"

    def __init__(self, lum_star: float, galaxy_catalogue: Dict[str, List], scaling_params_power=None, order=3,
                 chunk_size=None):
        if scaling_params_power is None:
            scaling_params_power = {"theta_E": 0.5, "r_core": 0.5, "r_cut": 0.5}
        super().__init__(profile=DPIESeries(order=order), order=order, lum_star=lum_star,
                         scaling_params=["theta_E", "r_core", "r_cut"], scaling_params_power=scaling_params_power,
                         galaxy_catalogue=galaxy_catalogue, chunk_size=chunk_size)
        self._name = self.name = "Scaled-SeriesExpansion-dPIE"
