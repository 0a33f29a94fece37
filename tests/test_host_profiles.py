"""Host logic of the plugin classes added for the cluster workload and the linear solve (CPU, no GPU): names and
parameter lists as in the reference, catalogue tables, packing of least-squares amplitudes, series bookkeeping."""
import numpy as np
import pytest
import torch

from gigalens_amd.model import PhysicalModel
from gigalens_amd.profiles.light.sersic import Sersic, SersicEllipse
from gigalens_amd.profiles.light.shapelets import Shapelets
from gigalens_amd.profiles.mass.dpie_series import DPIESeries, DPIESubhaloSeries
from gigalens_amd.profiles.mass.dpie_subhalo import DPIESubhalo
from gigalens_amd.profiles.mass.epl import EPL
from gigalens_amd.profiles.mass.piemd import DPIE, DPIS
from gigalens_amd.profiles.mass.piep import DPIEP
from gigalens_amd.profiles.mass.scaling_relation import ScalingRelation
from oracle import ref_torch as ref

CAT = dict(lum=[1.0, 2.0, 0.5], center_x=[0.0, 1.0, -1.0], center_y=[0.5, -0.5, 0.0], e1=[0.1, 0.2, 0.0],
           e2=[0.0, 0.1, -0.1])


def test_names_and_parameter_lists_follow_the_reference():
    assert (DPIS().name, DPIS().params) == ("dPIS", ["theta_E", "r_core", "r_cut", "center_x", "center_y"])  # piemd.py:26-27
    assert (DPIE().name, DPIE().params) == ("dPIE", ["theta_E", "r_core", "r_cut", "center_x", "center_y", "e1", "e2"])  # :98-99
    assert (DPIEP().name, DPIEP().params) == ("dPIE", ["theta_E", "Ra", "Rs", "center_x", "center_y", "e1", "e2"])  # piep.py:22-23
    sub = DPIESubhalo(lum_star=1.0, galaxy_catalogue=CAT)
    assert sub.name == "Scaled-dPIE" and sub.params == ["theta_E", "r_core", "r_cut"]  # scaling_relation.py:19-24
    assert sub.n_galaxy == 3 and sub.chunk_size == 3 and sub.not_scaling_params == ["center_x", "center_y", "e1", "e2"]
    ser = DPIESubhaloSeries(lum_star=1.0, galaxy_catalogue=CAT, order=3)
    assert ser.name == "Scaled-SeriesExpansion-dPIE" and sorted(ser.params) == ["r_cut", "theta_E"]
    assert (ser.series_param, ser.amplitude_param, ser.order) == ("r_cut", "theta_E", 3)
    assert DPIESeries(2).name == "SeriesExpansion-dPIE" and DPIESeries(2).params == ["r_cut", "theta_E"]  # dpie_series.py:10-14


def test_catalogue_table_matches_the_oracle_constants():
    sub = ScalingRelation(DPIS(), ["r_cut", "theta_E"], 1.3, {"theta_E": 0.5, "r_cut": 0.4},
                          dict(CAT, r_core=[0.03, 0.03, 0.03]))
    kind, cols, t = sub._catalogue()
    assert kind == 6 and cols == [1, -1, 0] and t.shape == (3, 7) and t.dtype == np.float32
    un = ref.scaled_unscaled_factors(sub)  # (L/L*)^power in float32, scaling_relation.py:52-55
    assert np.allclose(t[:, 0], un["theta_E"].numpy(), rtol=2e-7) and np.allclose(t[:, 2], un["r_cut"].numpy(), rtol=2e-7)
    assert np.allclose(t[:, 1], 0.03) and np.allclose(t[:, 3], CAT["center_x"]) and np.all(t[:, 5:] == 0)  # dPIS: no e1, e2
    # any base profile / any of its parameters may scale (scaling_relation.py:8-19): such populations are served at the plugin level
    # (the generic sum over galaxies) and refused only inside a PhysicalModel, where the fused kernels take the dPIE family
    with pytest.raises(KeyError):
        ScalingRelation(EPL(), ["theta_E"], 1.0, {"theta_E": 0.5}, dict(lum=[1.0]))  # the EPL's other parameters are not in the catalogue
    odd = ScalingRelation(DPIE(), ["center_x"], 1.0, {"center_x": 0.5},
                          dict(CAT, theta_E=[1.0] * 3, r_core=[0.03] * 3, r_cut=[1.0] * 3, e1=[0.0] * 3, e2=[0.0] * 3))
    assert odd._generic
    from gigalens_amd import _native
    with pytest.raises(_native.NativeLibraryError, match="dPIS, dPIE, dPIEP"):
        odd._component()
    with pytest.raises(KeyError):
        ScalingRelation(DPIE(), ["theta_E"], 1.0, {"theta_E": 0.5}, CAT)  # r_core / r_cut columns missing


def test_packing_of_least_squares_amplitudes():
    phys = PhysicalModel([EPL()], [SersicEllipse(use_lstsq=True)], [Sersic(use_lstsq=True), Shapelets(2, use_lstsq=True)])
    assert "Ie" not in phys.lens_light[0].params and phys.source_light[1].params == ["beta", "center_x", "center_y"]
    lay = phys._packing()
    # native rows keep the amplitude columns: 6 + 7 + 5 + (3 + 6)
    assert lay.P == 27 and lay.linear == [12, 17, 21, 22, 23, 24, 25, 26]
    params = dict(lens_mass=[dict(theta_E=1.0, gamma=2.0, e1=0.0, e2=0.0, center_x=0.0, center_y=0.0)],
                  lens_light=[dict(R_sersic=1.0, n_sersic=2.0, e1=0.0, e2=0.0, center_x=0.0, center_y=0.0)],
                  source_light=[dict(R_sersic=0.2, n_sersic=1.0, center_x=0.0, center_y=0.0),
                                dict(beta=0.1, center_x=0.0, center_y=0.0)])
    packed = lay.pack(params, 2, torch.device("cpu"))
    assert packed.shape == (2, 27) and torch.all(packed[:, lay.linear] == 1.0)  # unit placeholders until the solve


def test_series_bookkeeping():
    ser = DPIESubhaloSeries(lum_star=1.0, galaxy_catalogue=CAT, order=2)
    with pytest.raises(ValueError):
        ser.set_deriv()  # series_profile.py:61-62 needs grid and constants
    ser.set_constants(dict(theta_E=0.3, r_core=0.02, r_cut=2.0))
    assert ser.series_var_0 == 2.0 and ser.constants_dict["r_core"] == 0.02
    kind, cols, table, scales = ser._series_inputs()
    assert kind == 7 and cols == [0, 1, 2] and table.shape == (3, 7) and scales[0] == 1.0 and scales[2] == 2.0
    assert ser._component() == (10, 2, 0) and ser._native_params() == ["theta_E", "r_cut"]
    with pytest.raises(ValueError):
        DPIESeries(order=6)
    one = DPIESeries(order=1)
    one.set_constants(dict(r_core=0.1, center_x=0.0, center_y=0.0, e1=0.1, e2=0.0, r_cut=3.0, theta_E=1.0))
    kind, cols, table, scales = one._series_inputs()
    assert cols == [-1, -1, 0] and table.shape == (1, 7) and scales == [3.0]


def test_get_coords_centres_the_grid():
    """LensSimulatorInterface.get_coords (src/gigalens/simulator.py:129-162): mean coordinate (0, 0); for the diagonal
    transform it is the simulator's own grid (simulator.py:47-55) at supersample 1 scale."""
    from gigalens_amd.simulator import LensSimulatorInterface, LensWCS
    d, n, ss = 0.065, 8, 2
    T = np.array([[d / ss, 0.0], [0.0, d / ss]])
    ra0, dec0, X, Y = LensSimulatorInterface.get_coords(ss, n, T)
    assert X.shape == (16, 16) and X.dtype == np.float32
    assert abs(X.mean()) < 1e-7 and abs(Y.mean()) < 1e-7
    assert np.isclose(ra0, X[0, 0]) and np.isclose(dec0, Y[0, 0])
    gx, gy = LensWCS(n=n, supersample=ss, pix_scale=d).pixel_grid()
    assert np.allclose(X, gx.reshape(16, 16), atol=1e-7) and np.allclose(Y, gy.reshape(16, 16), atol=1e-7)
    # a rotated / sheared transform keeps the centring
    T2 = np.array([[0.03, 0.01], [-0.012, 0.031]])
    _, _, X2, Y2 = LensSimulatorInterface.get_coords(1, 9, T2)
    assert abs(X2.mean()) < 1e-6 and abs(Y2.mean()) < 1e-6
    assert np.isclose(X2[0, 1] - X2[0, 0], 0.03) and np.isclose(Y2[1, 0] - Y2[0, 0], 0.031)


def test_nfw_table_in_s_reproduces_the_closed_form(hostmath):
    """The NFW table of gl_clusterw_kernel (H(s) = h(sqrt s), s = X^2, intervals on the float format of s, cubic Hermite
    coefficients per interval; csrc/gl_host_tables.h::build_nfw_table_s) read in float32 exactly as the kernel reads it
    (csrc/gl_clusterw.hip.h::nfw_fwd_s: interval and position from the bits of s, Horner steps as fused multiply-adds, the
    slope scaled by a power of two from the exponent) against the float64 closed form over the whole table: the value to
    4e-7, the slope dH/ds = h'(X) / (2X) -- what the gradients w.r.t. centre and scale radius are made of -- to 2e-6 (the
    node-pair table in X of the other kernels: 7e-5 in the slope).  Ref: tf/profiles/mass/nfw.py:26-52."""
    import ctypes
    from ctypes import POINTER, c_double, c_float, c_int
    rng = np.random.default_rng(3)
    X = np.exp2(rng.uniform(-6.0, 6.0, 400_000))
    X = X[np.abs(X - 1.0) > 1e-6]
    s = (X.astype(np.float32) ** 2).astype(np.float32)
    Xs = np.sqrt(s.astype(np.float64))  # the X the float32 s stands for
    ok = (Xs >= 2.0 ** -6) & (Xs < 2.0 ** 6)
    s, Xs = np.ascontiguousarray(s[ok]), np.ascontiguousarray(Xs[ok])
    H, dH = np.zeros_like(s), np.zeros_like(s)
    hostmath.hm_nfw_table_s_f32(c_int(len(s)), s.ctypes.data_as(POINTER(c_float)), H.ctypes.data_as(POINTER(c_float)),
                                dH.ctypes.data_as(POINTER(c_float)))
    h, hp = np.zeros_like(Xs), np.zeros_like(Xs)
    hostmath.hm_nfw_h_f64(c_int(len(Xs)), Xs.ctypes.data_as(POINTER(c_double)), h.ctypes.data_as(POINTER(c_double)),
                          hp.ctypes.data_as(POINTER(c_double)))
    assert np.isfinite(H).all() and np.isfinite(dH).all()
    assert np.max(np.abs(H - h) / np.abs(h)) < 4e-7
    dHds = hp / (2.0 * Xs)
    assert np.max(np.abs(dH - dHds) / np.abs(dHds)) < 2e-6
