"""Every refusal of the native library is a clear, typed error -- never a silent fallback, never a crash.

The closed-world limits (INTEGRATION.md "What does not transfer"): shapelets ``n_max > 20`` (the reference takes any
``n_max``, shapelets.py:20-24), linear systems above 255 coefficients, ``ScalingRelation`` over profiles outside the dPIE
family (scaling_relation.py:8-19 accepts any ``MassProfile``), user-defined ``deriv`` / ``light`` bodies inside a model
(profile.py:58-82 are abstract extension points in the reference; here a body written in Python cannot run -- one written as a
``hip_body`` is compiled at run time and served at the plugin level and inside models, tests/test_gpu_user_profile.py)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GL_EINVAL, GL_EUNSUPPORTED = -1, -2


def _create(comps, n_lens, n_ll, n_src, n=16):
    from gigalens_amd import _native
    L = _native.lib()
    arr = (_native.gl_component * len(comps))(*[_native.gl_component(*c, 0) for c in comps])
    gx = np.zeros(n * n, np.float32)
    g = _native.gl_grid()
    g.height, g.width, g.supersample, g.n_region = n, n, 1, n * n
    g.grid_x = gx.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    g.grid_y = gx.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    g.conversion_factor = 1.0
    h = ctypes.c_void_p()
    rc = L.gl_model_create(arr, n_lens, n_ll, n_src, ctypes.byref(g), ctypes.byref(h))
    msg = L.gl_last_error().decode()
    if rc == 0:
        L.gl_model_destroy(h)
    return rc, msg


def test_shapelets_above_the_cap_are_refused():
    from gigalens_amd import _native
    from gigalens_amd.profiles.light.shapelets import Shapelets
    rc, msg = _create([(1, 50, 0), (18, 21, 0)], 1, 0, 1)
    assert rc == GL_EUNSUPPORTED and "n_max=21" in msg
    for ok_nmax in (10, 11, 20):  # above 10 the runtime-order path of the interpreter kernel serves the model
        rc, msg = _create([(1, 50, 0), (18, ok_nmax, 0)], 1, 0, 1)
        assert rc == 0, msg
    rc, msg = _create([(7, 0, 0), (18, 12, 0)], 1, 0, 1)  # ... but only beside the basic profile families (here: a dPIE lens)
    assert rc == GL_EUNSUPPORTED and "n_max > 10" in msg
    shp = Shapelets(n_max=22, interpolate=False)
    x = torch.zeros(4)
    kw = {k: 1.0 for k in shp.params}
    with pytest.raises(_native.NativeLibraryError, match="n_max=22"):
        shp.light(x, x, **kw)
    with pytest.raises(_native.NativeLibraryError, match="n_max=22"):
        Shapelets(n_max=22, interpolate=False, use_lstsq=True).light(x, x, beta=1.0, center_x=0.0, center_y=0.0)


def test_linear_systems_above_255_coefficients_are_refused():
    """Two n_max = 15 shapelet sources solved linearly = 272 coefficients: the solve serves 255 (above 127 with its matrices in
    the workspace instead of LDS); round 2 stopped at 79."""
    from gigalens_amd import _native
    from gigalens_amd.model import PhysicalModel
    from gigalens_amd.profiles.light.shapelets import Shapelets
    from gigalens_amd.profiles.mass.sis import SIS
    from gigalens_amd.simulator import LensSimulator, SimulatorConfig
    phys = PhysicalModel([SIS()], [], [Shapelets(15, use_lstsq=True, interpolate=False), Shapelets(15, use_lstsq=True, interpolate=False)])
    sim = LensSimulator(phys, SimulatorConfig(delta_pix=0.1, num_pix=20), bs=2)
    params = {"lens_mass": [dict(theta_E=1.0, center_x=0.0, center_y=0.0)],
              "source_light": [dict(beta=0.2, center_x=0.0, center_y=0.0), dict(beta=0.3, center_x=0.1, center_y=0.0)]}
    obs, err = np.ones((20, 20), np.float32), np.ones((20, 20), np.float32)
    with pytest.raises(_native.NativeLibraryError, match="272 linear coefficients"):
        sim.lstsq_simulate(params, obs, err)
    # the stack itself (no solve) is still served
    assert sim.lstsq_simulate(params, obs, err, return_stacked=True).shape == (2, 20, 20, 272)


def test_scaling_relation_outside_the_dpie_family_is_refused():
    from gigalens_amd import _native
    from gigalens_amd.profiles.mass.nfw import NFW
    from gigalens_amd.profiles.mass.scaling_relation import ScalingRelation
    from gigalens_amd.model import PhysicalModel
    from gigalens_amd.profiles.light.sersic import Sersic
    from gigalens_amd.simulator import LensSimulator, SimulatorConfig
    cat = dict(lum=np.ones(3, np.float32), center_x=np.zeros(3, np.float32), center_y=np.zeros(3, np.float32))
    # populations of NFW / SIS / user-written members are compiled at run time (tests/test_gpu_user_profile.py); a base kind with
    # neither a fused kernel nor a member body is served at the plugin level only
    from gigalens_amd.profiles.mass.tnfw import TNFW
    pop = ScalingRelation(TNFW(), ["Rs"], 1.0, {"Rs": 0.5}, dict(cat, alpha_Rs=np.ones(3, np.float32), r_trunc=np.ones(3, np.float32)))
    with pytest.raises(_native.NativeLibraryError, match="dPIS, dPIE, dPIEP"):
        LensSimulator(PhysicalModel([pop], [], [Sersic()]), SimulatorConfig(delta_pix=0.1, num_pix=8), bs=1)
    L = _native.lib()
    one = torch.zeros(4, device="cuda")
    cols = (ctypes.c_int32 * 3)(0, -1, -1)
    rc = L.gl_scaled_eval(3, 3, cols, _native._ptr(one), _native._ptr(one), _native._ptr(one), 4, 1, 0, _native._ptr(one), 1,
                          _native._ptr(one), _native._ptr(one), None)
    assert rc == GL_EUNSUPPORTED and "not built" in L.gl_last_error().decode()


def test_user_defined_profile_bodies_are_refused_with_a_clear_message():
    """The reference's extension point: subclass MassProfile / LightProfile and write ``deriv`` / ``light`` in TensorFlow
    (profile.py:58-82).  Here those bodies are HIP templates inside the library; a subclass without a ``gl_kind`` cannot be
    evaluated and says so (adding a profile = adding a kind: csrc/gl_profiles.h + the dispatch switches)."""
    from gigalens_amd import _native
    from gigalens_amd.model import PhysicalModel
    from gigalens_amd.profile import MassProfile
    from gigalens_amd.profiles.light.sersic import Sersic
    from gigalens_amd.simulator import LensSimulator, SimulatorConfig

    class MyLens(MassProfile):
        _name, _params = "MINE", ["a", "center_x", "center_y"]

        def deriv(self, x, y, a, center_x, center_y):
            return a * (x - center_x), a * (y - center_y)

    with pytest.raises(_native.NativeLibraryError, match="user-defined"):
        LensSimulator(PhysicalModel([MyLens()], [], [Sersic()]), SimulatorConfig(delta_pix=0.1, num_pix=8), bs=1)
    rc, msg = _create([(99, 0, 0), (16, 0, 0)], 1, 0, 1)
    assert rc == GL_EINVAL and "kind 99" in msg
    rc, msg = _create([(16, 0, 0), (16, 0, 0)], 1, 0, 1)  # a light kind in a lens slot
    assert rc == GL_EINVAL and "not a mass profile" in msg


def test_oversized_models_are_refused_not_truncated():
    """More accumulators than the 64 KiB LDS columns hold: refused at create time."""
    rc, msg = _create([(1, 50, 0)] + [(18, 10, 0)] * 4, 1, 0, 4)  # 4 shapelet sources: 4 x 69 accumulators x 64 columns > 64 KiB
    assert rc in (0, GL_EUNSUPPORTED)
    if rc:
        assert "LDS" in msg
    rc, msg = _create([(1, 50, 0)] + [(18, 10, 0)] * 16, 1, 0, 16)
    assert rc == GL_EUNSUPPORTED and "LDS" in msg
