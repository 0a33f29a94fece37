"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
symbol include/gigalens_hip.h declares; argument validation works without a GPU (no compute is launched)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    import __graft_entry__ as ge
    ge.build()
    from gigalens_amd import _native
    return _native


def test_header_symbols_exported(native):
    hdr = open(os.path.join(ROOT, "include", "gigalens_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gl_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 14
    lib = native.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/gigalens_hip.h but not exported"
    assert declared == set(native.SYMBOLS), "ctypes table out of sync with the header"
    assert b"gfx950" in lib.gl_version()


def test_argument_validation_without_gpu(native):
    lib = native.lib()
    comp = native.gl_component(1, 0, 0, 0)
    assert lib.gl_kind_num_params(ctypes.byref(comp)) == 6
    comp = native.gl_component(18, 10, 0, 0)
    assert lib.gl_kind_num_params(ctypes.byref(comp)) == 69
    comp = native.gl_component(99, 0, 0, 0)
    assert lib.gl_kind_num_params(ctypes.byref(comp)) < 0
    assert b"unknown profile kind" in lib.gl_last_error()
    out = ctypes.c_void_p()
    assert lib.gl_model_create(None, 0, 0, 0, None, ctypes.byref(out)) == -1  # GL_EINVAL: grid is null
    assert lib.gl_workspace_bytes(None, 4) == 0


def test_product_has_no_cpu_fallback(native):
    """Without a GPU every compute entry of the product raises instead of silently computing elsewhere."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gigalens_amd.profiles.mass.sis import SIS
    with pytest.raises(native.NativeLibraryError):
        SIS().deriv(x=[0.1], y=[0.2], theta_E=1.0, center_x=0.0, center_y=0.0)
    from gigalens_amd import workloads
    from gigalens_amd.simulator import LensSimulator
    wl = workloads.make("C1", num_pix=8, batch=1)
    with pytest.raises(native.NativeLibraryError):
        LensSimulator(wl.phys_model, wl.sim_config, bs=1)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "gigalens_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
