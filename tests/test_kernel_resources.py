"""Register budgets of the kernels the BASELINE configs dispatch: no VGPR spills, no scratch traffic.

CPU half: the code-object metadata of the shipped library (tools/isa_flops.py reads the AMDGPU notes of every translation
unit's gfx950 code object) for the instantiations the dispatch tables of csrc/gl_launch.hip.h select for configs C1-C5 --
the names are the ones rocprofv3 lists in profiles/ and the GPU half re-derives from live launches
(tests/test_gpu_parity.py::test_dispatched_kernels_do_not_spill asks the library which kernel it launched)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

V2 = "float __vector(2)"
EPLSHEAR, SIE, NONE, SERSIC, SHAPELETS = "glk::KindList<1, 4>", "glk::KindList<2>", "glk::KindList<>", "glk::KindList<16>", "glk::KindList<18>"

# config -> kernels of simulate() [IMG_FWD=0], its VJP [IMG_BWD=1], log-likelihood [LL_FWD=2] and fused forward+gradient [LL_GRAD=3]
DISPATCHED = {
    "C1 SIE | Sersic": [f"gl_pair_kernel<{m}, {V2}, {w}, {SIE}, {NONE}, {SERSIC} >" for m, w in ((0, 4), (1, 3), (2, 4), (3, 3))],
    "C2 EPL+Shear | Sersic": [f"gl_pair_kernel<{m}, {V2}, {w}, {EPLSHEAR}, {NONE}, {SERSIC} >" for m, w in ((0, 4), (1, 3), (2, 4), (3, 3))],
    "C3 EPL+Shear | Shapelets": [f"gl_shp_kernel<{m}, 2, {EPLSHEAR}, {NONE}, 6, true, false>" for m in (0, 1, 2, 3)],  # table mode (the default), whole tiles
    "C4 / C5 8 NFW | 20 Sersic": ["gl_main_kernel<0, 4, false, 0, false>", "gl_clusterw_kernel<1, glk::CwLensNfw<2>, 5, false, 3>",
                                  "gl_main_kernel<2, 4, false, 0, false>", "gl_clusterw_kernel<3, glk::CwLensNfw<2>, 5, false, 3>"],
}


@pytest.fixture(scope="module")
def metadata():
    import isa_flops as isa
    if not os.path.exists(isa.LIB):
        pytest.skip("library not built")
    return isa.kernel_metadata(isa.code_object())


@pytest.mark.parametrize("config", sorted(DISPATCHED))
def test_dispatched_instantiations_do_not_spill(metadata, config):
    for pat in DISPATCHED[config]:
        hits = [k for k in metadata if pat in k]
        assert len(hits) == 1, (pat, hits)
        md = metadata[hits[0]]
        assert md["vgpr_spill_count"] == 0, (hits[0], md)
        assert md["vgpr_count"] <= 256
        if md["scratch_bytes"]:  # a reserved private segment is tolerated only if no instruction touches it
            import isa_flops as isa
            ins = isa.disassemble(md["co"], md["symbol"])
            assert not [i for i in ins if i[1].startswith("scratch_")], (hits[0], md["scratch_bytes"])


# Instantiations that are allowed to spill, and why none of them is on a default dispatch path of a BASELINE config
# (everything else in the library must compile without VGPR spills):
ALLOWED_SPILLS = [
    (r"gl_main_kernel<\d, \d, (true|false), 2, false>", "FAM 2: the profile families beyond SURVEY section 8 (NFW_ELLIPSE, TNFW with its float64 core, CoreSersic)"),
    (r"gl_main_kernel<[13], 2, true, [01], false>", "shapelets through the interpreter, gradient modes (models outside the specialised compositions)"),
    (r"gl_static_kernel<[13], [24], ", "pre-pair tile variants of the specialised kernels in gradient modes: reached only with GIGALENS_HIP_PAIR=0 / GIGALENS_HIP_TILE*"),
    (r"gl_series_hessian_precompute_kernel<", "one-off float64 jet precompute of the Hessian series (not on the per-step path)"),
]


def test_no_unexpected_spills(metadata):
    import re
    bad = []
    for name, md in metadata.items():
        if md["vgpr_spill_count"] and not any(re.search(p, name) for p, _ in ALLOWED_SPILLS):
            bad.append((name[:120], md["vgpr_spill_count"]))
    assert not bad, bad
