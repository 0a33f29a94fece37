"""The open plugin boundary, the part that needs no GPU: a user-written profile body (profile.py `hip_body`) is compiled by hiprtc
for gfx950 -- a correct body compiles, a wrong one comes back with the compiler's own message and the user's line numbers."""
import pytest

from gigalens_amd import _native

SIS_BODY = """
template <class R> __device__ void deriv(R x, R y, const R* p, R& fx, R& fy) {
  // p = theta_E, center_x, center_y (src/gigalens/tf/profiles/mass/sis.py: alpha = theta_E (dx, dy) / r)
  R dx = x - p[1], dy = y - p[2];
  R r = sqrt(dx * dx + dy * dy);
  fx = p[0] * dx / r;
  fy = p[0] * dy / r;
}
"""


def test_a_user_body_compiles_without_a_gpu():
    _native.user_profile_check(SIS_BODY, False, 3)
    _native.user_profile_check("template <class R> __device__ R light(R x, R y, const R* p) { return p[2] * exp(-(x * x + y * y) / (2.f * p[0] * p[1])); }",
                               True, 3)


def test_compile_errors_come_back_verbatim():
    with pytest.raises(_native.NativeLibraryError) as e:
        _native.user_profile_check("template <class R> __device__ R light(R x, R y, const R* p) {\n  return no_such_function(x);\n}", True, 1)
    msg = str(e.value)
    assert "does not compile" in msg and "user_profile:2" in msg and "no_such_function" in msg
    with pytest.raises(_native.NativeLibraryError, match="n_params"):
        _native.user_profile_check(SIS_BODY, False, 100)
    with pytest.raises(_native.NativeLibraryError):  # a mass body offered as a light profile: no `light` to instantiate
        _native.user_profile_check(SIS_BODY, True, 3)
