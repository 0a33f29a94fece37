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


def test_member_loop_bodies_of_populations_compile_without_a_gpu():
    """ScalingRelation outside the dPIE family inside a model: the generated member loop around the base profile's body
    (profiles/mass/scaling_relation.py `_member_loop_body`) is a valid user body for every base it is offered for."""
    import numpy as np
    from gigalens_amd.profile import MassProfile
    from gigalens_amd.profiles.mass.nfw import NFW
    from gigalens_amd.profiles.mass.sis import SIS
    from gigalens_amd.profiles.mass.tnfw import TNFW
    from gigalens_amd.profiles.mass.scaling_relation import ScalingRelation

    class UserSIS(MassProfile):
        _name, _params = "USER_SIS", ["theta_E", "center_x", "center_y"]
        hip_body = SIS_BODY

    cat = dict(lum=np.array([0.5, 1.0, 2.0], np.float32), center_x=np.array([0.1, -0.4, 0.9], np.float32),
               center_y=np.array([0.3, 0.2, -0.7], np.float32), alpha_Rs=np.array([0.5, 0.6, 0.7], np.float32),
               r_trunc=np.ones(3, np.float32))
    from gigalens_amd.profiles.mass.sie import SIE
    cat = dict(cat, e1=np.array([0.1, -0.2, 0.05], np.float32), e2=np.array([0.0, 0.1, -0.1], np.float32))
    for pop in (ScalingRelation(NFW(), ["Rs", "alpha_Rs"], 1.0, {"Rs": 0.4, "alpha_Rs": 0.5}, cat),
                ScalingRelation(SIE(), ["theta_E"], 1.0, {"theta_E": 0.5}, cat),
                ScalingRelation(SIS(), ["theta_E"], 1.0, {"theta_E": 0.5}, cat),
                ScalingRelation(UserSIS(), ["theta_E"], 1.0, {"theta_E": 0.5}, cat)):
        assert pop._generic and "sr_member_deriv" in pop.hip_body and "sr_cat[3]" in pop.hip_body
        assert pop._component() == (0, 0, 0) and pop._native_params() == pop.scaling_params
        _native.user_profile_check(pop.hip_body, False, len(pop.scaling_params))
    # a base with neither a fused kernel nor a member body: plugin level only, and the message says what is served
    pop = ScalingRelation(TNFW(), ["Rs"], 1.0, {"Rs": 0.5}, cat)
    assert pop._generic and not pop.hip_body
    with pytest.raises(_native.NativeLibraryError, match="NFW.*SIE.*SIS"):
        pop._component()


def test_mass_bodies_compile_for_the_point_kernels_without_a_gpu():
    """The image-position likelihood and the lens maps evaluate a user-written lens on NESTED duals (gld::Dual<float, 2>,
    gld::Dual<gld::Dual<float, 1>, 2>, csrc/gl_dual.h): the body's vocabulary -- arithmetic with plain numbers, comparisons, value(),
    the math functions -- must resolve on those types too."""
    import numpy as np
    from gigalens_amd.profiles.mass.nfw import NFW
    from gigalens_amd.profiles.mass.scaling_relation import ScalingRelation
    _native.user_points_check(SIS_BODY, 3)
    cat = dict(lum=np.array([0.5, 2.0], np.float32), center_x=np.array([0.1, -0.4], np.float32), center_y=np.array([0.3, 0.2], np.float32),
               alpha_Rs=np.array([0.5, 0.6], np.float32))
    _native.user_points_check(ScalingRelation(NFW(), ["Rs"], 1.0, {"Rs": 0.4}, cat).hip_body, 1)
    everything = """
    template <class R> __device__ void deriv(R x, R y, const R* p, R& fx, R& fy) {
      R dx = x - p[1], dy = 2.f * (y - p[2]) / 2.f;
      R r = sqrt(dx * dx + dy * dy) + 1e-3f;
      R a = p[0] * (exp(-r) + log(1.f + r) + pow(r, 1.5f) + pow(r, p[3]) + pow(2.f, -r) + sin(r) * cos(r) + tan(0.1f * r) + atan(r) +
                    atan2(dy, dx + 3.f) + sinh(0.1f * r) + cosh(0.1f * r) + tanh(r) + atanh(0.1f * tanh(r)) + abs(dx) + fmin(r, p[0]) + fmax(r, p[0]));
      if (value(r) > 2.f || dx < 0.f || 0.5f >= dy || r == p[0]) a += 1.f;
      a *= 0.5f;  a -= 0.1f;  a /= 1.5f;  a += r;  a = -a;  a = +a;
      fx = a * dx / r;  fy = a * dy / r;
    }"""
    _native.user_profile_check(everything, False, 4)
    _native.user_points_check(everything, 4)
    with pytest.raises(_native.NativeLibraryError, match="does not compile"):
        _native.user_points_check("template <class R> __device__ void deriv(R x, R y, const R* p, R& fx, R& fy) { fx = nope; }", 1)
