// gl_math.h -- scalar math wrappers for the lens hot path.
//
// Every per-pixel / per-sample function in gl_profiles.h is a template on the
// real type R.  On the GPU R = float and the wrappers below lower to single
// CDNA4 transcendental instructions (v_rcp_f32, v_sqrt_f32, v_exp_f32,
// v_log_f32: ~1 ulp, quarter rate) or to the precise ROCm device-library calls
// where conditioning demands it.  The same templates instantiate with
// R = double on the host so that tests/hostmath can check the hand-written
// VJPs against autograd to 1e-10 (test harness only; never a product path).
#pragma once
#if !defined(__HIPCC_RTC__)  // (the run-time compiler of user profiles brings its own device headers and has no host library)
#include <cmath>
#include <cstdint>
#else
typedef signed long long int64_t;
typedef unsigned long long uint64_t;
typedef signed int int32_t;
typedef unsigned int uint32_t;
typedef unsigned long size_t_rtc_unused;
#endif

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GL_HD __host__ __device__ __forceinline__
#else
#define GL_HD inline
#endif

namespace glm {

constexpr double kLn2 = 0.693147180559945309417232121458;
constexpr double kLog2e = 1.442695040888963407359924681002;
constexpr double kPi = 3.141592653589793238462643383280;

// ---- double (host harness) -------------------------------------------------
GL_HD double rcp(double x) { return 1.0 / x; }
GL_HD double sqrt_(double x) { return ::sqrt(x); }
GL_HD double exp2_(double x) { return ::exp2(x); }
GL_HD double log2_(double x) { return ::log2(x); }
GL_HD double exp_(double x) { return ::exp(x); }
GL_HD double log_(double x) { return ::log(x); }
GL_HD double atan_(double x) { return ::atan(x); }
GL_HD double atanh_(double x) { return ::atanh(x); }
GL_HD double fabs_(double x) { return ::fabs(x); }
GL_HD double fmin_(double a, double b) { return a < b ? a : b; }  // NaN in a -> b (matches clip semantics below)
GL_HD double fmax_(double a, double b) { return a > b ? a : b; }
GL_HD double floor_(double x) { return ::floor(x); }
GL_HD bool isnan_(double x) { return x != x; }

// ---- float -------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
GL_HD float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
GL_HD float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
GL_HD float exp2_(float x) { return __builtin_amdgcn_exp2f(x); }
GL_HD float log2_(float x) { return __builtin_amdgcn_logf(x); }
#else
GL_HD float rcp(float x) { return 1.0f / x; }
GL_HD float sqrt_(float x) { return ::sqrtf(x); }
GL_HD float exp2_(float x) { return ::exp2f(x); }
GL_HD float log2_(float x) { return ::log2f(x); }
#endif
// exp with the x*log2(e) rounding error folded back in (|rel err| ~2 ulp for any |x|)
GL_HD float exp_(float x) {
  const float hi = (float)kLog2e;
  const float lo = (float)(kLog2e - (double)(float)kLog2e);
  float t = x * hi;
  float e = fmaf(x, hi, -t) + x * lo;
  float p = exp2_(t);
  return fmaf(p, e * (float)kLn2, p);
}
GL_HD float log_(float x) { return log2_(x) * (float)kLn2; }
GL_HD float atan_(float x) { return ::atanf(x); }
GL_HD float atanh_(float x) { return ::atanhf(x); }
GL_HD float fabs_(float x) { return ::fabsf(x); }
GL_HD float fmin_(float a, float b) { return a < b ? a : b; }
GL_HD float fmax_(float a, float b) { return a > b ? a : b; }
GL_HD float floor_(float x) { return ::floorf(x); }
GL_HD bool isnan_(float x) { return x != x; }

template <class R> GL_HD R clamp_(R x, R lo, R hi) { return fmin_(fmax_(x, lo), hi); }

// the wider twin of a real type, for the few expressions whose cancellation exceeds fp32 (float -> double; a type that
// is already wide maps to itself; gl_dual.h extends this to dual numbers component-wise)
template <class R> struct Wide {
  using type = R;
  static GL_HD type up(const R& x) { return x; }
  static GL_HD R down(const type& x) { return x; }
};
template <> struct Wide<float> {
  using type = double;
  static GL_HD double up(float x) { return (double)x; }
  static GL_HD float down(double x) { return (float)x; }
};

}  // namespace glm
