// gl_user.hip -- the open plugin boundary: profile bodies written by the USER, compiled at run time into a point kernel.
//
// The reference's extension point is a Python subclass: MassProfile.deriv / LightProfile.light are abstract (profile.py:58-82) and
// a user writes them in TensorFlow.  Here a profile is normally a gl_kind the library implements; this file opens the same door
// for the plugin-level calls (deriv / light on arbitrary points, with derivatives): the user supplies ONE function template in
// HIP C++ over a number type R,
//     template <class R> __device__ void deriv(R x, R y, const R* p, R& fx, R& fy);     // mass profile: deflection
//     template <class R> __device__ R    light(R x, R y, const R* p);                   // light profile: surface brightness
// hiprtc compiles it once for gfx950 together with a forward-mode dual number type (namespace gl: arithmetic, comparisons, sqrt
// exp log pow sin cos tan atan atan2 sinh cosh tanh atanh abs, found by argument-dependent lookup, so the body just writes
// sqrt(x)) and a kernel that instantiates it on float (values) and on gl::Dual<n_params + 2> (values and the Jacobian with
// respect to x, y and every parameter in one pass).  Nothing is refused but compile errors, which come back verbatim.
// The pixel kernels of the likelihood path still take built-in kinds only (gl_model_create): see INTEGRATION.md.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gigalens_hip.h"
#include "gl_model.h"

namespace {

const char* kPrelude = R"GLSRC(
namespace gl {
template <int N> struct Dual {
  float v;
  float d[N];
  __device__ Dual() : v(0.f) { for (int i = 0; i < N; ++i) d[i] = 0.f; }
  __device__ Dual(float x) : v(x) { for (int i = 0; i < N; ++i) d[i] = 0.f; }
  __device__ Dual(double x) : v((float)x) { for (int i = 0; i < N; ++i) d[i] = 0.f; }
  __device__ Dual(int x) : v((float)x) { for (int i = 0; i < N; ++i) d[i] = 0.f; }
  __device__ static Dual var(float x, int k) { Dual r(x); r.d[k] = 1.f; return r; }
};
// f(a) with derivative fp:  chain rule
template <int N> __device__ inline Dual<N> chain(const Dual<N>& a, float f, float fp) {
  Dual<N> r; r.v = f; for (int i = 0; i < N; ++i) r.d[i] = fp * a.d[i]; return r;
}
template <int N> __device__ inline Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> __device__ inline Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> __device__ inline Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> __device__ inline Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
  const float ib = 1.f / b.v, q = a.v * ib;
  Dual<N> r; r.v = q; for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - q * b.d[i]) * ib; return r;
}
template <int N> __device__ inline Dual<N> operator-(const Dual<N>& a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> __device__ inline Dual<N> operator+(const Dual<N>& a) { return a; }
#define GL_MIXED(OP)                                                                                                   \
  template <int N, class S> __device__ inline auto operator OP(const Dual<N>& a, S b) -> decltype(a OP Dual<N>((float)b)) { return a OP Dual<N>((float)b); } \
  template <int N, class S> __device__ inline auto operator OP(S a, const Dual<N>& b) -> decltype(Dual<N>((float)a) OP b) { return Dual<N>((float)a) OP b; }
GL_MIXED(+) GL_MIXED(-) GL_MIXED(*) GL_MIXED(/)
#undef GL_MIXED
template <int N> __device__ inline Dual<N>& operator+=(Dual<N>& a, const Dual<N>& b) { a = a + b; return a; }
template <int N> __device__ inline Dual<N>& operator-=(Dual<N>& a, const Dual<N>& b) { a = a - b; return a; }
template <int N> __device__ inline Dual<N>& operator*=(Dual<N>& a, const Dual<N>& b) { a = a * b; return a; }
template <int N> __device__ inline Dual<N>& operator/=(Dual<N>& a, const Dual<N>& b) { a = a / b; return a; }
template <int N, class S> __device__ inline Dual<N>& operator+=(Dual<N>& a, S b) { a.v += (float)b; return a; }
template <int N, class S> __device__ inline Dual<N>& operator-=(Dual<N>& a, S b) { a.v -= (float)b; return a; }
template <int N, class S> __device__ inline Dual<N>& operator*=(Dual<N>& a, S b) { a = a * Dual<N>((float)b); return a; }
template <int N, class S> __device__ inline Dual<N>& operator/=(Dual<N>& a, S b) { a = a / Dual<N>((float)b); return a; }
#define GL_CMP(OP)                                                                                           \
  template <int N> __device__ inline bool operator OP(const Dual<N>& a, const Dual<N>& b) { return a.v OP b.v; } \
  template <int N, class S> __device__ inline bool operator OP(const Dual<N>& a, S b) { return a.v OP (float)b; } \
  template <int N, class S> __device__ inline bool operator OP(S a, const Dual<N>& b) { return (float)a OP b.v; }
GL_CMP(<) GL_CMP(>) GL_CMP(<=) GL_CMP(>=) GL_CMP(==) GL_CMP(!=)
#undef GL_CMP
template <int N> __device__ inline Dual<N> sqrt(const Dual<N>& a) { const float s = ::sqrtf(a.v); return chain(a, s, 0.5f / s); }
template <int N> __device__ inline Dual<N> exp(const Dual<N>& a) { const float e = ::expf(a.v); return chain(a, e, e); }
template <int N> __device__ inline Dual<N> log(const Dual<N>& a) { return chain(a, ::logf(a.v), 1.f / a.v); }
template <int N> __device__ inline Dual<N> sin(const Dual<N>& a) { return chain(a, ::sinf(a.v), ::cosf(a.v)); }
template <int N> __device__ inline Dual<N> cos(const Dual<N>& a) { return chain(a, ::cosf(a.v), -::sinf(a.v)); }
template <int N> __device__ inline Dual<N> tan(const Dual<N>& a) { const float t = ::tanf(a.v); return chain(a, t, 1.f + t * t); }
template <int N> __device__ inline Dual<N> atan(const Dual<N>& a) { return chain(a, ::atanf(a.v), 1.f / (1.f + a.v * a.v)); }
template <int N> __device__ inline Dual<N> sinh(const Dual<N>& a) { return chain(a, ::sinhf(a.v), ::coshf(a.v)); }
template <int N> __device__ inline Dual<N> cosh(const Dual<N>& a) { return chain(a, ::coshf(a.v), ::sinhf(a.v)); }
template <int N> __device__ inline Dual<N> tanh(const Dual<N>& a) { const float t = ::tanhf(a.v); return chain(a, t, 1.f - t * t); }
template <int N> __device__ inline Dual<N> atanh(const Dual<N>& a) { return chain(a, ::atanhf(a.v), 1.f / (1.f - a.v * a.v)); }
template <int N> __device__ inline Dual<N> abs(const Dual<N>& a) { return a.v < 0.f ? -a : a; }
template <int N> __device__ inline Dual<N> fabs(const Dual<N>& a) { return a.v < 0.f ? -a : a; }
template <int N> __device__ inline Dual<N> atan2(const Dual<N>& y, const Dual<N>& x) {
  const float r2 = x.v * x.v + y.v * y.v, ir2 = 1.f / r2;
  Dual<N> r; r.v = ::atan2f(y.v, x.v); for (int i = 0; i < N; ++i) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) * ir2; return r;
}
template <int N> __device__ inline Dual<N> pow(const Dual<N>& a, const Dual<N>& b) {  // a > 0
  const float p = ::powf(a.v, b.v), la = ::logf(a.v);
  Dual<N> r; r.v = p; for (int i = 0; i < N; ++i) r.d[i] = p * (b.d[i] * la + b.v * a.d[i] / a.v); return r;
}
template <int N, class S> __device__ inline Dual<N> pow(const Dual<N>& a, S b) { const float e = (float)b; return chain(a, ::powf(a.v, e), e * ::powf(a.v, e - 1.f)); }
template <int N, class S> __device__ inline Dual<N> pow(S a, const Dual<N>& b) { const float p = ::powf((float)a, b.v); return chain(b, p, p * ::logf((float)a)); }
template <int N> __device__ inline Dual<N> fmin(const Dual<N>& a, const Dual<N>& b) { return a.v < b.v ? a : b; }
template <int N> __device__ inline Dual<N> fmax(const Dual<N>& a, const Dual<N>& b) { return a.v > b.v ? a : b; }
// the value of a number, for branches the body wants to take on plain floats
__device__ inline float value(float a) { return a; }
template <int N> __device__ inline float value(const Dual<N>& a) { return a.v; }
}  // namespace gl
using gl::value;  // value(float) has no namespace to be found through
)GLSRC";

const char* kKernel = R"GLSRC(
extern "C" __global__ void __launch_bounds__(256) gl_user_point_kernel(const float* __restrict__ x, const float* __restrict__ y, long long n,
                                                                       int B, int xy_batched, const float* __restrict__ params,
                                                                       float* __restrict__ out0, float* __restrict__ out1,
                                                                       float* __restrict__ jac) {
  constexpr int NPAR = GL_USER_NP, N = GL_USER_NP + 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x, total = n * (long long)B;
  if (i >= total) return;
  const int b = (int)(i % B);
  const long long ip = xy_batched ? i : i / B;
  const float* p = params + (size_t)b * NPAR;
  if (!jac) {
    float pp[NPAR > 0 ? NPAR : 1];
    for (int k = 0; k < NPAR; ++k) pp[k] = p[k];
#if GL_USER_LIGHT
    out0[i] = light<float>(x[ip], y[ip], pp);
#else
    float fx = 0.f, fy = 0.f;
    deriv<float>(x[ip], y[ip], pp, fx, fy);
    out0[i] = fx;
    out1[i] = fy;
#endif
    return;
  }
  typedef gl::Dual<N> D;
  D pp[NPAR > 0 ? NPAR : 1];
  for (int k = 0; k < NPAR; ++k) pp[k] = D::var(p[k], 2 + k);
  const D X = D::var(x[ip], 0), Y = D::var(y[ip], 1);
#if GL_USER_LIGHT
  const D f = light<D>(X, Y, pp);
  out0[i] = f.v;
  for (int j = 0; j < N; ++j) jac[(size_t)j * total + i] = f.d[j];
#else
  D fx, fy;
  deriv<D>(X, Y, pp, fx, fy);
  out0[i] = fx.v;
  out1[i] = fy.v;
  for (int j = 0; j < N; ++j) {
    jac[(size_t)j * total + i] = fx.d[j];
    jac[(size_t)(N + j) * total + i] = fy.d[j];
  }
#endif
}
)GLSRC";

}  // namespace

struct gl_user_profile {
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;
  int is_light = 0, n_params = 0;
};

namespace {
// body -> gfx950 code object (no device needed); GL_OK or the compiler's own words
int compile_user_profile(const char* body, int is_light, int n_params, std::vector<char>& code) {
  using glk::fail;
  if (!body) return fail(GL_EINVAL, "body is null");
  if (n_params < 0 || n_params > 62) return fail(GL_EINVAL, "n_params %d outside [0, 62]", n_params);
  std::string src = std::string(kPrelude) + "\n#line 1 \"user_profile\"\n" + body + "\n" + kKernel;
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, src.c_str(), "gl_user_profile.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
    return fail(GL_ELAUNCH, "hiprtcCreateProgram failed");
  const std::string d_np = "-DGL_USER_NP=" + std::to_string(n_params), d_l = std::string("-DGL_USER_LIGHT=") + (is_light ? "1" : "0");
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", d_np.c_str(), d_l.c_str()};
  const hiprtcResult rc = hiprtcCompileProgram(prog, 5, opts);
  if (rc != HIPRTC_SUCCESS) {
    size_t n = 0;
    (void)hiprtcGetProgramLogSize(prog, &n);
    std::string log(n ? n : 1, '\0');
    if (n) (void)hiprtcGetProgramLog(prog, &log[0]);
    (void)hiprtcDestroyProgram(&prog);
    // the first error line onward, as much as the message buffer holds
    const size_t at = log.find("error");
    const size_t from = at == std::string::npos ? 0 : log.rfind('\n', at) == std::string::npos ? 0 : log.rfind('\n', at) + 1;
    return fail(GL_EINVAL, "user profile does not compile (%s):\n%.380s", hiprtcGetErrorString(rc), log.c_str() + from);
  }
  size_t code_size = 0;
  if (hiprtcGetCodeSize(prog, &code_size) != HIPRTC_SUCCESS || !code_size) {
    (void)hiprtcDestroyProgram(&prog);
    return fail(GL_ELAUNCH, "hiprtcGetCodeSize failed");
  }
  code.resize(code_size);
  const hiprtcResult rc2 = hiprtcGetCode(prog, code.data());
  (void)hiprtcDestroyProgram(&prog);
  if (rc2 != HIPRTC_SUCCESS) return fail(GL_ELAUNCH, "hiprtcGetCode failed");
  return GL_OK;
}
}  // namespace

// ---- user bodies inside a model: the interpreter kernel compiled with them ----------------------------------------------------
#include "gl_kernels.hip.h"  // MainArgs, Mode, kinds (host view; the device code of this translation unit is unused)
#include "gl_static.hip.h"   // KindList codes of user-written profiles, static_nacc

namespace glk {

// The kernel headers the run-time compile includes travel INSIDE the library: __graft_entry__.build() writes every csrc/*.h into
// build/gl_embedded_headers.inc as string literals (kEmbeddedNames / kEmbeddedSources / kEmbeddedCount) and hiprtcCreateProgram
// gets them as named headers -- an installed library needs no source checkout.  GIGALENS_HIP_CSRC=<dir> (development) reads the
// headers from a directory instead.
#include "gl_embedded_headers.inc"

// one compiled interpreter per distinct program text (user bodies, parameter counts, shapelet / family switches are all part
// of the text): ModellingSequence builds a LensSimulator per stage and per batch size, each calling gl_model_create_user with
// the same bodies -- the seconds of hiprtc are paid once per process, a model only loads the code object
struct UserCode {
  std::vector<char> code;
  std::vector<std::string> lowered;  // of the name expressions, in the caller's order
};
static std::mutex g_user_mu;
static std::map<std::string, std::shared_ptr<const UserCode>>& user_cache() {
  static std::map<std::string, std::shared_ptr<const UserCode>> c;
  return c;
}
static std::atomic<long long> g_user_compiles{0};

// program text + name expressions -> code object and lowered names, compiled once per distinct text and process
static int rtc_compile_cached(const std::string& src, const std::vector<std::string>& names, std::shared_ptr<const UserCode>& out) {
  const char* dev_dir = getenv("GIGALENS_HIP_CSRC");
  std::lock_guard<std::mutex> lock(g_user_mu);  // (also serialises concurrent compiles of the same text: the second finds the first's)
  const std::string key = src + (dev_dir ? std::string("\n//csrc=") + dev_dir : std::string());
  auto it = user_cache().find(key);
  if (it != user_cache().end()) {
    out = it->second;
    return GL_OK;
  }
  hiprtcProgram prog = nullptr;
  const hiprtcResult rcc = dev_dir ? hiprtcCreateProgram(&prog, src.c_str(), "gl_user_model.hip", 0, nullptr, nullptr)
                                   : hiprtcCreateProgram(&prog, src.c_str(), "gl_user_model.hip", kEmbeddedCount, kEmbeddedSources, kEmbeddedNames);
  if (rcc != HIPRTC_SUCCESS) return fail(GL_ELAUNCH, "hiprtcCreateProgram failed");
  for (const std::string& n : names) (void)hiprtcAddNameExpression(prog, n.c_str());
  const std::string inc = std::string("-I") + (dev_dir ? dev_dir : ".");
  // (-fno-slp-vectorize: as for the interpreter of the library itself, gl_launch_generic.hip.h)
  // (-fno-hip-fp32-correctly-rounded-divide-sqrt: a user body's `/` and sqrt as the hardware reciprocal / square root with
  // one Newton step -- 2.5 ulp, what the built-in kinds' v_rcp / v_sqrt wrappers deliver -- instead of the correctly rounded
  // sequences: a body is evaluated on n_params + 2 tangents, every division of which was ~10 instructions)
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fno-hip-fp32-correctly-rounded-divide-sqrt", inc.c_str()};
  const hiprtcResult rc = hiprtcCompileProgram(prog, dev_dir ? 6 : 5, opts);
  if (rc != HIPRTC_SUCCESS) {
    size_t n = 0;
    (void)hiprtcGetProgramLogSize(prog, &n);
    std::string log(n ? n : 1, '\0');
    if (n) (void)hiprtcGetProgramLog(prog, &log[0]);
    (void)hiprtcDestroyProgram(&prog);
    const size_t at = log.find("error");
    const size_t from = at == std::string::npos ? 0 : log.rfind('\n', at) == std::string::npos ? 0 : log.rfind('\n', at) + 1;
    return fail(GL_EINVAL, "user profile does not compile (%s; kernel headers %s):\n%.330s", hiprtcGetErrorString(rc),
                dev_dir ? dev_dir : "embedded in the library", log.c_str() + from);
  }
  auto fresh = std::make_shared<UserCode>();
  for (const std::string& n : names) {
    const char* ln = nullptr;
    if (hiprtcGetLoweredName(prog, n.c_str(), &ln) != HIPRTC_SUCCESS || !ln) {
      (void)hiprtcDestroyProgram(&prog);
      return fail(GL_ELAUNCH, "hiprtcGetLoweredName failed for %s", n.c_str());
    }
    fresh->lowered.push_back(ln);
  }
  size_t code_size = 0;
  if (hiprtcGetCodeSize(prog, &code_size) != HIPRTC_SUCCESS || !code_size) {
    (void)hiprtcDestroyProgram(&prog);
    return fail(GL_ELAUNCH, "hiprtcGetCodeSize failed");
  }
  fresh->code.resize(code_size);
  const hiprtcResult rc2 = hiprtcGetCode(prog, fresh->code.data());
  (void)hiprtcDestroyProgram(&prog);
  if (rc2 != HIPRTC_SUCCESS) return fail(GL_ELAUNCH, "hiprtcGetCode failed");
  g_user_compiles.fetch_add(1);
  user_cache()[key] = fresh;
  out = fresh;
  return GL_OK;
}

// What a user body needs from the nested duals of the point kernels (gl_dual.h: gld::Dual<T, N>, T = float or a dual again): the
// vocabulary of the prelude above under the same names -- found by argument-dependent lookup when the body is instantiated on
// gld::Dual<float, 2> (Hessians) and gld::Dual<gld::Dual<float, 1>, 2> (their parameter derivatives).
static const char* kGldVocabulary = R"GLSRC(
#include "gl_dual.h"
namespace gld {
#define GLD_U template <class T, int N> __device__ inline
GLD_U Dual<T, N> sqrt(const Dual<T, N>& a) { return l_sqrt(a); }
GLD_U Dual<T, N> exp(const Dual<T, N>& a) { return l_exp(a); }
GLD_U Dual<T, N> log(const Dual<T, N>& a) { return l_log(a); }
GLD_U Dual<T, N> sin(const Dual<T, N>& a) { return l_sin(a); }
GLD_U Dual<T, N> cos(const Dual<T, N>& a) { return l_cos(a); }
GLD_U Dual<T, N> tan(const Dual<T, N>& a) { return l_sin(a) / l_cos(a); }
GLD_U Dual<T, N> atan(const Dual<T, N>& a) { return l_atan(a); }
GLD_U Dual<T, N> atanh(const Dual<T, N>& a) { return l_atanh(a); }
GLD_U Dual<T, N> atan2(const Dual<T, N>& y, const Dual<T, N>& x) { return l_atan2(y, x); }
GLD_U Dual<T, N> sinh(const Dual<T, N>& a) { const Dual<T, N> e = l_exp(a); return (e - Dual<T, N>(T(1)) / e) * Dual<T, N>(T(0.5f)); }
GLD_U Dual<T, N> cosh(const Dual<T, N>& a) { const Dual<T, N> e = l_exp(a); return (e + Dual<T, N>(T(1)) / e) * Dual<T, N>(T(0.5f)); }
GLD_U Dual<T, N> tanh(const Dual<T, N>& a) { const Dual<T, N> e = l_exp(a + a); return (e - Dual<T, N>(T(1))) / (e + Dual<T, N>(T(1))); }
GLD_U Dual<T, N> abs(const Dual<T, N>& a) { return fabs_(a); }
GLD_U Dual<T, N> fabs(const Dual<T, N>& a) { return fabs_(a); }
GLD_U Dual<T, N> fmin(const Dual<T, N>& a, const Dual<T, N>& b) { return fmin_(a, b); }
GLD_U Dual<T, N> fmax(const Dual<T, N>& a, const Dual<T, N>& b) { return fmax_(a, b); }
GLD_U Dual<T, N> pow(const Dual<T, N>& a, const Dual<T, N>& b) { return l_pow(a, b); }
template <class T, int N, class S> __device__ inline auto pow(const Dual<T, N>& a, S b) -> decltype(Dual<T, N>((float)b)) { return l_pow(a, Dual<T, N>((float)b)); }
template <class T, int N, class S> __device__ inline auto pow(S a, const Dual<T, N>& b) -> decltype(Dual<T, N>((float)a)) { return l_exp(b * Dual<T, N>(::logf((float)a))); }
GLD_U float value(const Dual<T, N>& a) { return (float)val(a); }
#define GLD_MIXED(OP) \
  template <class T, int N, class S> __device__ inline auto operator OP(const Dual<T, N>& a, S b) -> decltype(a OP Dual<T, N>((float)b)) { return a OP Dual<T, N>((float)b); } \
  template <class T, int N, class S> __device__ inline auto operator OP(S a, const Dual<T, N>& b) -> decltype(Dual<T, N>((float)a) OP b) { return Dual<T, N>((float)a) OP b; }
GLD_MIXED(+) GLD_MIXED(-) GLD_MIXED(*) GLD_MIXED(/)
GLD_MIXED(<) GLD_MIXED(>) GLD_MIXED(<=) GLD_MIXED(>=) GLD_MIXED(==)
#undef GLD_MIXED
GLD_U bool operator!=(const Dual<T, N>& a, const Dual<T, N>& b) { return val(a) != val(b); }
GLD_U Dual<T, N>& operator/=(Dual<T, N>& a, const Dual<T, N>& b) { a = a / b; return a; }
GLD_U Dual<T, N> operator+(const Dual<T, N>& a) { return a; }
template <class T, int N, class S> __device__ inline auto operator+=(Dual<T, N>& a, S b) -> decltype(a = a + Dual<T, N>((float)b)) { a = a + Dual<T, N>((float)b); return a; }
template <class T, int N, class S> __device__ inline auto operator-=(Dual<T, N>& a, S b) -> decltype(a = a - Dual<T, N>((float)b)) { a = a - Dual<T, N>((float)b); return a; }
template <class T, int N, class S> __device__ inline auto operator*=(Dual<T, N>& a, S b) -> decltype(a = a * Dual<T, N>((float)b)) { a = a * Dual<T, N>((float)b); return a; }
template <class T, int N, class S> __device__ inline auto operator/=(Dual<T, N>& a, S b) -> decltype(a = a / Dual<T, N>((float)b)) { a = a / Dual<T, N>((float)b); return a; }
#undef GLD_U
}  // namespace gld
)GLSRC";

// program text of the point kernels for a set of bodies (npar[w] < 0: body w unused; light[w]: it is a light profile)
static std::string point_program(const std::string& body_text, const std::vector<int>& npar, const std::vector<int>& light) {
  std::string ps = std::string("#define GL_HAVE_USER_POINT 1\n") + kGldVocabulary + kPrelude + body_text +
                   "#line 1 \"gl_user_point_glue\"\nnamespace glu {\n"
                   "template <class R> __device__ inline void mass_point(int which, const R* p, R x, R y, R& ax, R& ay) {\n  ax = R(0.f);  ay = R(0.f);\n  switch (which) {\n";
  for (size_t w = 0; w < npar.size(); ++w)
    if (npar[w] >= 0 && light[w] == 0) ps += "    case " + std::to_string(w) + ": glu_body" + std::to_string(w) + "::deriv<R>(x, y, p, ax, ay); break;\n";
  ps += "  }\n}\n}  // namespace glu\n#include \"gl_positions.hip.h\"\n";
  return ps;
}
static const char* const kPointKernels[5] = {"glk::gl_pos_p1_kernel", "glk::gl_pos_p2_kernel", "glk::gl_pos_p3_kernel", "glk::gl_pos_p4_kernel",
                                             "glk::gl_lens_maps_kernel"};

// a mass body through the point kernels' compile, no device needed (gl_user_points_check)
int check_user_points(const char* body, int n_params) {
  if (!body) return fail(GL_EINVAL, "body is null");
  if (n_params < 0 || n_params > 16) return fail(GL_EINVAL, "n_params %d outside [0, 16]", n_params);
  const std::string text = std::string("namespace glu_body0 {\n#line 1 \"user_profile_0\"\n") + body + "\n}\n";
  std::shared_ptr<const UserCode> uc;
  return rtc_compile_cached(point_program(text, {n_params}, {0}), std::vector<std::string>(kPointKernels, kPointKernels + 5), uc);
}

int compile_user_model(gl_model* m, const char* const* bodies, int n_bodies) {
  // which body serves which kind, with how many parameters (a body used by several components must agree with itself)
  std::vector<int> npar(n_bodies, -1), light(n_bodies, -1);
  for (const CompDesc& cd : m->comps) {
    if (cd.kind != K_USER_MASS && cd.kind != K_USER_LIGHT) continue;
    const int w = (int)cd.flags, l = cd.kind == K_USER_LIGHT;
    if ((npar[w] >= 0 && npar[w] != cd.iparam) || (light[w] >= 0 && light[w] != l))
      return fail(GL_EINVAL, "user body %d is used with different parameter counts / as mass and light profile", w);
    npar[w] = cd.iparam;
    light[w] = l;
  }
  std::string src = "#define GL_HAVE_USER 1\n";
  src += kPrelude;
  std::string body_text;
  for (int w = 0; w < n_bodies; ++w) {
    if (npar[w] < 0) continue;
    body_text += "namespace glu_body" + std::to_string(w) + " {\n#line 1 \"user_profile_" + std::to_string(w) + "\"\n" + bodies[w] + "\n}\n";
  }
  src += body_text;
  // the program of the point kernels (image-position likelihood, lens maps; compiled when first asked for: compile_user_points)
  m->user_point_src = point_program(body_text, npar, light);
  auto cases = [&](int want_light, const char* fmt_body) {
    std::string out;
    for (int w = 0; w < n_bodies; ++w) {
      if (npar[w] < 0 || light[w] != want_light) continue;
      std::string b = fmt_body;
      auto rep = [&](const std::string& k, const std::string& v) { for (size_t at; (at = b.find(k)) != std::string::npos;) b.replace(at, k.size(), v); };
      rep("@W", std::to_string(w));
      rep("@N1", std::to_string(npar[w] > 0 ? npar[w] : 1));
      rep("@N2", std::to_string(npar[w] + 2));
      rep("@N", std::to_string(npar[w]));
      out += b;
    }
    return out;
  };
  src += "#line 1 \"gl_user_glue\"\nnamespace glu {\n"
         "__device__ inline void mass_fwd(unsigned which, const float* d, float x, float y, float& ax, float& ay) {\n  ax = ay = 0.f;\n  switch (which) {\n";
  src += cases(0, "    case @W: { float p[@N1]; for (int k = 0; k < @N; ++k) p[k] = d[k]; glu_body@W::deriv<float>(x, y, p, ax, ay); } break;\n");
  src += "  }\n}\n"
         "__device__ inline void mass_vjp(unsigned which, const float* d, float x, float y, float gx, float gy, float* acc) {\n  switch (which) {\n";
  src += cases(0, "    case @W: { typedef gl::Dual<@N1> D; D p[@N1]; for (int k = 0; k < @N; ++k) p[k] = D::var(d[k], k); D fx, fy;\n"
                  "      glu_body@W::deriv<D>(D(x), D(y), p, fx, fy); for (int k = 0; k < @N; ++k) acc[k] += gx * fx.d[k] + gy * fy.d[k]; } break;\n");
  src += "  }\n}\n"
         "__device__ inline float light_fwd(unsigned which, const float* d, float x, float y) {\n  switch (which) {\n";
  src += cases(1, "    case @W: { float p[@N1]; for (int k = 0; k < @N; ++k) p[k] = d[k]; return glu_body@W::light<float>(x, y, p); }\n");
  src += "  }\n  return 0.f;\n}\n"
         "__device__ inline void light_vjp(unsigned which, const float* d, float x, float y, float g, float* acc, float& dgx, float& dgy) {\n  switch (which) {\n";
  src += cases(1, "    case @W: { typedef gl::Dual<@N2> D; D p[@N1]; for (int k = 0; k < @N; ++k) p[k] = D::var(d[k], 2 + k);\n"
                  "      const D f = glu_body@W::light<D>(D::var(x, 0), D::var(y, 1), p); dgx += g * f.d[0]; dgy += g * f.d[1];\n"
                  "      for (int k = 0; k < @N; ++k) acc[k] += g * f.d[2 + k]; } break;\n");
  src += "  }\n}\n}  // namespace glu\n#include \"gl_kernels.hip.h\"\n";
  // the instantiations launch_main would pick for an interpreter model: T = 2, the model's shapelet / family switches
  // (mode 4 = IMG_BASIS, the basis stack of the linear-amplitude solve: only for models with linear columns)
  const int n_modes = m->lin_cols.empty() ? 4 : 5;
  std::string names[5];
  for (int mode = 0; mode < n_modes; ++mode) {
    names[mode] = "glk::gl_main_kernel<" + std::to_string(mode) + ", 2, " + (m->has_shapelets ? "true" : "false") + ", " + std::to_string(m->fam) + ", false>";
    src += "template __global__ void " + names[mode] + "(glk::MainArgs);\n";
  }
  // Round 4: the SPECIALISED pixel-pair kernel for the model's own component list (the same gl_pair_kernel the built-in
  // compositions run: component loops unrolled at compile time, accumulators in registers, forward state kept for the VJPs,
  // packed fp32 for the built-in members) with the user bodies as members of the kind lists -- when every lens is EPL / SIE /
  // Shear / SIS or user-written, every light Sersic / SersicEllipse or user-written, the counts are small and the accumulator
  // row is the plain [stats | lenses | lights] sequence the kernel addresses in closed form.
  std::string pair_names[4];
  bool pair_ok = getenv("GIGALENS_HIP_USER_PAIR") == nullptr || atoi(getenv("GIGALENS_HIP_USER_PAIR")) != 0;
  {
    const int n_lens = m->n_lens, n_ll = m->n_ll, n_src = m->n_src;
    pair_ok = pair_ok && !m->has_shapelets && m->fam == 0 && n_lens >= 1 && n_lens <= 4 && n_ll <= 2 && n_src >= 1 && n_src <= 3;
    int off = NSTAT;
    std::string lists[3];
    for (int i = 0; i < (int)m->comps.size() && pair_ok; ++i) {
      const CompDesc& cd = m->comps[i];
      const bool lens = i < n_lens;
      int code = -1;
      if (lens && (cd.kind == K_EPL || cd.kind == K_SIE || cd.kind == K_SHEAR || cd.kind == K_SIS)) code = cd.kind;
      if (!lens && (cd.kind == K_SERSIC || cd.kind == K_SERSIC_ELLIPSE)) code = cd.kind;
      if (lens && cd.kind == K_USER_MASS && cd.iparam <= 16) code = user_code(false, (int)cd.flags, cd.iparam);
      if (!lens && cd.kind == K_USER_LIGHT && cd.iparam <= 16) code = user_code(true, (int)cd.flags, cd.iparam);
      pair_ok = code >= 0 && cd.a_off == off && cd.n_acc == static_nacc(code);
      off += cd.n_acc;
      std::string& l = lists[lens ? 0 : (i < n_lens + n_ll ? 1 : 2)];
      l += (l.empty() ? "" : ", ") + std::to_string(code);
    }
    pair_ok = pair_ok && off == m->A;
    if (pair_ok) {
      src += "#include \"gl_pair.hip.h\"\n";
      for (int mode = 0; mode < 4; ++mode) {
        pair_names[mode] = "glk::gl_pair_kernel<" + std::to_string(mode) + ", glk::v2f, 2, glk::KindList<" + lists[0] + ">, glk::KindList<" +
                           lists[1] + ">, glk::KindList<" + lists[2] + "> >";
        src += "template __global__ void " + pair_names[mode] + "(glk::MainArgs);\n";
      }
    }
  }
  std::vector<std::string> all_names(names, names + n_modes);
  if (pair_ok) all_names.insert(all_names.end(), pair_names, pair_names + 4);
  std::shared_ptr<const UserCode> uc;
  if (int rc = rtc_compile_cached(src, all_names, uc)) return rc;
  hipError_t e = hipModuleLoadData(&m->user_module, uc->code.data());
  for (int mode = 0; mode < n_modes && e == hipSuccess; ++mode)
    e = hipModuleGetFunction(&m->user_fn[mode], m->user_module, uc->lowered[mode].c_str());
  for (int mode = 0; mode < 4 && pair_ok && e == hipSuccess; ++mode)
    e = hipModuleGetFunction(&m->user_pair_fn[mode], m->user_module, uc->lowered[n_modes + mode].c_str());
  if (e != hipSuccess) return fail(GL_ELAUNCH, "loading the compiled user model failed: %s", hipGetErrorString(e));
  return GL_OK;
}

// The point kernels of a model with user-written lenses (gl_positions.hip.h: the four kernels of the image-position likelihood
// and the lens-maps kernel), compiled with the bodies on the nested duals when first asked for -- most models never are.
int compile_user_points(const gl_model* cm) {
  gl_model* m = const_cast<gl_model*>(cm);
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (m->user_point_fn[0]) return GL_OK;
  if (m->user_point_src.empty()) return fail(GL_EINVAL, "model without user-written profiles");
  std::shared_ptr<const UserCode> uc;
  if (int rc = rtc_compile_cached(m->user_point_src, std::vector<std::string>(kPointKernels, kPointKernels + 5), uc)) return rc;
  hipFunction_t fn[5] = {};
  hipError_t e = hipModuleLoadData(&m->user_point_module, uc->code.data());
  for (int k = 0; k < 5 && e == hipSuccess; ++k) e = hipModuleGetFunction(&fn[k], m->user_point_module, uc->lowered[k].c_str());
  if (e != hipSuccess) return fail(GL_ELAUNCH, "loading the compiled point kernels failed: %s", hipGetErrorString(e));
  for (int k = 4; k >= 0; --k) m->user_point_fn[k] = fn[k];  // [0] last: it is the "compiled" flag
  return GL_OK;
}

}  // namespace glk

extern "C" {

long long gl_user_model_compile_count(void) { return glk::g_user_compiles.load(); }

int gl_user_points_check(const char* body, int n_params) { return glk::check_user_points(body, n_params); }

int gl_user_profile_check(const char* body, int is_light, int n_params) {
  std::vector<char> code;
  return compile_user_profile(body, is_light, n_params, code);
}

int gl_user_profile_create(const char* body, int is_light, int n_params, gl_user_profile** out) {
  using glk::fail;
  if (!out) return fail(GL_EINVAL, "out is null");
  *out = nullptr;
  std::vector<char> code;
  if (int rc = compile_user_profile(body, is_light, n_params, code)) return rc;
  gl_user_profile* u = new gl_user_profile;
  u->is_light = is_light ? 1 : 0;
  u->n_params = n_params;
  hipError_t e = hipModuleLoadData(&u->module, code.data());
  if (e == hipSuccess) e = hipModuleGetFunction(&u->fn, u->module, "gl_user_point_kernel");
  if (e != hipSuccess) {
    if (u->module) (void)hipModuleUnload(u->module);
    delete u;
    return fail(GL_ELAUNCH, "loading the compiled user profile failed: %s", hipGetErrorString(e));
  }
  *out = u;
  return GL_OK;
}

int gl_user_profile_eval(const gl_user_profile* u, const float* x, const float* y, int64_t n_pts, int B, int xy_batched,
                         const float* params, float* out0, float* out1, float* jac_or_null, void* hip_stream) {
  using glk::fail;
  if (!u) return fail(GL_EINVAL, "user profile is null");
  if (!x || !y || !out0 || (!u->is_light && !out1)) return fail(GL_EINVAL, "x / y / out is null");
  if (u->n_params > 0 && !params) return fail(GL_EINVAL, "params is null");
  if (n_pts < 0 || B < 1) return fail(GL_EINVAL, "bad n_pts / B");
  if (n_pts == 0) return GL_OK;
  long long n = n_pts;
  const long long total = n * (long long)B;
  if ((total + 255) / 256 > 0x7fffffffLL) return fail(GL_EINVAL, "too many points for one launch");
  void* args[] = {(void*)&x, (void*)&y, (void*)&n, (void*)&B, (void*)&xy_batched, (void*)&params, (void*)&out0, (void*)&out1, (void*)&jac_or_null};
  const hipError_t e = hipModuleLaunchKernel(u->fn, (unsigned)((total + 255) / 256), 1, 1, 256, 1, 1, 0, (hipStream_t)hip_stream, args, nullptr);
  if (e != hipSuccess) return fail(GL_ELAUNCH, "launch of the user profile kernel failed: %s", hipGetErrorString(e));
  return GL_OK;
}

void gl_user_profile_destroy(gl_user_profile* u) {
  if (!u) return;
  if (u->module) (void)hipModuleUnload(u->module);
  delete u;
}

}  // extern "C"
