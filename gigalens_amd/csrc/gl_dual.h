// gl_dual.h -- forward-mode dual numbers for the image-position likelihood (tf/model.py:103-124).
//
// The reference obtains lensing Hessians by differentiating `deriv` with a persistent GradientTape
// (tf/profile.py:9-43) and then differentiates the magnification again w.r.t. the parameters.  Here the
// per-profile templates of gl_profiles.h (they are generic in the real type R) are simply instantiated with
// nested duals:  Dual<Dual<float,P>,2>  carries  alpha, d alpha/d(x,y)  and the parameter derivatives of both
// (mixed second derivatives) in one evaluation -- exact to rounding, no finite differences, no tape.
// Only the tiny (n_images x batch) position kernels use this; the pixel path never does.
#pragma once
#include "gl_math.h"

namespace gld {

template <class T, int N> struct Dual {
  T v;
  T d[N];
  GL_HD Dual() : v(T(0)) { for (int i = 0; i < N; ++i) d[i] = T(0); }
  GL_HD Dual(const T& x) : v(x) { for (int i = 0; i < N; ++i) d[i] = T(0); }
  template <class S, class = decltype(T(S()))> GL_HD Dual(S x) : v(T(x)) { for (int i = 0; i < N; ++i) d[i] = T(0); }
};

// value of the innermost scalar (for comparisons / branches: decisions are taken on values only)
GL_HD float val(float x) { return x; }
GL_HD double val(double x) { return x; }
template <class T, int N> GL_HD auto val(const Dual<T, N>& x) { return val(x.v); }

#define GLD_T template <class T, int N> GL_HD
GLD_T Dual<T, N> operator+(const Dual<T, N>& a, const Dual<T, N>& b) { Dual<T, N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
GLD_T Dual<T, N> operator-(const Dual<T, N>& a, const Dual<T, N>& b) { Dual<T, N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
GLD_T Dual<T, N> operator-(const Dual<T, N>& a) { Dual<T, N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
GLD_T Dual<T, N> operator*(const Dual<T, N>& a, const Dual<T, N>& b) { Dual<T, N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
GLD_T Dual<T, N> operator/(const Dual<T, N>& a, const Dual<T, N>& b) {
  Dual<T, N> r;
  T ib = T(1) / b.v;
  r.v = a.v * ib;
  for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
  return r;
}
GLD_T Dual<T, N>& operator+=(Dual<T, N>& a, const Dual<T, N>& b) { a = a + b; return a; }
GLD_T Dual<T, N>& operator-=(Dual<T, N>& a, const Dual<T, N>& b) { a = a - b; return a; }
GLD_T Dual<T, N>& operator*=(Dual<T, N>& a, const Dual<T, N>& b) { a = a * b; return a; }
GLD_T bool operator<(const Dual<T, N>& a, const Dual<T, N>& b) { return val(a) < val(b); }
GLD_T bool operator>(const Dual<T, N>& a, const Dual<T, N>& b) { return val(a) > val(b); }
GLD_T bool operator<=(const Dual<T, N>& a, const Dual<T, N>& b) { return val(a) <= val(b); }
GLD_T bool operator>=(const Dual<T, N>& a, const Dual<T, N>& b) { return val(a) >= val(b); }
GLD_T bool operator==(const Dual<T, N>& a, const Dual<T, N>& b) { return val(a) == val(b); }

// chain rule helper: f(a) with f'(a) = fp
GLD_T Dual<T, N> chain(const Dual<T, N>& a, const T& f, const T& fp) { Dual<T, N> r; r.v = f; for (int i = 0; i < N; ++i) r.d[i] = fp * a.d[i]; return r; }

}  // namespace gld

// ---- the math vocabulary of gl_math.h / gl_profiles.h for duals (found by ADL) ---------------------------------
namespace gld {
// scalar leaf functions in precise form (this path is tiny; accuracy first)
GL_HD float l_sqrt(float x) { return ::sqrtf(x); }
GL_HD float l_log(float x) { return ::logf(x); }
GL_HD float l_exp(float x) { return ::expf(x); }
GL_HD float l_atan(float x) { return ::atanf(x); }
GL_HD float l_atan2(float y, float x) { return ::atan2f(y, x); }
GL_HD float l_sin(float x) { return ::sinf(x); }
GL_HD float l_cos(float x) { return ::cosf(x); }
GL_HD float l_pow(float x, float y) { return ::powf(x, y); }
GL_HD float l_atanh(float x) { return ::atanhf(x); }
GL_HD double l_sqrt(double x) { return ::sqrt(x); }
GL_HD double l_log(double x) { return ::log(x); }
GL_HD double l_exp(double x) { return ::exp(x); }
GL_HD double l_atan(double x) { return ::atan(x); }
GL_HD double l_atan2(double y, double x) { return ::atan2(y, x); }
GL_HD double l_sin(double x) { return ::sin(x); }
GL_HD double l_cos(double x) { return ::cos(x); }
GL_HD double l_pow(double x, double y) { return ::pow(x, y); }
GL_HD double l_atanh(double x) { return ::atanh(x); }

GLD_T Dual<T, N> l_sqrt(const Dual<T, N>& a) { T s = l_sqrt(a.v); return chain(a, s, T(0.5) / s); }
GLD_T Dual<T, N> l_log(const Dual<T, N>& a) { return chain(a, l_log(a.v), T(1) / a.v); }
GLD_T Dual<T, N> l_exp(const Dual<T, N>& a) { T e = l_exp(a.v); return chain(a, e, e); }
GLD_T Dual<T, N> l_atan(const Dual<T, N>& a) { return chain(a, l_atan(a.v), T(1) / (T(1) + a.v * a.v)); }
GLD_T Dual<T, N> l_atanh(const Dual<T, N>& a) { return chain(a, l_atanh(a.v), T(1) / (T(1) - a.v * a.v)); }
GLD_T Dual<T, N> l_sin(const Dual<T, N>& a) { return chain(a, l_sin(a.v), l_cos(a.v)); }
GLD_T Dual<T, N> l_cos(const Dual<T, N>& a) { return chain(a, l_cos(a.v), -l_sin(a.v)); }
GLD_T Dual<T, N> l_atan2(const Dual<T, N>& y, const Dual<T, N>& x) {
  Dual<T, N> r;
  r.v = l_atan2(y.v, x.v);
  T den = x.v * x.v + y.v * y.v;
  for (int i = 0; i < N; ++i) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) / den;
  return r;
}
GLD_T Dual<T, N> l_pow(const Dual<T, N>& x, const Dual<T, N>& y) {  // x > 0
  Dual<T, N> r;
  r.v = l_pow(x.v, y.v);
  T dx = y.v * l_pow(x.v, y.v - T(1)), dy = r.v * l_log(x.v);
  for (int i = 0; i < N; ++i) r.d[i] = dx * x.d[i] + dy * y.d[i];
  return r;
}

// names used by the profile templates
GLD_T Dual<T, N> rcp(const Dual<T, N>& a) { return Dual<T, N>(T(1)) / a; }
GLD_T Dual<T, N> sqrt_(const Dual<T, N>& a) { return l_sqrt(a); }
GLD_T Dual<T, N> log_(const Dual<T, N>& a) { return l_log(a); }
GLD_T Dual<T, N> exp_(const Dual<T, N>& a) { return l_exp(a); }
GLD_T Dual<T, N> log2_(const Dual<T, N>& a) { return l_log(a) * Dual<T, N>(T(glm::kLog2e)); }
GLD_T Dual<T, N> exp2_(const Dual<T, N>& a) { return l_exp(a * Dual<T, N>(T(glm::kLn2))); }
GLD_T Dual<T, N> atan_(const Dual<T, N>& a) { return l_atan(a); }
GLD_T Dual<T, N> atanh_(const Dual<T, N>& a) { return l_atanh(a); }
GLD_T Dual<T, N> fabs_(const Dual<T, N>& a) { return val(a) < 0 ? -a : a; }
GLD_T Dual<T, N> fmin_(const Dual<T, N>& a, const Dual<T, N>& b) { return a < b ? a : b; }
GLD_T Dual<T, N> fmax_(const Dual<T, N>& a, const Dual<T, N>& b) { return a > b ? a : b; }
GLD_T bool isnan_(const Dual<T, N>& a) { auto x = val(a); return x != x; }
GLD_T Dual<T, N> p_sqrt(const Dual<T, N>& a) { return l_sqrt(a); }
GLD_T Dual<T, N> p_log(const Dual<T, N>& a) { return l_log(a); }
GLD_T Dual<T, N> p_sin(const Dual<T, N>& a) { return l_sin(a); }
GLD_T Dual<T, N> p_cos(const Dual<T, N>& a) { return l_cos(a); }
GLD_T Dual<T, N> p_atan2(const Dual<T, N>& y, const Dual<T, N>& x) { return l_atan2(y, x); }
GLD_T Dual<T, N> p_pow(const Dual<T, N>& x, const Dual<T, N>& y) { return l_pow(x, y); }
#undef GLD_T
}  // namespace gld

namespace glm {
template <class T, int N> struct Wide<gld::Dual<T, N>> {
  using type = gld::Dual<typename Wide<T>::type, N>;
  static GL_HD type up(const gld::Dual<T, N>& x) {
    type r;
    r.v = Wide<T>::up(x.v);
    for (int i = 0; i < N; ++i) r.d[i] = Wide<T>::up(x.d[i]);
    return r;
  }
  static GL_HD gld::Dual<T, N> down(const type& x) {
    gld::Dual<T, N> r;
    r.v = Wide<T>::down(x.v);
    for (int i = 0; i < N; ++i) r.d[i] = Wide<T>::down(x.d[i]);
    return r;
  }
};
}  // namespace glm
