// gl_jet.h -- truncated Taylor series in ONE variable (Taylor-mode automatic differentiation).
//
// The reference accelerates the cluster-member population by expanding its deflection in the population's cut
// radius, alpha(r_cut) = sum_n f_n (r_cut - r_cut0)^n / n!  (tf/series/series_profile.py:66-95), with f_n produced by
// 2.6 k lines of sympy-generated code (tf/series/profiles/dpie.py, generator series_codegen/sympy_codegen.py).
// Here the same coefficients come from instantiating the ordinary profile templates (gl_dpie.h, generic in the
// real type R) on Jet<float, N>: arithmetic on jets IS the series arithmetic, exact to rounding, any order.
//   c[k] = coefficient of h^k  (= k-th derivative / k!).
#pragma once
#include "gl_math.h"

namespace glj {

template <class T, int N> struct Jet {
  T c[N + 1];
  GL_HD Jet() { for (int i = 0; i <= N; ++i) c[i] = T(0); }
  GL_HD Jet(T v) { c[0] = v; for (int i = 1; i <= N; ++i) c[i] = T(0); }
  template <class S, class = decltype(T(S()))> GL_HD Jet(S v) { c[0] = T(v); for (int i = 1; i <= N; ++i) c[i] = T(0); }
};

#define GLJ_T template <class T, int N> GL_HD
GLJ_T Jet<T, N> operator+(const Jet<T, N>& a, const Jet<T, N>& b) { Jet<T, N> r; for (int i = 0; i <= N; ++i) r.c[i] = a.c[i] + b.c[i]; return r; }
GLJ_T Jet<T, N> operator-(const Jet<T, N>& a, const Jet<T, N>& b) { Jet<T, N> r; for (int i = 0; i <= N; ++i) r.c[i] = a.c[i] - b.c[i]; return r; }
GLJ_T Jet<T, N> operator-(const Jet<T, N>& a) { Jet<T, N> r; for (int i = 0; i <= N; ++i) r.c[i] = -a.c[i]; return r; }
GLJ_T Jet<T, N> operator*(const Jet<T, N>& a, const Jet<T, N>& b) {
  Jet<T, N> r;
  for (int k = 0; k <= N; ++k) {
    T s = T(0);
    for (int i = 0; i <= k; ++i) s += a.c[i] * b.c[k - i];
    r.c[k] = s;
  }
  return r;
}
GLJ_T Jet<T, N> operator/(const Jet<T, N>& a, const Jet<T, N>& b) {
  Jet<T, N> q;
  const T ib = T(1) / b.c[0];
  for (int k = 0; k <= N; ++k) {
    T s = a.c[k];
    for (int i = 0; i < k; ++i) s -= q.c[i] * b.c[k - i];
    q.c[k] = s * ib;
  }
  return q;
}
GLJ_T Jet<T, N>& operator+=(Jet<T, N>& a, const Jet<T, N>& b) { a = a + b; return a; }
GLJ_T Jet<T, N>& operator-=(Jet<T, N>& a, const Jet<T, N>& b) { a = a - b; return a; }
GLJ_T Jet<T, N>& operator*=(Jet<T, N>& a, const Jet<T, N>& b) { a = a * b; return a; }
// decisions are taken on values
GLJ_T bool operator<(const Jet<T, N>& a, const Jet<T, N>& b) { return a.c[0] < b.c[0]; }
GLJ_T bool operator>(const Jet<T, N>& a, const Jet<T, N>& b) { return a.c[0] > b.c[0]; }
GLJ_T bool operator<=(const Jet<T, N>& a, const Jet<T, N>& b) { return a.c[0] <= b.c[0]; }
GLJ_T bool operator>=(const Jet<T, N>& a, const Jet<T, N>& b) { return a.c[0] >= b.c[0]; }
GLJ_T bool operator==(const Jet<T, N>& a, const Jet<T, N>& b) { return a.c[0] == b.c[0]; }

GLJ_T Jet<T, N> j_sqrt(const Jet<T, N>& a) {
  Jet<T, N> s;
  s.c[0] = (T)::sqrt((double)a.c[0]);
  const T h = T(0.5) / s.c[0];
  for (int k = 1; k <= N; ++k) {
    T t = a.c[k];
    for (int i = 1; i < k; ++i) t -= s.c[i] * s.c[k - i];
    s.c[k] = t * h;
  }
  return s;
}
GLJ_T Jet<T, N> j_log(const Jet<T, N>& a) {
  Jet<T, N> l;
  l.c[0] = (T)::log((double)a.c[0]);
  const T ia = T(1) / a.c[0];
  for (int k = 1; k <= N; ++k) {
    T t = T(0);
    for (int i = 1; i < k; ++i) t += T(i) * l.c[i] * a.c[k - i];
    l.c[k] = (a.c[k] - t / T(k)) * ia;
  }
  return l;
}
// theta' = (x y' - y x') / (x^2 + y^2), integrated term by term
GLJ_T Jet<T, N> j_atan2(const Jet<T, N>& y, const Jet<T, N>& x) {
  Jet<T, N> dy, dx;  // derivative series (degree N-1, top coefficient unused)
  for (int k = 0; k < N; ++k) { dy.c[k] = T(k + 1) * y.c[k + 1]; dx.c[k] = T(k + 1) * x.c[k + 1]; }
  dy.c[N] = T(0);
  dx.c[N] = T(0);
  const Jet<T, N> w = (x * dy - y * dx) / (x * x + y * y);
  Jet<T, N> th;
  th.c[0] = (T)::atan2((double)y.c[0], (double)x.c[0]);
  for (int k = 1; k <= N; ++k) th.c[k] = w.c[k - 1] / T(k);
  return th;
}

// the vocabulary of gl_math.h / gl_profiles.h (found by ADL)
GLJ_T Jet<T, N> rcp(const Jet<T, N>& a) { return Jet<T, N>(T(1)) / a; }
GLJ_T Jet<T, N> sqrt_(const Jet<T, N>& a) { return j_sqrt(a); }
GLJ_T Jet<T, N> p_sqrt(const Jet<T, N>& a) { return j_sqrt(a); }
GLJ_T Jet<T, N> log_(const Jet<T, N>& a) { return j_log(a); }
GLJ_T Jet<T, N> p_log(const Jet<T, N>& a) { return j_log(a); }
GLJ_T Jet<T, N> log2_(const Jet<T, N>& a) { return j_log(a) * Jet<T, N>(T(glm::kLog2e)); }
GLJ_T Jet<T, N> p_atan2(const Jet<T, N>& y, const Jet<T, N>& x) { return j_atan2(y, x); }
GLJ_T Jet<T, N> fabs_(const Jet<T, N>& a) { return a.c[0] < T(0) ? -a : a; }
GLJ_T Jet<T, N> fmin_(const Jet<T, N>& a, const Jet<T, N>& b) { return a < b ? a : b; }
GLJ_T Jet<T, N> fmax_(const Jet<T, N>& a, const Jet<T, N>& b) { return a > b ? a : b; }
// leaf names of gl_dual.h, so that Dual<Jet<F, N>, 2> (space derivatives of a series in r_cut: the Hessian half of
// the accelerator, series_profile.py:64-65) instantiates the same profile templates
GLJ_T T val(const Jet<T, N>& a) { return a.c[0]; }
GLJ_T Jet<T, N> l_sqrt(const Jet<T, N>& a) { return j_sqrt(a); }
GLJ_T Jet<T, N> l_log(const Jet<T, N>& a) { return j_log(a); }
GLJ_T Jet<T, N> l_atan2(const Jet<T, N>& y, const Jet<T, N>& x) { return j_atan2(y, x); }
#undef GLJ_T

}  // namespace glj
