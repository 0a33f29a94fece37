// gl_eigh.h -- symmetric eigendecomposition and pseudo-inverse solve of the normal matrix of
// LensSimulator.lstsq_simulate (tf/simulator.py:226-236:  coeffs = pinv(X^T X, rcond=1e-6) X^T Y).
//
// One 64-lane wavefront owns one system (n <= 79) in LDS: Householder reduction to tridiagonal form, explicit
// accumulation of the reflectors, implicit-shift QL on the tridiagonal with the rotations applied to the eigenvector
// matrix, then  coeffs = V diag(1/lambda_k if |lambda_k| > rcond max|lambda|) V^T rhs  -- tf.linalg.pinv's cutoff on
// the singular values, which for the symmetric normal matrix are |eigenvalues|.  ~(4/3 + 4/3 + ~3) n^3 flops against
// ~50 n^3 of a cyclic Jacobi solve, and no workgroup barrier in the QL phase: every lane carries the scalar rotation
// chain redundantly (identical values, identical writes) and owns one row of V.
//
// The routine is a template on an execution context so that the SAME code runs serially on the host
// (tests/hostmath: checked against numpy's float64 pinv) and lane-parallel on the device.
#pragma once
#include "gl_math.h"
#ifndef GL_STAMP
#define GL_STAMP(k)
#endif

namespace gle {

constexpr int EIG_MAXN = 80;

// Execution context of the host harness: one "lane" owns every row.  The QL phase keeps the tridiagonal (d, es) in the
// context: plain arrays here, lane-distributed registers in the device context (gl_lstsq.hip.h), where an LDS
// round-trip per scalar access would dominate the rotation chain.
struct SerialCtx {
  static constexpr int ROWS = EIG_MAXN;  // rows of Z one lane owns at most
  float dd_[EIG_MAXN + 1], ee_[EIG_MAXN + 1];
  GL_HD int lane() const { return 0; }
  GL_HD int lanes() const { return 1; }
  GL_HD float sum(float v) const { return v; }
  GL_HD float max(float v) const { return v; }
  GL_HD void sync() const {}
  GL_HD float rsq(float x) const { return 1.0f / ::sqrtf(x); }
  GL_HD void load_tridiagonal(const float* d, const float* es, int n) {
    for (int i = 0; i < n; ++i) { dd_[i] = d[i]; ee_[i] = es[i]; }
  }
  GL_HD void store_diagonal(float* d, int n) const { for (int i = 0; i < n; ++i) d[i] = dd_[i]; }
  GL_HD float d(int i) const { return dd_[i]; }
  GL_HD float e(int i) const { return ee_[i]; }
  GL_HD void set_d(int i, float v) { dd_[i] = v; }
  GL_HD void set_e(int i, float v) { ee_[i] = v; }
  // smallest m >= l with m == n-1 or a negligible coupling es[m]
  GL_HD int first_split(int l, int n) const {
    int m = l;
    for (; m < n - 1; ++m) {
      const float s = ::fabsf(dd_[m]) + ::fabsf(dd_[m + 1]);
      if (::fabsf(ee_[m]) + s == s) break;
    }
    return m;
  }
};

// A [n][ld] full symmetric (destroyed), Z [n][ld] (eigenvectors in columns on return), d [n] eigenvalues,
// e [n+1], v, p, bet [n]: scratch.  ld odd keeps lane-strided row accesses on distinct LDS banks.
// The context must be ONE lock-step wavefront (or serial): the QL phase relies on every lane executing the same
// scalar instruction stream, so that uniform writes to d / e need no ordering between lanes.
template <class C>
GL_HD void sym_eig(C& cx, float* A, float* Z, int n, int ld, float* d, float* e, float* v, float* p, float* bet) {
  const int lane = cx.lane(), NL = cx.lanes();
  // ---- Householder tridiagonalisation, rows n-1 .. 2; reflector i acts on coordinates 0..i-1 ----
  for (int i = n - 1; i >= 2; --i) {
    float ss = 0.f;
    for (int k = lane; k < i - 1; k += NL) { const float t = A[i * ld + k]; ss += t * t; }
    ss = cx.sum(ss);
    const float xl = A[i * ld + i - 1];
    if (ss == 0.f) {  // the row is already tridiagonal
      e[i] = xl;
      bet[i] = 0.f;
      continue;
    }
    const float norm = ::sqrtf(ss + xl * xl);
    const float alpha = xl >= 0.f ? -norm : norm;
    const float vl = xl - alpha;
    const float b = 2.0f / (ss + vl * vl);
    for (int k = lane; k < i; k += NL) v[k] = (k == i - 1) ? vl : A[i * ld + k];
    cx.sync();
    for (int j = lane; j < i; j += NL) {  // p = b A v
      float s = 0.f;
      for (int k = 0; k < i; ++k) s += A[j * ld + k] * v[k];
      p[j] = b * s;
    }
    cx.sync();
    float vp = 0.f;
    for (int k = lane; k < i; k += NL) vp += v[k] * p[k];
    vp = cx.sum(vp);
    const float K = 0.5f * b * vp;
    for (int j = lane; j < i; j += NL) p[j] -= K * v[j];  // w = p - K v
    cx.sync();
    for (int j = lane; j < i; j += NL) {  // A -= v w^T + w v^T on the leading i x i block
      const float vj = v[j], wj = p[j];
      for (int k = 0; k < i; ++k) A[j * ld + k] -= vj * p[k] + wj * v[k];
    }
    for (int k = lane; k < i; k += NL) A[i * ld + k] = v[k];  // keep the reflector in the (now unused) row i
    e[i] = alpha;
    bet[i] = b;
    cx.sync();
  }
  GL_STAMP(2);
  e[0] = 0.f;
  e[n] = 0.f;
  if (n > 1) e[1] = A[ld];
  for (int k = lane; k < n; k += NL) d[k] = A[k * ld + k];
  cx.sync();
  // ---- Q = H_{n-1} ... H_2, built from the small end so that only the leading block is touched ----
  for (int j = lane; j < n; j += NL)
    for (int k = 0; k < n; ++k) Z[k * ld + j] = (k == j) ? 1.f : 0.f;
  cx.sync();
  for (int i = 2; i < n; ++i) {
    const float b = bet[i];
    if (b == 0.f) continue;
    for (int j = lane; j < i; j += NL) {
      float g = 0.f;
      for (int k = 0; k < i; ++k) g += A[i * ld + k] * Z[k * ld + j];
      g *= b;
      for (int k = 0; k < i; ++k) Z[k * ld + j] -= g * A[i * ld + k];
    }
  }
  cx.sync();  // columns of Z were owned by lanes above, rows below
  GL_STAMP(3);
  // ---- implicit QL on (d, es); every lane runs the scalar chain, lane k rotates rows k, k + lanes, .. of Z ----
  // es[i] = e[i+1] couples d[i] and d[i+1]; es[n-1] = 0
  cx.load_tridiagonal(d, e + 1, n);
  for (int l = 0; l < n; ++l) {
    int iter = 0;
    for (;;) {
      const int m = cx.first_split(l, n);
      if (m == l || iter++ == 60) break;
      const float el = cx.e(l), dl = cx.d(l);
      float g = (cx.d(l + 1) - dl) / (2.0f * el);
      float r = ::sqrtf(g * g + 1.0f);
      g = cx.d(m) - dl + el / (g + (g >= 0.f ? r : -r));
      float s = 1.f, c = 1.f, pp = 0.f;
      float zh[C::ROWS];  // column i+1 of the lane's rows, carried from rotation to rotation
      float z0[C::ROWS];  // column i, fetched one rotation ahead so that its LDS latency hides behind the scalar chain
#pragma unroll
      for (int t = 0; t < C::ROWS; ++t) {
        const int k = lane + t * NL;
        zh[t] = k < n ? Z[k * ld + m] : 0.f;
        z0[t] = k < n ? Z[k * ld + m - 1] : 0.f;
      }
      int i = m - 1;
      bool under = false;
      for (; i >= l; --i) {
        float zn[C::ROWS];
#pragma unroll
        for (int t = 0; t < C::ROWS; ++t) {
          const int k = lane + t * NL;
          zn[t] = (k < n && i > l) ? Z[k * ld + i - 1] : 0.f;
        }
        const float ei = cx.e(i);
        const float f = s * ei, bb = c * ei;
        const float h2 = f * f + g * g;
        if (h2 == 0.f) {  // both underflowed: deflate here
          cx.set_d(i + 1, cx.d(i + 1) - pp);
          cx.set_e(m, 0.f);
          under = true;
          break;
        }
        const float ri = cx.rsq(h2);
        cx.set_e(i + 1, h2 * ri);
        s = f * ri;
        c = g * ri;
        g = cx.d(i + 1) - pp;
        r = (cx.d(i) - g) * s + 2.0f * c * bb;
        pp = s * r;
        cx.set_d(i + 1, g + pp);
        g = c * r - bb;
#pragma unroll
        for (int t = 0; t < C::ROWS; ++t) {
          const int k = lane + t * NL;
          if (k < n) {
            Z[k * ld + i + 1] = s * z0[t] + c * zh[t];
            zh[t] = c * z0[t] - s * zh[t];
          }
          z0[t] = zn[t];
        }
      }
#pragma unroll
      for (int t = 0; t < C::ROWS; ++t) {  // the carried column: i+1 (== l after a complete pass)
        const int k = lane + t * NL;
        if (k < n) Z[k * ld + i + 1] = zh[t];
      }
      if (under) continue;
      cx.set_d(l, cx.d(l) - pp);
      cx.set_e(l, g);
      cx.set_e(m, 0.f);
    }
  }
  cx.store_diagonal(d, n);
  cx.sync();
}

// coeffs = V diag(pinv) V^T rhs with the relative cutoff of tf.linalg.pinv; g: [n] scratch; scale: what A was divided by
template <class C>
GL_HD void pinv_apply(const C& cx, const float* Z, int n, int ld, const float* d, const float* rhs, float rcond,
                      float inv_scale, float* g, float* coeffs) {
  const int lane = cx.lane(), NL = cx.lanes();
  float m = 0.f;
  for (int k = lane; k < n; k += NL) m = fmaxf(m, ::fabsf(d[k]));
  m = cx.max(m);
  for (int k = lane; k < n; k += NL) {
    float val = 0.f;
    const float lam = d[k];
    if (::fabsf(lam) > rcond * m) {
      float dot = 0.f;
      for (int i = 0; i < n; ++i) dot += Z[i * ld + k] * rhs[i];
      val = dot / lam;
    }
    g[k] = val;
  }
  cx.sync();
  for (int i = lane; i < n; i += NL) {
    float val = 0.f;
    for (int k = 0; k < n; ++k) val += Z[i * ld + k] * g[k];
    coeffs[i] = val * inv_scale;
  }
}

}  // namespace gle
