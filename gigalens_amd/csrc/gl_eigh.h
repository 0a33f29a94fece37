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
  GL_HD float rcp(float x) const { return 1.0f / x; }
  GL_HD int first_lane(bool pred) const { return pred ? 0 : -1; }  // smallest lane whose predicate holds, or -1
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
// ---- Householder tridiagonalisation, rows n-1 .. 2; reflector i acts on coordinates 0..i-1 and stays in row i of
// A (columns < i) with its factor in bet[i]:  T = Q^T A Q,  Q = H_{n-1} ... H_2;  d = diag(T), e[i] = T[i][i-1] ----
template <class C>
GL_HD void tridiagonalize(C& cx, float* A, int n, int ld, float* d, float* e, float* v, float* p, float* bet) {
  const int lane = cx.lane(), NL = cx.lanes();
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
    for (int j = lane; j < i; j += NL) {  // p = b A v; four independent chains keep the loads in flight
      const float* row = A + j * ld;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int k = 0;
      for (; k + 4 <= i; k += 4) {
        s0 += row[k] * v[k];
        s1 += row[k + 1] * v[k + 1];
        s2 += row[k + 2] * v[k + 2];
        s3 += row[k + 3] * v[k + 3];
      }
      for (; k < i; ++k) s0 += row[k] * v[k];
      p[j] = b * ((s0 + s1) + (s2 + s3));
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
      float* row = A + j * ld;
      int k = 0;
      for (; k + 8 <= i; k += 8) {  // loads of a group before its stores: the row does not alias v / p
        float a_[8], p_[8], v_[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { a_[u] = row[k + u]; p_[u] = p[k + u]; v_[u] = v[k + u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) row[k + u] = a_[u] - (vj * p_[u] + wj * v_[u]);
      }
      for (; k < i; ++k) row[k] -= vj * p[k] + wj * v[k];
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
}

// ---- Q = H_{n-1} ... H_2 explicitly, built from the small end so that only the leading block is touched ----
template <class C> GL_HD void accumulate_q(C& cx, const float* A, float* Z, int n, int ld, const float* bet) {
  const int lane = cx.lane(), NL = cx.lanes();
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
}

// ---- implicit QL on (d, es); every lane runs the scalar chain, lane k rotates rows k, k + lanes, .. of Z ----
// es[i] = e[i+1] couples d[i] and d[i+1]; es[n-1] = 0.  On return d holds the eigenvalues, Z's columns the vectors.
template <class C> GL_HD void ql_implicit(C& cx, float* Z, int n, int ld, float* d, const float* e) {
  const int lane = cx.lane(), NL = cx.lanes();
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

// ---- the well-conditioned short cut ---------------------------------------------------------------------------------
// When no eigenvalue is cut, pinv(A) b = A^{-1} b = Q T^{-1} Q^T b: no eigenvectors are needed, only the proof that the
// spectrum of T lies above the cutoff.  Sturm counts give it: the largest eigenvalue by multi-section (every lane
// counts at its own shift), then ONE count at  rcond * lambda_max:  zero eigenvalues below it means T is positive
// definite with nothing to cut, and the solve is a tridiagonal LDL^T between two sweeps of the reflectors.

// number of eigenvalues of T below x (T = tridiag(d, e), e[i] couples i-1 and i)
template <class C> GL_HD int sturm_count(const C& cx, const float* d, const float* e, int n, float x) {
  float q = d[0] - x;
  int cnt = q < 0.f;
  for (int i = 1; i < n; ++i) {
    if (::fabsf(q) < 1e-30f) q = -1e-30f;
    q = (d[i] - x) - e[i] * e[i] * cx.rcp(q);
    cnt += q < 0.f;
  }
  return cnt;
}

// an upper bound of the largest eigenvalue, tight to ~1e-4 relative (of the Gershgorin span)
template <class C> GL_HD float largest_eigenvalue(const C& cx, const float* d, const float* e, int n) {
  const int lane = cx.lane(), NL = cx.lanes();
  float lo = 3.0e38f, hi = -3.0e38f;
  for (int i = lane; i < n; i += NL) {
    const float r = ::fabsf(e[i]) + ::fabsf(e[i + 1]);  // e[0] = e[n] = 0
    lo = ::fminf(lo, d[i] - r);
    hi = ::fmaxf(hi, d[i] + r);
  }
  lo = -cx.max(-lo);
  hi = cx.max(hi);
  const float span = hi - lo;
  hi += 1e-6f * span;  // strictly above the spectrum: count(hi) == n
  for (int round = 0; round < 32 && hi - lo > 1e-4f * span; ++round) {
    const float step = (hi - lo) / (float)(NL + 1);
    const float x = lo + step * (float)(lane + 1);
    const int j = cx.first_lane(sturm_count(cx, d, e, n, x) == n);  // smallest shift already above everything
    if (j < 0) lo = lo + step * (float)NL;
    else { hi = lo + step * (float)(j + 1); lo = lo + step * (float)j; }
  }
  return hi;
}

// y = Q^T y (forward = true) or y = Q y, Q = H_{n-1} ... H_2 with the reflectors in the rows of A; y [n] shared
template <class C> GL_HD void apply_reflectors(const C& cx, const float* A, int n, int ld, const float* bet, float* y, bool transpose) {
  const int lane = cx.lane(), NL = cx.lanes();
  for (int s = 0; s < n - 2; ++s) {
    const int i = transpose ? n - 1 - s : 2 + s;
    const float b = bet[i];
    if (b == 0.f) continue;
    float dot = 0.f;
    for (int k = lane; k < i; k += NL) dot += A[i * ld + k] * y[k];
    dot = b * cx.sum(dot);
    for (int k = lane; k < i; k += NL) y[k] -= dot * A[i * ld + k];
    cx.sync();
  }
}

// T y = c for a positive-definite tridiagonal T (LDL^T, no pivoting needed); c is overwritten with y; w [n] scratch.
// Every lane runs the same chain (uniform values, uniform writes).
GL_HD void tridiagonal_spd_solve(const float* d, const float* e, int n, float* c, float* w) {
  w[0] = d[0];
  for (int i = 1; i < n; ++i) {
    const float m = e[i] / w[i - 1];
    w[i] = d[i] - m * e[i];
    c[i] -= m * c[i - 1];
  }
  c[n - 1] = c[n - 1] / w[n - 1];
  for (int i = n - 2; i >= 0; --i) c[i] = (c[i] - e[i + 1] * c[i + 1]) / w[i];
}

// coeffs = pinv(A, rcond) rhs for the symmetric A [n][ld] (already divided by 1/inv_scale; destroyed).
// Z [n][ld] and the [n]-sized scratch arrays d, v, p, bet, g, y and e [n+1] live in the same (LDS) space as A.
// Returns 1 when the short cut was taken (diagnostics / tests).
template <class C>
GL_HD int pinv_solve(C& cx, float* A, float* Z, int n, int ld, const float* rhs, float rcond, float inv_scale, float* d,
                     float* e, float* v, float* p, float* bet, float* g, float* y, float* coeffs, bool allow_shortcut) {
  const int lane = cx.lane(), NL = cx.lanes();
  tridiagonalize(cx, A, n, ld, d, e, v, p, bet);
  GL_STAMP(2);
  if (allow_shortcut) {
    const float lmax = largest_eigenvalue(cx, d, e, n);
    // 4x margin: a spectrum that comes within a factor 4 of the cutoff is left to the eigenvalue path, whose cut
    // decision is taken on converged eigenvalues (the Sturm count uses an approximate reciprocal and a bound of lmax)
    if (lmax > 0.f && sturm_count(cx, d, e, n, 4.0f * rcond * lmax) == 0) {
      for (int k = lane; k < n; k += NL) y[k] = rhs[k];
      cx.sync();
      apply_reflectors(cx, A, n, ld, bet, y, true);
      tridiagonal_spd_solve(d, e, n, y, v);
      cx.sync();
      apply_reflectors(cx, A, n, ld, bet, y, false);
      for (int k = lane; k < n; k += NL) coeffs[k] = y[k] * inv_scale;
      return 1;
    }
  }
  accumulate_q(cx, A, Z, n, ld, bet);
  GL_STAMP(3);
  ql_implicit(cx, Z, n, ld, d, e);
  pinv_apply(cx, Z, n, ld, d, rhs, rcond, inv_scale, g, coeffs);
  return 0;
}

}  // namespace gle
