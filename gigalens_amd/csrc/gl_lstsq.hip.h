// gl_lstsq.hip.h -- the linear-amplitude solve of LensSimulator.lstsq_simulate (tf/simulator.py:226-236):
//   W = 1/err_map,  Y = obs W,  X = stack W  (bs, HW, depth),  coeffs = pinv(X^T X, rcond=1e-6) X^T Y.
// Three kernels on the stack of basis images  S[b][d][p]  the IMG_BASIS pass (+ PSF / pooling) produced:
//   gl_normal_*_kernel   per (sample, pixel chunk): the symmetric normal matrix of the augmented system [X | Y]
//                        (last row/column = X^T Y, Y^T Y), register-tiled packed-fp32 SYRK
//   gl_pinv_solve_kernel per sample: sum the chunk partials (fixed order), parallel-ordered cyclic Jacobi
//                        eigendecomposition in LDS, pseudo-inverse with the reference's relative cutoff, coefficients
//   gl_combine_kernel    image = sum_d coeffs_d S_d   (tf/simulator.py:239)
// fp32 MFMA and packed fp32 FMA have the same peak on CDNA4 (157 TFLOP/s), so the SYRK stays on the vector ALU:
// 4x4 register tiles over pixel PAIRS (v_pk_fma_f32), operands staged through LDS as [pixel pair][channel][2].
#pragma once
#include <hip/hip_runtime.h>

#include "gl_kernels.hip.h"

namespace glk {

constexpr int LS_MAXD = 80;   // channels incl. the observation column (LDS: A and V of the Jacobi solve)
constexpr int LS_TPP = 32;    // pixel pairs per LDS tile
constexpr int LS_SMALL = 8;   // <= this many channels (incl. Y): pixel-parallel kernel with register accumulators

struct NormalArgs {
  const float* stack;  // [B][D][HW]
  const float* obs;    // [HW]
  const float* err;    // [HW]
  int D, Dp;           // basis channels; Dp = D + 1 rounded up to a multiple of 4 (channel D is Y)
  int HW, chunk, n_chunks;
  float* partial;      // [B][n_chunks][Dp*Dp]  (lower triangle valid)
};

// ---- few channels: one thread = strided pixels, all (Dp choose 2) sums in registers ------------------------------
template <int DM>
__global__ void __launch_bounds__(256) gl_normal_small_kernel(NormalArgs a) {
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int p0 = chunk * a.chunk, p1 = min(p0 + a.chunk, a.HW);
  float acc[DM * (DM + 1) / 2];
#pragma unroll
  for (int k = 0; k < DM * (DM + 1) / 2; ++k) acc[k] = 0.f;
  const float* S = a.stack + (size_t)b * a.D * a.HW;
  const int C = a.D + 1;
  for (int p = p0 + tid; p < p1; p += 256) {
    const float w = 1.0f / a.err[p];
    float v[DM];
#pragma unroll
    for (int d = 0; d < DM; ++d) v[d] = (d < a.D) ? S[(size_t)d * a.HW + p] * w : (d == a.D ? a.obs[p] * w : 0.f);
    int k = 0;
#pragma unroll
    for (int i = 0; i < DM; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) acc[k++] += v[i] * v[j];
  }
  __shared__ float red[4][DM * (DM + 1) / 2];
#pragma unroll
  for (int k = 0; k < DM * (DM + 1) / 2; ++k) {
    float s = wave_sum63(acc[k]);
    if ((tid & 63) == 63) red[tid >> 6][k] = s;
  }
  __syncthreads();
  float* out = a.partial + ((size_t)b * a.n_chunks + chunk) * a.Dp * a.Dp;
  for (int k = tid; k < DM * (DM + 1) / 2; k += 256) {
    int i = 0;
    while ((i + 1) * (i + 2) / 2 <= k) ++i;
    const int j = k - i * (i + 1) / 2;
    if (i < C) out[i * a.Dp + j] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
  }
}

// ---- many channels: 4x4 register tiles of the lower triangle, pixel pairs packed --------------------------------
__global__ void __launch_bounds__(256) gl_normal_tiled_kernel(NormalArgs a) {
  extern __shared__ float2 s_x[];  // [LS_TPP][Dp]
  __shared__ float s_w[2 * LS_TPP];  // 1/err of the tile's pixels (0 beyond the chunk)
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int Dp = a.Dp, nt = Dp / 4, ntiles = nt * (nt + 1) / 2;
  int ti = 0, tj = 0;
  const bool active = tid < ntiles;
  if (active) {
    while ((ti + 1) * (ti + 2) / 2 <= tid) ++ti;
    tj = tid - ti * (ti + 1) / 2;
  }
  v2f acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = v2f(0.f);
  const float* S = a.stack + (size_t)b * a.D * a.HW;
  const int p0 = chunk * a.chunk, p1 = min(p0 + a.chunk, a.HW);
  for (int base = p0; base < p1; base += 2 * LS_TPP) {
    __syncthreads();
    if (tid < 2 * LS_TPP) s_w[tid] = (base + tid < p1) ? 1.0f / a.err[base + tid] : 0.f;
    __syncthreads();
    // stage [channel][64 pixels] -> LDS [pixel pair][channel] as (even pixel, odd pixel), weighted by 1/err
    for (int e = tid; e < Dp * 2 * LS_TPP; e += 256) {
      const int d = e / (2 * LS_TPP), q = e - d * (2 * LS_TPP);
      const int p = min(base + q, p1 - 1);
      const float v = d < a.D ? S[(size_t)d * a.HW + p] : (d == a.D ? a.obs[p] : 0.f);
      reinterpret_cast<float*>(s_x)[((q >> 1) * Dp + d) * 2 + (q & 1)] = v * s_w[q];
    }
    __syncthreads();
    if (active) {
#pragma unroll 4
      for (int pp = 0; pp < LS_TPP; ++pp) {
        const float4* ra = reinterpret_cast<const float4*>(s_x + pp * Dp + 4 * ti);
        const float4* rb = reinterpret_cast<const float4*>(s_x + pp * Dp + 4 * tj);
        const float4 a01 = ra[0], a23 = ra[1], b01 = rb[0], b23 = rb[1];
        const v2f av[4] = {v2f{a01.x, a01.y}, v2f{a01.z, a01.w}, v2f{a23.x, a23.y}, v2f{a23.z, a23.w}};
        const v2f bv[4] = {v2f{b01.x, b01.y}, v2f{b01.z, b01.w}, v2f{b23.x, b23.y}, v2f{b23.z, b23.w}};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[r][c] = __builtin_elementwise_fma(av[r], bv[c], acc[r][c]);
      }
    }
  }
  if (active) {
    float* out = a.partial + ((size_t)b * a.n_chunks + chunk) * Dp * Dp;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) out[(4 * ti + r) * Dp + 4 * tj + c] = acc[r][c].x + acc[r][c].y;
  }
}

// ---- per sample: A = sum of partials, eigendecomposition, coeffs = pinv(A_DD, rcond) A_DY -------------------------
// Parallel-ordered cyclic Jacobi (round-robin tournament: n-1 steps of n/2 disjoint rotations per sweep); all n/2
// rotations of a step are applied together: columns (A J, V J), then rows (J^T A).  tf.linalg.pinv cuts singular
// values <= rcond * max (here |eigenvalues| of the symmetric normal matrix).
template <int NT>
__global__ void __launch_bounds__(NT) gl_pinv_solve_kernel(const float* __restrict__ partial, int n_chunks, int D,
                                                            int Dp, float rcond, int sweeps, float* __restrict__ coeffs) {
  extern __shared__ float sm[];
  const int n = (D + 1) & ~1;  // even size for the tournament (a padded row/column of zeros is inert)
  const int ld = n + 1;        // odd row stride: column sweeps hit distinct LDS banks
  float* A = sm;               // [n][ld]
  float* V = A + n * ld;       // [n][ld]
  float* cs = V + n * ld;      // [n/2][2] rotations, then scratch
  float* rhs = cs + n;         // [n]
  int* pr = reinterpret_cast<int*>(rhs + n);  // [n/2][2] pairs
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* src = partial + (size_t)b * n_chunks * Dp * Dp;
  for (int e = tid; e < n * n; e += NT) {
    const int i = e / n, j = e - i * n;
    float v = 0.f;
    if (i < D && j < D) {
      const int hi = max(i, j), lo = min(i, j);
      for (int ch = 0; ch < n_chunks; ++ch) v += src[(size_t)ch * Dp * Dp + hi * Dp + lo];
    }
    A[i * ld + j] = v;
    V[i * ld + j] = (i == j) ? 1.f : 0.f;
  }
  for (int i = tid; i < n; i += NT) {
    float v = 0.f;
    if (i < D)
      for (int ch = 0; ch < n_chunks; ++ch) v += src[(size_t)ch * Dp * Dp + D * Dp + i];  // row D = X^T Y
    rhs[i] = v;
  }
  __syncthreads();
  const int half = n / 2;
  __shared__ int s_rot;
  for (int sw = 0; sw < sweeps; ++sw) {
    if (tid == 0) s_rot = 0;
    __syncthreads();
    for (int r = 0; r < n - 1; ++r) {
      if (tid < half) {
        int p, q;
        if (tid == 0) { p = n - 1; q = r; }
        else { p = (r + tid) % (n - 1); q = (r - tid + (n - 1)) % (n - 1); }
        if (p > q) { int t = p; p = q; q = t; }
        const float app = A[p * ld + p], aqq = A[q * ld + q], apq = A[p * ld + q];
        float c = 1.f, s = 0.f;
        // rotations below fp32 resolution of the two diagonal entries change nothing: skip, and stop sweeping once a
        // whole sweep consisted of such rotations
        if (fabsf(apq) > 3e-8f * sqrtf(fabsf(app * aqq)) && fabsf(apq) > 1e-30f) {
          s_rot = 1;
          const float tau = (aqq - app) / (2.f * apq);
          const float t = (tau >= 0.f ? 1.f : -1.f) / (fabsf(tau) + sqrtf(1.f + tau * tau));
          c = 1.f / sqrtf(1.f + t * t);
          s = t * c;
        }
        cs[2 * tid] = c; cs[2 * tid + 1] = s;
        pr[2 * tid] = p; pr[2 * tid + 1] = q;
      }
      __syncthreads();
      // columns: (x_p, x_q) <- (c x_p - s x_q, s x_p + c x_q) for every row of A and V
      for (int e = tid; e < half * n; e += NT) {
        const int k = e / n, i = e - k * n;
        const float c = cs[2 * k], s = cs[2 * k + 1];
        const int p = pr[2 * k], q = pr[2 * k + 1];
        const float ap = A[i * ld + p], aq = A[i * ld + q];
        A[i * ld + p] = c * ap - s * aq;
        A[i * ld + q] = s * ap + c * aq;
        const float vp = V[i * ld + p], vq = V[i * ld + q];
        V[i * ld + p] = c * vp - s * vq;
        V[i * ld + q] = s * vp + c * vq;
      }
      __syncthreads();
      // rows of A
      for (int e = tid; e < half * n; e += NT) {
        const int k = e / n, j = e - k * n;
        const float c = cs[2 * k], s = cs[2 * k + 1];
        const int p = pr[2 * k], q = pr[2 * k + 1];
        const float ap = A[p * ld + j], aq = A[q * ld + j];
        A[p * ld + j] = c * ap - s * aq;
        A[q * ld + j] = s * ap + c * aq;
      }
      __syncthreads();
    }
    if (!s_rot) break;  // uniform: read after the step's last barrier
  }
  // eigenvalues on the diagonal, eigenvectors in the columns of V
  float* g = cs;  // reuse: g_k = (V^T rhs)_k / lambda_k  or 0
  __shared__ float s_max;
  if (tid == 0) {
    float m = 0.f;
    for (int k = 0; k < D; ++k) m = fmaxf(m, fabsf(A[k * ld + k]));
    s_max = m;
  }
  __syncthreads();
  for (int k = tid; k < n; k += NT) {
    float v = 0.f;
    if (k < D) {
      const float lam = A[k * ld + k];
      if (fabsf(lam) > rcond * s_max) {
        float dot = 0.f;
        for (int i = 0; i < D; ++i) dot += V[i * ld + k] * rhs[i];
        v = dot / lam;
      }
    }
    g[k] = v;
  }
  __syncthreads();
  for (int i = tid; i < D; i += NT) {
    float v = 0.f;
    for (int k = 0; k < D; ++k) v += V[i * ld + k] * g[k];
    coeffs[(size_t)b * D + i] = v;
  }
}

// image[b][p] = sum_d coeffs[b][d] stack[b][d][p]
__global__ void __launch_bounds__(256) gl_combine_kernel(const float* __restrict__ stack, const float* __restrict__ coeffs,
                                                         int D, int HW, float* __restrict__ image) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const float* S = stack + (size_t)b * D * HW;
  const float* c = coeffs + (size_t)b * D;
  float v = 0.f;
  for (int d = 0; d < D; ++d) v += c[d] * S[(size_t)d * HW + p];
  image[(size_t)b * HW + p] = v;
}

// params copy with every amplitude column set to 1 (the basis images carry unit amplitude)
__global__ void __launch_bounds__(256) gl_unit_amplitudes_kernel(const float* __restrict__ params, int P, int B,
                                                                 const int* __restrict__ lin_cols, int D,
                                                                 float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * P) return;
  const int col = i % P;
  float v = params[i];
  for (int k = 0; k < D; ++k)
    if (lin_cols[k] == col) v = 1.f;
  out[i] = v;
}

}  // namespace glk
