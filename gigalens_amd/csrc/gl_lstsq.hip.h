// gl_lstsq.hip.h -- the linear-amplitude solve of LensSimulator.lstsq_simulate (tf/simulator.py:226-236):
//   W = 1/err_map,  Y = obs W,  X = stack W  (bs, HW, depth),  coeffs = pinv(X^T X, rcond=1e-6) X^T Y.
// Three kernels on the stack of basis images  S[b][d][p]  the IMG_BASIS pass (+ PSF / pooling) produced:
//   gl_normal_*_kernel   per (sample, pixel chunk): the symmetric normal matrix of the augmented system [X | Y]
//                        (last row/column = X^T Y, Y^T Y); MFMA SYRK, or register accumulators for <= 7 channels
//   gl_eigh_solve_kernel per sample, one wavefront: sum the chunk partials (fixed order), Householder + implicit-QL
//                        eigendecomposition in LDS (gl_eigh.h), pseudo-inverse with the reference's relative cutoff
//   gl_combine_kernel    image = sum_d coeffs_d S_d   (tf/simulator.py:239)
// The SYRK runs on the matrix cores in exact fp32 (v_mfma_f32_16x16x4_f32): same peak as packed fp32 FMA on CDNA4, but
// one operand register per lane instead of LDS-staged 4x4 register tiles, and the vector ALU stays free for the weights.
#pragma once
#include <hip/hip_runtime.h>

#ifdef GL_EIGH_STAMPS  // debug build only: wall-clock stamps of the solve's phases (block 0)
namespace glk { __device__ long long g_eigh_stamps[8]; }
#define GL_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) ::glk::g_eigh_stamps[k] = wall_clock64(); } while (0)
#endif
#include "gl_eigh.h"
#include "gl_kernels.hip.h"

namespace glk {

constexpr int LS_MAXD = 80;   // channels incl. the observation column up to which the single-launch SYRK (<= 5 tile rows) runs
constexpr int LS_LDS_MAXN = 127;  // unknowns whose eigen solve keeps A and V in LDS (2 n (n | 1) floats: 129 KB of the CU's 160)
constexpr int LS_MAXN = 255;      // unknowns served at all: above LS_LDS_MAXN the two matrices live in the workspace (L2)
constexpr int LS_SB = 4;          // tile rows (16 channels each) per super-block of the block-pair SYRK
constexpr int LS_TPP = 32;    // chunk granularity: 2 * LS_TPP = 64 pixels (4 waves x 16-pixel MFMA groups)
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

// ---- many channels: the SYRK on the matrix cores -------------------------------------------------------------------
// C = X^T X with X = [pixels][channels]: v_mfma_f32_16x16x4_f32 takes A = X^T (16 channels x 4 pixels) and B = X
// (4 pixels x 16 channels) in the SAME lane layout -- lane l holds channel (l & 15) of pixel slot (l >> 4) -- so one
// register per 16-channel block serves as the A operand of its tile row and the B operand of its tile column.  A lane
// fetches 4 consecutive pixels of its channel with one 16-byte load straight from the stack (16 lanes x 64 B per
// channel block; no LDS staging), weighs them by 1/err and feeds them as 4 k-steps.  Only the NT (NT+1) / 2 lower
// tiles are accumulated (4 accumulator registers each).  fp32 MFMA: every product is rounded once, sums are k-ordered
// fma chains (the weights 1/err come from v_rcp_f32).  The 4 waves of a workgroup take interleaved 16-pixel groups of the chunk and are summed through LDS.
// (Round-1 history: the packed-fp32 VALU version of this kernel, 4x4 register tiles fed from LDS, ran 2.9 ms on C3L.)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT, bool VEC>  // VEC: every channel row, obs and err sit on a 16-byte pitch
__global__ void __launch_bounds__(256) gl_normal_mfma_kernel(NormalArgs a) {
  constexpr int NTILES = NT * (NT + 1) / 2;
  __shared__ float s_red[NTILES * 256];
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, l = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: group bounds below are wave-uniform
  const int c = l & 15, q = l >> 4;
  const int p0 = chunk * a.chunk, p1 = min(p0 + a.chunk, a.HW);
  const float* S = a.stack + (size_t)b * a.D * a.HW;
  // the lane's channel of block t: a basis row, the observation (channel D) or padding (reads obs, multiplied by 0)
  const float* row[NT];
  float keep[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ch = 16 * t + c;
    row[t] = ch < a.D ? S + (size_t)ch * a.HW : a.obs;
    keep[t] = ch <= a.D ? 1.f : 0.f;
  }
  f32x4 acc[NTILES];
#pragma unroll
  for (int k = 0; k < NTILES; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

  // 4 consecutive pixels from pb.  Whole groups (the wave-uniform common case) load without any per-lane condition, so
  // that the NT + 1 loads of a group are in flight together; the last, partial group of a chunk is guarded per pixel.
  auto fetch_whole = [&](const float* src, int pb) -> float4 {
    if constexpr (VEC) return *reinterpret_cast<const float4*>(src + pb);
    else return float4{src[pb], src[pb + 1], src[pb + 2], src[pb + 3]};
  };
  auto fetch_part = [&](const float* src, int pb) -> float4 {
    float4 v;
    v.x = pb < p1 ? src[pb] : 0.f;
    v.y = pb + 1 < p1 ? src[pb + 1] : 0.f;
    v.z = pb + 2 < p1 ? src[pb + 2] : 0.f;
    v.w = pb + 3 < p1 ? src[pb + 3] : 0.f;
    return v;
  };
  auto load_group = [&](int pg, float4* x, float4& er) {
    const int pb = pg + 4 * q;
    if (pg + 16 <= p1) {  // uniform over the wave
      er = fetch_whole(a.err, pb);
#pragma unroll
      for (int t = 0; t < NT; ++t) x[t] = fetch_whole(row[t], pb);
    } else {
      er = fetch_part(a.err, pb);
#pragma unroll
      for (int t = 0; t < NT; ++t) x[t] = fetch_part(row[t], pb);
    }
  };

  float4 x[NT], xn[NT], er, ern;
  int pg = p0 + 16 * wave;
  if (pg < p1) load_group(pg, x, er);
  for (; pg < p1; pg += 64) {
    const int nxt = pg + 64;
    if (nxt < p1) load_group(nxt, xn, ern);  // in flight while this group multiplies
    const int pb = pg + 4 * q;
    // weights 1/err by v_rcp_f32 (1 ulp; the same w multiplies X and Y): the IEEE division sequence costs ~10 VALU
    // issue slots per pixel beside the MFMAs (1.27 -> 1.21 ms).  Only the last block can contain padding channels.
    float w[4];
    w[0] = pb < p1 ? __builtin_amdgcn_rcpf(er.x) : 0.f;
    w[1] = pb + 1 < p1 ? __builtin_amdgcn_rcpf(er.y) : 0.f;
    w[2] = pb + 2 < p1 ? __builtin_amdgcn_rcpf(er.z) : 0.f;
    w[3] = pb + 3 < p1 ? __builtin_amdgcn_rcpf(er.w) : 0.f;
    float xv[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float k_ = t == NT - 1 ? keep[t] : 1.f;
      xv[t][0] = x[t].x * (t == NT - 1 ? w[0] * k_ : w[0]);
      xv[t][1] = x[t].y * (t == NT - 1 ? w[1] * k_ : w[1]);
      xv[t][2] = x[t].z * (t == NT - 1 ? w[2] * k_ : w[2]);
      xv[t][3] = x[t].w * (t == NT - 1 ? w[3] * k_ : w[3]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int k = 0;
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj, ++k)
          acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[ti][r], xv[tj][r], acc[k], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) x[t] = xn[t];
    er = ern;
  }
  // sum the four waves (fixed order), then write the tiles: element (row 4 (l >> 4) + r, column l & 15) of tile k
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int k = 0; k < NTILES; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int idx = (k * 4 + r) * 64 + l;
          s_red[idx] = (wv == 0 ? 0.f : s_red[idx]) + acc[k][r];
        }
    }
    __syncthreads();
  }
  float* out = a.partial + ((size_t)b * a.n_chunks + chunk) * a.Dp * a.Dp;
  for (int e = tid; e < NTILES * 256; e += 256) {
    const int k = e >> 8, r = (e >> 6) & 3, ll = e & 63;
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= k) ++ti;
    const int tj = k - ti * (ti + 1) / 2;
    const int i = 16 * ti + 4 * (ll >> 4) + r, j = 16 * tj + (ll & 15);
    if (i < a.Dp && j < a.Dp) out[i * a.Dp + j] = s_red[e];
  }
}

// ---- more than LS_MAXD channels: the same SYRK by PAIRS of super-blocks (LS_SB tile rows = 64 channels each) -----------------
// grid.z enumerates the pairs (I, J <= I); a workgroup multiplies super-block I's channels with super-block J's over its pixel
// chunk -- 16 accumulator tiles (10 on the diagonal) instead of NT (NT + 1) / 2, which no longer fits the register file from
// NT = 9 on.  The stack is read once per pair a super-block takes part in (2.5 x on average at four super-blocks).
template <bool VEC>
__global__ void __launch_bounds__(256) gl_normal_pair_kernel(NormalArgs a) {
  constexpr int SB = LS_SB;
  __shared__ float s_red[SB * SB * 256];
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, l = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int I = 0;
  while ((I + 1) * (I + 2) / 2 <= (int)blockIdx.z) ++I;
  const int J = (int)blockIdx.z - I * (I + 1) / 2;
  const bool diag = I == J;
  const int c = l & 15, q = l >> 4;
  const int p0 = chunk * a.chunk, p1 = min(p0 + a.chunk, a.HW);
  const float* S = a.stack + (size_t)b * a.D * a.HW;
  const float* rowA[SB];
  const float* rowB[SB];
  float keepA[SB], keepB[SB];
#pragma unroll
  for (int t = 0; t < SB; ++t) {
    const int ca = 16 * (SB * I + t) + c, cb = 16 * (SB * J + t) + c;
    rowA[t] = ca < a.D ? S + (size_t)ca * a.HW : a.obs;
    keepA[t] = ca <= a.D ? 1.f : 0.f;
    rowB[t] = cb < a.D ? S + (size_t)cb * a.HW : a.obs;
    keepB[t] = cb <= a.D ? 1.f : 0.f;
  }
  f32x4 acc[SB][SB];
#pragma unroll
  for (int i = 0; i < SB; ++i)
#pragma unroll
    for (int j = 0; j < SB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](const float* src, int pb) -> float4 {
    if (VEC && pb + 4 <= p1) return *reinterpret_cast<const float4*>(src + pb);
    float4 v;
    v.x = pb < p1 ? src[pb] : 0.f;
    v.y = pb + 1 < p1 ? src[pb + 1] : 0.f;
    v.z = pb + 2 < p1 ? src[pb + 2] : 0.f;
    v.w = pb + 3 < p1 ? src[pb + 3] : 0.f;
    return v;
  };
  for (int pg = p0 + 16 * wave; pg < p1; pg += 64) {
    const int pb = pg + 4 * q;
    const float4 er = fetch(a.err, pb);
    float4 xa[SB], xb[SB];
#pragma unroll
    for (int t = 0; t < SB; ++t) xa[t] = fetch(rowA[t], pb);
    if (!diag) {
#pragma unroll
      for (int t = 0; t < SB; ++t) xb[t] = fetch(rowB[t], pb);
    }
    float w[4];
    w[0] = pb < p1 ? __builtin_amdgcn_rcpf(er.x) : 0.f;
    w[1] = pb + 1 < p1 ? __builtin_amdgcn_rcpf(er.y) : 0.f;
    w[2] = pb + 2 < p1 ? __builtin_amdgcn_rcpf(er.z) : 0.f;
    w[3] = pb + 3 < p1 ? __builtin_amdgcn_rcpf(er.w) : 0.f;
    float va[SB][4], vb[SB][4];
#pragma unroll
    for (int t = 0; t < SB; ++t) {
      va[t][0] = xa[t].x * (w[0] * keepA[t]); va[t][1] = xa[t].y * (w[1] * keepA[t]);
      va[t][2] = xa[t].z * (w[2] * keepA[t]); va[t][3] = xa[t].w * (w[3] * keepA[t]);
      if (diag) {
#pragma unroll
        for (int r = 0; r < 4; ++r) vb[t][r] = va[t][r];
      } else {
        vb[t][0] = xb[t].x * (w[0] * keepB[t]); vb[t][1] = xb[t].y * (w[1] * keepB[t]);
        vb[t][2] = xb[t].z * (w[2] * keepB[t]); vb[t][3] = xb[t].w * (w[3] * keepB[t]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ti = 0; ti < SB; ++ti)
#pragma unroll
        for (int tj = 0; tj < SB; ++tj)
          if (!diag || tj <= ti) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[ti][r], vb[tj][r], acc[ti][tj], 0, 0, 0);
  }
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int ti = 0; ti < SB; ++ti)
#pragma unroll
        for (int tj = 0; tj < SB; ++tj)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int idx = ((ti * SB + tj) * 4 + r) * 64 + l;
            s_red[idx] = (wv == 0 ? 0.f : s_red[idx]) + acc[ti][tj][r];
          }
    }
    __syncthreads();
  }
  float* out = a.partial + ((size_t)b * a.n_chunks + chunk) * a.Dp * a.Dp;
  for (int e = tid; e < SB * SB * 256; e += 256) {
    const int k = e >> 8, r = (e >> 6) & 3, ll = e & 63;
    const int ti = k / SB, tj = k % SB;
    if (diag && tj > ti) continue;
    const int i = 16 * (SB * I + ti) + 4 * (ll >> 4) + r, j = 16 * (SB * J + tj) + (ll & 15);
    if (i < a.Dp && j < a.Dp) out[i * a.Dp + j] = s_red[e];
  }
}

// ---- per sample: A = sum of partials, eigendecomposition, coeffs = pinv(A_DD, rcond) A_DY -------------------------
// One wavefront per system (gl_eigh.h): Householder tridiagonalisation in LDS; when Sturm counts prove that no
// eigenvalue falls under tf.linalg.pinv's cutoff (singular values <= rcond * max, here |eigenvalues| of the symmetric
// normal matrix) the solve is a tridiagonal LDL^T between two reflector sweeps, otherwise implicit QL with vectors.  The workgroup IS the wave, so
// __syncthreads() is a single-wave barrier.  (The first version of this step was a parallel-ordered cyclic Jacobi
// solve with 1024 threads per system: 3.8 ms for 1024 systems of 66 unknowns, barrier- and LDS-bound; this one does
// ~1/10 of the arithmetic and has no barrier in its longest phase.)
template <int R>  // R registers of 64 lanes hold the tridiagonal: n <= 64 R; a lane owns rows k, k + 64, ... of V
struct WaveCtxN {
  static constexpr int ROWS = R;
  // the tridiagonal, lane-distributed: entry i lives in lane i & 63 of register (i >> 6); uniform reads are
  // v_readlane (a few cycles) instead of an LDS round trip on the critical path of every rotation
  float dr[R], er[R];
  int n_ = 0;
  __device__ __forceinline__ int lane() const { return (int)threadIdx.x; }
  __device__ __forceinline__ int lanes() const { return 64; }
  __device__ __forceinline__ float sum(float v) const { return rl(wave_sum63(v), 63); }
  __device__ __forceinline__ float max(float v) const {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
  }
  __device__ __forceinline__ void sync() const { __syncthreads(); }
  __device__ __forceinline__ float rsq(float x) const { return __builtin_amdgcn_rsqf(x); }
  __device__ __forceinline__ float rcp(float x) const { return __builtin_amdgcn_rcpf(x); }
  __device__ __forceinline__ int first_lane(bool pred) const {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(pred);
    return m ? (int)__builtin_ctzll(m) : -1;
  }
  static __device__ __forceinline__ float rl(float v, int i) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i));
  }
  static __device__ __forceinline__ float wl(float reg, int i, float v) { return (int)threadIdx.x == i ? v : reg; }
  __device__ __forceinline__ void load_tridiagonal(const float* d, const float* es, int n) {
    const int k = (int)threadIdx.x;
    n_ = n;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      dr[r] = k + 64 * r < n ? d[k + 64 * r] : 0.f;
      er[r] = k + 64 * r < n ? es[k + 64 * r] : 0.f;
    }
  }
  __device__ __forceinline__ void store_diagonal(float* d, int n) const {
    const int k = (int)threadIdx.x;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (k + 64 * r < n) d[k + 64 * r] = dr[r];
  }
  // branch-free: every register is read / conditionally written, the index picks
  __device__ __forceinline__ float d(int i) const {
    float v = rl(dr[0], i & 63);
#pragma unroll
    for (int r = 1; r < R; ++r) { const float t = rl(dr[r], i & 63); v = (i >> 6) == r ? t : v; }
    return v;
  }
  __device__ __forceinline__ float e(int i) const {
    float v = rl(er[0], i & 63);
#pragma unroll
    for (int r = 1; r < R; ++r) { const float t = rl(er[r], i & 63); v = (i >> 6) == r ? t : v; }
    return v;
  }
  __device__ __forceinline__ void set_d(int i, float v) {
#pragma unroll
    for (int r = 0; r < R; ++r) dr[r] = wl(dr[r], i - 64 * r, v);
  }
  __device__ __forceinline__ void set_e(int i, float v) {
#pragma unroll
    for (int r = 0; r < R; ++r) er[r] = wl(er[r], i - 64 * r, v);
  }
  // all couplings are tested at once: lane k of register r looks at es[64 r + k] against |d[.]| + |d[. + 1]|, ballots find
  // the first split at or after l
  __device__ __forceinline__ int first_split(int l, int n) const {
    const int k = (int)threadIdx.x;
    unsigned long long m[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float a = fabsf(dr[r]);
      float nx = __shfl_down(a, 1, 64);                     // |d[64 r + k + 1]| for k < 63
      const float first_next = r + 1 < R ? rl(fabsf(dr[r + 1 < R ? r + 1 : r]), 0) : 0.f;
      if (k == 63) nx = first_next;                          // |d[64 (r + 1)]|
      const float sgm = a + nx;
      const bool t = 64 * r + k < n - 1 && (fabsf(er[r]) + sgm == sgm);
      m[r] = __builtin_amdgcn_ballot_w64(t);
    }
    int found = n - 1;
    bool done = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (done || l >= 64 * (r + 1)) continue;               // l lies beyond this register
      const int sh = l > 64 * r ? l - 64 * r : 0;
      const unsigned long long mm = m[r] >> sh;
      if (mm) { found = 64 * r + sh + __builtin_ctzll(mm); done = true; }
    }
    return found;
  }
};
using WaveCtx = WaveCtxN<2>;

// partial[b][0] += partial[b][1..n_chunks-1]  (fixed order), so that the solve reads one matrix per sample
__global__ void __launch_bounds__(256) gl_partial_sum_kernel(float* __restrict__ partial, int n_chunks, int DpDp) {
  const int b = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (e >= DpDp) return;
  float* src = partial + (size_t)b * n_chunks * DpDp + e;
  float v = src[0];
  for (int ch = 1; ch < n_chunks; ++ch) v += src[(size_t)ch * DpDp];
  src[0] = v;
}

// R: registers of the lane-distributed tridiagonal (n <= 64 R).  GLOBAL: A and V live in `mats` ([B][2][n][ld] floats of the
// workspace, L2-resident) instead of LDS -- systems above LS_LDS_MAXN unknowns; the vectors stay in LDS either way.
template <int R, bool GLOBAL>
__global__ void __launch_bounds__(64) gl_eigh_solve_kernel(const float* __restrict__ partial, int n_chunks, int n_sum,
                                                           int D, int Dp, float rcond, float* __restrict__ coeffs,
                                                           float* __restrict__ mats, const int* __restrict__ todo) {
  if (todo && !todo[blockIdx.x]) return;  // gl_chol_solve_kernel proved the cut idle and solved this system
  extern __shared__ float sm[];
  const int n = D, ld = n | 1;
  float* A = GLOBAL ? mats + (size_t)blockIdx.x * 2 * n * ld : sm;  // [n][ld]
  float* Z = A + n * ld;                                             // [n][ld]
  float* d = GLOBAL ? sm : Z + n * ld;                               // [n]
  float* e = d + n;            // [n+1]
  float* v = e + n + 1;        // [n]
  float* p = v + n;            // [n]
  float* bet = p + n;          // [n]
  float* rhs = bet + n;        // [n]
  float* g = rhs + n;          // [n]
  float* y = g + n;            // [n]
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* src = partial + (size_t)b * n_chunks * Dp * Dp;
  GL_STAMP(0);
  // lower triangle (valid in the partials) -> full symmetric A; row D = X^T Y
  for (int i = 0; i < n; ++i)
    for (int j = lane; j <= i; j += 64) {
      float s = 0.f;
      for (int ch = 0; ch < n_sum; ++ch) s += src[(size_t)ch * Dp * Dp + i * Dp + j];
      A[i * ld + j] = s;
      A[j * ld + i] = s;
    }
  for (int i = lane; i < n; i += 64) {
    float s = 0.f;
    for (int ch = 0; ch < n_sum; ++ch) s += src[(size_t)ch * Dp * Dp + D * Dp + i];
    rhs[i] = s;
    bet[i] = 0.f;
  }
  __syncthreads();
  WaveCtxN<R> cx;
  float scale = 0.f;
  for (int k = lane; k < n; k += 64) scale = fmaxf(scale, fabsf(A[k * ld + k]));
  scale = cx.max(scale);
  if (!(scale > 0.f) || !(scale < 3.0e38f)) {  // empty system: the pseudo-inverse of 0 is 0; non-finite input: NaN
    const float out = scale == 0.f ? 0.f : __builtin_nanf("");
    for (int i = lane; i < n; i += 64) coeffs[(size_t)b * D + i] = out;
    return;
  }
  const float inv = 1.0f / scale;
  for (int i = lane; i < n; i += 64)
    for (int k = 0; k < n; ++k) A[i * ld + k] *= inv;
  __syncthreads();
  GL_STAMP(1);
  gle::pinv_solve(cx, A, Z, n, ld, rhs, rcond, inv, d, e, v, p, bet, g, y, coeffs + (size_t)b * D, true);
  GL_STAMP(6);
}

// ---- per sample, first attempt: pinv(A) = inverse(A) when no eigenvalue lies under the cutoff ------------------------------
// tf.linalg.pinv(A, rcond) drops the eigenvalues <= rcond * lambda_max of the symmetric normal matrix (tf/simulator.py:238).  By
// Sylvester's law of inertia, A - mu I has a Cholesky factorisation (all pivots positive) exactly when every eigenvalue of A
// exceeds mu; with mu = 4 rcond ||A||_F >= 4 rcond * lambda_max that proves the cut idle, and the pseudo-inverse is the
// inverse.  The workgroup factorises A - mu I (the proof) and the augmented [A | b] (the solve, root-free LDL^T: the row of b
// becomes L^-1 b on the way) side by side.  256 threads own the lower triangles 16-cyclically in REGISTERS (thread (r, c): rows
// r + 16 ii, columns c + 16 kk); per column j its owners publish the column through LDS -- for [A | b] into the column's own
// slot, which leaves the whole factor in LDS for the back substitution (one wave) -- one barrier, and every thread updates its
// elements.  (First version: both triangles in LDS, read-modify-write per element: 119 us for 1024 systems of 66 unknowns, all
// of it LDS latency; this one 64 us.)  The normal matrices of real fits are far on the safe side (C3L: condition numbers
// 2..24, tools/dev/lstsq_condition_probe.py); systems that fail the proof -- duplicated components, empty bases, non-finite
// input -- set todo[b] and go to gl_eigh_solve_kernel, which decides the cut on converged eigenvalues.
template <int NB>  // 16 NB >= n + 1
__global__ void __launch_bounds__(256) gl_chol_solve_kernel(const float* __restrict__ partial, int n_chunks, int n_sum, int D,
                                                            int Dp, float rcond, float* __restrict__ coeffs,
                                                            int* __restrict__ todo) {
  extern __shared__ float sm[];
  constexpr int ld = 16 * NB + 1;
  const int n = D;
  float* Lc = sm;            // [n][ld]: slot j = column j of the factor, T[i][j] at Lc[j * ld + i] for j <= i <= n (row n: L^-1 b)
  float* c2 = Lc + n * ld;   // [2][ld]: column j of the proof matrix, double-buffered on the parity of j
  float* red = c2 + 2 * ld;  // [4]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tr = tid >> 4, tc = tid & 15;
  const float* src = partial + (size_t)b * n_chunks * Dp * Dp;
  float T1[NB][NB], T2[NB][NB];
  float fro = 0.f;
#pragma unroll
  for (int ii = 0; ii < NB; ++ii)
#pragma unroll
    for (int kk = 0; kk <= ii; ++kk) {
      const int i = 16 * ii + tr, k = 16 * kk + tc;
      float v = 0.f;
      if (k <= i && k < n && i <= n)
        for (int ch = 0; ch < n_sum; ++ch) v += src[(size_t)ch * Dp * Dp + i * Dp + k];
      T1[ii][kk] = v;
      if (i < n) fro = __builtin_fmaf(i == k ? v : 2.f * v, v, fro);
    }
  fro = wave_sum63(fro);
  if (lane == 63) red[wave] = fro;
  __syncthreads();
  const float mu = 4.0f * rcond * sqrtf((red[0] + red[1]) + (red[2] + red[3]));
  bool ok = mu > 0.f && mu < 3.0e38f;  // an empty or non-finite system is the eigenvalue path's business
#pragma unroll
  for (int ii = 0; ii < NB; ++ii)
#pragma unroll
    for (int kk = 0; kk <= ii; ++kk) T2[ii][kk] = T1[ii][kk] - ((ii == kk && tr == tc) ? mu : 0.f);
#pragma unroll
  for (int jb = 0; jb < NB; ++jb) {
    for (int jj = 0; jj < 16 && ok; ++jj) {
      const int j = 16 * jb + jj;
      if (j >= n) break;
      float* cj1 = Lc + j * ld;
      float* cj2 = c2 + (j & 1) * ld;
      if (tc == jj) {
#pragma unroll
        for (int ii = jb; ii < NB; ++ii) {
          const int i = 16 * ii + tr;
          if (i >= j && i <= n) { cj1[i] = T1[ii][jb]; cj2[i] = T2[ii][jb]; }
        }
      }
      __syncthreads();
      const float p1 = cj1[j], p2 = cj2[j];
      ok = p1 > 0.f && p2 > 0.f;  // uniform: every thread reads the same two pivots
      if (!ok) break;
      const float ip1 = 1.0f / p1, ip2 = 1.0f / p2;
      // root-free trailing update with the unscaled column: T[i][k] -= T[i][j] T[k][j] / T[j][j],  j < k <= i.  No guards: a
      // register outside that range (columns already published, the upper halves of the diagonal blocks, rows and columns past
      // the matrix) takes a meaningless update from stale LDS, and is never published or read afterwards.
      float ck1[NB], ck2[NB], ri1[NB], ri2[NB];
#pragma unroll
      for (int q = jb; q < NB; ++q) {
        ck1[q] = cj1[16 * q + tc];
        ck2[q] = cj2[16 * q + tc];
        ri1[q] = cj1[16 * q + tr] * ip1;
        ri2[q] = cj2[16 * q + tr] * ip2;
      }
#pragma unroll
      for (int ii = jb; ii < NB; ++ii)
#pragma unroll
        for (int kk = jb; kk <= ii; ++kk) {
          T1[ii][kk] = __builtin_fmaf(-ri1[ii], ck1[kk], T1[ii][kk]);
          T2[ii][kk] = __builtin_fmaf(-ri2[ii], ck2[kk], T2[ii][kk]);
        }
    }
  }
  __syncthreads();
  if (tid == 0) todo[b] = ok ? 0 : 1;
  if (!ok || wave != 0) return;
  // A = L D L^T with D = the pivots and L[m][i] = T[m][i] / D_i; row n holds w = L^-1 b.  x_i = (w_i - sum_{m>i} T[m][i] x_m) / D_i
  constexpr int R = (16 * NB - 1 + 63) / 64;
  float acc[R], w[R], idv[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = lane + 64 * r < n ? lane + 64 * r : 0;
    acc[r] = 0.f;
    w[r] = Lc[i * ld + n];
    idv[r] = 1.0f / Lc[i * ld + i];
  }
  for (int k = n - 1; k >= 0; --k) {
    float xk = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
      if ((k >> 6) == r) {
        const float mine = (w[r] - acc[r]) * idv[r];
        xk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), k & 63));
      }
    if (lane == 0) coeffs[(size_t)b * D + k] = xk;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = lane + 64 * r;
      if (i < k) acc[r] = __builtin_fmaf(Lc[i * ld + k], xk, acc[r]);
    }
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

// params copy with amplitude column k set to the solved coefficient k (the fitted image of the stack-free path)
__global__ void __launch_bounds__(256) gl_set_amplitudes_kernel(const float* __restrict__ params, int P, int B,
                                                                const int* __restrict__ lin_cols, int D,
                                                                const float* __restrict__ coeffs, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * P) return;
  const int col = i % P, b = i / P;
  float v = params[i];
  for (int k = 0; k < D; ++k)
    if (lin_cols[k] == col) v = coeffs[(size_t)b * D + k];
  out[i] = v;
}

}  // namespace glk
