// gl_shp.hip.h -- the specialised kernel for "lenses | [Sersic lens lights] | ONE shapelet source" models (BASELINE config 3 and
// the reference's shapelets-demo model; tf/profiles/light/shapelets.py:20-85).
//
// Round-3 redesign of the shapelet render / VJP (the scalar one-pixel-per-thread kernel it replaces spent 574 VALU
// instructions per pixel, 69 accumulator registers per lane and ran at 2 waves per SIMD):
//   * lens side in pixel-PAIR form (gl_vec.hip.h): the ray-shoot, the chi^2 terms and the lens VJP issue as packed fp32.
//   * shapelet side one pixel at a time, packed over the pixel's TWO COORDINATES: (X_n(u), Y_n(v)) advance through the
//     three-term recurrence in one packed instruction stream.  Table mode generates the values at the two bracketing nodes of
//     the reference's 6000-node grid the same way (ShpNodeGen below) instead of loading them.
//   * ONE contraction serves the value and both derivatives: with s_n2 = sum_n1 a(n1,n2) X_n1 and s'_n2 the same sum over
//     the derivative basis X'_n1,   S = sum Y s,  dS/du = sum Y s',  dS/dv = sum Y' s.   Row n1 of the contraction runs as soon
//     as order n1 exists (X_n1 is a broadcast half of a register pair, the row two n2 per register from the workgroup's LDS copy
//     of the zero-padded square matrix shapelets_prep writes behind the triangle): 2 x 36 packed FMAs + 33 scalar ones per
//     pixel where the separable form of round 2 spent 242 scalar ones.
//   * the amplitude gradient  G(n1,n2) = sum_pixels gS X_n1 Y_n2  is a rank-1 update per pixel -- a GEMM with the pixels as
//     the contraction index.  It runs on the matrix pipe in exact fp32 (v_mfma_f32_16x16x4_f32: same arithmetic as an FMA
//     chain): each wave parks (X_n, Y_n) of its pixels in LDS, one plane [pixel][2] per order (conflict-free 8-byte stores),
//     and reads them back transposed -- lane (m, k) = order m of pixel 4 kb + k, 4-byte reads, plane stride = 4 mod 32 banks --
//     as the A and B operands of one MFMA per four pixels.  G lives in FOUR accumulator registers per lane instead of 66
//     VGPRs, and the 66-value epilogue reduction of the scalar kernel disappears (one 16 x 16 tile per wave, summed over the
//     four waves in fixed order).
// Bitwise reproducible like every other kernel of the path: no atomics, fixed summation order.
#pragma once
#include "gl_pair.hip.h"

namespace glk {

typedef float v4f __attribute__((ext_vector_type(4)));

// dwords per order-pair plane of a wave's exchange buffer: 128 pixels (two per lane) x 2 + 4 pad: stride = 4 (mod 32 banks)
constexpr int SHX_PLANE = 128 * 2 + 4;
// per wave: the order planes, gS of the 128 slots, and the compaction buffers of table mode -- (u, v) of the live pixels in rank
// order [128][2] and their (S, dS/du, dS/dv, -) on the way back [128][4]
constexpr int SHX_GS = 128, SHX_CIN = 256, SHX_COUT = 512;
__host__ __device__ constexpr int shp_exchange_floats(int np) { return 2 /*X, Y*/ * np * SHX_PLANE + SHX_GS + SHX_CIN + SHX_COUT; }
__host__ __device__ constexpr size_t shp_exchange_bytes(int np) { return (size_t)4 /*waves*/ * shp_exchange_floats(np) * sizeof(float); }

template <int NP> struct ShpPix { float S, Su, Sv, u, v, dx, dy, fac; };  // what a pixel's VJP needs of its forward pass

// ---- the bases of one pixel, (u, v)-packed: val[n] = (X_n(u), Y_n(v)), two coordinates per packed instruction -------------------
// The orders are produced one at a time -- value() = (X_n(u), Y_n(v)), slope() = (X'_n, Y'_n), then advance(n) -- so that the
// caller consumes each (parks it, runs its row of the contraction) instead of holding 2 x 2 x 12 basis registers.

// Table mode (shapelets.py:39-40, 55-62: tfp.math.interp_regular_1d_grid over phi_n(linspace(-5, 5, 6000)), fill 0 outside).
// The node values are not LOADED, they are generated: phi_0 at the node below by one exponential, the higher orders by the
// normalised three-term recurrence -- and the DIFFERENCE to the next node by its own recurrence
//     D_{n+1} = a_n (u0 + h) D_n - b_n D_{n-1} + a_n h V_n,      D_0 = V_0 expm1(-h (u0 + h / 2)),
// so that the slope of the interpolant (what the position gradients see) carries no cancellation: both are good to ~1e-6 of
// the basis amplitude, the distance between the reference's own float32 nodes and exact ones (checked against the float64
// table over all 6000 nodes in tests/).  Round 3 first gathered [values | differences] rows of a 576 KB table, 96 bytes per
// coordinate and lane: 120 M vector-L1 accesses per C3 launch = 72 % of the L1's 64 bytes per cycle, with the VALU a third busy.
// value() = the interpolant, slope() = its difference per node spacing; a pixel with either coordinate outside the table has
// every basis value zero (V_0 = 0 propagates), which is what fill 0 x anything gives.  Orders above the model's n_max are
// generated too: their amplitudes are zeros of the padded matrix, their gradient entries and normal-matrix channels never stored.
struct ShpNodeGen {
  static constexpr float top = (float)(SH_NODES - 1), h = 10.f / top;
  v2f u0, u1, t, Vp, Dp, Vc, Dc;
  __device__ __forceinline__ void init(float u, float v) {
    const v2f fi = (v2f{u, v} + 5.f) * (top / 10.f);
    const bool live = (fi.x >= 0.f) && (fi.x <= top) && (fi.y >= 0.f) && (fi.y <= top);  // NaN: not live
    const v2f fic = v2f{clamp_(fi.x, 0.f, top), clamp_(fi.y, 0.f, top)};
    const v2f fb = v2f{fmin_(floor_(fic.x), top - 1.f), fmin_(floor_(fic.y), top - 1.f)};  // the last node belongs to the last interval (t = 1)
    t = fic - fb;
    u0 = (fb - 0.5f * top) * h;  // node below: (i - 2999.5) h, symmetric about 0
    u1 = u0 + h;
    const v2f e0 = exp2_(u0 * u0 * (float)(-0.5 * kLog2e));
    const v2f z = __builtin_elementwise_fma(u0, v2f(-h), v2f(-0.5f * h * h));  // -(u1^2 - u0^2) / 2 with u1 = u0 + h; |z| < 8.4e-3
    v2f em1 = __builtin_elementwise_fma(z, v2f(1.f / 6.f), v2f(0.5f));
    em1 = __builtin_elementwise_fma(z, em1, v2f(1.f)) * z;  // expm1(z) to 2e-10
    Vp = v2f(0.f);
    Dp = v2f(0.f);
    Vc = live ? e0 * 0.75112554446494248286f : v2f(0.f);
    Dc = Vc * em1;
  }
  __device__ __forceinline__ v2f value() const { return __builtin_elementwise_fma(t, Dc, Vc); }
  __device__ __forceinline__ v2f slope() const { return Dc; }
  // order n -> n + 1 (n a compile-time constant after unrolling), in the monic scaling phi_n = SH_K[n] P_n:
  //   P_{n+1} = u0 P_n - (n / 2) P_{n-1},      Q_{n+1} = (u0 + h) Q_n - (n / 2) Q_{n-1} + h P_n      (Q: difference to the next node)
  __device__ __forceinline__ void advance(int n) {
    const float bn = 0.5f * (float)n;
    const v2f vn = n == 0 ? u0 * Vc : __builtin_elementwise_fma(u0, Vc, -(Vp * bn));
    v2f dn = n == 0 ? u1 * Dc : __builtin_elementwise_fma(u1, Dc, -(Dp * bn));
    dn = __builtin_elementwise_fma(v2f(h), Vc, dn);
    Vp = Vc; Dp = Dc;
    Vc = vn; Dc = dn;
  }
};

// direct mode (shapelets.py:67-85): the Hermite polynomial part without the Gaussian, same monic scaling; X'_n = sqrt(2n) X_{n-1}
// becomes P'_n = n P_{n-1}
struct ShpDirectGen {
  v2f uv, Vp, Vc, Dc;
  __device__ __forceinline__ void init(float u, float v) {
    uv = v2f{u, v};
    Vp = v2f(0.f);
    Vc = v2f(0.75112554446494248286f);
    Dc = v2f(0.f);
  }
  __device__ __forceinline__ v2f value() const { return Vc; }
  __device__ __forceinline__ v2f slope() const { return Dc; }
  __device__ __forceinline__ void advance(int n) {  // P_{n+1} = u P_n - (n / 2) P_{n-1},  P'_{n+1} = (n + 1) P_n
    const v2f vn = n == 0 ? uv * Vc : __builtin_elementwise_fma(uv, Vc, -(Vp * (0.5f * (float)n)));
    Dc = Vc * (float)(n + 1);
    Vp = Vc;
    Vc = vn;
  }
};
template <bool INTERP> using ShpGen = std::conditional_t<INTERP, ShpNodeGen, ShpDirectGen>;

template <int NP> __device__ __forceinline__ void shp_pixel_coords(const float* d, float px, float py, ShpPix<NP>& st) {
  const float ib = d[SHP_IB];
  st.dx = px - d[SHP_CX];
  st.dy = py - d[SHP_CY];
  st.u = st.dx * ib;
  st.v = st.dy * ib;
}
// the table's support, with the expression shp_basis_nodes itself uses: outside it a basis and its slope are exactly zero
__device__ __forceinline__ bool shp_in_table(float u) {
  const float fi = (u + 5.f) * ((float)(SH_NODES - 1) / 10.f);
  return (fi >= 0.f) && (fi <= (float)(SH_NODES - 1));
}

// ---- forward of one pixel ----------------------------------------------------------------------------------------------------
//   gA : the sample's zero-padded square amplitude matrix [2NP][2NP] in GLOBAL memory (wave-uniform address: scalar loads)
//   buf: this lane's slot of the wave's exchange planes (plane n = (X_n, Y_n) of every pixel): parked there for the MFMA pass
template <int NP, bool INTERP, bool GRAD>
__device__ __forceinline__ float shp_pixel_fwd(const float* d, const v2f* __restrict__ gA, float* __restrict__ buf, ShpPix<NP>& st) {
  constexpr int NO = 2 * NP;
  // s_n2 = sum_n1 a(n1, n2) X_n1 (and s'_n2 with X'_n1), two n2 per register; rows beyond the triangle are zero and skipped.  Row
  // n1 of the contraction runs as soon as order n1 exists; of the basis only (Y_n, Y'_n) is kept for the sums over n2.
  // (The 72 matrix entries are scalar loads the compiler issues together; with the kernel's pointers that is more SGPRs than a
  // wave has, so some pointers are parked in VGPR lanes: ~60 v_readlane per tile.  Splitting the rows into dependent groups, a
  // scheduling barrier, or a laundered pointer all made the allocation worse: measured, see DESIGN.md.)
  v2f s[NP], sd[NP], yk[NO - 1];
#pragma unroll
  for (int j = 0; j < NP; ++j) { s[j] = v2f(0.f); sd[j] = v2f(0.f); }
  // Row n1 + 1 of the matrix is read (LDS broadcast) while row n1 is in use, and no earlier: its address carries an opaque zero
  // derived from order n1 (left alone, the compiler issues all 72 reads first and the kernel no longer fits its registers).
  auto load_row = [&](int n1, float dep, v2f (&row)[NP]) {
    int z = 0;
    if constexpr (INTERP) asm("v_and_b32 %0, 0, %1" : "=v"(z) : "v"(dep));  // (direct mode: the compiler's own order fits)
    const v2f* __restrict__ p = reinterpret_cast<const v2f*>(reinterpret_cast<const char*>(gA) + z) + n1 * NP;
#pragma unroll
    for (int j = 0; 2 * j + n1 < NO - 1; ++j) row[j] = p[j];
  };
  ShpGen<INTERP> gen;
  gen.init(st.u, st.v);
  v2f row[2][NP];  // (two rows ahead: no faster)
  load_row(0, st.u, row[0]);
#pragma unroll
  for (int n1 = 0; n1 < NO - 1; ++n1) {
    const v2f W = gen.value(), D = gen.slope();
    if (n1 + 1 < NO - 1) load_row(n1 + 1, W.x, row[(n1 + 1) & 1]);
    if constexpr (GRAD) *reinterpret_cast<v2f*>(buf + n1 * SHX_PLANE) = W;  // parked for the MFMA pass right away
    yk[n1] = v2f{W.y, D.y};
#pragma unroll
    for (int j = 0; 2 * j + n1 < NO - 1; ++j) {  // n2 = 2j, 2j+1 with n1 + n2 <= 2 NP - 2 (the largest n_max this NP serves)
      const v2f arow = row[n1 & 1][j];
      s[j] = __builtin_elementwise_fma(v2f(W.x), arow, s[j]);
      if (GRAD) sd[j] = __builtin_elementwise_fma(v2f(D.x), arow, sd[j]);
    }
    if (n1 + 1 < NO - 1) gen.advance(n1);
  }
  if constexpr (GRAD) *reinterpret_cast<v2f*>(buf + (NO - 1) * SHX_PLANE) = v2f(0.f);  // the pad order (NO = 2 NP serves n_max <= NO - 2)
  st.fac = INTERP ? 1.f : exp_(-(st.u * st.u + st.v * st.v) * 0.5f);  // shapelets.py:70
  v2f SSv = v2f(0.f);  // (S, dS/dv): (Y_n2, Y'_n2) against s_n2
  float Su = 0.f;
#pragma unroll
  for (int n2 = 0; n2 < NO - 1; ++n2) {
    const float sn = (n2 & 1) ? s[n2 >> 1].y : s[n2 >> 1].x;
    SSv = __builtin_elementwise_fma(yk[n2], v2f(sn), SSv);
    if (GRAD) {
      const float sdn = (n2 & 1) ? sd[n2 >> 1].y : sd[n2 >> 1].x;
      Su = __builtin_fmaf(yk[n2].x, sdn, Su);
    }
  }
  const float S = SSv.x, Sv = SSv.y;
  st.S = S;
  if (GRAD) { st.Su = Su; st.Sv = Sv; }
  return st.fac * st.S;
}

// ---- table mode: pixels PROVABLY outside the shapelet table, decided before the lens is evaluated -----------------------------
// The table's support is |u|, |v| <= 5 with (u, v) = (beta - c) / beta_s: a pixel whose source-plane position is further than
// 5 sqrt(2) beta_s from the source centre renders exactly zero with zero slopes, and nothing of its lens evaluation reaches any
// output (image value 0, cotangent 0).  beta = x - alpha(x), so |beta - c| >= |x - c| - |alpha(x)|, and |alpha| has a bound that
// needs no series:
//   EPL   alpha = P Omega,  P = 2b/(1+q) (b/R)^(t-1),  |Omega| <= sum |c_n| <= 1/(1 - f) = (1+q)/(2q)  (|c_n| <= f^n for 0 < t < 2)
//         t >= 1:  (b/R)^(t-1) <= 1 where the elliptical radius R >= b, which |x - c_lens| >= b/q guarantees:  |alpha| <= b/q;
//         t <  1:  (R/b)^(1-t) <= (D/b)^(1-t) with D the largest distance of a pixel from the lens centre:   |alpha| <= (b/q)(D/b)^(1-t)
//   Shear |alpha| = |gamma| |x| <= |gamma| r_max;   SIS |alpha| = theta_E;   SIE |alpha| <= A hypot(pi/2, atanh(sqrt(1-q^2)))
// (r_max: MainArgs.grid_rmax).  A wave-tile ALL of whose 128 pixels pass  |x - c|^2 > (1.001 (5 sqrt2 beta_s + sum of bounds))^2
// skips the lens, the chains and the VJPs: on the C3 prior (theta_E ~ 1.25", beta_s ~ 0.1", an 8.3" field, a wave-tile = one image
// row) that is every row further than ~2.5" from the source centre, 40 % of the wave-tiles.  Parameters outside the ranges the
// bounds assume (q, b, t, beta_s; NaN) switch the test off for the sample.
template <int NL> struct ShpCull {
  float T2, cx, cy;                                        // threshold^2 (+inf: off), source centre
  float lx[NL > 0 ? NL : 1], ly[NL > 0 ? NL : 1], rb2[NL > 0 ? NL : 1];  // per lens: centre and the squared radius its bound holds outside of (0: everywhere)
  __device__ __forceinline__ bool tile_outside(v2f x, v2f y) const {
    const v2f dx = x - cx, dy = y - cy, d2 = dx * dx + dy * dy;
    bool far = d2.x > T2 && d2.y > T2;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const v2f ex = x - lx[i], ey = y - ly[i], e2 = ex * ex + ey * ey;
      far = far && e2.x >= rb2[i] && e2.y >= rb2[i];
    }
    return __builtin_amdgcn_ballot_w64(!far) == 0ull;
  }
};
template <class LK, bool INTERP, int NL = LK::n>
__device__ __forceinline__ ShpCull<NL> shp_cull_setup(const float* const* dL, const float* dS, float r_max) {
  ShpCull<NL> c;
  c.T2 = __builtin_inff();
  c.cx = c.cy = 0.f;
#pragma unroll
  for (int i = 0; i < (NL > 0 ? NL : 1); ++i) c.lx[i] = c.ly[i] = c.rb2[i] = 0.f;
  if constexpr (INTERP) {
    const float ib = dS[SHP_IB];
    bool ok = r_max >= 0.f && ib > 0.f;
    float A = 0.f;
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int kind = LK::kinds[i];
      const float* d = dL[i];
      if constexpr (kind == K_EPL) {
        const float q = d[EPL_Q], b = d[EPL_B], tm1 = d[EPL_TM1];
        ok = ok && q > 0.f && q <= 1.f && b > 0.f && tm1 > -1.f && tm1 < 1.f;
        const float bq = b / q, D = r_max + __builtin_sqrtf(d[EPL_CX] * d[EPL_CX] + d[EPL_CY] * d[EPL_CY]);
        const float grow = tm1 < 0.f ? exp2_(-tm1 * log2_(__builtin_fmaxf(D / b, 1.f))) : 1.f;
        A += bq * grow;
        c.lx[i] = d[EPL_CX];
        c.ly[i] = d[EPL_CY];
        c.rb2[i] = tm1 > 0.f ? 1.002f * bq * bq : 0.f;
      } else if constexpr (kind == K_SHEAR) {
        A += __builtin_sqrtf(d[SHR_G1] * d[SHR_G1] + d[SHR_G2] * d[SHR_G2]) * r_max;
      } else if constexpr (kind == K_SIE) {
        const float sq = d[SIE_SQ];
        ok = ok && sq >= 0.f && sq < 1.f;
        const float ah = 0.5f * __builtin_logf((1.f + sq) / (1.f - sq));
        A += __builtin_fabsf(d[SIE_A]) * __builtin_sqrtf(2.4674011f + ah * ah);
      } else {
        A += __builtin_fabsf(d[SIS_TE]);
      }
    }, std::make_integer_sequence<int, NL>{});
    const float T = 1.001f * (7.0710678f / ib + A);
    if (ok && T > 0.f && T < 1e15f) c.T2 = T * T;
    c.cx = dS[SHP_CX];
    c.cy = dS[SHP_CY];
  }
  return c;
}

// ---- the kernel -------------------------------------------------------------------------------------------------------------
// Component list: LK lenses, LLK lens lights (Sersic kinds), then exactly one K_SHAPELETS source.  Every thread owns pixels
// (j, j + 256) of a 512-pixel tile; the two shapelet chains of a lane run one after the other.
// RAGGED = false: the launch site guarantees whole 512-pixel tiles, no mask and no pixel list (every BASELINE config) -- the kernel
// then carries no ragged-end tile code at all (beside the steady-state body that instantiation does not fit 256 registers).
template <int MODE, int WAVES, class LK, class LLK, int NP, bool INTERP, bool RAGGED>
__global__ void __launch_bounds__(WG, WAVES) gl_shp_kernel(MainArgs a) {
  using V = v2f;
  constexpr int NL = LK::n, NLL = LLK::n;
  constexpr bool GRAD = (MODE == IMG_BWD || MODE == LL_GRAD);
  extern __shared__ float smem[];
  float* s_d = smem;
  float* s_x = smem + ((a.D + 3) & ~3);  // exchange planes (gradient modes); the epilogue's rows alias them
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = a.order ? a.order[blockIdx.y] : blockIdx.y, chunk = blockIdx.x;
  const CompDesc* __restrict__ comps = a.comps;
  const float* __restrict__ gder = a.derived + (size_t)b * a.D;
  for (int i = tid; i < a.D; i += WG) s_d[i] = gder[i];
  __syncthreads();
  constexpr int NACC_L = [] { int n = 0; for (int i = 0; i < NL; ++i) n += static_nacc(LK::kinds[i]); return n; }();
  constexpr int NACC_C = [] { int n = 0; for (int i = 0; i < NLL; ++i) n += static_nacc(LLK::kinds[i]); return n; }();
  V accL[NACC_L > 0 ? NACC_L : 1];
  V accC[NACC_C > 0 ? NACC_C : 1];
#pragma unroll
  for (int k = 0; k < NACC_L; ++k) accL[k] = V(0.f);
#pragma unroll
  for (int k = 0; k < NACC_C; ++k) accC[k] = V(0.f);
  V st0 = V(0.f), st1 = V(0.f);
  V acc_cx = V(0.f), acc_cy = V(0.f), acc_ib = V(0.f);  // shapelet centre / 1/beta sums (pixel pair)
  v4f G = {0.f, 0.f, 0.f, 0.f}, G2 = {0.f, 0.f, 0.f, 0.f};  // this wave's 16 x 16 tile of the amplitude gradient (even / odd pixel groups)
  const float* dL[NL > 0 ? NL : 1];
  const float* dC[NLL > 0 ? NLL : 1];
#pragma unroll
  for (int i = 0; i < NL; ++i) dL[i] = s_d + comps[i].d_off;
#pragma unroll
  for (int i = 0; i < NLL; ++i) dC[i] = s_d + comps[NL + i].d_off;
  const CompDesc shp = comps[NL + NLL];
  const float* dS = s_d + shp.d_off;
  // the zero-padded square amplitude matrix, from the workgroup's LDS copy of the sample's constants: uniform addresses, so every
  // read is a broadcast (as 72 vector registers filled by global loads it was a third of the kernel's L1 traffic)
  const v2f* __restrict__ gA = INTERP ? reinterpret_cast<const v2f*>(dS + SHP_SQ) : reinterpret_cast<const v2f*>(gder + shp.d_off + SHP_SQ);
  const bool has_err = a.err != nullptr, has_mask = a.mask != nullptr, has_pix = a.pix != nullptr;
  const ShpCull<NL> cull = shp_cull_setup<LK, INTERP>(dL, dS, a.grid_rmax);
  // exchange planes of this wave: plane n = (X_n, Y_n) of each pixel; a lane parks pixel slot w at pixel index 64 w + lane
  float* xw = s_x + wave * shp_exchange_floats(NP);
  float* wr_xy = xw + 2 * lane;
  // transposed read: lane (m, k) = (lane & 15, lane >> 4) takes order m of pixel 4 kb + k; orders beyond 2 NP - 1 re-read the last
  // (their rows / columns of the tile are never stored)
  const int mm = min(lane & 15, 2 * NP - 1);
  const float* rd_gx = xw + mm * SHX_PLANE + 2 * (lane >> 4);
  const float* rd_y = rd_gx + 1;
  float* wr_gs0 = xw + 2 * NP * SHX_PLANE;                  // gS of the pixel with rank k at slot k (direct mode: 64 w + lane)
  const float* rd_gs = xw + 2 * NP * SHX_PLANE + (lane >> 4);  // pixel 4 kb + k of the operand lane (m, k)

  const int p0 = chunk * a.chunk;
  const int p1 = min(p0 + a.chunk, a.N);
  // Pixel of slot w (0, 1) of this lane in the workgroup step that starts at `base` (whole tiles).  Linear: base + 256 w + tid.
  // Blocked (a.blk_w = image width, table mode on a whole image): the step's four wave-tiles are the image's 8-row x 16-column
  // blocks 4 (base / 512) + wave in row-major block order, slot w = rows 4 w .. 4 w + 3 of the block.  The support of the shapelet
  // table is a band along the arcs: with blocks instead of single rows 24 % of the wave-tiles hold a live pixel instead of 45 %,
  // the chains run 0.39 rounds per tile instead of 0.47 and the cull test of shp_cull_setup fires on 52 % instead of 31 % (C3 prior).
  const int blk_w = a.blk_w, wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto pixel_of = [&](int base, int w) -> int {
    if (blk_w == 0) return base + w * WG + tid;
    const int nbx = blk_w >> 4, t = (base >> 7) + wave_u;
    const int by = t / nbx, bx = t - by * nbx;
    return (by * 8 + 4 * w + (lane >> 4)) * blk_w + bx * 16 + (lane & 15);
  };
  int n_tiles = 0, n_live = 0, n_lensed = 0;  // wave-tiles seen / chain rounds run / tiles that ran the lens (wave-uniform): a measurement aid in the row's two pad slots
  // (x_pre, y_pre: the tile's grid coordinates, requested while the previous tile was being worked on -- steady-state tiles only.
  // With two waves per SIMD nothing hides the round trip of a tile's first loads: ~600 of a dead tile's ~1500 cycles.)
  auto tile = [&](int base, auto check_tag, auto pre_tag, V x_pre, V y_pre, V obs_pre, V err_pre) {
    constexpr bool CHECK = decltype(check_tag)::value;
    constexpr bool PRE = decltype(pre_tag)::value;
    unsigned jj[2], pidx[2];
    bool valid[2];
    V vmask = V(1.f);
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      int j = CHECK ? base + w * WG + tid : pixel_of(base, w);
      valid[w] = CHECK ? (j < p1) : true;
      jj[w] = (unsigned)(valid[w] ? j : p1 - 1);
      pidx[w] = (CHECK && has_pix) ? (unsigned)a.pix[jj[w]] : jj[w];  // CHECK=false tiles run only without a pixel list
    }
    // 32-bit byte offsets from the scalar plane bases (one shift per pixel serves grid, observation and error planes)
    const unsigned jo0 = jj[0] << 2, jo1 = jj[1] << 2, po0 = pidx[0] << 2, po1 = pidx[1] << 2;
    auto ldf = [](const float* base, unsigned byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
    const V x = PRE ? x_pre : V{ldf(a.gx, jo0), ldf(a.gx, jo1)}, y = PRE ? y_pre : V{ldf(a.gy, jo0), ldf(a.gy, jo1)};
    // the likelihood's planes are requested here, ahead of the lens and the chains that do not need them
    // (table mode only, like the coordinate prefetch: the direct-mode kernel is at its register limit and measured 2 % slower with them)
    V o_pre = obs_pre, e_pre = err_pre;
    if constexpr (INTERP && !PRE && (MODE == LL_FWD || MODE == LL_GRAD)) {
      o_pre = V{ldf(a.obs, po0), ldf(a.obs, po1)};
      e_pre = has_err ? V{ldf(a.err, po0), ldf(a.err, po1)} : V(1.f);
    }
    if (CHECK) vmask = V{valid[0] ? 1.f : 0.f, valid[1] ? 1.f : 0.f};
    V bx = x, by = y, m = V(0.f);
    EplStateV<V> est[NL > 0 ? NL : 1];
    // (table mode) a wave-tile every pixel of which is PROVABLY outside the shapelet table skips the lens altogether: shp_cull_setup
    const bool culled = (INTERP && cull.tile_outside(x, y)) || GL_DBG(a.dbg, 1024);  // (dissection builds: 1024 no lens at all)
    if (!culled)
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int kind = LK::kinds[i];
      if constexpr (kind == K_EPL) epl_fwd_v<V, GRAD, gptr4>(dL[i], (gptr4)(gder + comps[i].d_off), x, y, bx, by, est[i]);  // (constant address space: gl_vec.hip.h)
      else if constexpr (kind == K_SIE) sie_fwd_v<V>(dL[i], x, y, bx, by);
      else if constexpr (kind == K_SHEAR) shear_fwd_v<V>(dL[i], x, y, bx, by);
      else sis_fwd_v<V>(dL[i], x, y, bx, by);
    }, std::make_integer_sequence<int, NL>{});
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      SerStateV<V> sst_i;
      m += sersic_fwd_v<V, true>(dC[i], x, y, sst_i);
    }, std::make_integer_sequence<int, NLL>{});
    ShpPix<NP> ps0, ps1;
    shp_pixel_coords<NP>(dS, bx.x, by.x, ps0);
    shp_pixel_coords<NP>(dS, bx.y, by.y, ps1);
    if (culled) ps0.u = ps1.u = 100.f;  // outside the table whatever beta would have been
    n_lensed += culled ? 0 : 1;
    // Table mode: a pixel whose u OR v lies outside the table's support renders exactly zero with zero slopes (fill 0 / 0,
    // shapelets.py:58-60).  Round 3 skipped the wave-tiles none of whose 128 pixels is inside (54 % on the C3 prior) and ran both
    // chains of every lane on the others -- but on a lensed field the support is a band along the arcs, and a live wave-tile
    // typically holds 30-60 live pixels of 128.  Round 4 COMPACTS them: every live pixel takes its rank in the wave (ballot +
    // mbcnt), its (u, v) goes to the rank's slot of a wave-private LDS list, and the chains run on the list -- lane i of round r
    // takes entry 64 r + i -- in ceil(count / 64) rounds (0, 1 or 2, wave-uniform) instead of always two; the order planes are
    // parked by RANK, so the matrix-pipe pass walks 16 pixel groups per round instead of 32 per tile; (S, dS/du, dS/dv) travel
    // back to the owning lane through the list.  Entries beyond the count run the chain outside the table (all zeros).
    bool shp_live = true;
    int rounds = 2, r0 = lane, r1 = 64 + lane;  // direct mode: every pixel is live, slot w of lane l has rank 64 w + l
    bool in0 = true, in1 = true;
    ++n_tiles;
    if constexpr (INTERP) {
      in0 = shp_in_table(ps0.u) && shp_in_table(ps0.v);
      in1 = shp_in_table(ps1.u) && shp_in_table(ps1.v);
      const unsigned long long m0 = __builtin_amdgcn_ballot_w64(in0), m1 = __builtin_amdgcn_ballot_w64(in1);
      const int c0 = __builtin_popcountll(m0), count = c0 + __builtin_popcountll(m1);
      r0 = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, 0u));
      r1 = c0 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u));
      rounds = GL_DBG(a.dbg, 128) ? 0 : (count + 63) >> 6;  // (dissection builds: 128 no chain rounds, 256 no matrix-pipe pass, 512 no VJPs of a live tile)
      shp_live = count != 0 && !GL_DBG(a.dbg, 128);
      ps0.S = ps0.Su = ps0.Sv = 0.f; ps0.fac = 1.f;
      ps1.S = ps1.Su = ps1.Sv = 0.f; ps1.fac = 1.f;
      if (shp_live) {
        float2* cin = reinterpret_cast<float2*>(xw + 2 * NP * SHX_PLANE + SHX_GS);
        float4* cout = reinterpret_cast<float4*>(xw + 2 * NP * SHX_PLANE + SHX_GS + SHX_CIN);
        float* gs_all = xw + 2 * NP * SHX_PLANE;
        if (in0) cin[r0] = float2{ps0.u, ps0.v};
        if (in1) cin[r1] = float2{ps1.u, ps1.v};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
        for (int r = 0; r < rounds; ++r) {
          const int idx = 64 * r + lane;
          const float2 uv = cin[idx];
          ShpPix<NP> pc;
          pc.u = idx < count ? uv.x : 100.f;  // (beyond the count: stale list entries -- outside the table instead)
          pc.v = idx < count ? uv.y : 100.f;
          pc.Su = pc.Sv = 0.f;
          (void)shp_pixel_fwd<NP, true, GRAD>(dS, gA, xw + 2 * idx, pc);
          cout[idx] = float4{pc.S, pc.Su, pc.Sv, 0.f};
          if constexpr (GRAD) gs_all[idx] = 0.f;  // a slot without an owner multiplies its planes by zero in the matrix-pipe pass
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (in0) { const float4 q = cout[r0]; ps0.S = q.x; ps0.Su = q.y; ps0.Sv = q.z; }
        if (in1) { const float4 q = cout[r1]; ps1.S = q.x; ps1.Su = q.y; ps1.Sv = q.z; }
        m += V{ps0.S, ps1.S};
      }
      n_live += rounds;
    } else {
      n_live += 2;
      const float l0 = shp_pixel_fwd<NP, INTERP, GRAD>(dS, gA, wr_xy, ps0);
      const float l1 = shp_pixel_fwd<NP, INTERP, GRAD>(dS, gA, wr_xy + 128, ps1);
      m += V{l0, l1};
    }
    auto nanp = m != m;
    m = (nanp ? V(0.f) : m) * a.out_scale;  // NaN -> 0 (tf/simulator.py:140), then x det(T) (:156)
    if (MODE == IMG_FWD) {
      float* row = a.img + (size_t)b * a.img_stride;
      if (valid[0]) row[pidx[0]] = m.x;
      if (valid[1]) row[pidx[1]] = m.y;
      return;
    }
    V gm;
    if (MODE == IMG_BWD) {
      const float* row = a.gimg + (size_t)b * a.img_stride;
      const V g = V{row[pidx[0]], row[pidx[1]]};
      gm = nanp ? V(0.f) : (CHECK ? g * vmask : g) * a.out_scale;
    } else {
      V o = o_pre, w = vmask, e = e_pre;
      if constexpr (!INTERP) {
        o = V{ldf(a.obs, po0), ldf(a.obs, po1)};
        if (has_err) e = V{ldf(a.err, po0), ldf(a.err, po1)};
      }
      if (CHECK && has_mask) w = w * V{ldf(a.mask, po0), ldf(a.mask, po1)};
      V dmo = m - o;  // tf/model.py:92-99; sigma^2 = bg^2 + m/t (no clip: negative -> NaN like the sqrt of a negative)
      V s2 = has_err ? e * e : m * a.inv_t + a.bg2;
      V is2 = rcp(s2);
      V nm = vlog<V>(s2 * (float)(2 * kPi));
      V c2 = __builtin_elementwise_fma(nm, V(0.f), dmo * dmo * is2);  // + 0 * nm: carries that NaN into chi^2
      if (CHECK) {
        auto use = w != V(0.f);
        st0 += use ? c2 * w : V(0.f);
        st1 += use ? nm * w : V(0.f);
      } else {
        st0 += c2;
        st1 += nm;
      }
      if (MODE == LL_GRAD) {
        // (0.5 / t as a scalar operand and an inline constant: folded into one packed register it was the kernel's only spill)
        V g = has_err ? -(dmo * is2) : (dmo * dmo * is2 - 1.f) * ((is2 * a.inv_t) * 0.5f) - dmo * is2;
        gm = nanp ? V(0.f) : (CHECK ? g * w : g) * a.out_scale;
      }
    }
    if constexpr (GRAD) {
      V gbx = V(0.f), gby = V(0.f);
      static_for([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr int off = [] { int n = 0; for (int j = 0; j < i; ++j) n += static_nacc(LLK::kinds[j]); return n; }();
        // the lens light's forward state is re-evaluated here (three transcendentals per pixel) rather than carried across the
        // shapelet chains: with it the kernel needs more than 256 VGPRs
        SerStateV<V> sst_i;
        (void)sersic_fwd_v<V, true>(dC[i], x, y, sst_i);
        sersic_vjp_v<V, false, true>(dC[i], sst_i, gm, accC + off, gbx, gby);
      }, std::make_integer_sequence<int, NLL>{});
      // ---- shapelet VJP of the pair: positions in packed form, amplitudes through the matrix pipe ----
      if (shp_live) {
        const float ib = dS[SHP_IB];
        // the source-plane offsets and (u, v) are recomputed from beta (two packed instructions each) instead of kept
        // through the likelihood terms; table mode has no Gaussian factor to keep either
        const V pdx = bx - dS[SHP_CX], pdy = by - dS[SHP_CY];
        const V fac = INTERP ? V(1.f) : V{ps0.fac, ps1.fac};
        const V gS = INTERP ? gm : gm * fac;
        // the pixel's cotangent of S, at its RANK's slot: applied to the Y operand after the transposed read
        if (in0) wr_gs0[r0] = gS.x;
        if (in1) wr_gs0[r1] = gS.y;
        const float ds = INTERP ? (float)(SH_NODES - 1) / 10.f : 1.f;  // table mode: the differences are per node spacing
        V gu = gS * (V{ps0.Su, ps1.Su} * ds), gv = gS * (V{ps0.Sv, ps1.Sv} * ds);
        if constexpr (!INTERP) {  // d fac / du = -u fac
          const V gIf = gm * (fac * V{ps0.S, ps1.S});
          gu -= gIf * (pdx * ib);
          gv -= gIf * (pdy * ib);
        }
        const V gdx = gu * ib, gdy = gv * ib;
        acc_cx -= gdx;
        acc_cy -= gdy;
        acc_ib += gu * pdx + gv * pdy;
        gbx += gdx;
        gby += gdy;
      }
      gbx = -gbx;
      gby = -gby;
      // A tile none of whose pixels is inside the shapelet support sends no cotangent to the lens (the image does not depend on
      // beta there: value 0, slope 0; a lens light is evaluated on the grid, not at beta): the lens VJPs of such a tile add exact
      // zeros and are skipped (wave-uniform; 54 % of the wave-tiles on the C3 prior).
      if (shp_live && !GL_DBG(a.dbg, 512))
      static_for([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr int kind = LK::kinds[i];
        constexpr int off = [] { int n = 0; for (int j = 0; j < i; ++j) n += static_nacc(LK::kinds[j]); return n; }();
        if constexpr (kind == K_EPL) epl_vjp_v<V>(dL[i], gbx, gby, est[i], accL + off);
        else if constexpr (kind == K_SIE) sie_vjp_v<V>(dL[i], x, y, gbx, gby, accL + off);
        else if constexpr (kind == K_SHEAR) shear_vjp_v<V>(x, y, gbx, gby, accL + off);
        else sis_vjp_v<V>(dL[i], x, y, gbx, gby, accL + off);
      }, std::make_integer_sequence<int, NL>{});
      // ---- G += (gS X)(Y)^T over the wave's 128 pixels: 32 MFMAs of four pixels each.  The planes are private to the wave and
      // LDS serves a wave's requests in order, so stores -> transposed loads need no barrier, only the compiler's ordering. ----
      if (shp_live && !GL_DBG(a.dbg, 256)) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // operands in batches of MB pixel groups (all reads of a batch in flight before its first MFMA; the next
      // batch's reads are issued ahead of this batch's MFMAs), two accumulator tiles in turn: the chain of dependent MFMAs
      // (40 cycles each) is half as long and no MFMA waits for its own LDS read
      constexpr int MB = 4;  // pixel groups per batch (a batch's 2 MB operand registers are double-buffered)
      auto mfma_groups = [&](const float* gx, const float* gy, const float* gg, auto n_batches) {
        constexpr int NB = decltype(n_batches)::value;
        float opa[2][MB], opb[2][MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) { opa[0][i] = gx[8 * i]; opb[0][i] = gy[8 * i] * gg[4 * i]; }
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) {
          if (bt + 1 < NB) {
#pragma unroll
            for (int i = 0; i < MB; ++i) {
              opa[(bt + 1) & 1][i] = gx[8 * (MB * (bt + 1) + i)];
              opb[(bt + 1) & 1][i] = gy[8 * (MB * (bt + 1) + i)] * gg[4 * (MB * (bt + 1) + i)];
            }
          }
#pragma unroll
          for (int i = 0; i < MB; i += 2) {
            G = __builtin_amdgcn_mfma_f32_16x16x4f32(opa[bt & 1][i], opb[bt & 1][i], G, 0, 0, 0);
            G2 = __builtin_amdgcn_mfma_f32_16x16x4f32(opa[bt & 1][i + 1], opb[bt & 1][i + 1], G2, 0, 0, 0);
          }
        }
      };
      if constexpr (INTERP) {
#pragma unroll 1
        for (int r = 0; r < rounds; ++r)  // 16 pixel groups (64 ranked slots) per round of the chains
          mfma_groups(rd_gx + 128 * r, rd_y + 128 * r, rd_gs + 64 * r, std::integral_constant<int, 16 / MB>{});
      } else {
        mfma_groups(rd_gx, rd_y, rd_gs, std::integral_constant<int, 32 / MB>{});  // all 128 slots, one unrolled stream
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the next tile's stores stay behind these loads
      __builtin_amdgcn_wave_barrier();
      }
    }
  };
  {
    int base = p0;
    // whole tiles without a pixel list: the coordinates of tile k + 1 are requested before tile k is worked on (the request past
    // the last whole tile re-reads that tile's: inside the planes, never used)
    auto whole_tiles = [&]() {
      if constexpr (!INTERP) {
        for (; base + WG * 2 <= p1; base += WG * 2) tile(base, std::false_type{}, std::false_type{}, V(0.f), V(0.f), V(0.f), V(1.f));
        return;
      }
      if (base + WG * 2 > p1) return;
      // (the observation and error planes too: a tile that skips the lens has nothing else to wait behind -- dissected in round 4,
      // the bare tile loop without lens or chains took 37 us of the kernel's 175, ~1400 cycles per wave-tile for ~60 instructions)
      auto planes = [&](int bs, V& xo, V& yo, V& oo, V& eo) {
        const int j0 = pixel_of(bs, 0), j1 = pixel_of(bs, 1);
        xo = V{a.gx[j0], a.gx[j1]};
        yo = V{a.gy[j0], a.gy[j1]};
        oo = V(0.f);
        eo = V(1.f);
        if constexpr (MODE == LL_FWD || MODE == LL_GRAD) {
          oo = V{a.obs[j0], a.obs[j1]};
          if (has_err) eo = V{a.err[j0], a.err[j1]};
        }
      };
      V xn, yn, on, en;
      planes(base, xn, yn, on, en);
      for (; base + WG * 2 <= (GL_DBG(a.dbg, 2048) ? p0 : p1); base += WG * 2) {  // (dissection builds: 2048 no tiles)
        const V xc = xn, yc = yn, oc = on, ec = en;
        planes(base + WG * 4 <= p1 ? base + WG * 2 : base, xn, yn, on, en);
        tile(base, std::false_type{}, std::true_type{}, xc, yc, oc, ec);
      }
    };
    if constexpr (!RAGGED) {
      whole_tiles();
    } else {
      if (!has_mask && !has_pix) whole_tiles();
      for (; base < p1; base += WG * 2) tile(base, std::true_type{}, std::false_type{}, V(0.f), V(0.f), V(0.f), V(1.f));
    }
  }
  if (MODE == IMG_FWD) return;
  float* out = a.partial + ((size_t)b * gridDim.x + chunk) * a.A;
  __syncthreads();  // every wave is done with its exchange planes: the epilogue rows alias them
  float* s_acc = s_x;                  // [16 rows][Apad] (gradient modes) / [4 rows][Apad]
  float* s_g = s_x + 16 * a.Apad;      // [4 waves][16][17] amplitude tiles
  if constexpr (!GRAD) {
    for (int i = tid; i < 4 * a.Apad; i += WG) s_acc[i] = 0.f;
    __syncthreads();
    float* s_row = s_acc + wave * a.Apad;
    const float c2 = wave_sum63(hsum(st0)), nm = wave_sum63(hsum(st1));
    if (lane == 63) { s_row[0] = c2; s_row[1] = nm; }
    __syncthreads();
    for (int k = tid; k < a.A; k += WG) out[k] = (s_acc[k] + s_acc[a.Apad + k]) + (s_acc[2 * a.Apad + k] + s_acc[3 * a.Apad + k]);
    return;
  } else {
    // [chi2, norm, 0, 0 | lenses | lens lights | shapelet cx, cy, 1/beta] four values per register (gl_pair.hip.h), then the tile
    constexpr int NV = NSTAT + NACC_L + NACC_C + SHPA_AMP, NVP = (NV + 3) & ~3;
    float vals[NVP];
#pragma unroll
    for (int k = 0; k < NVP; ++k) vals[k] = 0.f;
    vals[0] = (MODE == LL_GRAD) ? hsum(st0) : 0.f;
    vals[1] = (MODE == LL_GRAD) ? hsum(st1) : 0.f;
    vals[2] = lane == 0 ? (float)n_live : 0.f;   // pad slots of the row (finalize does not read them): rounds of the shapelet
    vals[3] = lane == 0 ? (float)(n_tiles + 4096 * n_lensed) : 0.f;  // chains this wave ran (0-2 per tile), its tiles and (x 4096) those of them that ran the lens -- bench.py's work model
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int kind = LK::kinds[i];
      constexpr int off = [] { int n = 0; for (int j = 0; j < i; ++j) n += static_nacc(LK::kinds[j]); return n; }();
      constexpr int Gn = static_nacc(kind);
      float tmp[Gn];
#pragma unroll
      for (int k = 0; k < Gn; ++k) tmp[k] = hsum(accL[off + k]);
      if constexpr (kind == K_EPL) {  // deferred per-sample factors of epl_vjp_v (see gl_pair.hip.h)
        tmp[EPLA_B] = tmp[EPLA_P0] * (dL[i][EPL_TM1] * dL[i][EPL_INVB]);
        tmp[EPLA_P0] *= rcp(dL[i][EPL_P0]);
        const float gxr = tmp[EPLA_CX], gyr = tmp[EPLA_CY], cc = dL[i][EPL_C], ss = dL[i][EPL_S];
        tmp[EPLA_CX] = -(gxr * cc - gyr * ss);
        tmp[EPLA_CY] = -(gxr * ss + gyr * cc);
      }
#pragma unroll
      for (int k = 0; k < Gn; ++k) vals[NSTAT + off + k] = tmp[k];
    }, std::make_integer_sequence<int, NL>{});
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int off = [] { int n = 0; for (int j = 0; j < i; ++j) n += static_nacc(LLK::kinds[j]); return n; }();
#pragma unroll
      for (int k = 0; k < SER_NACC; ++k) vals[NSTAT + NACC_L + off + k] = hsum(accC[off + k]) * (k == SERA_INVN ? (float)kLn2 : 1.f);
    }, std::make_integer_sequence<int, NLL>{});
    vals[NSTAT + NACC_L + NACC_C + SHPA_CX] = hsum(acc_cx);
    vals[NSTAT + NACC_L + NACC_C + SHPA_CY] = hsum(acc_cy);
    vals[NSTAT + NACC_L + NACC_C + SHPA_IB] = hsum(acc_ib);
    const bool odd = tid & 1, hi = tid & 2;
    float* s_row16 = s_acc + (tid >> 4) * a.Apad;
#pragma unroll
    for (int g = 0; g < NVP / 4; ++g) {
      float r = quad_transpose_sum(vals[4 * g], vals[4 * g + 1], vals[4 * g + 2], vals[4 * g + 3], odd, hi);
      r = dpp_add(r, 0x114, 0xF);  // row_shr:4
      r = dpp_add(r, 0x118, 0xF);  // row_shr:8 -> lanes 12..15 of the row: the row's sums of values 4g .. 4g + 3
      if ((tid & 15) >= 12 && 4 * g + (tid & 3) < NV) s_row16[4 * g + (tid & 3)] = r;
    }
    // the wave's tile: lane holds rows 4 (lane / 16) + i, column lane % 16 (v_mfma_f32_16x16x4_f32 accumulator layout)
    float* s_gw = s_g + wave * (16 * 17);
#pragma unroll
    for (int i = 0; i < 4; ++i) s_gw[(4 * (lane >> 4) + i) * 17 + (lane & 15)] = G[i] + G2[i];
    __syncthreads();
    for (int k = tid; k < NV; k += WG) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) v += s_acc[j * a.Apad + k];
      out[k] = v;
    }
    // amplitude i = n (n + 1) / 2 + n2 with n = n1 + n2 (shapelets.py:41-46)
    const int n_amp = shp.n_acc - SHPA_AMP;
    for (int i = tid; i < n_amp; i += WG) {
      int n = 0;
      while ((n + 1) * (n + 2) / 2 <= i) ++n;
      const int n2 = i - n * (n + 1) / 2, n1 = n - n2;
      const float* t = s_g + n1 * 17 + n2;
      // the planes hold the monic basis: G(n1, n2) = SH_K[n1] SH_K[n2] sum gS P_n1 P_n2
      out[shp.a_off + SHPA_AMP + i] = ((t[0] + t[16 * 17]) + (t[2 * 16 * 17] + t[3 * 16 * 17])) * (SH_K[n1] * SH_K[n2]);
    }
  }
}


// ---- lstsq_simulate without the basis stack (tf/simulator.py:158-240) -------------------------------------------------------
// The normal matrix of the linear-amplitude solve,  N = [X | Y]^T W^2 [X | Y]  with X the unit-amplitude shapelet basis images,
// is formed straight from the bases: the round-2 path rendered the (D, H W) stack of every sample to HBM (4.4 GB at C3L, 0.98
// ms) and read it back in the SYRK (1.21 ms).  Here a wave keeps X_n(u) w and Y_n(v) of its 128 pixels in LDS planes (the
// exchange layout of gl_shp_kernel), and lane (c, k) of the MFMA operand layout multiplies its channel's factors on the fly:
// channel (n1, n2) of pixel P = X'[n1][P] Y[n2][P]; the observation column reads obs w against a plane of ones, padding channels
// a plane of zeros -- no selects in the loop.  NT tile rows of 16 channels, lower tiles only, exact fp32 MFMA like the SYRK.
// Tiles none of whose pixels lies inside the shapelet table only add their sum of (obs w)^2 to the (Y, Y) entry.
struct ShpNormalArgs {
  const float* obs;   // [N]
  const float* err;   // [N]
  float* partial;     // [B][n_chunks][Dp * Dp], lower triangle valid (the layout of gl_normal_mfma_kernel)
  int Dl, Dp;         // linear channels (shapelet layers); Dp = Dl + 1 rounded up to a multiple of 4
};

constexpr int SHN_PLANE = 132;  // floats per plane of the normal kernel: 64 pixels x 2 orders, + 4 (plane stride = 4 mod 32 banks)
constexpr int SHN_LIST = 512;   // table mode: the wave-tile's live pixels in rank order, [128] x (u, v, w, obs w)
__host__ __device__ constexpr int shn_wave_floats(int np) { return (2 * np + 1) * SHN_PLANE + SHN_LIST; }  // X', Y planes + the (obs w, 1) plane + the list

template <int NT, int WAVES, class LK, int NP, bool INTERP>
__global__ void __launch_bounds__(WG, WAVES) gl_shp_normal_kernel(MainArgs a, ShpNormalArgs na) {
  using V = v2f;
  constexpr int NL = LK::n;
  constexpr int NTILES = NT * (NT + 1) / 2;
  constexpr int XPL = shn_wave_floats(NP);
  extern __shared__ float smem[];
  float* s_d = smem;
  float* s_x = smem + ((a.D + 3) & ~3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = a.order ? a.order[blockIdx.y] : blockIdx.y, chunk = blockIdx.x;
  const CompDesc* __restrict__ comps = a.comps;
  const float* __restrict__ gder = a.derived + (size_t)b * a.D;
  for (int i = tid; i < a.D; i += WG) s_d[i] = gder[i];
  float* xw = s_x + wave * XPL;
  float* pl_ow = xw + 2 * NP * SHN_PLANE;  // [pixel][2] = (obs w, 1): the observation column's two "factors"
  __syncthreads();
  const float* dL[NL > 0 ? NL : 1];
#pragma unroll
  for (int i = 0; i < NL; ++i) dL[i] = s_d + comps[i].d_off;
  const CompDesc shp = comps[NL];
  const float* dS = s_d + shp.d_off;
  float* wr_xy = xw + 2 * lane;  // plane n = (X_n w, Y_n) of each pixel
  // the lane's channel of every tile row -- two factor addresses in the planes ([pixel][2] pairs, 8 floats per pixel group):
  // amplitude (n1, n2): X'[n1] and Y[n2]; the observation column: (obs w, 1); padding: order 2 NP - 1 of X and Y, which is
  // zero for every n_max this kernel serves (<= 2 NP - 2)
  const int c = lane & 15, k = lane >> 4;
  const float* rd_a[NT];
  const float* rd_b[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ch = 16 * t + c;
    int n1 = 2 * NP - 1, n2 = 2 * NP - 1;
    if (ch < na.Dl) {
      int n = 0;
      while ((n + 1) * (n + 2) / 2 <= ch) ++n;
      n2 = ch - n * (n + 1) / 2;
      n1 = n - n2;
    }
    rd_a[t] = xw + n1 * SHN_PLANE + 2 * k;
    rd_b[t] = xw + n2 * SHN_PLANE + 1 + 2 * k;
    if (ch == na.Dl) { rd_a[t] = pl_ow + 2 * k; rd_b[t] = pl_ow + 1 + 2 * k; }
  }
  v4f acc[NTILES];
#pragma unroll
  for (int q = 0; q < NTILES; ++q) acc[q] = v4f{0.f, 0.f, 0.f, 0.f};
  float yy = 0.f;  // sum of (obs w)^2 over the runs that skip the MFMAs
  // 512-pixel tiles dealt round-robin to the sample's workgroups: the shapelet support is a compact region of the image, and
  // contiguous chunks would leave some workgroups all MFMA passes and others none
  // Inside a tile the four waves swap 64-pixel runs from trip to trip: with a fixed assignment waves 0, 2 would see left halves
  // of 128-pixel rows only and waves 1, 3 right halves, and a lensed source is rarely symmetric -- the workgroup then waits for
  // its busiest wave at the end.
  const int p1 = a.N;
  int rot = wave;
  for (int base = chunk * (WG * 2); base < p1; base += (int)gridDim.x * (WG * 2), ++rot) {
    // the lens on two pixels per lane (packed fp32 like the other kernels); the bases and the MFMA pass then take the wave's two
    // 64-pixel runs one after the other, so that the planes of a wave are 64 pixels deep (27 KB / workgroup: three per CU)
    unsigned jj[2];
    bool valid[2];
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      int j = base + w * WG + ((rot & 3) << 6) + lane;
      if (a.blk_w) {  // the wave's two 64-pixel runs are rows 4 w .. 4 w + 3 of the image's 8 x 16 block (base / 128) + (rot & 3)
        const int nbx = a.blk_w >> 4, t = (base >> 7) + (rot & 3);
        const int by = t / nbx, bx = t - by * nbx;
        j = (by * 8 + 4 * w + (lane >> 4)) * a.blk_w + bx * 16 + (lane & 15);
      }
      valid[w] = j < p1;
      jj[w] = (unsigned)(valid[w] ? j : p1 - 1);
    }
    const unsigned jo0 = jj[0] << 2, jo1 = jj[1] << 2;
    auto ldf = [](const float* bp, unsigned byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(bp) + byte_off); };
    const V x = V{ldf(a.gx, jo0), ldf(a.gx, jo1)}, y = V{ldf(a.gy, jo0), ldf(a.gy, jo1)};
    const V er = V{ldf(na.err, jo0), ldf(na.err, jo1)}, ob = V{ldf(na.obs, jo0), ldf(na.obs, jo1)};
    V wgt = rcp(er);  // weights 1/err by v_rcp_f32, like the SYRK
    wgt = V{valid[0] ? wgt.x : 0.f, valid[1] ? wgt.y : 0.f};
    const V ow = ob * wgt;
    V bx = x, by = y;
    EplStateV<V> est;
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int kind = LK::kinds[i];
      if constexpr (kind == K_EPL) epl_fwd_v<V, false, gptr4>(dL[i], (gptr4)(gder + comps[i].d_off), x, y, bx, by, est);
      else if constexpr (kind == K_SIE) sie_fwd_v<V>(dL[i], x, y, bx, by);
      else if constexpr (kind == K_SHEAR) shear_fwd_v<V>(dL[i], x, y, bx, by);
      else sis_fwd_v<V>(dL[i], x, y, bx, by);
    }, std::make_integer_sequence<int, NL>{});
    // the bases of 64 pixels -> planes, then the MFMA pass over them: one ROUND
    auto round_ = [&](float u, float v, float w1, float ow1) {
      {
        constexpr int NO = 2 * NP;
        const float fac = INTERP ? 1.f : exp_(-(u * u + v * v) * 0.5f);
        const v2f xs = v2f{w1 * fac, 1.f};  // weight (and the Gaussian of direct mode) folded into the X factor
        ShpGen<INTERP> gen;
        gen.init(u, v);
#pragma unroll
        for (int n = 0; n < NO - 1; ++n) {
          v2f xy = gen.value() * (xs * SH_K[n]);  // phi_n = SH_K[n] P_n: the channels of the normal matrix are the normalised bases
          xy.x = xy.x == xy.x ? xy.x : 0.f;  // NaN -> 0 like the stack (tf/simulator.py:140 on each basis image)
          *reinterpret_cast<v2f*>(wr_xy + n * SHN_PLANE) = xy;
          if (n + 1 < NO - 1) gen.advance(n);
        }
        *reinterpret_cast<v2f*>(wr_xy + (NO - 1) * SHN_PLANE) = v2f(0.f);  // the pad order: what the padding channels read
        *reinterpret_cast<v2f*>(pl_ow + 2 * lane) = v2f{ow1, 1.f};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {  // fully unrolled: every LDS offset is an immediate, no address arithmetic beside the MFMAs
        float v_[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) v_[t] = rd_a[t][8 * kb] * rd_b[t][8 * kb];
        int q = 0;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
          for (int tj = 0; tj <= ti; ++tj, ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(v_[ti], v_[tj], acc[q], 0, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    };
    ShpPix<NP> ps0, ps1;
    shp_pixel_coords<NP>(dS, bx.x, by.x, ps0);
    shp_pixel_coords<NP>(dS, bx.y, by.y, ps1);
    if constexpr (INTERP) {
      // Table mode: a pixel outside the table has every basis image zero -- only Y^T Y grows by its (obs w)^2.  Round 3 decided
      // that per 64-pixel run; round 4 compacts the wave-tile's LIVE pixels by rank (as gl_shp_kernel does) and runs
      // ceil(live / 64) rounds of bases + MFMA pass over the list: on a lensed field a live wave-tile holds 30-60 live pixels.
      const bool in0 = shp_in_table(ps0.u) && shp_in_table(ps0.v), in1 = shp_in_table(ps1.u) && shp_in_table(ps1.v);
      const unsigned long long m0 = __builtin_amdgcn_ballot_w64(in0), m1 = __builtin_amdgcn_ballot_w64(in1);
      const int c0 = __builtin_popcountll(m0), count = c0 + __builtin_popcountll(m1);
      if (!in0) yy = __builtin_fmaf(ow.x, ow.x, yy);
      if (!in1) yy = __builtin_fmaf(ow.y, ow.y, yy);
      if (count != 0) {
        float4* list = reinterpret_cast<float4*>(xw + (2 * NP + 1) * SHN_PLANE);
        const int r0 = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, 0u));
        const int r1 = c0 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u));
        if (in0) list[r0] = float4{ps0.u, ps0.v, wgt.x, ow.x};
        if (in1) list[r1] = float4{ps1.u, ps1.v, wgt.y, ow.y};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int rounds = (count + 63) >> 6;
#pragma unroll 1
        for (int r = 0; r < rounds; ++r) {
          const int idx = 64 * r + lane;
          const float4 e = list[idx];
          const bool ok = idx < count;  // beyond the count: a stale entry -- outside the table, weight 0
          round_(ok ? e.x : 100.f, ok ? e.y : 100.f, ok ? e.z : 0.f, ok ? e.w : 0.f);
        }
      }
    } else {
      round_(ps0.u, ps0.v, wgt.x, ow.x);
      round_(ps1.u, ps1.v, wgt.y, ow.y);
    }
  }
  // ---- the four waves' tiles summed in fixed order through LDS (aliases the planes), then the lower tiles written ----
  __syncthreads();
  float* s_red = s_x;
  // (all reads of a wave's turn first, then the adds and the stores: written as one read-modify-write per element the compiler
  // keeps them in order, 60 dependent LDS round trips per turn -- 40k cycles per workgroup, 15% of its life, measured)
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
      float prev[NTILES * 4];
#pragma unroll
      for (int e = 0; e < NTILES * 4; ++e) prev[e] = wv == 0 ? 0.f : s_red[e * 64 + lane];
#pragma unroll
      for (int e = 0; e < NTILES * 4; ++e) s_red[e * 64 + lane] = prev[e] + acc[e >> 2][e & 3];
    }
    __syncthreads();
  }
  float* s_yy = s_red + NTILES * 256;
  {
    const float t = wave_sum63(yy);
    if (lane == 63) s_yy[wave] = t;
  }
  __syncthreads();
  const float yy_skipped = (s_yy[0] + s_yy[1]) + (s_yy[2] + s_yy[3]);
  float* out = na.partial + ((size_t)b * gridDim.x + chunk) * na.Dp * na.Dp;
  // one 16-byte store per lane: lane (R, c4) of a wave takes columns 4 c4 .. 4 c4 + 3 of row R of a tile (four consecutive floats
  // of the reduction buffer); Dp is a multiple of 4, so a quad is inside the matrix or outside as a whole
  for (int q = wave; q < NTILES; q += 4) {
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= q) ++ti;
    const int tj = q - ti * (ti + 1) / 2;
    const int R = lane >> 2, c4 = lane & 3;
    const int i = 16 * ti + R, j = 16 * tj + 4 * c4;
    float4 v = *reinterpret_cast<const float4*>(s_red + (q * 4 + (R & 3)) * 64 + 16 * (R >> 2) + 4 * c4);
    if (i == na.Dl && (na.Dl >> 2) == (j >> 2)) {
      const int o = na.Dl & 3;
      v.x += o == 0 ? yy_skipped : 0.f;
      v.y += o == 1 ? yy_skipped : 0.f;
      v.z += o == 2 ? yy_skipped : 0.f;
      v.w += o == 3 ? yy_skipped : 0.f;
    }
    if (i < na.Dp && j < na.Dp) *reinterpret_cast<float4*>(out + i * na.Dp + j) = v;
  }
}

}  // namespace glk
