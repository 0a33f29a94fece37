// gl_cluster.hip.h -- forward+gradient kernel for "N x same kind" cluster models: up to NH NFW halos lensing up to NS
// Sersic sources (BASELINE config 4: 8 halos + 20 sources, 256 x 256 px; tf/profiles/mass/nfw.py:15-52,
// tf/profiles/light/sersic.py:29-80, tf/simulator.py:109-156, tf/model.py:89-101).
//
// Why not the interpreter (gl_main_kernel): for such a model it spends its time on (a) every profile's forward pass a
// second time inside its VJP -- 3 quarter-rate transcendentals per source and pixel, the branchy NFW g(X) per halo and
// pixel -- (b) the component switch and LDS constant loads per (component, tile), and it spills (81 VGPRs under a
// 128-register budget).  Here
//   * one thread owns ONE pixel pair (packed fp32 throughout, like gl_pair_kernel) and keeps the forward state of EVERY
//     component of that pair in registers until its VJP has consumed it: 3 values per source (E, u, log2(R/Rs)) and 3 per
//     halo (h = g/X^2 and the two factors its VJP needs) -- 6 VGPRs each, 168 for 8 + 20, inside a 256-register budget
//     (2 waves per SIMD), no spills, nothing evaluated twice;
//   * the component loops are compile-time loops over the capacity (NH, NS) guarded by wave-uniform counts, so the state is
//     statically indexed; per-component constants come through scalar loads from the sample's derived row (wave-uniform
//     address -> SGPR operands): no LDS reads, no descriptor loads in the pixel loop;
//   * the 4..8 gradient values per component (152 for 8 + 20) cannot each own a register, and summing them into LDS every
//     tile is what sinks this design: ds_add_f32 by the quad leaders (measured: 3.3 ms, SQ_WAIT_INST_LDS = 31 % of the wave
//     cycles, VALU busy 39 %; the same kernel without the atomics 1.6 ms) or read-modify-write chains at two waves per SIMD.
//     Instead FOUR different values share one register: a 4 x 4 transpose-reduction inside every quad (6 selects + 3 DPP
//     adds per four values) leaves in lane q of each quad the quad's sum of value q, and that register is added to a
//     per-lane running sum -- 38 VGPRs for all 152 accumulators, no LDS instruction in the pixel loop, one store per
//     running sum at the end of the chunk; spherical Sersic sources skip the two ellipticity accumulators altogether and
//     are packed in pairs (2 x 6 values = 3 registers).  Summation order is fixed: bitwise reproducible.
// Same derived-constant layout, accumulator layout, partial rows and finalize as the other kernels.
#pragma once
#include "gl_pair.hip.h"

namespace glk {

#ifdef GL_CLUSTER_FENCE
#define GL_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define GL_SCHED_FENCE() ((void)0)
#endif

template <class V> struct NfwStateC { V h, w, uu; };
template <class V> struct SerStateC { V E, u; };  // log2(R/Rs) is recomputed in the VJP (1 transcendental): 40 VGPRs for 20 sources
// ... except for the first NKEEP spherical sources of a model, whose log2(R/Rs) is kept as well (the five-sum VJP left the
// registers free): their VJP takes 1/r2 = 2^(-2 L2) / Rs^2 from it instead of recomputing r2, its logarithm and reciprocal
template <class V> struct SerStateL { V L2; };

// NFW forward on a pixel pair, leaving what the VJP needs (nfw_vjp_v written once, split at the state):
//   a = K0 h;  cot(K0) = ga h;  gX0 = ga p, p = [X0 > 1e-6] K0 h';  t = gR0 / R0 = ga w, w = [R0 > 1e-7] p / (Rs R0);
//   cot(Rs) = -gX0 X0 / Rs = -ga uu, uu = p X0 / Rs        (ga = g . (dx, dy))
template <class V, class P = const float*>
__device__ __forceinline__ void nfw_fwd_c(P d, const float* __restrict__ s_tab, V x, V y, V& bx, V& by,
                                          NfwStateC<V>& st) {
  const float invrs = d[NFW_INVRS], K0 = d[NFW_K0];
  V dx = x - d[NFW_CX], dy = y - d[NFW_CY];
  V r2 = dx * dx + dy * dy;
  V R0 = sqrt_(r2);
  V iR0 = (r2 > V(0.f)) ? rcp(R0) : V(0.f);
  V X0 = vmax(R0, V(1e-7f)) * invrs;  // nfw.py:26
  V X = vmax(X0, V(1e-6f));           // nfw.py:37
  V iX = rcp(X);
  V hp;
  nfw_h_pair(s_tab, X, iX, st.h, hp);
  V p = (X0 > V(1e-6f)) ? hp * K0 : V(0.f);
  st.w = (R0 > V(1e-7f)) ? p * invrs * iR0 : V(0.f);
  st.uu = p * X0 * invrs;
  V a = st.h * K0;
  bx -= a * dx;
  by -= a * dy;
}
template <class V, class P = const float*>
__device__ __forceinline__ void nfw_vjp_c(P d, V x, V y, V gx, V gy, const NfwStateC<V>& st,
                                          V (&va)[NFW_NACC]) {
  V dx = x - d[NFW_CX], dy = y - d[NFW_CY];
  V ga = gx * dx + gy * dy;
  V a = st.h * d[NFW_K0];
  V t = ga * st.w;
  va[NFWA_CX] = -(gx * a + t * dx);
  va[NFWA_CY] = -(gy * a + t * dy);
  va[NFWA_RS] = -(ga * st.uu);
  va[NFWA_K0] = ga * st.h;
}

// Sersic forward / VJP with the state split (sersic_fwd_v / sersic_vjp_v of gl_vec.hip.h); ELL = false is the spherical
// profile (sersic.py:23-66 passes e1 = e2 = 0: no rotation, no axis-ratio stretch, no ellipticity gradients)
template <class V, bool ELL, class P = const float*>
__device__ __forceinline__ V sersic_fwd_c(P d, V x, V y, SerStateC<V>& st, V* keepL2 = nullptr) {
  V dx = x - d[SER_CX], dy = y - d[SER_CY];
  V r2;
  if constexpr (ELL) {
    const float c = d[SER_C], s = d[SER_S];
    V xt1 = (dx * c + dy * s) * d[SER_SQ], xt2 = (dy * c - dx * s) * d[SER_ISQ];
    r2 = xt1 * xt1 + xt2 * xt2;
  } else if (keepL2) {
    // (a pixel exactly on the source centre: 1e-30 keeps the kept logarithm finite, see sersic_vjp5_c; added inside the fused
    // multiply-adds -- below the last bit of any r2 > 1e-23 -- instead of two v_max per pixel pair)
    r2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, V(1e-30f)));
  } else {
    r2 = dx * dx + dy * dy;
  }
  if (ELL && keepL2) r2 = vmax(r2, V(1e-30f));
  V L2 = log2_(r2) * 0.5f + d[SER_L2IRS];  // log2(R / R_sersic) without the square root
  if (keepL2) *keepL2 = L2;
  st.u = exp2_(L2 * d[SER_INVN]);
  st.E = vexp<V>((st.u - 1.f) * -d[SER_BN]);
  return st.E * d[SER_IE];
}
template <class V, bool ELL, class P = const float*>
__device__ __forceinline__ void sersic_vjp_c(P d, V x, V y, const SerStateC<V>& st, V gI,
                                             V (&va)[SER_NACC], V& gpx, V& gpy) {
  V dx = x - d[SER_CX], dy = y - d[SER_CY];
  V gE = gI * st.E;
  V tI = gE * d[SER_IE];
  V guu = -(tI * st.u) * d[SER_BN];
  V gL = guu * d[SER_INVN];
  V gdx, gdy;
  if constexpr (ELL) {
    const float c = d[SER_C], s = d[SER_S], sq = d[SER_SQ], isq = d[SER_ISQ];
    V a1 = dx * c + dy * s, a2 = dy * c - dx * s;
    V xt1 = a1 * sq, xt2 = a2 * isq;
    V r2 = xt1 * xt1 + xt2 * xt2;
    auto pos = r2 > V(0.f);
    V k = pos ? gL * rcp(r2) : V(0.f);
    V gxt1 = k * xt1, gxt2 = k * xt2;
    V ga1 = gxt1 * sq, ga2 = gxt2 * isq;
    gdx = ga1 * c - ga2 * s;
    gdy = ga1 * s + ga2 * c;
    va[SERA_PHI] = ga1 * a2 - ga2 * a1;
    va[SERA_SQ] = gxt1 * a1 - gxt2 * a2 * (isq * isq);
    V L2 = log2_(r2) * 0.5f + d[SER_L2IRS];
    va[SERA_INVN] = pos ? guu * L2 : V(0.f);  // x ln2 when it is summed
  } else {
    V r2 = dx * dx + dy * dy;
    auto pos = r2 > V(0.f);
    V k = pos ? gL * rcp(r2) : V(0.f);
    gdx = k * dx;
    gdy = k * dy;
    va[SERA_PHI] = V(0.f);
    va[SERA_SQ] = V(0.f);
    V L2 = log2_(r2) * 0.5f + d[SER_L2IRS];
    va[SERA_INVN] = pos ? guu * L2 : V(0.f);
  }
  va[SERA_CX] = -gdx;
  va[SERA_CY] = -gdy;
  va[SERA_L] = gL;
  va[SERA_BN] = -(tI * (st.u - 1.f));
  va[SERA_IE] = gE;
  gpx += gdx;
  gpy += gdy;
}

// Spherical Sersic VJP reduced to FIVE raw sums per source.  Every accumulator of sersic_vjp_c is a per-sample constant times a
// sum of  w = gI E  against (u, u L2, u dx / r2, u dy / r2, 1):
//     CX = c Sx,  CY = c Sy,  L = -c A,  INVN = -Ie bn ln2 D,  BN = -Ie (A - B),  IE = B,       c = Ie bn / n,
//     A = sum w u,  B = sum w,  D = sum w u L2,  Sx = sum w u dx / r2,  Sy = sum w u dy / r2,
// so the pixel loop forms w, w u, their three products and one reciprocal -- 13 packed instructions per source and pixel pair
// instead of 25 and no selects -- five values (not six) go through the transpose-reduction, and the constants are applied
// once per workgroup when the partial row is written (cluster_sersic5_finish).  r2 is clamped at 1e-30 (a pixel exactly on a
// source centre: the reference's own gradient is 0 * inf there) instead of selected on.
enum { S5_SX = 0, S5_SY, S5_A, S5_D, S5_B, S5_N };
template <class V, class P = const float*>
__device__ __forceinline__ void sersic_vjp5_c(P d, V x, V y, const SerStateC<V>& st, V gI, V (&va)[S5_N],
                                              V& gpx, V& gpy) {
  V dx = x - d[SER_CX], dy = y - d[SER_CY];
  V r2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, V(1e-30f)));  // the forward pass's floor
  V L2 = log2_(r2) * 0.5f + d[SER_L2IRS];
  V w = gI * st.E;
  V wu = w * st.u;
  V q = wu * rcp(r2);
  V qx = q * dx, qy = q * dy;
  va[S5_SX] = qx;
  va[S5_SY] = qy;
  va[S5_A] = wu;
  va[S5_D] = wu * L2;
  va[S5_B] = w;
  const float c = d[SER_CG];
  gpx -= qx * c;  // d I / d beta = -c q (dx, dy)
  gpy -= qy * c;
}
// the same with log2(R / Rs) kept from the forward pass: 1 / r2 = (1 / Rs^2) 2^(-2 L2)   (SER_INVRS = 1 / R_sersic)
template <class V, class P = const float*>
__device__ __forceinline__ void sersic_vjp5_keep_c(P d, V x, V y, const SerStateC<V>& st, V L2, V gI,
                                                   V (&va)[S5_N], V& gpx, V& gpy) {
  V dx = x - d[SER_CX], dy = y - d[SER_CY];
  V ir2 = exp2_(L2 * -2.f) * d[SER_IRS2];  // finite: the forward pass of a kept source clamps r2 at 1e-30
  V w = gI * st.E;
  V wu = w * st.u;
  V q = wu * ir2;
  V qx = q * dx, qy = q * dy;
  va[S5_SX] = qx;
  va[S5_SY] = qy;
  va[S5_A] = wu;
  va[S5_D] = wu * L2;
  va[S5_B] = w;
  const float c = d[SER_CG];
  gpx -= qx * c;
  gpy -= qy * c;
}
// raw sums of one source (already summed over the workgroup) -> the accumulator slots finalize expects
__device__ __forceinline__ float cluster_sersic5_finish(const float* __restrict__ d, const float* raw, int k) {
  const float Ie = d[SER_IE], bn = d[SER_BN], c = Ie * bn * d[SER_INVN];
  switch (k) {
    case SERA_CX: return c * raw[S5_SX];
    case SERA_CY: return c * raw[S5_SY];
    case SERA_L: return -c * raw[S5_A];
    case SERA_INVN: return -(Ie * bn * (float)kLn2) * raw[S5_D];
    case SERA_BN: return -Ie * (raw[S5_A] - raw[S5_B]);
    case SERA_IE: return raw[S5_B];
    default: return 0.f;
  }
}

// accumulator slot (inside the sample's accumulator row) of lane q of running sum g -- the inverse of the packing in the loop
template <int NH, bool ELL> __device__ __forceinline__ int cluster_slot(int g, int q, int n_h, int n_s) {
  if (g < NH) return g < n_h ? NSTAT + NFW_NACC * g + q : -1;
  const int aS = NSTAT + NFW_NACC * n_h;
  g -= NH;
  if (ELL) {
    const int s = g >> 1;
    return s < n_s ? aS + SER_NACC * s + 4 * (g & 1) + q : -1;
  }
  const int flat = 4 * (g % 5) + q;  // 0..19 inside a group of four spherical sources: five raw sums each (sersic_vjp5_c)
  const int s = 4 * (g / 5) + flat / 5, k = flat % 5;
  const int map[5] = {SERA_CX, SERA_CY, SERA_L, SERA_INVN, SERA_IE};  // where the raw sums (Sx, Sy, A, D, B) are parked in the column
  return s < n_s ? aS + SER_NACC * s + map[k] : -1;
}

template <int MODE, int NH, int NS, bool ELL, int WAVES>
__global__ void __launch_bounds__(WG, WAVES) gl_cluster_kernel(MainArgs a, int n_h, int n_s) {
  static_assert(MODE == IMG_BWD || MODE == LL_GRAD, "gradient modes only (forward modes: gl_main_kernel)");
  using V = v2f;
  constexpr int W = 2;
  constexpr int SERP = (SER_NDX + 3) & ~3, NFWP = (NFW_ND + 3) & ~3;
  extern __shared__ float smem[];
  float* s_acc = smem;                  // [64][Apad]
  float* s_tab = smem + 64 * a.Apad;    // [NFW_TAB_NODES][2]: h(X), dh/du du
  const int tid = threadIdx.x;
  const int b = a.order ? a.order[blockIdx.y] : blockIdx.y, chunk = blockIdx.x;
  for (int i = tid; i < 64 * a.Apad; i += WG) s_acc[i] = 0.f;
  for (int i = tid; i < 2 * NFW_TAB_NODES; i += WG) s_tab[i] = a.nfw_tab[i];
  __syncthreads();
  // this sample's derived constants: wave-uniform address -> scalar loads
  const float* __restrict__ gder = a.derived + (size_t)b * a.D;
  const float* __restrict__ dH = gder;
  const float* __restrict__ dS = gder + NFWP * n_h;
  const int lane = tid & 63, wave = tid >> 6;
  const bool odd = lane & 1, hi = lane & 2;
  float* col = s_acc + (wave * 16 + (lane >> 2)) * a.Apad;
  // running sums: lane q of every quad owns value q of four accumulators per register (see quad_transpose_sum)
  constexpr int NR = NH + (ELL ? 2 * NS : 5 * ((NS + 3) / 4));
  float racc[NR];
#pragma unroll
  for (int g = 0; g < NR; ++g) racc[g] = 0.f;
  const bool has_err = a.err != nullptr, has_mask = a.mask != nullptr, has_pix = a.pix != nullptr;
  V st0 = V(0.f), st1 = V(0.f);
  const int p0 = chunk * a.chunk;
  const int p1 = min(p0 + a.chunk, a.N);

  auto tile = [&](int base, auto check_tag) {
    constexpr bool CHECK = decltype(check_tag)::value;
    unsigned jj[W], pidx[W];
    bool valid[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
      int j = base + w * WG + tid;
      valid[w] = CHECK ? (j < p1) : true;
      jj[w] = (unsigned)(valid[w] ? j : p1 - 1);
      pidx[w] = (CHECK && has_pix) ? (unsigned)a.pix[jj[w]] : jj[w];  // CHECK=false tiles run only without a pixel list
    }
    // 32-bit byte offsets from the scalar plane bases (one shift per pixel serves grid, observation and error planes)
    const unsigned jo0 = jj[0] << 2, jo1 = jj[1] << 2, po0 = pidx[0] << 2, po1 = pidx[1] << 2;
    auto ldf = [](const float* base, unsigned byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
    const V x = V{ldf(a.gx, jo0), ldf(a.gx, jo1)}, y = V{ldf(a.gy, jo0), ldf(a.gy, jo1)};
    V vmask = V(1.f);
    if (CHECK) vmask = V{valid[0] ? 1.f : 0.f, valid[1] ? 1.f : 0.f};
    V bx = x, by = y, m = V(0.f);
    NfwStateC<V> hst[NH];
    SerStateC<V> sst[NS];
    constexpr int NKEEP = ELL ? 0 : (NS > 8 ? 18 : NS);  // sources whose log2(R/Rs) stays in registers (budget: 256 VGPRs at 8 + 20)
    V sL2[NKEEP > 0 ? NKEEP : 1];
    // ---- ray-shoot: beta = (x, y) - sum_h alpha_h   (tf/simulator.py:72-78) ----
#pragma unroll
    for (int h = 0; h < NH; ++h)
      if (h < n_h) {
        nfw_fwd_c<V>(dH + NFWP * h, s_tab, x, y, bx, by, hst[h]);
        GL_SCHED_FENCE();
      }
    // ---- render the sources at beta (tf/simulator.py:128-138) ----
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (s < n_s) {
        m += sersic_fwd_c<V, ELL>(dS + SERP * s, bx, by, sst[s], s < NKEEP ? &sL2[s < NKEEP ? s : 0] : nullptr);
        GL_SCHED_FENCE();
      }
    auto nanp = m != m;
    m = (nanp ? V(0.f) : m) * a.out_scale;  // NaN -> 0 (tf/simulator.py:140), then x det(T) (:156)
    V gm;
    if (MODE == IMG_BWD) {
      const float* row = a.gimg + (size_t)b * a.img_stride;
      V g = V{row[pidx[0]], row[pidx[1]]};
      gm = nanp ? V(0.f) : (CHECK ? g * vmask : g) * a.out_scale;
    } else {
      V o = V{ldf(a.obs, po0), ldf(a.obs, po1)}, w = vmask, e = V(1.f);
      if (CHECK && has_mask) w = w * V{ldf(a.mask, po0), ldf(a.mask, po1)};
      if (has_err) e = V{ldf(a.err, po0), ldf(a.err, po1)};
      V dmo = m - o;
      V s2 = has_err ? e * e : m * a.inv_t + a.bg2;  // tf/model.py:92-95
      V is2 = rcp(s2);
      V nm = vlog<V>(s2 * (float)(2 * kPi));
      V c2 = __builtin_elementwise_fma(nm, V(0.f), dmo * dmo * is2);  // + 0 * nm: a NaN sigma reaches chi^2 like in the reference
      if (CHECK) {
        auto use = w != V(0.f);
        st0 += use ? c2 * w : V(0.f);
        st1 += use ? nm * w : V(0.f);
      } else {
        st0 += c2;
        st1 += nm;
      }
      V g = has_err ? -(dmo * is2) : (dmo * dmo * is2 - 1.f) * (is2 * (0.5f * a.inv_t)) - dmo * is2;
      gm = nanp ? V(0.f) : (CHECK ? g * w : g) * a.out_scale;
    }
    // ---- source VJPs: parameter gradients and the cotangent of beta ----
    V gbx = V(0.f), gby = V(0.f);
    if constexpr (ELL) {
#pragma unroll
      for (int s = 0; s < NS; ++s)
        if (s < n_s) {
          V va[SER_NACC];
          sersic_vjp_c<V, true>(dS + SERP * s, bx, by, sst[s], gm, va, gbx, gby);
          float v[SER_NACC];
#pragma unroll
          for (int k = 0; k < SER_NACC; ++k) v[k] = va[k].x + va[k].y;
          v[SERA_INVN] *= (float)kLn2;
          racc[NH + 2 * s] += quad_transpose_sum(v[0], v[1], v[2], v[3], odd, hi);
          racc[NH + 2 * s + 1] += quad_transpose_sum(v[4], v[5], v[6], v[7], odd, hi);
        }
    } else {
#pragma unroll
      for (int p = 0; p < (NS + 3) / 4; ++p)
        if (4 * p < n_s) {  // four spherical sources: 4 x (Sx, Sy, A, D, B) = five registers
          float v[20];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int s = 4 * p + j;
            if (s < NS && s < n_s) {
              V va[S5_N];
              if (s < NKEEP) sersic_vjp5_keep_c<V>(dS + SERP * s, bx, by, sst[s < NS ? s : 0], sL2[s < NKEEP ? s : 0], gm, va, gbx, gby);
              else sersic_vjp5_c<V>(dS + SERP * s, bx, by, sst[s < NS ? s : 0], gm, va, gbx, gby);
#pragma unroll
              for (int k = 0; k < S5_N; ++k) v[5 * j + k] = va[k].x + va[k].y;
            } else {
#pragma unroll
              for (int k = 0; k < S5_N; ++k) v[5 * j + k] = 0.f;
            }
          }
#pragma unroll
          for (int r = 0; r < 5; ++r) racc[NH + 5 * p + r] += quad_transpose_sum(v[4 * r], v[4 * r + 1], v[4 * r + 2], v[4 * r + 3], odd, hi);
          GL_SCHED_FENCE();
        }
    }
    // ---- halo VJPs with the cotangent -g_beta (beta = x - sum alpha) ----
    gbx = -gbx;
    gby = -gby;
#pragma unroll
    for (int h = 0; h < NH; ++h)
      if (h < n_h) {
        V va[NFW_NACC];
        nfw_vjp_c<V>(dH + NFWP * h, x, y, gbx, gby, hst[h], va);
        racc[h] += quad_transpose_sum(va[0].x + va[0].y, va[1].x + va[1].y, va[2].x + va[2].y, va[3].x + va[3].y, odd, hi);
        GL_SCHED_FENCE();
      }
  };
  {
    const bool plain = !has_mask && !has_pix;
    int base = p0;
    if (plain)
      for (; base + WG * W <= p1; base += WG * W) tile(base, std::false_type{});
    for (; base < p1; base += WG * W) tile(base, std::true_type{});
  }
  // ---- epilogue: every lane stores its running sums into its quad's column, then the 64 columns are summed ----
#pragma unroll
  for (int g = 0; g < NR; ++g) {
    const int slot = cluster_slot<NH, ELL>(g, lane & 3, n_h, n_s);
    if (slot >= 0) col[slot] = racc[g];
  }
  if (MODE == LL_GRAD) {
    const float c2 = quad_sum(st0.x + st0.y), nm = quad_sum(st1.x + st1.y);
    if ((lane & 3) == 0) { col[0] = c2; col[1] = nm; }
  }
  __syncthreads();
  float* out = a.partial + ((size_t)b * gridDim.x + chunk) * a.A;
  if constexpr (ELL) {
    for (int k = tid; k < a.A; k += WG) {
      float v = 0.f;
      for (int j = 0; j < 64; ++j) v += s_acc[j * a.Apad + k];
      out[k] = v;
    }
  } else {
    // spherical sources: the columns hold RAW sums (sersic_vjp5_c); sum the 64 columns in fixed order into column 0, then turn
    // each source's five sums into its accumulator slots with the sample's constants
    for (int k = tid; k < a.A; k += WG) {
      float v = 0.f;
      for (int j = 0; j < 64; ++j) v += s_acc[j * a.Apad + k];
      s_acc[k] = v;  // column 0, slot k: only this thread reads column 0's slot k above
    }
    __syncthreads();
    const int aS = NSTAT + NFW_NACC * n_h;
    for (int k = tid; k < a.A; k += WG) {
      if (k < aS) { out[k] = s_acc[k]; continue; }
      const int sidx = (k - aS) / SER_NACC, kk = (k - aS) % SER_NACC;
      const float* r = s_acc + aS + SER_NACC * sidx;
      const float raw[S5_N] = {r[SERA_CX], r[SERA_CY], r[SERA_L], r[SERA_INVN], r[SERA_IE]};
      out[k] = cluster_sersic5_finish(dS + SERP * sidx, raw, kk);
    }
  }
}

}  // namespace glk
