// gl_clusterw.hip.h -- the cluster kernel with the COMPONENTS dealt over the workgroup's four waves (round 4).
//
// gl_cluster_kernel (gl_cluster.hip.h) gives every lane one pixel pair and ALL components: 152 gradient sums per lane cannot own
// registers, so four sums share one through a 4 x 4 transpose-reduction per tile (~300 of its 1022 vector instructions per pixel),
// 28 components' constants do not fit the scalar file (160 SGPR spills through v_readlane / v_writelane) and the 168 + 38 VGPRs of
// state and running sums leave two waves per SIMD (PMC: a third of the wave-cycles wait on a dependency).
//
// Here the four waves of a workgroup walk the SAME 128 pixels (64 lanes x one pixel pair) and each owns a quarter of the
// components -- halos h = wave, wave + 4, ..., sources s = wave, wave + 4, ... (tf/simulator.py:72-78, 128-138 are sums over
// components, so the split is exact up to summation order):
//   * a wave's <= 2 halos x 4 + <= 5 sources x 5 (8 elliptical) sums are plain per-lane accumulators: one packed add per sum and
//     pixel pair, no transposes, no selects; one wave reduction per sum per CHUNK;
//   * its constants (<= 48 dwords) arrive through scalar loads, a component at a time, from neutral-padded slots: no count guards
//     in the pixel loop, 26 SGPR spills where the pixel-split kernel has 160;
//   * the NFW function comes from a table in s = X^2 (nfw_fwd_s below): no square root, no reciprocal, ~30 instead of ~85
//     vector instructions per halo and pixel pair, and a slope 80 x more accurate than the node-pair table's;
//   * the three quantities that couple the components -- the deflection sum, the model image, the cotangent of beta -- are
//     exchanged through LDS: each wave writes its partial (16 / 8 / 16 bytes per lane), one barrier, every wave reads the four
//     partials and adds them in wave order (so all four hold bitwise the same beta, image and cotangent; fixed order:
//     reproducible).  Three barriers per 128 pixels; the other workgroups of the CU (3-4 per CU) run through them;
//   * the pixel statistics (chi^2, d loglike / d image: ~35 instructions per pixel pair) are computed by all four waves.
// Same derived-constant layout, accumulator row, partial rows and finalize as gl_cluster_kernel.
#pragma once
#include "gl_cluster.hip.h"

namespace glk {

constexpr int CW_PX = 128;  // pixels per step of a workgroup: 64 lanes x one pixel pair, the same for its four waves
// LDS of the exchange: alpha partials [4][64] float4, g_beta partials [4][64] float4, image partials [4][64] float2
constexpr size_t CW_XCHG_FLOATS = 4 * 64 * 4 + 4 * 64 * 4 + 4 * 64 * 2;

// A component's constant block as a pointer into CONSTANT memory (address space 4: nothing writes the derived rows while a main
// kernel runs, and only such loads stay scalar next to an opaque asm; a laundered pointer is also opaque to the compiler's
// address-space inference, and a flat pointer is loaded through the vector path).
typedef const float __attribute__((address_space(4)))* cw_gptr;
__device__ __forceinline__ cw_gptr cw_launder(cw_gptr p) {
  unsigned long long u = (unsigned long long)p;
  asm volatile("" : "+s"(u));
  return (cw_gptr)u;
}
#ifndef CW_NOFENCE
#define CW_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define CW_FENCE() ((void)0)
#endif

// ---- NFW forward on a pixel pair from the table in s = X^2 (gl_host_tables.h::build_nfw_table_s) ----------------------------
// alpha = K0 h(X) d with X = R / Rs, and the VJP needs h'(X) only through  w = K0 h'(X) / (Rs R) = 2 K0 H'(s) / Rs^2  (H(s) = h(sqrt s),
// s = r^2 / Rs^2) and  uu = K0 h'(X) X / Rs = w r^2 / Rs -- so no square root, no reciprocal, no transcendental at all: the
// interval and the position inside it come from the bits of s, four 4-byte LDS reads per pixel fetch the interval's cubic, five
// packed multiply-adds give H and dH/dtau, the exponent bits give dtau/ds.  ~30 vector instructions per pixel pair where
// nfw_fwd_c (sqrt, two reciprocals, the clamps' selects, node pairs shuffled into pixel pairs) takes ~85.
// Lanes outside the table (X outside [2^-6, 2^6)), within 4 ulp of s = 1 (the reference's g(1) = 1 point, nfw.py:38) or within
// 1e-7 of the centre (nfw.py:26) take the closed form with the reference's clamps, lane by lane, exactly as nfw_fwd_c does.
constexpr int NFWS_LOG2_LO = -12, NFWS_LOG2_HI = 12, NFWS_PER_OCT = 64, NFWS_N = (NFWS_LOG2_HI - NFWS_LOG2_LO) * NFWS_PER_OCT;
constexpr int CW_NEUTRAL_OFF = 2 * NFW_TAB_NODES;      // floats into MainArgs::nfw_tab: [h(X) table | neutral blocks | H(s) table]
constexpr int CW_TABS_OFF = CW_NEUTRAL_OFF + 4 + 16;
static_assert(CW_TABS_OFF % 4 == 0, "the H(s) table is copied to LDS with 16-byte loads");

__device__ __forceinline__ void nfw_slow_lane(float r2, float invrs, float K0, float& h, float& w, float& uu) {
  const float R0 = glm::sqrt_(r2);
  const float iR0 = r2 > 0.f ? glm::rcp(R0) : 0.f;
  const float X0 = __builtin_fmaxf(R0, 1e-7f) * invrs;  // nfw.py:26
  const float X = __builtin_fmaxf(X0, 1e-6f);           // nfw.py:37
  const float iX = glm::rcp(X);
  float g, gp;
  nfw_gw<float>(X, g, gp);
  const float i2 = iX * iX;
  h = g * i2;
  const float hp = gp * i2 - (h + h) * iX;
  const float p = X0 > 1e-6f ? hp * K0 : 0.f;
  w = R0 > 1e-7f ? p * invrs * iR0 : 0.f;
  uu = p * X0 * invrs;
}

template <class P>
__device__ __forceinline__ void nfw_fwd_s(P d, const float* __restrict__ s_tab, v2f x, v2f y, v2f& bx, v2f& by, NfwStateC<v2f>& st) {
  using V = v2f;
  const float invrs = d[NFW_INVRS], K0 = d[NFW_K0];
  const float invrs2 = invrs * invrs, w0 = 2.f * K0 * invrs2;  // wave-uniform (two vector instructions per halo and step)
  const V dx = x - d[NFW_CX], dy = y - d[NFW_CY];
  const V r2 = dx * dx + dy * dy;
  const V s = r2 * invrs2;
  const float s0 = s.x, s1 = s.y;  // (element copies first: see nfw_h_pair)
  const unsigned b0 = __float_as_uint(s0), b1 = __float_as_uint(s1);
  constexpr unsigned BASE = (unsigned)(127 + NFWS_LOG2_LO) << 6;  // (bits >> 17) of 2^NFWS_LOG2_LO
  const unsigned i0 = (b0 >> 17) - BASE, i1 = (b1 >> 17) - BASE;  // s = 0, negative zero, NaN: out of range
  const bool slow0 = !(i0 < (unsigned)NFWS_N) || (b0 - 0x3F7FFFFCu) < 8u || !(r2.x > 1e-14f);
  const bool slow1 = !(i1 < (unsigned)NFWS_N) || (b1 - 0x3F7FFFFCu) < 8u || !(r2.y > 1e-14f);
  const unsigned j0 = min(i0, (unsigned)NFWS_N - 1), j1 = min(i1, (unsigned)NFWS_N - 1);
  const V c0{s_tab[j0], s_tab[j1]}, c1{s_tab[NFWS_N + j0], s_tab[NFWS_N + j1]};
  const V c2{s_tab[2 * NFWS_N + j0], s_tab[2 * NFWS_N + j1]}, c3{s_tab[3 * NFWS_N + j0], s_tab[3 * NFWS_N + j1]};
  const V tau{(float)(b0 & 0x1FFFFu), (float)(b1 & 0x1FFFFu)};  // the low 17 mantissa bits: exact
  const V p1 = __builtin_elementwise_fma(c3, tau, c2), p2 = __builtin_elementwise_fma(p1, tau, c1);
  V H = __builtin_elementwise_fma(p2, tau, c0);
  const V q2 = __builtin_elementwise_fma(c3, tau, p1), dH = __builtin_elementwise_fma(q2, tau, p2);  // dH/dtau
  // dtau/ds = 2^(23 - e): a power of two built from the exponent bits
  const V dtds{__uint_as_float(0x8A800000u - (b0 & 0x7F800000u)), __uint_as_float(0x8A800000u - (b1 & 0x7F800000u))};
  V w = dH * dtds * w0;
  V uu = w * (r2 * invrs);
  if (slow0) {
    float h_, w_, u_;
    nfw_slow_lane(r2.x, invrs, K0, h_, w_, u_);
    H.x = h_; w.x = w_; uu.x = u_;
  }
  if (slow1) {
    float h_, w_, u_;
    nfw_slow_lane(r2.y, invrs, K0, h_, w_, u_);
    H.y = h_; w.y = w_; uu.y = u_;
  }
  st.h = H;
  st.w = w;
  st.uu = uu;
  const V a = H * K0;
  bx -= a * dx;
  by -= a * dy;
}

// ---- lens policies: what a wave does in the two lens phases of a step -----------------------------------------------------------
// init (before the first barrier of the kernel), fwd (adds MINUS this wave's share of the deflection of the pixel pair), vjp (with
// the cotangent of alpha = -d loglike / d beta, summed over the waves), finish (parks the wave's sums in the workgroup's row or in
// the 16-float scratch), finish2 (after a barrier: sums that need all four waves).

// N x NFW halos (BASELINE configs 4 / 5): halo h = wave + 4 i, constants through scalar loads from neutral-padded slots
template <int HPW> struct CwLensNfw {
  using V = v2f;
  static constexpr int kLdsFloats = 4 * NFWS_N;  // the cubics of H(s), coefficient planes
  static constexpr int NFWP = (NFW_ND + 3) & ~3;
  cw_gptr pH[HPW];
  NfwStateC<V> hst[HPW];
  V acc[HPW][NFW_NACC];
  const float* s_tab;
  __device__ __forceinline__ void init(const MainArgs& a, float* lds, int tid, int wave, int b, int n_lens, const float* gder) {
    s_tab = lds;
    const float4* __restrict__ src = reinterpret_cast<const float4*>(a.nfw_tab + CW_TABS_OFF);  // (16-byte aligned: CW_TABS_OFF % 4 == 0)
    for (int i = tid; i < NFWS_N; i += WG) reinterpret_cast<float4*>(lds)[i] = src[i];
    const float* neutral = a.neutral;
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
      pH[i] = (cw_gptr)(wave + 4 * i < n_lens ? gder + NFWP * (wave + 4 * i) : neutral);
#pragma unroll
      for (int k = 0; k < NFW_NACC; ++k) acc[i][k] = V(0.f);
    }
  }
  __device__ __forceinline__ void fwd(V x, V y, V& pax, V& pay) {
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
      nfw_fwd_s<cw_gptr>(cw_launder(pH[i]), s_tab, x, y, pax, pay, hst[i]);
      CW_FENCE();
    }
  }
  __device__ __forceinline__ void vjp(V x, V y, V gx, V gy) {
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
      V va[NFW_NACC];
      nfw_vjp_c<V, cw_gptr>(cw_launder(pH[i]), x, y, gx, gy, hst[i], va);
#pragma unroll
      for (int k = 0; k < NFW_NACC; ++k) acc[i][k] += va[k];
      CW_FENCE();
    }
  }
  __device__ __forceinline__ void finish(const MainArgs&, float* s_row, float*, int wave, bool last, int n_lens) {
#pragma unroll
    for (int i = 0; i < HPW; ++i) {
      const int h = wave + 4 * i;
#pragma unroll
      for (int k = 0; k < NFW_NACC; ++k) {
        const float v = wave_sum63(acc[i][k].x + acc[i][k].y);
        if (last && h < n_lens) s_row[NSTAT + NFW_NACC * h + k] = v;
      }
    }
  }
  __device__ __forceinline__ void finish2(float*, const float*, int) {}
};

// (Round 4 also built a policy for the cluster-lens workload C6 -- free-standing dPIE halos + one galaxy catalogue with its 200
// members dealt over the four waves, tangents contracted per wave -- and measured it SLOWER than the interpreter: 4.54 vs 4.31 ms
// per 128 samples.  The member loop is 143 vector instructions per member and pixel pair in both kernels = 89 % of the
// interpreter's 16 100 instructions per pixel, so serving the 20 sources efficiently can save 6 % at most, and the interpreter
// runs that loop at four waves per SIMD (120 VGPRs, VALU busy 0.90) where this kernel's per-wave accumulators leave two.)

template <int MODE, class LENS, int SPW, bool ELL, int WAVES>
__global__ void __launch_bounds__(WG, WAVES) gl_clusterw_kernel(MainArgs a, int n_lens, int n_s) {
  static_assert(MODE == IMG_BWD || MODE == LL_GRAD, "gradient modes only (forward modes: gl_main_kernel)");
  using V = v2f;
  constexpr int NFWP = (NFW_ND + 3) & ~3;
  constexpr int NSA = ELL ? SER_NACC : S5_N;  // sums per source
  extern __shared__ float smem[];
  float4* s_xa = reinterpret_cast<float4*>(smem);  // [4][64]: -(sum of this wave's alpha) of the lane's pixel pair (x0, x1, y0, y1)
  float4* s_xg = s_xa + 4 * 64;                    // [4][64]: this wave's part of d loglike / d beta
  float2* s_xm = reinterpret_cast<float2*>(s_xg + 4 * 64);  // [4][64]: this wave's part of the model image
  float* s_lens = reinterpret_cast<float*>(s_xm + 4 * 64);  // the lens policy's tables
  float* s_aux = s_lens + LENS::kLdsFloats;                  // [16]: sums that need all four waves
  float* s_row = s_aux + 16;                                  // [A]: the workgroup's accumulator row (epilogue)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = a.order ? a.order[blockIdx.y] : blockIdx.y, chunk = blockIdx.x;
  // this sample's derived constants: wave-uniform addresses -> scalar loads
  const float* __restrict__ gder = a.derived + (size_t)b * a.D;
  LENS lens;
  lens.init(a, s_lens, tid, wave, b, n_lens, gder);
  for (int i = tid; i < a.A; i += WG) s_row[i] = 0.f;
  __syncthreads();
  // A slot beyond the model's counts points at a neutral block (zero amplitude; behind the NFW table): it adds exact zeros to
  // the deflection, the image and the cotangent of beta, and its own sums are never written -- the pixel loop carries no count
  // guards.  The pointers are laundered per step: left alone the compiler hoists all 48 constants out of the loop and spills
  // them (178 SGPRs, 111 VGPRs).  Also tried and dropped (round 4, each built and measured at C4: 0.946 ms as shipped): the constants
  // held in scalar registers for the whole chunk by hand-written s_load (148-176 spilled SGPRs: the scalar file also holds the
  // plane bases, the packed instructions' literals and the lane masks); the sources' loads software-pipelined by hand, two
  // buffers in turn, the load of source j + 1 in flight under source j's arithmetic (0.959 ms: the scalar round trips were not
  // what the waves wait for); the next step's grid / observation lines pulled into L1 a phase ahead (0.986 ms); wave-uniform
  // count guards instead of neutral blocks (1.05 ms).
  const float* neutral = a.neutral;  // [NFW block (4) | Sersic block (16)]
  cw_gptr pS[SPW];
  int aS[SPW];  // accumulator slot of source wave + 4 j inside the row (-1: no such source)
#pragma unroll
  for (int j = 0; j < SPW; ++j) {
    const bool on = wave + 4 * j < n_s;
    const CompDesc cd = a.comps[n_lens + (on ? wave + 4 * j : 0)];
    pS[j] = (cw_gptr)(on ? gder + cd.d_off : neutral + NFWP);
    aS[j] = on ? cd.a_off : -1;
  }
  V accS[SPW][NSA];
#pragma unroll
  for (int j = 0; j < SPW; ++j)
#pragma unroll
    for (int k = 0; k < NSA; ++k) accS[j][k] = V(0.f);
  const bool has_err = a.err != nullptr, has_mask = a.mask != nullptr, has_pix = a.pix != nullptr;
  float st0 = 0.f, st1 = 0.f;  // (one wave's sums, updated behind a branch: as pairs the compiler spilled them)
  const int p0 = chunk * a.chunk;
  const int p1 = min(p0 + a.chunk, a.N);
  const int my = tid;  // = wave * 64 + lane

  auto step = [&](int base, auto check_tag) {
    constexpr bool CHECK = decltype(check_tag)::value;
    unsigned jj[2], pidx[2];
    bool valid[2];
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      int j = base + w * 64 + lane;
      valid[w] = CHECK ? (j < p1) : true;
      jj[w] = (unsigned)(valid[w] ? j : p1 - 1);
      pidx[w] = (CHECK && has_pix) ? (unsigned)a.pix[jj[w]] : jj[w];  // CHECK=false steps run only without a pixel list
    }
    const unsigned jo0 = jj[0] << 2, jo1 = jj[1] << 2, po0 = pidx[0] << 2, po1 = pidx[1] << 2;
    auto ldf = [](const float* base, unsigned byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
    const V x = V{ldf(a.gx, jo0), ldf(a.gx, jo1)}, y = V{ldf(a.gy, jo0), ldf(a.gy, jo1)};
    V vmask = V(1.f);
    if (CHECK) vmask = V{valid[0] ? 1.f : 0.f, valid[1] ? 1.f : 0.f};
    using PS = cw_gptr;
    cw_gptr cS[SPW];
#pragma unroll
    for (int j = 0; j < SPW; ++j) cS[j] = cw_launder(pS[j]);
    // ---- ray-shoot: this wave's lenses (tf/simulator.py:72-78) ----
    V pax = V(0.f), pay = V(0.f);  // -(sum of this wave's alpha)
    lens.fwd(x, y, pax, pay);
    s_xa[my] = float4{pax.x, pax.y, pay.x, pay.y};
    __syncthreads();
    V bx = x, by = y;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float4 t = s_xa[w * 64 + lane];
      bx += V{t.x, t.y};
      by += V{t.z, t.w};
    }
    // ---- render this wave's sources at beta (tf/simulator.py:128-138) ----
    SerStateC<V> sst[SPW];
    V sL2[ELL ? 1 : SPW];
    V pm = V(0.f);
#pragma unroll
    for (int j = 0; j < SPW; ++j) {
      const PS d = cS[j];
      pm += sersic_fwd_c<V, ELL, PS>(d, bx, by, sst[j], ELL ? nullptr : &sL2[ELL ? 0 : j]);
      CW_FENCE();
    }
    s_xm[my] = float2{pm.x, pm.y};
    __syncthreads();
    // the planes the statistics need (requested here: any earlier they hold registers through the source phase)
    V o = V(0.f), wgt = vmask, e = V(1.f), gin = V(0.f);
    if (MODE == IMG_BWD) {
      const float* row = a.gimg + (size_t)b * a.img_stride;
      gin = V{row[pidx[0]], row[pidx[1]]};
    } else {
      o = V{ldf(a.obs, po0), ldf(a.obs, po1)};
      if (CHECK && has_mask) wgt = wgt * V{ldf(a.mask, po0), ldf(a.mask, po1)};
      if (has_err) e = V{ldf(a.err, po0), ldf(a.err, po1)};
    }
    V m = V(0.f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float2 t = s_xm[w * 64 + lane];
      m += V{t.x, t.y};
    }
    auto nanp = m != m;
    m = (nanp ? V(0.f) : m) * a.out_scale;  // NaN -> 0 (tf/simulator.py:140), then x det(T) (:156)
    V gm;
    if (MODE == IMG_BWD) {
      gm = nanp ? V(0.f) : (CHECK ? gin * vmask : gin) * a.out_scale;
    } else {
      V dmo = m - o;
      V s2 = has_err ? e * e : m * a.inv_t + a.bg2;  // tf/model.py:92-95
      V is2 = rcp(s2);
      if (wave == 0) {  // the statistics themselves: one wave's sums (the others need only the cotangent)
        V nm = vlog<V>(s2 * (float)(2 * kPi));
        V c2 = __builtin_elementwise_fma(nm, V(0.f), dmo * dmo * is2);  // + 0 * nm: a NaN sigma reaches chi^2 like in the reference
        if (CHECK) {
          auto use = wgt != V(0.f);
          st0 += hsum(use ? c2 * wgt : V(0.f));
          st1 += hsum(use ? nm * wgt : V(0.f));
        } else {
          st0 += hsum(c2);
          st1 += hsum(nm);
        }
      }
      // (is2 x inv_t x 0.5 as two packed multiplies by a scalar and an inline constant: 0.5 inv_t itself is a product of
      // wave-uniform floats, which only the vector unit can form -- the compiler kept it in a register pair and spilled it)
      V g = has_err ? -(dmo * is2) : (dmo * dmo * is2 - 1.f) * (is2 * a.inv_t * 0.5f) - dmo * is2;
      gm = nanp ? V(0.f) : (CHECK ? g * wgt : g) * a.out_scale;
    }
    // ---- source VJPs: parameter gradients and this wave's part of the cotangent of beta ----
    V gbx = V(0.f), gby = V(0.f);
#pragma unroll
    for (int j = 0; j < SPW; ++j) {
      const PS d = cS[j];
      if constexpr (ELL) {
        V va[SER_NACC];
        sersic_vjp_c<V, true, PS>(d, bx, by, sst[j], gm, va, gbx, gby);
#pragma unroll
        for (int k = 0; k < SER_NACC; ++k) accS[j][k] += va[k];
      } else {
        V va[S5_N];
        sersic_vjp5_keep_c<V, PS>(d, bx, by, sst[j], sL2[ELL ? 0 : j], gm, va, gbx, gby);
#pragma unroll
        for (int k = 0; k < S5_N; ++k) accS[j][k] += va[k];
      }
      CW_FENCE();
    }
    s_xg[my] = float4{gbx.x, gbx.y, gby.x, gby.y};
    __syncthreads();
    V tgx = V(0.f), tgy = V(0.f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float4 t = s_xg[w * 64 + lane];
      tgx += V{t.x, t.y};
      tgy += V{t.z, t.w};
    }
    // ---- lens VJPs with the cotangent -g_beta (beta = x - sum alpha) ----
    tgx = -tgx;
    tgy = -tgy;
    // the grid coordinates are read again (L1 hits) instead of holding four registers through the source phases
    const V xh = V{ldf(a.gx, jo0), ldf(a.gx, jo1)}, yh = V{ldf(a.gy, jo0), ldf(a.gy, jo1)};
    lens.vjp(xh, yh, tgx, tgy);
  };
  {
    const bool plain = !has_mask && !has_pix;
    int base = p0;
    if (plain)
      for (; base + CW_PX <= p1; base += CW_PX) step(base, std::false_type{});
    for (; base < p1; base += CW_PX) step(base, std::true_type{});
  }
  // ---- epilogue: one wave reduction per sum; lane 63 parks it in the workgroup's row (every slot has one owner) ----
  const bool last = lane == 63;
  lens.finish(a, s_row, s_aux, wave, last, n_lens);
#pragma unroll
  for (int j = 0; j < SPW; ++j) {
#pragma unroll
    for (int k = 0; k < NSA; ++k) {
      float v = wave_sum63(accS[j][k].x + accS[j][k].y);
      int slot = k;
      if constexpr (ELL) {
        if (k == SERA_INVN) v *= (float)kLn2;
      } else {
        constexpr int map[S5_N] = {SERA_CX, SERA_CY, SERA_L, SERA_INVN, SERA_IE};  // where the raw sums (Sx, Sy, A, D, B) are parked
        slot = map[k];
      }
      if (last && aS[j] >= 0) s_row[aS[j] + slot] = v;
    }
  }
  if (MODE == LL_GRAD && wave == 0) {
    const float c2 = wave_sum63(st0), nm = wave_sum63(st1);
    if (last) { s_row[0] = c2; s_row[1] = nm; }
  }
  __syncthreads();
  lens.finish2(s_row, s_aux, tid);
  if constexpr (!ELL) {
    // spherical sources: the row holds RAW sums (sersic_vjp5_c); each source's five sums -> its accumulator slots, in place
    // (one thread per source: reads its five raw sums, writes its eight slots)
    for (int s = tid; s < n_s; s += WG) {
      const CompDesc cd = a.comps[n_lens + s];
      float* r = s_row + cd.a_off;
      const float raw[S5_N] = {r[SERA_CX], r[SERA_CY], r[SERA_L], r[SERA_INVN], r[SERA_IE]};
#pragma unroll
      for (int kk = 0; kk < SER_NACC; ++kk) r[kk] = cluster_sersic5_finish(gder + cd.d_off, raw, kk);
    }
  }
  __syncthreads();
  float* out = a.partial + ((size_t)b * gridDim.x + chunk) * a.A;
  for (int k = tid; k < a.A; k += WG) out[k] = s_row[k];
}

}  // namespace glk
