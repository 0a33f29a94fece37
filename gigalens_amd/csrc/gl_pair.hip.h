// gl_pair.hip.h -- the specialised kernels in "pixel-pair" form: every per-pixel quantity is a 2-vector
// (pixel j, pixel j+256) so that the arithmetic of the whole fused forward+gradient pass -- not only the EPL
// series loop -- issues as packed fp32 instructions (v_pk_mul/add/fma_f32).  On CDNA4 a wave64 VALU
// instruction occupies its SIMD for 4 cycles whether it is packed or not (measured: SQ_ACTIVE_INST_VALU /
// SQ_INSTS_VALU = 4.17 with 21 % packed instructions), so packed issue is the only way to the 157 TFLOP/s
// vector peak, and this path is VALU-issue bound (profiles/r1_summary.json: VALU busy 0.87).
//
// The per-profile maths is the same as gl_profiles.h (same derived-constant layout, same accumulator
// layout, same finalize), written once over a value type V in {float, v2f}: comparisons yield lane masks and
// `m ? a : b` selects per lane.  Transcendentals (v_rcp/sqrt/log/exp) have no packed form and are applied
// per lane.  Supported components: EPL, SIE, SHEAR, SIS lenses; SERSIC / SERSIC_ELLIPSE lights.
#pragma once
#include "gl_static.hip.h"

namespace glk {

// bring the scalar wrappers into this scope so that they overload with the pair versions below
using glm::rcp;
using glm::sqrt_;
using glm::exp2_;
using glm::log2_;
using glm::atan_;
using glm::atanh_;

// ---- per-lane transcendentals on pairs -----------------------------------------------------------
__device__ __forceinline__ v2f rcp(v2f a) { return v2f{glm::rcp(a.x), glm::rcp(a.y)}; }
__device__ __forceinline__ v2f sqrt_(v2f a) { return v2f{glm::sqrt_(a.x), glm::sqrt_(a.y)}; }
__device__ __forceinline__ float rsq_(float a) { return __builtin_amdgcn_rsqf(a); }
__device__ __forceinline__ v2f rsq_(v2f a) { return v2f{__builtin_amdgcn_rsqf(a.x), __builtin_amdgcn_rsqf(a.y)}; }
__device__ __forceinline__ v2f exp2_(v2f a) { return v2f{glm::exp2_(a.x), glm::exp2_(a.y)}; }
__device__ __forceinline__ v2f log2_(v2f a) { return v2f{glm::log2_(a.x), glm::log2_(a.y)}; }
__device__ __forceinline__ v2f atan_(v2f a) { return v2f{glm::atan_(a.x), glm::atan_(a.y)}; }
__device__ __forceinline__ v2f atanh_(v2f a) { return v2f{glm::atanh_(a.x), glm::atanh_(a.y)}; }
template <class V> __device__ __forceinline__ V vexp(V x) {  // exp with the product rounding folded back in
  const float hi = (float)kLog2e;
  const float lo = (float)(kLog2e - (double)(float)kLog2e);
  V t = x * hi;
  V e = __builtin_elementwise_fma(x, V(hi), -t) + x * lo;
  V p = exp2_(t);
  return __builtin_elementwise_fma(p, e * (float)kLn2, p);
}
template <> __device__ __forceinline__ float vexp<float>(float x) { return glm::exp_(x); }
template <class V> __device__ __forceinline__ V vlog(V x) { return log2_(x) * (float)kLn2; }
template <class V> __device__ __forceinline__ V vmin(V a, V b) { return a < b ? a : b; }
template <class V> __device__ __forceinline__ V vmax(V a, V b) { return a > b ? a : b; }
__device__ __forceinline__ float hsum(float a) { return a; }
__device__ __forceinline__ float hsum(v2f a) { return a.x + a.y; }

// ---- EPL ------------------------------------------------------------------------------------------------
template <class V> struct EplStateV {
  V xr, yr, inv, invc, L2, P, Ox, Oy, Sx, Sy, Fx, Fy, Tx, Ty;
};

template <class V, bool GRAD>
__device__ __forceinline__ void epl_fwd_v(const float* d, const float* __restrict__ gd, V x, V y, V& bx, V& by,
                                          EplStateV<V>& st) {
  const float c = d[EPL_C], s = d[EPL_S], q = d[EPL_Q];
  V dx = x - d[EPL_CX], dy = y - d[EPL_CY];
  st.xr = dx * c + dy * s;
  st.yr = dy * c - dx * s;
  V X = st.xr * q;
  V r2 = X * X + st.yr * st.yr;
  // one transcendental instead of sqrt + 2 rcp: r = 1/R0 (inf at R0 = 0), and 1/clip(R0, 1e-10, 1e10) is the
  // clip of 1/R0 to [1e-10, 1e10] (epl.py:31); R0 itself is only needed through 1/R0
  V r = rsq_(r2);
  auto pos = r2 > V(0.f);
  st.inv = pos ? r : V(0.f);
  V Cs = pos ? X * r : V(1.f);
  V Ss = st.yr * st.inv;
  V iRc = vmin(vmax(r, V(1e-10f)), V(1e10f));
  st.invc = (iRc == r) ? r : V(0.f);  // clip_by_value passes gradient only inside the clamp
  V E2x = Cs * Cs - Ss * Ss, E2y = (Cs + Cs) * Ss;
  V Ex = Cs, Ey = Ss;
  st.Ox = Cs; st.Oy = Ss;
  if (GRAD) {
    st.Sx = Cs; st.Sy = Ss;
    st.Fx = V(0.f); st.Fy = V(0.f); st.Tx = V(0.f); st.Ty = V(0.f);
  }
  // scalar-loaded trip count and coefficients (wave-uniform address): SGPR operands, scalar loop control
  const int K = reinterpret_cast<const int*>(gd)[EPL_KI];
  const float4* __restrict__ gtab = reinterpret_cast<const float4*>(gd + EPL_TAB);
  auto step = [&](const float4 cc) {
    V tx = E2x * Ex - E2y * Ey;
    Ey = E2y * Ex + E2x * Ey;
    Ex = tx;
    st.Ox += cc.x * Ex; st.Oy += cc.x * Ey;
    if (GRAD) {
      st.Sx += cc.y * Ex; st.Sy += cc.y * Ey;
      st.Fx += cc.z * Ex; st.Fy += cc.z * Ey;
      st.Tx += cc.w * Ex; st.Ty += cc.w * Ey;
    }
  };
  int n = 1;
#if defined(GL_EXP_NOTABLE)  // timing experiment only: coefficients hoisted out of the loop (wrong numbers)
  const float4 c1 = gtab[1], c2 = gtab[2];
  for (; n + 1 <= K; n += 2) { step(c1); step(c2); }
  if (n <= K) step(c1);
#else
  for (; n + 1 <= K; n += 2) {
    const float4 ca = gtab[n], cb = gtab[n + 1];
    step(ca);
    step(cb);
  }
  if (n <= K) step(gtab[n]);
#endif
  st.L2 = log2_(iRc * d[EPL_B]);
  st.P = exp2_(st.L2 * d[EPL_TM1]) * d[EPL_P0];  // 2b/(1+q) (b/R)^(t-1), epl.py:55
  V arx = st.P * st.Ox, ary = st.P * st.Oy;
  bx -= arx * c - ary * s;
  by -= arx * s + ary * c;
}

template <class V>
__device__ __forceinline__ void epl_vjp_v(const float* d, V gx, V gy, const EplStateV<V>& st, V* acc) {
  const float c = d[EPL_C], s = d[EPL_S], q = d[EPL_Q], tm1 = d[EPL_TM1];
  V P = st.P;
  V arx = P * st.Ox, ary = P * st.Oy;
  V ax = arx * c - ary * s, ay = arx * s + ary * c;
  V grx = gx * c + gy * s, gry = gy * c - gx * s;
  V g_phi = gy * ax - gx * ay;
  V gP = grx * st.Ox + gry * st.Oy;
  V gOx = P * grx, gOy = P * gry;
  V g_ang = gOy * st.Sx - gOx * st.Sy;
  V g_t = gOx * st.Tx + gOy * st.Ty;
  V g_f = gOx * st.Fx + gOy * st.Fy;
  V gW_W = gP * P;
  g_t += gW_W * (st.L2 * (float)kLn2);
  V gWt = gW_W * tm1;
  V gR0 = -(gWt * st.invc);
  V Cs = st.xr * q * st.inv, Ss = st.yr * st.inv;  // (R0 == 0: inv = 0, and g_ang * inv = 0 as in the scalar code)
  V gai = g_ang * st.inv;
  V gX = gR0 * Cs - gai * Ss;
  V gyr = gR0 * Ss + gai * Cs;
  V gxr = gX * q;
  g_phi += gxr * st.yr - gyr * st.xr;
  acc[EPLA_CX] -= gxr * c - gyr * s;
  acc[EPLA_CY] -= gxr * s + gyr * c;
  acc[EPLA_PHI] += g_phi;
  acc[EPLA_Q] += gX * st.xr;
  acc[EPLA_B] += gWt;     // x 1/b in the epilogue
  acc[EPLA_T] += g_t;
  acc[EPLA_F] += g_f;
  acc[EPLA_P0] += gW_W;   // x 1/P0 in the epilogue (gP * W = gP * P / P0)
}

// ---- SIE / SHEAR / SIS (stateless: cheap to re-evaluate) ----------------------------------------------
template <class V> __device__ __forceinline__ void sie_fwd_v(const float* d, V x, V y, V& bx, V& by) {
  const float c = d[SIE_C], s = d[SIE_S], q = d[SIE_Q], sq = d[SIE_SQ], A = d[SIE_A];
  V dx = x - d[SIE_CX], dy = y - d[SIE_CY];
  V xr = dx * c + dy * s, yr = dy * c - dx * s;
  V ipsi = rcp(sqrt_(xr * xr * (q * q) + yr * yr));
  V arx = atan_(xr * ipsi * sq) * A, ary = atanh_(yr * ipsi * sq) * A;
  bx -= arx * c - ary * s;
  by -= arx * s + ary * c;
}
template <class V> __device__ __forceinline__ void sie_vjp_v(const float* d, V x, V y, V gx, V gy, V* acc) {
  const float c = d[SIE_C], s = d[SIE_S], q = d[SIE_Q], sq = d[SIE_SQ], A = d[SIE_A];
  V dx = x - d[SIE_CX], dy = y - d[SIE_CY];
  V xr = dx * c + dy * s, yr = dy * c - dx * s;
  V ipsi = rcp(sqrt_(xr * xr * (q * q) + yr * yr));
  V u = xr * ipsi * sq, v = yr * ipsi * sq;
  V fu = atan_(u), fv = atanh_(v);
  V arx = fu * A, ary = fv * A;
  V ax = arx * c - ary * s, ay = arx * s + ary * c;
  V grx = gx * c + gy * s, gry = gy * c - gx * s;
  V g_phi = gy * ax - gx * ay;
  V gu = grx * A * rcp(V(1.f) + u * u);
  V gv = gry * A * rcp(V(1.f) - v * v);
  V gpsi = -(gu * u + gv * v) * ipsi;
  V gxr = gu * ipsi * sq + gpsi * xr * ipsi * (q * q);
  V gyr = gv * ipsi * sq + gpsi * yr * ipsi;
  g_phi += gxr * yr - gyr * xr;
  acc[SIEA_CX] -= gxr * c - gyr * s;
  acc[SIEA_CY] -= gxr * s + gyr * c;
  acc[SIEA_PHI] += g_phi;
  acc[SIEA_Q] += gpsi * xr * xr * ipsi * q;
  acc[SIEA_SQ] += (gu * xr + gv * yr) * ipsi;
  acc[SIEA_A] += grx * fu + gry * fv;
}
template <class V> __device__ __forceinline__ void shear_fwd_v(const float* d, V x, V y, V& bx, V& by) {
  bx -= x * d[SHR_G1] + y * d[SHR_G2];
  by -= x * d[SHR_G2] - y * d[SHR_G1];
}
template <class V> __device__ __forceinline__ void shear_vjp_v(V x, V y, V gx, V gy, V* acc) {
  acc[0] += gx * x - gy * y;
  acc[1] += gx * y + gy * x;
}
template <class V> __device__ __forceinline__ void sis_fwd_v(const float* d, V x, V y, V& bx, V& by) {
  V dx = x - d[SIS_CX], dy = y - d[SIS_CY];
  V R0 = sqrt_(dx * dx + dy * dy);
  V a = (R0 == V(0.f)) ? V(0.f) : rcp(R0) * d[SIS_TE];
  bx -= a * dx;
  by -= a * dy;
}
template <class V> __device__ __forceinline__ void sis_vjp_v(const float* d, V x, V y, V gx, V gy, V* acc) {
  V dx = x - d[SIS_CX], dy = y - d[SIS_CY];
  V R0 = sqrt_(dx * dx + dy * dy);
  V iR = (R0 == V(0.f)) ? V(0.f) : rcp(R0);
  V a = iR * d[SIS_TE];
  V ga = gx * dx + gy * dy;
  V gR0 = -(ga * a * iR);
  acc[0] -= gx * a + gR0 * dx * iR;
  acc[1] -= gy * a + gR0 * dy * iR;
  acc[2] += ga * iR;
}

// ---- SERSIC ---------------------------------------------------------------------------------------------
template <class V> struct SerStateV { V a1, a2, r2, L2, u, E; };

template <class V> __device__ __forceinline__ V sersic_fwd_v(const float* d, V x, V y, SerStateV<V>& st) {
  const float c = d[SER_C], s = d[SER_S];
  V dx = x - d[SER_CX], dy = y - d[SER_CY];
  st.a1 = dx * c + dy * s;
  st.a2 = dy * c - dx * s;
  V xt1 = st.a1 * d[SER_SQ], xt2 = st.a2 * d[SER_ISQ];
  st.r2 = xt1 * xt1 + xt2 * xt2;
  st.L2 = log2_(st.r2) * 0.5f + d[SER_L2IRS];  // log2(R / R_sersic) without the square root
  st.u = exp2_(st.L2 * d[SER_INVN]);
  st.E = vexp<V>((st.u - 1.f) * -d[SER_BN]);
  return st.E * d[SER_IE];
}
template <class V, bool SRC>
__device__ __forceinline__ void sersic_vjp_v(const float* d, const SerStateV<V>& st, V gI, V* acc, V& gpx, V& gpy) {
  const float c = d[SER_C], s = d[SER_S], sq = d[SER_SQ], isq = d[SER_ISQ];
  V xt1 = st.a1 * sq, xt2 = st.a2 * isq;
  auto pos = st.r2 > V(0.f);
  V gE = gI * st.E;
  V tI = gE * d[SER_IE];
  V guu = -(tI * st.u) * d[SER_BN];
  V gL = guu * d[SER_INVN];
  V k = pos ? gL * rcp(st.r2) : V(0.f);
  V gxt1 = k * xt1, gxt2 = k * xt2;
  V ga1 = gxt1 * sq, ga2 = gxt2 * isq;
  V gdx = ga1 * c - ga2 * s, gdy = ga1 * s + ga2 * c;
  acc[SERA_CX] -= gdx;
  acc[SERA_CY] -= gdy;
  acc[SERA_PHI] += ga1 * st.a2 - ga2 * st.a1;
  acc[SERA_SQ] += gxt1 * st.a1 - gxt2 * st.a2 * (isq * isq);
  acc[SERA_L] += gL;
  acc[SERA_INVN] += pos ? guu * st.L2 : V(0.f);  // x ln2 in the epilogue
  acc[SERA_BN] -= tI * (st.u - 1.f);
  acc[SERA_IE] += gE;
  if (SRC) { gpx += gdx; gpy += gdy; }
}

// ---- the pair kernel ----------------------------------------------------------------------------------------
// V = v2f: each thread owns pixels (j, j + 256) of every 512-pixel tile; V = float: one pixel per thread.
template <int MODE, class V, int WAVES, class LK, class LLK, class SK>
__global__ void __launch_bounds__(WG, WAVES) gl_pair_kernel(MainArgs a) {
  constexpr int NL = LK::n, NLL = LLK::n, NS = SK::n, NLIGHT = NLL + NS;
  constexpr bool GRAD = (MODE == IMG_BWD || MODE == LL_GRAD);
  constexpr int W = sizeof(V) / sizeof(float);
  extern __shared__ float smem[];
  float* s_d = smem;
  float* s_acc = smem + ((a.D + 3) & ~3);
  const int tid = threadIdx.x;
  const int b = a.order ? a.order[blockIdx.y] : blockIdx.y, chunk = blockIdx.x;
  const CompDesc* __restrict__ comps = a.comps;
  const float* __restrict__ gder = a.derived + (size_t)b * a.D;
  {
    for (int i = tid; i < a.D; i += WG) s_d[i] = gder[i];
  }
  __syncthreads();
  constexpr int NACC_L = [] { int n = 0; for (int i = 0; i < NL; ++i) n += static_nacc(LK::kinds[i]); return n; }();
  constexpr int NACC_C = [] {
    int n = 0;
    for (int i = 0; i < NLL; ++i) n += static_nacc(LLK::kinds[i]);
    for (int i = 0; i < NS; ++i) n += static_nacc(SK::kinds[i]);
    return n;
  }();
  V accL[NACC_L > 0 ? NACC_L : 1];
  V accC[NACC_C > 0 ? NACC_C : 1];
#pragma unroll
  for (int k = 0; k < NACC_L; ++k) accL[k] = V(0.f);
#pragma unroll
  for (int k = 0; k < NACC_C; ++k) accC[k] = V(0.f);
  V st0 = V(0.f), st1 = V(0.f);
  const float* dL[NL > 0 ? NL : 1];
  const float* dC[NLIGHT > 0 ? NLIGHT : 1];
#pragma unroll
  for (int i = 0; i < NL; ++i) dL[i] = s_d + comps[i].d_off;
#pragma unroll
  for (int i = 0; i < NLIGHT; ++i) dC[i] = s_d + comps[NL + i].d_off;
  const bool has_err = a.err != nullptr, has_mask = a.mask != nullptr, has_pix = a.pix != nullptr;

  const int p0 = chunk * a.chunk;
  const int p1 = min(p0 + a.chunk, a.N);
  // One tile = W*256 pixels.  CHECK=false is the steady state (whole tile inside the chunk, no pixel mask):
  // no validity selects, no weights.  CHECK=true handles the ragged last tile and img_region weights.
  auto tile = [&](int base, auto check_tag) {
    constexpr bool CHECK = decltype(check_tag)::value;
    unsigned jj[W], pidx[W];
    bool valid[W];
    V x, y, vmask = V(1.f);
#pragma unroll
    for (int w = 0; w < W; ++w) {
      int j = base + w * WG + tid;
      valid[w] = CHECK ? (j < p1) : true;
      jj[w] = (unsigned)(valid[w] ? j : p1 - 1);
      pidx[w] = has_pix ? (unsigned)a.pix[jj[w]] : jj[w];
    }
    if constexpr (W == 2) {
      x = V{a.gx[jj[0]], a.gx[jj[1]]};
      y = V{a.gy[jj[0]], a.gy[jj[1]]};
      if (CHECK) vmask = V{valid[0] ? 1.f : 0.f, valid[1] ? 1.f : 0.f};
    } else {
      x = a.gx[jj[0]];
      y = a.gy[jj[0]];
      if (CHECK) vmask = valid[0] ? 1.f : 0.f;
    }
    V bx = x, by = y, m = V(0.f);
    EplStateV<V> est[NL > 0 ? NL : 1];
    SerStateV<V> sst[NLIGHT > 0 ? NLIGHT : 1];
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int kind = LK::kinds[i];
      if constexpr (kind == K_EPL) epl_fwd_v<V, GRAD>(dL[i], gder + comps[i].d_off, x, y, bx, by, est[i]);
      else if constexpr (kind == K_SIE) sie_fwd_v<V>(dL[i], x, y, bx, by);
      else if constexpr (kind == K_SHEAR) shear_fwd_v<V>(dL[i], x, y, bx, by);
      else sis_fwd_v<V>(dL[i], x, y, bx, by);
    }, std::make_integer_sequence<int, NL>{});
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr bool src = i >= NLL;
      m += sersic_fwd_v<V>(dC[i], src ? bx : x, src ? by : y, sst[i]);
    }, std::make_integer_sequence<int, NLIGHT>{});
    auto nanp = m != m;
    m = (nanp ? V(0.f) : m) * a.out_scale;  // NaN -> 0 (tf/simulator.py:140), then x det(T) (:156)
    if (MODE == IMG_FWD) {
      float* row = a.img + (size_t)b * a.img_stride;
      if constexpr (W == 2) {
        if (valid[0]) row[pidx[0]] = m.x;
        if (valid[1]) row[pidx[1]] = m.y;
      } else {
        if (valid[0]) row[pidx[0]] = m;
      }
      return;
    }
    V gm;
    if (MODE == IMG_BWD) {
      const float* row = a.gimg + (size_t)b * a.img_stride;
      V g;
      if constexpr (W == 2) g = V{row[pidx[0]], row[pidx[1]]}; else g = row[pidx[0]];
      gm = nanp ? V(0.f) : (CHECK ? g * vmask : g) * a.out_scale;
    } else {
      V o, w = vmask, e = V(1.f);
      if constexpr (W == 2) {
        o = V{a.obs[pidx[0]], a.obs[pidx[1]]};
        if (CHECK && has_mask) w = w * V{a.mask[pidx[0]], a.mask[pidx[1]]};
        if (has_err) e = V{a.err[pidx[0]], a.err[pidx[1]]};
      } else {
        o = a.obs[pidx[0]];
        if (CHECK && has_mask) w = w * a.mask[pidx[0]];
        if (has_err) e = a.err[pidx[0]];
      }
      // tf/model.py:92-99; sigma^2 = bg^2 + m/t (no clip: negative -> NaN like sqrt of a negative)
      V dmo = m - o;
      V s2 = has_err ? e * e : m * a.inv_t + a.bg2;
      V is2 = rcp(s2);
      V nm = vlog<V>(s2 * (float)(2 * kPi));  // NaN for sigma^2 < 0, like log(2 pi sqrt(.)^2) in the reference
      V c2 = __builtin_elementwise_fma(nm, V(0.f), dmo * dmo * is2);  // + 0 * nm: carries that NaN into chi^2
      if (CHECK) {  // invalid lanes carry weight 0; "x * 0" would keep a NaN, so select instead
        auto use = w != V(0.f);
        st0 += use ? c2 * w : V(0.f);
        st1 += use ? nm * w : V(0.f);
      } else {
        st0 += c2;
        st1 += nm;
      }
      if (MODE == LL_GRAD) {
        V g = has_err ? -(dmo * is2) : (dmo * dmo * is2 - 1.f) * (is2 * (0.5f * a.inv_t)) - dmo * is2;
        gm = nanp ? V(0.f) : (CHECK ? g * w : g) * a.out_scale;
      }
    }
    if constexpr (GRAD) {
      V gbx = V(0.f), gby = V(0.f);
      static_for([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr bool src = i >= NLL;
        constexpr int off = [] {
          int n = 0;
          for (int j = 0; j < i; ++j) n += static_nacc(j < NLL ? LLK::kinds[j < NLL ? j : 0] : SK::kinds[j >= NLL ? j - NLL : 0]);
          return n;
        }();
        sersic_vjp_v<V, src>(dC[i], sst[i], gm, accC + off, gbx, gby);
      }, std::make_integer_sequence<int, NLIGHT>{});
      gbx = -gbx;
      gby = -gby;
      static_for([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr int kind = LK::kinds[i];
        constexpr int off = [] { int n = 0; for (int j = 0; j < i; ++j) n += static_nacc(LK::kinds[j]); return n; }();
        if constexpr (kind == K_EPL) epl_vjp_v<V>(dL[i], gbx, gby, est[i], accL + off);
        else if constexpr (kind == K_SIE) sie_vjp_v<V>(dL[i], x, y, gbx, gby, accL + off);
        else if constexpr (kind == K_SHEAR) shear_vjp_v<V>(x, y, gbx, gby, accL + off);
        else sis_vjp_v<V>(dL[i], x, y, gbx, gby, accL + off);
      }, std::make_integer_sequence<int, NL>{});
    }
  };
  {
    const bool plain = !has_mask;
    int base = p0;
    if (plain)
      for (; base + WG * W <= p1; base += WG * W) tile(base, std::false_type{});
    for (; base < p1; base += WG * W) tile(base, std::true_type{});
  }
  if (MODE == IMG_FWD) return;
  // ---- epilogue: per-sample scale factors deferred out of the pixel loop, lane sum, one reduction ----
  // accumulators live in registers for the whole chunk, so ONE full wave64 reduction per value per workgroup
  // is cheap: lane 63 of each wave stores its sums into the wave's own LDS row (no zero-fill, no read-modify-write)
  if (!GRAD) {  // forward-only: the gradient slots of the partial row are defined (zero), never garbage
    for (int i = tid; i < 4 * a.Apad; i += WG) s_acc[i] = 0.f;
    __syncthreads();
  }
  float* s_row = s_acc + (tid >> 6) * a.Apad;
  const bool last_lane = (tid & 63) == 63;
  auto put = [&](float v, int idx) {
    v = wave_sum63(v);
    if (last_lane) s_row[idx] = v;
  };
  put((MODE == LL_FWD || MODE == LL_GRAD) ? hsum(st0) : 0.f, 0);
  put((MODE == LL_FWD || MODE == LL_GRAD) ? hsum(st1) : 0.f, 1);
  if (last_lane) { s_row[2] = 0.f; s_row[3] = 0.f; }
  if constexpr (GRAD) {
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int kind = LK::kinds[i];
      constexpr int off = [] { int n = 0; for (int j = 0; j < i; ++j) n += static_nacc(LK::kinds[j]); return n; }();
      constexpr int G = static_nacc(kind);
      float tmp[G];
#pragma unroll
      for (int k = 0; k < G; ++k) tmp[k] = hsum(accL[off + k]);
      if constexpr (kind == K_EPL) {
        tmp[EPLA_B] *= dL[i][EPL_INVB];
        tmp[EPLA_P0] *= rcp(dL[i][EPL_P0]);
      }
#pragma unroll
      for (int k = 0; k < G; ++k) put(tmp[k], comps[i].a_off + k);
    }, std::make_integer_sequence<int, NL>{});
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int off = [] {
        int n = 0;
        for (int j = 0; j < i; ++j) n += static_nacc(j < NLL ? LLK::kinds[j < NLL ? j : 0] : SK::kinds[j >= NLL ? j - NLL : 0]);
        return n;
      }();
      constexpr int G = SER_NACC;
      float tmp[G];
#pragma unroll
      for (int k = 0; k < G; ++k) tmp[k] = hsum(accC[off + k]);
      tmp[SERA_INVN] *= (float)kLn2;
#pragma unroll
      for (int k = 0; k < G; ++k) put(tmp[k], comps[NL + i].a_off + k);
    }, std::make_integer_sequence<int, NLIGHT>{});
  }
  __syncthreads();
  float* out = a.partial + ((size_t)b * gridDim.x + chunk) * a.A;
  for (int k = tid; k < a.A; k += WG) {
    out[k] = (s_acc[k] + s_acc[a.Apad + k]) + (s_acc[2 * a.Apad + k] + s_acc[3 * a.Apad + k]);
  }
}

}  // namespace glk
