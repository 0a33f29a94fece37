// gl_static.hip.h -- compile-time-specialised variants of the main kernel.
//
// gl_main_kernel (gl_kernels.hip.h) interprets an arbitrary component list: it cannot keep gradient
// accumulators in registers across tiles (their index is a run-time value) and it has to re-evaluate each
// profile's forward pass inside its VJP.  For the model compositions that dominate real fits (and the
// BASELINE configs) the component list is a template parameter here, so that
//   * the component loops unroll at compile time -- no switch, no descriptor loads in the pixel loop;
//   * every gradient accumulator is a named register for the whole chunk and is reduced ONCE per workgroup;
//   * the forward pass leaves its intermediates ("state": the four EPL angular series, the Sersic
//     radius / exponentials) in registers and the VJP consumes them -- nothing is evaluated twice.
// Same per-profile maths (gl_profiles.h), same launch geometry, same partial/finalize protocol as the
// generic kernel; the two are cross-checked against each other and against the oracle in tests/.
#pragma once
#if defined(__HIPCC_RTC__) && !__has_include(<utility>)
// the run-time compiler of models with user-written profiles (gl_user.hip) has no host library: what the kernels use of
// <utility> / <type_traits>
namespace std {
template <class T, T v> struct integral_constant {
  static constexpr T value = v;
  typedef T value_type;
  typedef integral_constant type;
  constexpr operator T() const { return v; }
};
typedef integral_constant<bool, true> true_type;
typedef integral_constant<bool, false> false_type;
template <class T, T... Is> struct integer_sequence {};
template <class T, T N> using make_integer_sequence = __make_integer_seq<integer_sequence, T, N>;
template <bool B, class T, class F> struct conditional { typedef T type; };
template <class T, class F> struct conditional<false, T, F> { typedef F type; };
template <bool B, class T, class F> using conditional_t = typename conditional<B, T, F>::type;
}  // namespace std
#else
#include <utility>
#endif

#include "gl_kernels.hip.h"
#include "gl_shapelets.hip.h"

namespace glk {

template <int... Ks> struct KindList {
  static constexpr int n = sizeof...(Ks);
  static constexpr int kinds[sizeof...(Ks) + 1] = {Ks..., 0};
};

// A user-written profile inside a kind list of the run-time compiled specialised kernels (gl_user.hip): the code carries which of
// the model's bodies it is and its parameter count, so that the kernel is specialised on both --
//   USER_CODE + 2048 light + 32 body + n_params          (n_params <= 16, body < 64)
constexpr int USER_CODE = 0x1000;
__host__ __device__ constexpr bool is_user_code(int kind) { return kind >= USER_CODE; }
__host__ __device__ constexpr bool user_code_light(int kind) { return ((kind - USER_CODE) >> 11) & 1; }
__host__ __device__ constexpr int user_code_body(int kind) { return ((kind - USER_CODE) >> 5) & 63; }
__host__ __device__ constexpr int user_code_npar(int kind) { return (kind - USER_CODE) & 31; }
__host__ __device__ constexpr int user_code(bool light, int body, int npar) { return USER_CODE + (light ? 2048 : 0) + 32 * body + npar; }

__host__ __device__ constexpr int static_nacc(int kind) {
  if (is_user_code(kind)) return user_code_npar(kind);  // one gradient sum per parameter, straight from the body's duals
  return kind == K_EPL ? EPL_NACC : kind == K_SIE ? SIE_NACC : kind == K_NFW ? NFW_NACC : kind == K_SHEAR ? SHR_NACC
         : kind == K_SIS ? SIS_NACC : (kind == K_SERSIC || kind == K_SERSIC_ELLIPSE) ? SER_NACC
         : kind == K_SHAPELETS ? (SHPA_AMP + SH_MAXL) : 0;
}

template <class F, int... Is> __device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}

// ---- EPL with state -----------------------------------------------------------------------------
template <int T> struct EplState {
  float xr[T], yr[T], inv[T], Cs[T], Ss[T], iRc[T], L2[T], P[T];
  float Ox[T], Oy[T], Fx[T], Fy[T], Tx[T], Ty[T];
  bool inclamp[T];
};

// `gtab` is the same coefficient table in GLOBAL memory (this sample's row of the derived buffer): its
// address is wave-uniform, so the compiler reads it with scalar loads (s_load_dwordx4 -> SGPR operands) and
// the series loop issues no LDS/vector-memory instruction at all.
template <int T, bool GRAD>
__device__ __forceinline__ void epl_fwd_state(const float* d, const float* __restrict__ gd, const float (&x)[T],
                                              const float (&y)[T], float (&bx)[T], float (&by)[T], EplState<T>& st) {
  // Complex numbers are kept as (re, im) register pairs so that every instruction of the series loop is a
  // packed fp32 op (v_pk_mul_f32 / v_pk_fma_f32: two lanes-worth of FMA per issue slot, which is what the
  // 157 TFLOP/s vector peak of the chip assumes), with the coefficients as SGPR operands:
  //   E_{n+1} = 2 cos(2 theta) E_n - E_{n-1}                           1 packed op (three-term recurrence, see epl_fwd_v)
  //   O += c0 E ; F += c2 E ; Tt += c3 E                              3 packed ops (1 in forward-only mode)
  // (dOmega/dtheta = i S, S = sum (2n+1) c_n E_n = O + 2 f F because c_n ~ f^n: no separate S sum)
  v2f E[T], Pv[T], O[T], F[T], Tt[T];
  float twoc[T];
  const float c = d[EPL_C], s = d[EPL_S], q = d[EPL_Q];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float dx = x[t] - d[EPL_CX], dy = y[t] - d[EPL_CY];
    st.xr[t] = dx * c + dy * s;
    st.yr[t] = dy * c - dx * s;
    float X = q * st.xr[t];
    float R0 = sqrt_(X * X + st.yr[t] * st.yr[t]);
    bool pos = R0 > 0.f;
    st.inv[t] = pos ? rcp(R0) : 0.f;
    st.Cs[t] = pos ? X * st.inv[t] : 1.f;
    st.Ss[t] = st.yr[t] * st.inv[t];
    st.inclamp[t] = (R0 >= 1e-10f) && (R0 <= 1e10f);
    st.iRc[t] = rcp(clamp_(R0, 1e-10f, 1e10f));
    twoc[t] = 2.f * (st.Cs[t] * st.Cs[t] - st.Ss[t] * st.Ss[t]);
    E[t] = v2f{st.Cs[t], st.Ss[t]};
    Pv[t] = v2f{st.Cs[t], -st.Ss[t]};
    O[t] = E[t];
    F[t] = v2f{0.f, 0.f};
    Tt[t] = v2f{0.f, 0.f};
  }
  // trip count and coefficients come from GLOBAL memory at a wave-uniform address: scalar loads, scalar
  // loop control, SGPR operands -- the loop issues no vector-memory or LDS instruction.
  const int K = reinterpret_cast<const int*>(gd)[EPL_KI];
  const float4* __restrict__ gtab = reinterpret_cast<const float4*>(gd + EPL_TAB);
  auto add = [&](const float4 cc, int t, const v2f& e) {
    O[t] += cc.x * e;
    if (GRAD) {
      F[t] += cc.z * e;
      Tt[t] += cc.w * e;
    }
  };
  int n = 1;
  for (; n + 1 <= K; n += 2) {  // two terms per trip: one s_load_dwordx8, one wait; E and Pv swap roles
    const float4 ca = gtab[n], cb = gtab[n + 1];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      Pv[t] = twoc[t] * E[t] - Pv[t];
      add(ca, t, Pv[t]);
      E[t] = twoc[t] * Pv[t] - E[t];
      add(cb, t, E[t]);
    }
  }
  if (n <= K) {
    const float4 ca = gtab[n];
#pragma unroll
    for (int t = 0; t < T; ++t) add(ca, t, twoc[t] * E[t] - Pv[t]);
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    st.Ox[t] = O[t].x; st.Oy[t] = O[t].y;
    if (GRAD) {
      st.Fx[t] = F[t].x; st.Fy[t] = F[t].y;
      st.Tx[t] = Tt[t].x; st.Ty[t] = Tt[t].y;
    }
    st.L2[t] = log2_(d[EPL_B] * st.iRc[t]);
    st.P[t] = d[EPL_P0] * exp2_(d[EPL_TM1] * st.L2[t]);
    float arx = st.P[t] * st.Ox[t], ary = st.P[t] * st.Oy[t];
    bx[t] -= arx * c - ary * s;
    by[t] -= arx * s + ary * c;
  }
}

template <int T>
__device__ __forceinline__ void epl_vjp_state(const float* d, const float (&gx)[T], const float (&gy)[T],
                                              const EplState<T>& st, float* acc) {
  const float c = d[EPL_C], s = d[EPL_S], q = d[EPL_Q], tm1 = d[EPL_TM1], iP0 = rcp(d[EPL_P0]);
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float P = st.P[t];
    float arx = P * st.Ox[t], ary = P * st.Oy[t];
    float ax = arx * c - ary * s, ay = arx * s + ary * c;
    float grx = gx[t] * c + gy[t] * s, gry = gy[t] * c - gx[t] * s;
    float g_phi = gy[t] * ax - gx[t] * ay;
    float gP = grx * st.Ox[t] + gry * st.Oy[t];
    float gOx = P * grx, gOy = P * gry;
    float g_ang = (gOy * st.Ox[t] - gOx * st.Oy[t]) + d[EPL_F2] * (gOy * st.Fx[t] - gOx * st.Fy[t]);
    float g_t = gOx * st.Tx[t] + gOy * st.Ty[t];
    float g_f = gOx * st.Fx[t] + gOy * st.Fy[t];
    float gW_W = gP * P;
    g_t += gW_W * (st.L2[t] * (float)kLn2);
    float g_b = gW_W * tm1 * d[EPL_INVB];
    float gR0 = st.inclamp[t] ? -gW_W * tm1 * st.iRc[t] : 0.f;
    float gX = gR0 * st.Cs[t] - g_ang * st.Ss[t] * st.inv[t];
    float gyr = gR0 * st.Ss[t] + g_ang * st.Cs[t] * st.inv[t];
    float g_q = gX * st.xr[t];
    float gxr = gX * q;
    float gdx = gxr * c - gyr * s, gdy = gxr * s + gyr * c;
    g_phi += gxr * st.yr[t] - gyr * st.xr[t];
    acc[EPLA_CX] -= gdx;
    acc[EPLA_CY] -= gdy;
    acc[EPLA_PHI] += g_phi;
    acc[EPLA_Q] += g_q;
    acc[EPLA_B] += g_b;
    acc[EPLA_T] += g_t;
    acc[EPLA_F] += g_f;
    acc[EPLA_P0] += gW_W * iP0;  // gP * W,  W = P / P0
  }
}

// ---- Sersic with state --------------------------------------------------------------------------
struct SerState { float a1, a2, r2, L2, u, E; };

__device__ __forceinline__ float sersic_fwd_state(const float* d, float x, float y, SerState& st) {
  float dx = x - d[SER_CX], dy = y - d[SER_CY];
  float c = d[SER_C], s = d[SER_S];
  st.a1 = c * dx + s * dy;
  st.a2 = c * dy - s * dx;
  float xt1 = st.a1 * d[SER_SQ], xt2 = st.a2 * d[SER_ISQ];
  st.r2 = xt1 * xt1 + xt2 * xt2;
  float Rr = sqrt_(st.r2);
  st.L2 = log2_(Rr * d[SER_INVRS]);
  st.u = exp2_(st.L2 * d[SER_INVN]);
  st.E = exp_(-d[SER_BN] * (st.u - 1.f));
  return d[SER_IE] * st.E;
}
__device__ __forceinline__ void sersic_vjp_state(const float* d, const SerState& st, float gI, float* acc,
                                                 float& gpx, float& gpy) {
  const float c = d[SER_C], s = d[SER_S], sq = d[SER_SQ], isq = d[SER_ISQ];
  float xt1 = st.a1 * sq, xt2 = st.a2 * isq;
  bool pos = st.r2 > 0.f;
  float tI = gI * d[SER_IE] * st.E;
  float guu = -tI * d[SER_BN] * st.u;
  float gL = guu * d[SER_INVN];
  float k = pos ? gL * rcp(st.r2) : 0.f;
  float gxt1 = k * xt1, gxt2 = k * xt2;
  float ga1 = gxt1 * sq, ga2 = gxt2 * isq;
  float gdx = ga1 * c - ga2 * s, gdy = ga1 * s + ga2 * c;
  acc[SERA_CX] -= gdx;
  acc[SERA_CY] -= gdy;
  acc[SERA_PHI] += ga1 * st.a2 - ga2 * st.a1;
  acc[SERA_SQ] += gxt1 * st.a1 - gxt2 * st.a2 * isq * isq;
  acc[SERA_L] += gL;
  acc[SERA_INVN] += pos ? guu * st.L2 * (float)kLn2 : 0.f;
  acc[SERA_BN] -= tI * (st.u - 1.f);
  acc[SERA_IE] += gI * st.E;
  gpx += gdx;
  gpy += gdy;
}

// ---- the specialised kernel ------------------------------------------------------------------------
template <int MODE, int T, int WAVES, class LK, class LLK, class SK>
__global__ void __launch_bounds__(WG, WAVES) gl_static_kernel(MainArgs a) {
  constexpr int NL = LK::n, NLL = LLK::n, NS = SK::n, NLIGHT = NLL + NS;
  constexpr bool GRAD = (MODE == IMG_BWD || MODE == LL_GRAD);
  extern __shared__ float smem[];
  float* s_d = smem;
  float* s_acc = smem + ((a.D + 3) & ~3);
  const int tid = threadIdx.x;
  const int b = a.order ? a.order[blockIdx.y] : blockIdx.y, chunk = blockIdx.x;
  const CompDesc* __restrict__ comps = a.comps;
  {
    const float* src = a.derived + (size_t)b * a.D;
    for (int i = tid; i < a.D; i += WG) s_d[i] = src[i];
    if (MODE != IMG_FWD)
      for (int i = tid; i < a.ncols * a.Apad; i += WG) s_acc[i] = 0.f;
  }
  __syncthreads();
  // compile-time accumulator layout (register file); run-time a_off only at the final LDS add
  constexpr int NACC_L = [] { int n = 0; for (int i = 0; i < NL; ++i) n += static_nacc(LK::kinds[i]); return n; }();
  constexpr int NACC_C = [] {
    int n = 0;
    for (int i = 0; i < NLL; ++i) n += static_nacc(LLK::kinds[i]);
    for (int i = 0; i < NS; ++i) n += static_nacc(SK::kinds[i]);
    return n;
  }();
  float accL[NACC_L > 0 ? NACC_L : 1];
  float accC[NACC_C > 0 ? NACC_C : 1];
#pragma unroll
  for (int k = 0; k < NACC_L; ++k) accL[k] = 0.f;
#pragma unroll
  for (int k = 0; k < NACC_C; ++k) accC[k] = 0.f;
  float st[2] = {0.f, 0.f};

  const float* dL[NL > 0 ? NL : 1];
  const float* dC[NLIGHT > 0 ? NLIGHT : 1];
#pragma unroll
  for (int i = 0; i < NL; ++i) dL[i] = s_d + comps[i].d_off;
#pragma unroll
  for (int i = 0; i < NLIGHT; ++i) dC[i] = s_d + comps[NL + i].d_off;

  const float* __restrict__ gder = a.derived + (size_t)b * a.D;
  const int p0 = chunk * a.chunk;
  const int p1 = min(p0 + a.chunk, a.N);
  for (int base = p0; base < p1; base += WG * T) {
    float x[T], y[T], bx[T], by[T], m[T];
    int pidx[T];
    bool valid[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      int j = base + t * WG + tid;
      valid[t] = j < p1;
      int jj = valid[t] ? j : p1 - 1;
      x[t] = a.gx[jj];
      y[t] = a.gy[jj];
      pidx[t] = a.pix ? a.pix[jj] : jj;
      bx[t] = x[t]; by[t] = y[t]; m[t] = 0.f;
    }
    EplState<T> est[NL > 0 ? NL : 1];
    SerState sst[NLIGHT > 0 ? NLIGHT : 1][T];
    constexpr bool HAS_SHP = [] {
      for (int i = 0; i < NLL; ++i) if (LLK::kinds[i < NLL ? i : 0] == K_SHAPELETS) return true;
      for (int i = 0; i < NS; ++i) if (SK::kinds[i < NS ? i : 0] == K_SHAPELETS) return true;
      return false;
    }();
    ShpState<SH_CAP> hst[HAS_SHP ? NLIGHT : 1][HAS_SHP ? T : 1];
    // ---- ray-shoot ----
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int kind = LK::kinds[i];
      const float* d = dL[i];
      if constexpr (kind == K_EPL) {
        epl_fwd_state<T, GRAD>(d, gder + comps[i].d_off, x, y, bx, by, est[i]);
      } else {
#pragma unroll
        for (int t = 0; t < T; ++t) {
          float ax, ay;
          if constexpr (kind == K_SIE) sie_fwd(d, x[t], y[t], ax, ay);
          else if constexpr (kind == K_NFW) nfw_fwd(d, x[t], y[t], ax, ay);
          else if constexpr (kind == K_SHEAR) shear_fwd(d, x[t], y[t], ax, ay);
          else sis_fwd(d, x[t], y[t], ax, ay);
          bx[t] -= ax; by[t] -= ay;
        }
      }
    }, std::make_integer_sequence<int, NL>{});
    // ---- render ----
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int kind = i < NLL ? LLK::kinds[i < NLL ? i : 0] : SK::kinds[i >= NLL ? i - NLL : 0];
      constexpr bool src = i >= NLL;
      const float* d = dC[i];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        float px = src ? bx[t] : x[t], py = src ? by[t] : y[t];
        if constexpr (kind == K_SHAPELETS)
          m[t] += shp_fwd_state<SH_CAP>(d, gder + comps[NL + i].d_off + SHP_AMP, a.shp_tab, comps[NL + i].flags & 1u, px, py,
                                        hst[i][t]);
        else
          m[t] += sersic_fwd_state(d, px, py, sst[i][t]);
      }
    }, std::make_integer_sequence<int, NLIGHT>{});
    bool nanp[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      nanp[t] = isnan_(m[t]);
      m[t] = (nanp[t] ? 0.f : m[t]) * a.out_scale;
    }
    if (MODE == IMG_FWD) {
      float* row = a.img + (size_t)b * a.img_stride;
#pragma unroll
      for (int t = 0; t < T; ++t)
        if (valid[t]) row[pidx[t]] = m[t];
      continue;
    }
    float gm[T];
    if (MODE == IMG_BWD) {
      const float* row = a.gimg + (size_t)b * a.img_stride;
#pragma unroll
      for (int t = 0; t < T; ++t) gm[t] = (valid[t] && !nanp[t]) ? row[pidx[t]] * a.out_scale : 0.f;
    } else {
      const bool has_err = a.err != nullptr;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        float o = a.obs[pidx[t]];
        float w = a.mask ? a.mask[pidx[t]] : 1.f;
        float e = has_err ? a.err[pidx[t]] : 1.f;
        float c2, nm;
        chi2_terms(m[t], o, w, has_err, e, a.bg2, a.inv_t, c2, nm);
        if (valid[t]) { st[0] += c2; st[1] += nm; }
        if (MODE == LL_GRAD)
          gm[t] = (valid[t] && !nanp[t]) ? chi2_gm(m[t], o, w, has_err, e, a.bg2, a.inv_t) * a.out_scale : 0.f;
      }
    }
    if constexpr (GRAD) {
      float gbx[T], gby[T];
#pragma unroll
      for (int t = 0; t < T; ++t) { gbx[t] = 0.f; gby[t] = 0.f; }
      static_for([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr int kind = i < NLL ? LLK::kinds[i < NLL ? i : 0] : SK::kinds[i >= NLL ? i - NLL : 0];
        constexpr bool src = i >= NLL;
        constexpr int off = [] {
          int n = 0;
          for (int j = 0; j < i; ++j) n += static_nacc(j < NLL ? LLK::kinds[j < NLL ? j : 0] : SK::kinds[j >= NLL ? j - NLL : 0]);
          return n;
        }();
        const float* d = dC[i];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          float dgx = 0.f, dgy = 0.f;
          if constexpr (kind == K_SHAPELETS)
            shp_vjp_state<SH_CAP>(d, gder + comps[NL + i].d_off + SHP_AMP, comps[NL + i].flags & 1u, hst[i][t], gm[t],
                                  accC + off, dgx, dgy);
          else
            sersic_vjp_state(d, sst[i][t], gm[t], accC + off, dgx, dgy);
          if (src) { gbx[t] += dgx; gby[t] += dgy; }
        }
      }, std::make_integer_sequence<int, NLIGHT>{});
#pragma unroll
      for (int t = 0; t < T; ++t) { gbx[t] = -gbx[t]; gby[t] = -gby[t]; }
      static_for([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr int kind = LK::kinds[i];
        constexpr int off = [] { int n = 0; for (int j = 0; j < i; ++j) n += static_nacc(LK::kinds[j]); return n; }();
        const float* d = dL[i];
        if constexpr (kind == K_EPL) {
          epl_vjp_state<T>(d, gbx, gby, est[i], accL + off);
        } else {
#pragma unroll
          for (int t = 0; t < T; ++t) {
            if constexpr (kind == K_SIE) sie_vjp(d, x[t], y[t], gbx[t], gby[t], accL + off);
            else if constexpr (kind == K_NFW) nfw_vjp(d, x[t], y[t], gbx[t], gby[t], accL + off);
            else if constexpr (kind == K_SHEAR) shear_vjp(d, x[t], y[t], gbx[t], gby[t], accL + off);
            else sis_vjp(d, x[t], y[t], gbx[t], gby[t], accL + off);
          }
        }
      }, std::make_integer_sequence<int, NL>{});
    }
  }
  if (MODE == IMG_FWD) return;
  // ---- one reduction per workgroup: registers -> quad/row sum -> LDS columns -> partial row ----
  const AccCol ac = acc_col(s_acc, a.Apad, a.ncols, tid);
  if (MODE == LL_FWD || MODE == LL_GRAD) wave_acc<2>(st, ac, 0);
  if constexpr (GRAD) {
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int off = [] { int n = 0; for (int j = 0; j < i; ++j) n += static_nacc(LK::kinds[j]); return n; }();
      constexpr int G = static_nacc(LK::kinds[i]);
      float tmp[G];
#pragma unroll
      for (int k = 0; k < G; ++k) tmp[k] = accL[off + k];
      wave_acc<G>(tmp, ac, comps[i].a_off, comps[i].n_acc);
    }, std::make_integer_sequence<int, NL>{});
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int kind = i < NLL ? LLK::kinds[i < NLL ? i : 0] : SK::kinds[i >= NLL ? i - NLL : 0];
      constexpr int off = [] {
        int n = 0;
        for (int j = 0; j < i; ++j) n += static_nacc(j < NLL ? LLK::kinds[j < NLL ? j : 0] : SK::kinds[j >= NLL ? j - NLL : 0]);
        return n;
      }();
      constexpr int G = static_nacc(kind);
      float tmp[G];
#pragma unroll
      for (int k = 0; k < G; ++k) tmp[k] = accC[off + k];
      wave_acc<G>(tmp, ac, comps[NL + i].a_off, comps[NL + i].n_acc);
    }, std::make_integer_sequence<int, NLIGHT>{});
  }
  __syncthreads();
  float* out = a.partial + ((size_t)b * gridDim.x + chunk) * a.A;
  for (int k = tid; k < a.A; k += WG) {
    float v = 0.f;
    for (int j = 0; j < a.ncols; ++j) v += s_acc[j * a.Apad + k];
    out[k] = v;
  }
}

// the compositions with a compile-time-specialised kernel (dispatch: gl_launch.hip.h)
using L_EplShear = KindList<K_EPL, K_SHEAR>;
using L_Sie = KindList<K_SIE>;
using L_SieShear = KindList<K_SIE, K_SHEAR>;
using C_None = KindList<>;
using C_Sersic = KindList<K_SERSIC>;           // pair kernels: spherical fast path (every light profile of the model spherical)
using C_SersicE = KindList<K_SERSIC_ELLIPSE>;  // pair kernels: the general elliptical code, serves spherical members too
using C_Shapelets = KindList<K_SHAPELETS>;

enum StaticId { ST_NONE = 0, ST_EPLSHEAR_SERSIC, ST_EPLSHEAR_SERSIC_SERSIC, ST_SIE_SERSIC, ST_EPLSHEAR_SHAPELETS,
                ST_SIESHEAR_SERSIC_SERSIC, ST_EPLSHEAR_SERSIC_SHAPELETS /* shapelets-demo.ipynb: lens light + shapelet source */ };


}  // namespace glk
