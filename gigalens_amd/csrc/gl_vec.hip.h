// gl_vec.hip.h -- per-profile maths written once over a value type V in {float, v2f}.
//
// v2f is a pixel PAIR (pixel j, pixel j+256): arithmetic on it issues as packed fp32 instructions
// (v_pk_mul/add/fma_f32).  On CDNA4 a wave64 VALU instruction occupies its SIMD for 4 cycles whether it is packed
// or not (measured: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.2), so packed issue is the only way to the
// 157 TFLOP/s vector peak, and this path is VALU-issue bound.  Comparisons yield lane masks, `m ? a : b` selects
// per lane; transcendentals (v_rcp/rsq/log/exp) have no packed form and are applied per lane.
// Same derived-constant layout, accumulator layout and finalize as gl_profiles.h.
#pragma once
#include <hip/hip_runtime.h>

#include "gl_profiles.h"

namespace glk {
using namespace glp;

typedef float v2f __attribute__((ext_vector_type(2)));

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
__device__ __forceinline__ float floor_at(float a, float lo) { return __builtin_fmaxf(a, lo); }
__device__ __forceinline__ v2f floor_at(v2f a, float lo) { return v2f{__builtin_fmaxf(a.x, lo), __builtin_fmaxf(a.y, lo)}; }
__device__ __forceinline__ float clamp3(float a, float lo, float hi) { return __builtin_amdgcn_fmed3f(a, lo, hi); }
__device__ __forceinline__ v2f clamp3(v2f a, float lo, float hi) { return v2f{__builtin_amdgcn_fmed3f(a.x, lo, hi), __builtin_amdgcn_fmed3f(a.y, lo, hi)}; }
template <class V> __device__ __forceinline__ V vmin(V a, V b) { return a < b ? a : b; }
template <class V> __device__ __forceinline__ V vmax(V a, V b) { return a > b ? a : b; }
__device__ __forceinline__ float hsum(float a) { return a; }
__device__ __forceinline__ float hsum(v2f a) { return a.x + a.y; }

// ---- EPL ------------------------------------------------------------------------------------------------
// The angular series  Omega = sum_n c_n e^{i(2n+1)theta}  (epl.py:39-54; c_n real, per sample) and its derivatives w.r.t.
// f and t (same form with dc_n/df, dc_n/dt) are summed by CLENSHAW's backward recurrence: cos((2n+1)theta) and
// sin((2n+1)theta) obey the same three-term recurrence phi_{n+1} = 2 cos(2 theta) phi_n - phi_{n-1}, so ONE real sequence
//     b_k = c_k + 2 cos(2 theta) b_{k+1} - b_{k+2}        (k = K .. 0,  b_{K+1} = b_{K+2} = 0)
// serves both components:  sum c_n cos((2n+1)theta) = (b_0 - b_1) cos(theta),  sum c_n sin((2n+1)theta) = (b_0 + b_1) sin(theta)
// (phi_0 = cos / sin theta, phi_{-1} = cos(-theta) / sin(-theta)).  Two packed instructions per term and series for a pixel
// pair, against two for advancing E_n plus two per series when the terms e^{i(2n+1)theta} are formed explicitly: 6 instead
// of 8 per term in gradient mode (three series), 2 instead of 4 forward-only.  The small terms are added first.
template <class V> struct EplStateV {
  V xr, yr, inv, invc, L2, P, Cs, Ss, arx, ary;  // (arx, ary) = P Omega: the deflection in the lens frame
  V f0, f1, t0, t1;  // Clenshaw tails (b_0, b_1) of the d/df and d/dt series
};

// Pointer to the sample's constants in GLOBAL memory, in two flavours: plain, and in the constant address space.  A wave-uniform
// read through the plain one becomes a scalar load only while the compiler can prove nothing writes memory in between -- one
// `asm volatile` in the kernel (the shapelet kernels order their LDS row reads with one, the cluster kernel its steps) and every
// such read turns into a VECTOR load of 64 identical addresses: ~500 cycles of L1 latency per four-term trip of the EPL series
// in round 3's table-mode shapelet kernel, found in round 4 from its 28 vector reads per wave-tile where 8 were expected.  Reads
// through the constant address space stay scalar whatever else the kernel holds.
typedef const float __attribute__((address_space(4)))* gptr4;
typedef float vf4 __attribute__((ext_vector_type(4)));
template <class T, class P> struct gl_rebind;
template <class T> struct gl_rebind<T, const float*> { using type = const T*; };
template <class T> struct gl_rebind<T, gptr4> { using type = const T __attribute__((address_space(4)))*; };

template <class V, bool GRAD, class P = const float*>
__device__ __forceinline__ void epl_fwd_v(const float* d, const P gd, V x, V y, V& bx, V& by,
                                          EplStateV<V>& st) {
  using PInt = typename gl_rebind<int, P>::type;
  using PRow = typename gl_rebind<vf4, P>::type;
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
  st.Cs = pos ? X * r : V(1.f);
  st.Ss = st.yr * st.inv;
  V iRc = clamp3(r, 1e-10f, 1e10f);  // one v_med3 per lane (a NaN gives the lower bound, like the two selects did)
  st.invc = (iRc == r) ? r : V(0.f);  // clip_by_value passes gradient only inside the clamp
  V twoc = (st.Cs * st.Cs - st.Ss * st.Ss) * 2.f;
  // scalar-loaded trip count and coefficients (wave-uniform address): SGPR operands, scalar loop control
  const int K = ((PInt)gd)[EPL_KI];
  const PRow gtab = (PRow)(gd + EPL_TAB);  // rows (c_n, (2n+1) c_n, dc_n/df, dc_n/dt)
  V o1, o2, f1 = V(0.f), f2 = V(0.f), t1 = V(0.f), t2 = V(0.f);  // b_{k+1}, b_{k+2}: set by the first trip (the tails of the two gradient series only with GRAD)
  auto term = [&](const vf4 ck, V& b1, V& b2, V& g1, V& g2, V& h1, V& h2) {  // b_k written over b_{k+2}
    b2 = __builtin_elementwise_fma(twoc, b1, V(ck.x)) - b2;
    if (GRAD) {
      g2 = __builtin_elementwise_fma(twoc, g1, V(ck.z)) - g2;
      h2 = __builtin_elementwise_fma(twoc, h1, V(ck.w)) - h2;
    }
  };
  auto four = [&](const vf4 r0, const vf4 r1, const vf4 r2, const vf4 r3) {  // rows k..k+3, highest first
    term(r3, o1, o2, f1, f2, t1, t2);
    term(r2, o2, o1, f2, f1, t2, t1);
    term(r1, o1, o2, f1, f2, t1, t2);
    term(r0, o2, o1, f2, f1, t2, t1);  // leaves b_k in (o1, f1, t1) and b_{k+1} in (o2, f2, t2)
  };
  // Four terms per trip, from the top of the table down to row 0; epl_prep zero-fills rows K+1..K+3, and zero coefficients
  // above K leave the recurrence at zero, so there is no remainder logic.  Two register sets in turn: the 16 dwords of
  // the NEXT trip are requested before this trip's 24 packed instructions (scalar loads return out of order, so a wait
  // means "all of them": one request in flight at a time).
  // The FIRST trip starts from b = 0: its top term is the (wave-uniform) coefficient itself and its second has no b_{k+2} --
  // 6 instead of 8 packed instructions per series, and no zero-fill of the six tails (12 instructions per pixel pair less).
  auto four_first = [&](const vf4 r0, const vf4 r1, const vf4 r2, const vf4 r3) {
    o1 = __builtin_elementwise_fma(twoc, V(r3.x), V(r2.x));           // b_{k+2} = twoc c_{k+3} + c_{k+2}
    o2 = __builtin_elementwise_fma(twoc, o1, V(r1.x)) - V(r3.x);       // b_{k+1}
    o1 = __builtin_elementwise_fma(twoc, o2, V(r0.x)) - o1;            // b_k      (written over b_{k+2})
    if (GRAD) {
      f1 = __builtin_elementwise_fma(twoc, V(r3.z), V(r2.z));
      f2 = __builtin_elementwise_fma(twoc, f1, V(r1.z)) - V(r3.z);
      f1 = __builtin_elementwise_fma(twoc, f2, V(r0.z)) - f1;
      t1 = __builtin_elementwise_fma(twoc, V(r3.w), V(r2.w));
      t2 = __builtin_elementwise_fma(twoc, t1, V(r1.w)) - V(r3.w);
      t1 = __builtin_elementwise_fma(twoc, t2, V(r0.w)) - t1;
    }
  };
  int trips = (K + 4) >> 2;  // ceil((K + 1) / 4)
  int row = 4 * trips - 4;   // first row of the current trip
  PRow p = gtab + row;
  vf4 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
  row = row >= 4 ? row - 4 : 0;  // the request made by the LAST trip is clamped to rows 0..3: inside the table, never used
  p = gtab + row;
  vf4 b0 = p[0], b1 = p[1], b2 = p[2], b3 = p[3];
  four_first(a0, a1, a2, a3);
  if (--trips != 0) {
    while (true) {
      row = row >= 4 ? row - 4 : 0;
      p = gtab + row;
      a0 = p[0]; a1 = p[1]; a2 = p[2]; a3 = p[3];
      four(b0, b1, b2, b3);
      if (--trips == 0) break;
      row = row >= 4 ? row - 4 : 0;
      p = gtab + row;
      b0 = p[0]; b1 = p[1]; b2 = p[2]; b3 = p[3];
      four(a0, a1, a2, a3);
      if (--trips == 0) break;
    }
  }
  const V o0 = o1, ob = o2;  // b_0, b_1 of Omega
  if (GRAD) { st.f0 = f1; st.f1 = f2; st.t0 = t1; st.t1 = t2; }
  const V Ox = (o0 - ob) * st.Cs, Oy = (o0 + ob) * st.Ss;
  st.L2 = log2_(iRc * d[EPL_B]);
  st.P = exp2_(st.L2 * d[EPL_TM1]) * d[EPL_P0];  // 2b/(1+q) (b/R)^(t-1), epl.py:55
  const V arx = st.P * Ox, ary = st.P * Oy;
  st.arx = arx;
  st.ary = ary;
  bx -= arx * c - ary * s;
  by -= arx * s + ary * c;
}

template <class V>
__device__ __forceinline__ void epl_vjp_v(const float* d, V gx, V gy, const EplStateV<V>& st, V* acc) {
  const float c = d[EPL_C], s = d[EPL_S], q = d[EPL_Q], tm1 = d[EPL_TM1];
  const V P = st.P, arx = st.arx, ary = st.ary;
  V grx = gx * c + gy * s, gry = gy * c - gx * s;  // the cotangent in the lens frame
  V g_phi = gry * arx - grx * ary;                 // g x alpha is rotation invariant: no need for alpha in the sky frame
  const V cross = g_phi;                           // = gOy Ox - gOx Oy below (alpha_r = P Omega): formed once
  V gW_W = grx * arx + gry * ary;                  // gP P with gP = g . Omega: the cotangent of W = (b/R)^(t-1), times W
  V gOx = P * grx, gOy = P * gry;
  // dot and cross products of (gOx, gOy) with a series (Sx, Sy) = ((b0 - b1) Cs, (b0 + b1) Ss) straight from its tails:
  //   gOx Sx + gOy Sy = b0 (A + B) + b1 (B - A),  gOy Sx - gOx Sy = b0 (C - D) - b1 (C + D)
  V A = gOx * st.Cs, B = gOy * st.Ss, C = gOy * st.Cs, D = gOx * st.Ss;
  V dotp = A + B, dotm = B - A, crsm = C - D, crsp = C + D;
  // d Omega/d theta = i S with S = sum (2n+1) c_n E_n = Omega + 2 f dOmega/df  (c_n ~ f^n): no separate S sum
  V g_ang = cross + (st.f0 * crsm - st.f1 * crsp) * d[EPL_F2];
  V g_t = st.t0 * dotp + st.t1 * dotm;
  V g_f = st.f0 * dotp + st.f1 * dotm;
  g_t += gW_W * (st.L2 * (float)kLn2);
  V gR0 = -(gW_W * st.invc) * tm1;
  V gai = g_ang * st.inv;
  V gX = gR0 * st.Cs - gai * st.Ss;
  V gyr = gR0 * st.Ss + gai * st.Cs;
  V gxr = gX * q;
  g_phi += gxr * st.yr - gyr * st.xr;
  acc[EPLA_CX] += gxr;    // lens-frame sums: rotated to the sky frame (and negated) once, in the epilogue
  acc[EPLA_CY] += gyr;
  acc[EPLA_PHI] += g_phi;
  acc[EPLA_Q] += gX * st.xr;
  acc[EPLA_T] += g_t;
  acc[EPLA_F] += g_f;
  acc[EPLA_P0] += gW_W;   // x 1/P0 in the epilogue (gP * W = gP * P / P0); EPLA_B = (t - 1) / b times the same sum
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

// ELL = false: the spherical profile (sersic.py:23-66 passes e1 = e2 = 0): no rotation, no axis-ratio stretch and no
// ellipticity gradients -- 21 packed instructions per pixel pair less over forward + VJP
template <class V, bool ELL = true> __device__ __forceinline__ V sersic_fwd_v(const float* d, V x, V y, SerStateV<V>& st) {
  V dx = x - d[SER_CX], dy = y - d[SER_CY];
  if constexpr (ELL) {
    const float c = d[SER_C], s = d[SER_S];
    st.a1 = dx * c + dy * s;
    st.a2 = dy * c - dx * s;
    V xt1 = st.a1 * d[SER_SQ], xt2 = st.a2 * d[SER_ISQ];
    st.r2 = xt1 * xt1 + xt2 * xt2;
  } else {
    st.a1 = dx;
    st.a2 = dy;
    st.r2 = dx * dx + dy * dy;
  }
  st.L2 = log2_(st.r2) * 0.5f + d[SER_L2IRS];  // log2(R / R_sersic) without the square root
  st.u = exp2_(st.L2 * d[SER_INVN]);
  st.E = vexp<V>((st.u - 1.f) * -d[SER_BN]);
  return st.E * d[SER_IE];
}
template <class V, bool SRC, bool ELL = true>
__device__ __forceinline__ void sersic_vjp_v(const float* d, const SerStateV<V>& st, V gI, V* acc, V& gpx, V& gpy) {
  // A pixel exactly on the centre (r2 = 0): u = 0 there, so g_u u and g_L are (signed) zeros and the reference's selects
  // (`where(x > 0, ...)` in TF's pow gradient) only keep 0 x inf from becoming NaN.  Flooring r2 and log2(R / Rs) does the same
  // with one v_max per lane each instead of a compare and two selects: 0 x (finite) = 0.
  V gE = gI * st.E;
  V tI = gE * d[SER_IE];
  V guu = -(tI * st.u) * d[SER_BN];
  V gL = guu * d[SER_INVN];
  V k = gL * rcp(floor_at(st.r2, 1e-37f));
  V gdx, gdy;
  if constexpr (ELL) {
    const float c = d[SER_C], s = d[SER_S], sq = d[SER_SQ], isq = d[SER_ISQ];
    V xt1 = st.a1 * sq, xt2 = st.a2 * isq;
    V gxt1 = k * xt1, gxt2 = k * xt2;
    V ga1 = gxt1 * sq, ga2 = gxt2 * isq;
    gdx = ga1 * c - ga2 * s;
    gdy = ga1 * s + ga2 * c;
    acc[SERA_PHI] += ga1 * st.a2 - ga2 * st.a1;
    acc[SERA_SQ] += gxt1 * st.a1 - gxt2 * st.a2 * (isq * isq);
  } else {
    gdx = k * st.a1;
    gdy = k * st.a2;
  }
  acc[SERA_CX] -= gdx;
  acc[SERA_CY] -= gdy;
  acc[SERA_L] += gL;
  acc[SERA_INVN] += guu * floor_at(st.L2, -1e30f);  // x ln2 in the epilogue
  acc[SERA_BN] -= tI * (st.u - 1.f);
  acc[SERA_IE] += gE;
  if (SRC) { gpx += gdx; gpy += gdy; }
}


// ---- NFW (tf/profiles/mass/nfw.py:15-52) ---------------------------------------------------------------------
// the shared table of h(X) = g(X) / X^2 (gl_host_tables.h: same constants)
constexpr int NFW_TAB_PER_OCTAVE = 128, NFW_TAB_LOG2_LO = -6, NFW_TAB_LOG2_HI = 6, NFW_TAB_STRIDE = NFW_TAB_PER_OCTAVE + 1;
constexpr int NFW_TAB_NODES = (NFW_TAB_LOG2_HI - NFW_TAB_LOG2_LO) * NFW_TAB_STRIDE;


// h(X) and dh/dX on a pixel pair.  g(X) in closed form costs ~45 VALU instructions and 6-8 quarter-rate transcendentals
// per LANE (nfw_gw: branchy, not packable) -- 92 per pixel and halo, half of the whole kernel at 8 halos.  h has no
// parameters, so inside [2^-6, 2^6) it is read from one LDS-resident table (gl_host_tables.h) by cubic Hermite
// interpolation in X: the interval and the position inside it come from the exponent and mantissa bits of X (exact, no
// logarithm), two 8-byte LDS reads per lane, a dozen packed instructions per pair.  Outside the table, and at exactly X = 1
// where the reference returns g = 1 (nfw.py:38), the closed form runs.
__device__ __forceinline__ void nfw_h_pair(const float* __restrict__ s_tab, v2f X, v2f iX, v2f& h, v2f& hp) {
  // (element copies first: __builtin_bit_cast applied to an ext-vector ELEMENT expression reads element 0 for both)
  const float x0 = X.x, x1 = X.y;
  const int b0 = __float_as_int(x0), b1 = __float_as_int(x1);
  const int e0 = (b0 >> 23) - 127, e1 = (b1 >> 23) - 127;  // X >= 1e-6 > 0: sign bit clear, normal numbers
  const bool in0 = (e0 >= NFW_TAB_LOG2_LO) && (e0 < NFW_TAB_LOG2_HI) && (X.x != 1.f);
  const bool in1 = (e1 >= NFW_TAB_LOG2_LO) && (e1 < NFW_TAB_LOG2_HI) && (X.y != 1.f);
  const int o0 = min(max(e0 - NFW_TAB_LOG2_LO, 0), NFW_TAB_LOG2_HI - NFW_TAB_LOG2_LO - 1);
  const int o1 = min(max(e1 - NFW_TAB_LOG2_LO, 0), NFW_TAB_LOG2_HI - NFW_TAB_LOG2_LO - 1);
  const int i0 = o0 * NFW_TAB_STRIDE + ((b0 >> 16) & 127), i1 = o1 * NFW_TAB_STRIDE + ((b1 >> 16) & 127);
  const float2 a0 = reinterpret_cast<const float2*>(s_tab)[i0], c0 = reinterpret_cast<const float2*>(s_tab)[i0 + 1];
  const float2 a1 = reinterpret_cast<const float2*>(s_tab)[i1], c1 = reinterpret_cast<const float2*>(s_tab)[i1 + 1];
  const v2f t = v2f{(float)(b0 & 0xFFFF), (float)(b1 & 0xFFFF)} * (1.f / 65536.f);  // the low 16 mantissa bits: exact
  const v2f f0{a0.x, a1.x}, d0{a0.y, a1.y}, f1{c0.x, c1.x}, d1{c0.y, c1.y};
  const v2f df = f1 - f0;
  const v2f c2 = df * 3.f - d0 * 2.f - d1, c3 = d0 + d1 - df * 2.f;
  h = ((c3 * t + c2) * t + d0) * t + f0;
  const v2f dh = (c3 * (3.f * t) + c2 * 2.f) * t + d0;  // dh/dt, t = (X - X_node) / dX_e
  // dh/dX = dh/dt / dX_e,  1 / dX_e = 128 * 2^-e: a power of two built from the exponent
  const v2f inv_dx{__int_as_float((127 + 7 - e0) << 23), __int_as_float((127 + 7 - e1) << 23)};
  hp = dh * inv_dx;
  // rare: a lane outside the table (or exactly on the reference's g(1) = 1 point) takes the closed form, lane by lane
  if (!in0) {
    float g, gp;
    nfw_gw<float>(X.x, g, gp);
    const float i2 = iX.x * iX.x, he = g * i2;
    h.x = he;
    hp.x = gp * i2 - (he + he) * iX.x;
  }
  if (!in1) {
    float g, gp;
    nfw_gw<float>(X.y, g, gp);
    const float i2 = iX.y * iX.y, he = g * i2;
    h.y = he;
    hp.y = gp * i2 - (he + he) * iX.y;
  }
}

// pixel-pair forward / VJP of the interpreter kernel (the cluster kernel splits the same maths at the forward state)
__device__ __forceinline__ void nfw_fwd_v(const float* d, const float* __restrict__ s_tab, v2f x, v2f y, v2f& bx, v2f& by) {
  v2f dx = x - d[NFW_CX], dy = y - d[NFW_CY];
  v2f R0 = sqrt_(dx * dx + dy * dy);
  v2f X = vmax(vmax(R0, v2f(1e-7f)) * d[NFW_INVRS], v2f(1e-6f));  // nfw.py:26,37
  v2f h, hp;
  nfw_h_pair(s_tab, X, rcp(X), h, hp);
  v2f a = h * d[NFW_K0];
  bx -= a * dx;
  by -= a * dy;
}
__device__ __forceinline__ void nfw_vjp_v(const float* d, const float* __restrict__ s_tab, v2f x, v2f y, v2f gx, v2f gy, v2f* acc) {
  using V = v2f;
  V dx = x - d[NFW_CX], dy = y - d[NFW_CY];
  V R0 = sqrt_(dx * dx + dy * dy);
  V X0 = vmax(R0, V(1e-7f)) * d[NFW_INVRS];
  V X = vmax(X0, V(1e-6f));
  V h, hp;
  nfw_h_pair(s_tab, X, rcp(X), h, hp);
  const float K0 = d[NFW_K0];
  V a = h * K0;
  V ga = gx * dx + gy * dy;
  V gX0 = (X0 > V(1e-6f)) ? ga * hp * K0 : V(0.f);
  V gR0 = (R0 > V(1e-7f)) ? gX0 * d[NFW_INVRS] : V(0.f);
  V iR0 = (R0 > V(0.f)) ? rcp(R0) : V(0.f);
  V t = gR0 * iR0;
  acc[NFWA_CX] -= gx * a + t * dx;
  acc[NFWA_CY] -= gy * a + t * dy;
  acc[NFWA_RS] -= gX0 * X0 * d[NFW_INVRS];
  acc[NFWA_K0] += ga * h;
}

}  // namespace glk
