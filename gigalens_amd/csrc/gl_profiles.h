// gl_profiles.h -- per-profile math of the lens hot path: forward evaluation and
// hand-written vector-Jacobian products, shared by every kernel.
//
// Structure (one sample == one set of constrained profile parameters):
//   *_prep      raw parameter row  -> "derived constants" block (per sample; precise math).
//               The main kernels stage this block in LDS and every pixel reads it by broadcast.
//   *_fwd       (derived, pixel coordinates) -> deflection / surface brightness.
//   *_vjp       (derived, pixel, cotangent)  -> adds d(cotangent . output)/d(derived) into a small
//               accumulator array (registers) and, for light profiles evaluated on the source
//               plane, the cotangent of the ray-shot position.
//   *_finalize  summed accumulators -> gradient w.r.t. the raw parameters (per sample chain rule).
//
// What each profile computes follows the reference's TF substrate (cited per function, paths
// relative to the reference repo), including its quirks.  HOW it is
// computed is new: fused per pixel in registers, algebraic cos/sin instead of atan2+cos+sin,
// a per-sample coefficient table for the EPL angular series with forward-mode sensitivities,
// and an NFW g(X) formulation that stays accurate through X = 1.
#pragma once
#include "gl_math.h"

namespace glp {
using namespace glm;

enum Kind : int {
  K_EPL = 1, K_SIE = 2, K_NFW = 3, K_SHEAR = 4, K_SIS = 5,
  K_DPIS = 6, K_DPIE = 7, K_DPIEP = 8,  // gl_dpie.h
  K_SCALED = 9,                         // ScalingRelation over a galaxy catalogue (gl_dpie.h); iparam = catalogue slot
  K_SERIES = 10,                        // series expansion of a (scaled) dPIE in r_cut (gl_series.h); iparam = order, flags = field slot
  K_NFW_ELLIPSE = 11, K_TNFW = 12,      // gl_extra.h
  K_USER_MASS = 13,                     // a body the user wrote (gl_user.hip): iparam = its parameter count, flags = body slot of the model
  K_SERSIC = 16, K_SERSIC_ELLIPSE = 17, K_SHAPELETS = 18,
  K_CORE_SERSIC = 19,                   // gl_extra.h
  K_USER_LIGHT = 20                     // same for a light profile (amplitude: whatever the body makes of its parameters)
};
constexpr int USER_MAXP = 16;  // parameters of a user-written profile inside a model (its gradient sums are register arrays)

constexpr int SH_CAP = 10;                              // largest n_max the register-resident / matrix-pipe shapelet kernels serve
constexpr int SH_MAXL = (SH_CAP + 1) * (SH_CAP + 2) / 2;  // 66
constexpr int SH_CAPB = 20;                             // GL_SHAPELETS_NMAX_CAP: orders above SH_CAP run the runtime-order path
constexpr int SH_MAXLB = (SH_CAPB + 1) * (SH_CAPB + 2) / 2;  // 231
constexpr int SH_NODES = 6000;                          // shapelets.py:39-40

// ---------------------------------------------------------------------------------------------
// layout of the derived-constant blocks and accumulator arrays
// ---------------------------------------------------------------------------------------------
// EPL
enum { EPL_CX = 0, EPL_CY, EPL_C, EPL_S, EPL_Q, EPL_B, EPL_TM1, EPL_P0, EPL_K, EPL_INVB, EPL_KI /* K as int bits */, EPL_F2 /* 2 f */, EPL_TAB = 12 };
enum { EPLA_CX = 0, EPLA_CY, EPLA_PHI, EPLA_Q, EPLA_B, EPLA_T, EPLA_F, EPLA_P0, EPL_NACC };
// SIE
enum { SIE_CX = 0, SIE_CY, SIE_C, SIE_S, SIE_Q, SIE_SQ, SIE_A, SIE_ND };
enum { SIEA_CX = 0, SIEA_CY, SIEA_PHI, SIEA_Q, SIEA_SQ, SIEA_A, SIE_NACC };
// NFW
enum { NFW_CX = 0, NFW_CY, NFW_INVRS, NFW_K0, NFW_ND };
enum { NFWA_CX = 0, NFWA_CY, NFWA_RS, NFWA_K0, NFW_NACC };
// SHEAR
enum { SHR_G1 = 0, SHR_G2, SHR_ND };
enum { SHR_NACC = 2 };
// SIS
enum { SIS_CX = 0, SIS_CY, SIS_TE, SIS_ND };
enum { SIS_NACC = 3 };
// SERSIC / SERSIC_ELLIPSE (one code path; the spherical profile is the e=0 member, sersic.py:52-55)
// (order: the eight constants of a spherical source first, then the four of the ellipse -- gl_clusterw_kernel fetches a source's
// block with one 8-dword scalar load, plus a 4-dword one for elliptical sources)
enum { SER_CX = 0, SER_CY, SER_INVN, SER_BN, SER_IE, SER_L2IRS /* log2(1/R_sersic) */, SER_CG /* Ie bn / n */, SER_IRS2 /* 1 / R_sersic^2 */,
       SER_C, SER_S, SER_SQ, SER_ISQ, SER_INVRS, SER_PAD, SER_NDX /* floats of the block */, SER_ND = SER_NDX };
enum { SERA_CX = 0, SERA_CY, SERA_PHI, SERA_SQ, SERA_L, SERA_INVN, SERA_BN, SERA_IE, SER_NACC };
// SHAPELETS
enum { SHP_CX = 0, SHP_CY, SHP_IB, SHP_NMAX, SHP_AMP = 4 };
// behind the zero-padded amplitude triangle: the same amplitudes as a zero-padded SQUARE matrix a[n1][n2], row-major with
// SH_SQ columns -- gl_shp.hip.h reads its rows as SGPR pairs (n2 = 2j, 2j + 1)
constexpr int SH_SQ = 12;
constexpr int SHP_SQ = SHP_AMP + ((SH_MAXL + 3) & ~3);
// phi_n = SH_K[n] P_n with the monic recurrence P_{n+1} = u P_n - (n / 2) P_{n-1} (SH_K[n] = sqrt(2^n / n!)): gl_shp.hip.h advances
// P -- two instructions per order instead of three -- and the square matrix behind SHP_SQ holds a(n1, n2) SH_K[n1] SH_K[n2]
constexpr float SH_K[12] = {1.f, 1.4142135623730951f, 1.4142135623730951f, 1.1547005383792515f, 0.81649658092772603f,
                            0.5163977794943222f, 0.29814239699997197f, 0.15936381457791915f, 0.079681907288959575f,
                            0.037562411321267405f, 0.016798421022632321f, 0.007162870791336714f};
enum { SHPA_CX = 0, SHPA_CY, SHPA_IB, SHPA_AMP = 3 };

GL_HD int sh_layers(int n_max) { return (n_max + 1) * (n_max + 2) / 2; }

GL_HD int kind_num_params(int kind, int iparam) {
  switch (kind) {
    case K_EPL: return 6;
    case K_SIE: return 5;
    case K_NFW: return 4;
    case K_SHEAR: return 2;
    case K_SIS: return 3;
    case K_DPIS: return 5;
    case K_DPIE:
    case K_DPIEP: return 7;
    case K_SCALED: return (iparam >= 1 && iparam <= 3) ? iparam : -1;  // the population scales
    case K_SERIES: return (iparam >= 0 && iparam <= 5) ? 2 : -1;       // theta_E, r_cut
    case K_NFW_ELLIPSE: return 6;
    case K_TNFW: return 5;
    case K_CORE_SERSIC: return 10;
    case K_SERSIC: return 5;
    case K_SERSIC_ELLIPSE: return 7;
    case K_SHAPELETS: return 3 + sh_layers(iparam);
    case K_USER_MASS:
    case K_USER_LIGHT: return (iparam >= 0 && iparam <= USER_MAXP) ? iparam : -1;
  }
  return -1;
}
GL_HD int kind_num_derived(int kind, int iparam) {
  switch (kind) {
    case K_EPL: return EPL_TAB + 4 * (iparam + 4);  // rows 0..cap plus three zero rows: the Clenshaw loop takes four rows per trip
    case K_SIE: return SIE_ND + 1;
    case K_NFW: return NFW_ND;
    case K_SHEAR: return SHR_ND + 2;
    case K_SIS: return SIS_ND + 1;
    case K_DPIS:
    case K_DPIE:
    case K_DPIEP: return 32;  // DPX_ND
    case K_SCALED: return 4;  // the member blocks live in the catalogue workspace, not in the sample's LDS block
    case K_SERIES: return 4;
    case K_NFW_ELLIPSE:
    case K_TNFW: return 8;
    case K_CORE_SERSIC: return 16;
    case K_SERSIC:
    case K_SERSIC_ELLIPSE: return SER_NDX;
    case K_SHAPELETS:  // n_max <= 10: amplitude triangle zero-padded to n_max = 10, then the square matrix; above: the triangle only
      return iparam <= SH_CAP ? SHP_SQ + SH_SQ * SH_SQ : SHP_AMP + ((SH_MAXLB + 3) & ~3);
    case K_USER_MASS:
    case K_USER_LIGHT: return iparam > 0 ? ((iparam + 3) & ~3) : 4;  // the parameters themselves
  }
  return -1;
}
// linear (amplitude) coefficients of a light profile, the reference's `depth` (profile.py:39; shapelets.py:22-23)
GL_HD int kind_num_linear(int kind, int iparam) {
  switch (kind) {
    case K_SERSIC:
    case K_SERSIC_ELLIPSE:
    case K_CORE_SERSIC: return 1;
    case K_SHAPELETS: return sh_layers(iparam);
  }
  return 0;
}
GL_HD int kind_linear_col(int kind, int iparam) {  // first amplitude column inside the component's parameter row
  switch (kind) {
    case K_SERSIC: return 4;
    case K_SERSIC_ELLIPSE: return 6;
    case K_SHAPELETS: return 3;
    case K_CORE_SERSIC: return 9;
  }
  (void)iparam;
  return -1;
}
GL_HD int kind_num_acc(int kind, int iparam) {
  switch (kind) {
    case K_EPL: return EPL_NACC;
    case K_SIE: return SIE_NACC;
    case K_NFW: return NFW_NACC;
    case K_SHEAR: return SHR_NACC;
    case K_SIS: return SIS_NACC;
    case K_DPIS:
    case K_DPIE:
    case K_DPIEP: return 7;  // DP_NACC
    case K_SCALED: return 3;
    case K_SERIES: return 2;
    case K_NFW_ELLIPSE: return 6;
    case K_TNFW: return 5;
    case K_CORE_SERSIC: return 10;
    case K_SERSIC:
    case K_SERSIC_ELLIPSE: return SER_NACC;
    case K_SHAPELETS: return SHPA_AMP + sh_layers(iparam);
    case K_USER_MASS:
    case K_USER_LIGHT: return iparam;  // one sum per parameter
  }
  return -1;
}

// precise per-sample helpers (cost irrelevant: once per sample)
GL_HD double p_sqrt(double x) { return ::sqrt(x); }
GL_HD double p_atan2(double y, double x) { return ::atan2(y, x); }
GL_HD double p_sin(double x) { return ::sin(x); }
GL_HD double p_cos(double x) { return ::cos(x); }
GL_HD double p_log(double x) { return ::log(x); }
GL_HD double p_pow(double x, double y) { return ::pow(x, y); }
// float: the precise (<= 1-2 ulp) single-precision library routines -- the reference is fp32 throughout
GL_HD float p_sqrt(float x) { return ::sqrtf(x); }
GL_HD float p_atan2(float y, float x) { return ::atan2f(y, x); }
GL_HD float p_sin(float x) { return ::sinf(x); }
GL_HD float p_cos(float x) { return ::cosf(x); }
GL_HD float p_log(float x) { return ::logf(x); }
GL_HD float p_pow(float x, float y) { return ::powf(x, y); }

// (e1,e2) -> (phi, c, q) with c = min(|e|, cmax); EPL passes cmax = 1 (epl.py:22), every other
// elliptical profile 0.9999 (sie.py:17, sersic.py:57).  Arithmetic in R like the reference's fp32.
// cos / sin of phi = atan2(e2, e1) / 2 in (-pi/2, pi/2] come from the half-angle identities on cos 2phi = e1 / |e| -- the
// cancellation-free branch of each, the other through sin 2phi = 2 sin phi cos phi -- instead of atan2 + cos + sin (three
// library calls of ~80 dependent instructions each on the one lane that prepares a sample's lens).  e = 0: phi = atan2(+-0, +-0) / 2.
// sign bit of the innermost value (dual / jet types decide on values: their val() is found by argument-dependent lookup)
GL_HD bool neg_sign(float x) { return __builtin_signbitf(x); }
GL_HD bool neg_sign(double x) { return __builtin_signbit(x); }
template <class R> GL_HD auto neg_sign(const R& x) -> decltype(neg_sign(val(x))) { return neg_sign(val(x)); }
template <class R> struct Ellip { R ee, c, q, cphi, sphi; };
template <class R> GL_HD Ellip<R> ellip_prep(R e1, R e2, R cmax) {
  Ellip<R> o;
  o.ee = p_sqrt(e1 * e1 + e2 * e2);
  o.c = fmin_(o.ee, cmax);
  o.q = ((R)1 - o.c) / ((R)1 + o.c);
  const bool neg2 = neg_sign(e2), neg1 = neg_sign(e1);
  if (o.ee > (R)0) {
    const R c2 = e1 / o.ee, s2 = e2 / o.ee;  // cos 2phi, sin 2phi
    if (c2 >= (R)0) {
      o.cphi = p_sqrt(((R)1 + c2) * (R)0.5);
      o.sphi = s2 / ((R)2 * o.cphi);
    } else {
      const R sa = p_sqrt(((R)1 - c2) * (R)0.5);  // |sin phi|
      o.sphi = neg2 ? -sa : sa;
      o.cphi = (neg2 ? -s2 : s2) / ((R)2 * sa);
    }
  } else {  // atan2 of signed zeros: 0, +-pi -> phi = 0 or +-pi/2
    o.cphi = neg1 ? (R)0 : (R)1;
    o.sphi = neg1 ? (neg2 ? (R)-1 : (R)1) : (R)0;
  }
  return o;
}
// b = theta_E / sqrt((1+q^2)/(2q)) * sqrt((1+q^2)/2)   (epl.py:24-25, sie.py:19-20; == theta_E sqrt(q))
template <class R> GL_HD R einstein_b(R theta_E, R q) {
  R conv = theta_E / p_sqrt(((R)1 + q * q) / ((R)2 * q));
  return conv * p_sqrt(((R)1 + q * q) / (R)2);
}
// chain (g_c, g_phi) -> (g_e1, g_e2), c = min(|e|, cmax), phi = atan2(e2, e1)/2
template <class R> GL_HD void ellip_chain_c(R e1, R e2, R cmax, R g_c, R g_phi, R& g_e1, R& g_e2) {
  R ee = p_sqrt(e1 * e1 + e2 * e2);
  R g_ee = (ee <= cmax) ? g_c : (R)0;
  if (ee > (R)0) {
    R ie = (R)1 / ee;
    R h = (R)0.5 * ie * ie;
    g_e1 = g_ee * e1 * ie - g_phi * e2 * h;
    g_e2 = g_ee * e2 * ie + g_phi * e1 * h;
  } else {
    g_e1 = (R)0;
    g_e2 = (R)0;
  }
}
// chain (g_b, g_q, g_phi) -> (g_theta_E, g_e1, g_e2)
template <class R>
GL_HD void ellip_chain(R theta_E, R e1, R e2, R cmax, R g_b, R g_q, R g_phi, R& g_te, R& g_e1, R& g_e2) {
  Ellip<R> el = ellip_prep(e1, e2, cmax);
  R sq = p_sqrt(el.q);
  g_te = g_b * sq;
  if (sq > (R)0) g_q += g_b * theta_E / ((R)2 * sq);
  R opc = (R)1 + el.c;
  R g_c = g_q * ((R)-2 / (opc * opc));
  ellip_chain_c(e1, e2, cmax, g_c, g_phi, g_e1, g_e2);
}

// =============================================================================================
// EPL  (tf/profiles/mass/epl.py:19-57; Tessore & Metcalf 2015 angular series)
// =============================================================================================
// The reference iterates  last <- p_n * rot(2 theta) * last,  acc += last  with
// p_n = -f (2n-(2-t))/(2n+(2-t)) (epl.py:39-45).  Because p_n does not depend on the pixel,
// Omega = sum_n c_n e^{i(2n+1)theta} with per-sample coefficients c_n = prod p_k; the table below
// holds c_n, (2n+1) c_n, dc_n/df and dc_n/dt so that one rotation per term yields Omega and its
// derivatives w.r.t. theta, f and t together (forward mode inside the loop, no per-pixel tape).
// Trip count: n < log(tol)/log(f) + 2, capped at niter (epl.py:37,47-54), evaluated per sample (the reference takes max f
// over the batch).  tol is the reference's 1e-12 in float64; the float32 kernels stop at 1e-9: |c_n| <= f^n, so the terms
// left out sum to < 1e-9 f^2 / (1 - f) of an O(1) value -- under 0.02 ulp of a float32 sum, which the reference's own
// float32 accumulation cannot register either -- and the f-derivative's tail (K + 1) f^K / (1 - f)^2 stays under 1 ulp.
template <class R> GL_HD constexpr double epl_series_tol() { return sizeof(R) == 4 ? 1e-9 : 1e-12; }
// scalars of the derived block (d[0..EPL_TAB)); returns the series length K and hands out f and 2 - t for the table
// the trip count alone (what the cost-ordered dispatch sorts on): same arithmetic as epl_prep_head below
template <class R> GL_HD int epl_cost(R e1, R e2, int cap) {
  const R ee = p_sqrt(e1 * e1 + e2 * e2), c = fmin_(ee, (R)1), q = ((R)1 - c) / ((R)1 + c), f = ((R)1 - q) / ((R)1 + q);
  const R niter = p_log((R)epl_series_tol<R>()) / p_log(f) + (R)2;
  return niter > (R)1 ? (int)fmin_(-floor_(-niter) - (R)1, (R)cap) : 0;
}
template <class R> GL_HD int epl_prep_head(const R* p, int cap, R* d, R& f_out, R& two_mt_out) {
  R theta_E = p[0], gamma = p[1], e1 = p[2], e2 = p[3];
  Ellip<R> el = ellip_prep(e1, e2, (R)1);
  R q = el.q;
  R b = einstein_b(theta_E, q);
  R t = gamma - (R)1;
  R f = ((R)1 - q) / ((R)1 + q);
  d[EPL_CX] = p[4];
  d[EPL_CY] = p[5];
  d[EPL_C] = el.cphi;
  d[EPL_S] = el.sphi;
  d[EPL_Q] = q;
  d[EPL_B] = b;
  d[EPL_TM1] = t - (R)1;
  d[EPL_P0] = ((R)2 * b) / ((R)1 + q);
  d[EPL_INVB] = (R)1 / b;
  d[10] = (R)0;
  d[EPL_F2] = (R)2 * f;
  R niter = p_log((R)epl_series_tol<R>()) / p_log(f) + (R)2;
  // terms n = 1..K with n < niter (epl.py:47-54), K <= cap:  K = ceil(niter) - 1
  int K = 0;
  if (niter > (R)1) K = (int)fmin_(-floor_(-niter) - (R)1, (R)cap);
  d[EPL_K] = (R)K;
  if (sizeof(R) == 4) {  // the same count as raw int bits, so a kernel can fetch it with a scalar load
    int* ki = reinterpret_cast<int*>(&d[EPL_KI]);
    *ki = K;
  }
  f_out = f;
  two_mt_out = (R)2 - t;
  return K;
}
// one row's factors: p_n = f r_n with r_n = -(2n - (2-t)) / (2n + (2-t)), and dp_n/dt
template <class R> GL_HD void epl_row_factors(int n, R f, R two_mt, R& r, R& pn, R& dpdt) {
  R iden = (R)1 / ((R)(2 * n) + two_mt);
  r = -((R)(2 * n) - two_mt) * iden;
  pn = f * r;
  dpdt = -f * (R)(4 * n) * (iden * iden);
}
template <class R> GL_HD void epl_prep(const R* p, int cap, R* d) {
  R f, two_mt;
  const int K = epl_prep_head(p, cap, d, f, two_mt);
  // the divisions of different n are independent; only the products chain (the GPU front end builds the same table with a
  // parallel scan over the lanes of a wave, gl_kernels.hip.h epl_table_wave)
  R* tab = d + EPL_TAB;
  R c = (R)1, cf = (R)0, ct = (R)0;
  tab[0] = (R)1; tab[1] = (R)1; tab[2] = (R)0; tab[3] = (R)0;
  for (int n = 1; n <= K; ++n) {
    R r, pn, dpdt;
    epl_row_factors(n, f, two_mt, r, pn, dpdt);
    cf = cf * pn + c * r;
    ct = ct * pn + c * dpdt;
    c = c * pn;
    tab[4 * n + 0] = c;
    tab[4 * n + 1] = (R)(2 * n + 1) * c;
    tab[4 * n + 2] = cf;
    tab[4 * n + 3] = ct;
  }
  for (int j = 0; j < 12; ++j) tab[4 * (K + 1) + j] = (R)0;  // zero rows K+1..K+3: the four-row trips of the Clenshaw loop may start above K
}

template <class R> GL_HD void epl_fwd(const R* d, R x, R y, R& ax, R& ay) {
  R dx = x - d[EPL_CX], dy = y - d[EPL_CY];
  R c = d[EPL_C], s = d[EPL_S], q = d[EPL_Q];
  R xr = dx * c + dy * s, yr = dy * c - dx * s;
  R X = q * xr;
  R R0 = sqrt_(X * X + yr * yr);
  bool pos = R0 > (R)0;
  R inv = pos ? rcp(R0) : (R)0;
  R Cs = pos ? X * inv : (R)1, Ss = yr * inv;  // == cos/sin(atan2(yr, q xr)), atan2(0,0)=0 (epl.py:32-34)
  R Rc = clamp_(R0, (R)1e-10, (R)1e10);        // epl.py:31
  R E2x = Cs * Cs - Ss * Ss, E2y = (R)2 * Cs * Ss;
  R Ex = Cs, Ey = Ss, Ox = Cs, Oy = Ss;
  const int K = (int)d[EPL_K];
  const R* tab = d + EPL_TAB;
  for (int n = 1; n <= K; ++n) {
    R tx = E2x * Ex - E2y * Ey;
    Ey = E2y * Ex + E2x * Ey;
    Ex = tx;
    R cn = tab[4 * n];
    Ox += cn * Ex;
    Oy += cn * Ey;
  }
  R L2 = log2_(d[EPL_B] * rcp(Rc));
  R P = d[EPL_P0] * exp2_(d[EPL_TM1] * L2);  // 2b/(1+q) (b/R)^(t-1), epl.py:55
  R arx = P * Ox, ary = P * Oy;
  ax = arx * c - ary * s;  // rotate by -phi, epl.py:57
  ay = arx * s + ary * c;
}

// cotangent (gx, gy) of (alpha_x, alpha_y); returns alpha as a by-product
template <class R> GL_HD void epl_vjp(const R* d, R x, R y, R gx, R gy, R* acc) {
  R dx = x - d[EPL_CX], dy = y - d[EPL_CY];
  R c = d[EPL_C], s = d[EPL_S], q = d[EPL_Q];
  R xr = dx * c + dy * s, yr = dy * c - dx * s;
  R X = q * xr;
  R R0 = sqrt_(X * X + yr * yr);
  bool pos = R0 > (R)0;
  R inv = pos ? rcp(R0) : (R)0;
  R Cs = pos ? X * inv : (R)1, Ss = yr * inv;
  bool inclamp = (R0 >= (R)1e-10) && (R0 <= (R)1e10);
  R Rc = clamp_(R0, (R)1e-10, (R)1e10);
  R E2x = Cs * Cs - Ss * Ss, E2y = (R)2 * Cs * Ss;
  R Ex = Cs, Ey = Ss;
  R Ox = Cs, Oy = Ss;  // Omega
  R Sx = Cs, Sy = Ss;  // sum (2n+1) c_n E_n   (dOmega/dtheta = i * S)
  R Fx = (R)0, Fy = (R)0;  // dOmega/df
  R Tx = (R)0, Ty = (R)0;  // dOmega/dt
  const int K = (int)d[EPL_K];
  const R* tab = d + EPL_TAB;
  for (int n = 1; n <= K; ++n) {
    R tx = E2x * Ex - E2y * Ey;
    Ey = E2y * Ex + E2x * Ey;
    Ex = tx;
    R c0 = tab[4 * n], c1 = tab[4 * n + 1], c2 = tab[4 * n + 2], c3 = tab[4 * n + 3];
    Ox += c0 * Ex; Oy += c0 * Ey;
    Sx += c1 * Ex; Sy += c1 * Ey;
    Fx += c2 * Ex; Fy += c2 * Ey;
    Tx += c3 * Ex; Ty += c3 * Ey;
  }
  R tm1 = d[EPL_TM1], P0 = d[EPL_P0];
  R iRc = rcp(Rc);
  R L2 = log2_(d[EPL_B] * iRc);
  R W = exp2_(tm1 * L2);
  R P = P0 * W;
  R arx = P * Ox, ary = P * Oy;
  R ax = arx * c - ary * s, ay = arx * s + ary * c;
  // back-rotation
  R grx = gx * c + gy * s, gry = gy * c - gx * s;
  R g_phi = gy * ax - gx * ay;
  R gP = grx * Ox + gry * Oy;
  R gOx = P * grx, gOy = P * gry;
  R g_ang = gOy * Sx - gOx * Sy;
  R g_t = gOx * Tx + gOy * Ty;
  R g_f = gOx * Fx + gOy * Fy;
  R gW_W = gP * P;                        // gW * W  with gW = gP * P0
  g_t += gW_W * (L2 * (R)kLn2);           // dW/dt = W ln(b/R)
  R g_b = gW_W * tm1 * d[EPL_INVB];       // dW/db = (t-1) W / b
  R gR0 = inclamp ? -gW_W * tm1 * iRc : (R)0;  // clip_by_value passes gradient only inside the clamp
  R gX = gR0 * Cs - g_ang * Ss * inv;
  R gyr = gR0 * Ss + g_ang * Cs * inv;
  R g_q = gX * xr;
  R gxr = gX * q;
  R gdx = gxr * c - gyr * s, gdy = gxr * s + gyr * c;
  g_phi += gxr * yr - gyr * xr;
  acc[EPLA_CX] -= gdx;
  acc[EPLA_CY] -= gdy;
  acc[EPLA_PHI] += g_phi;
  acc[EPLA_Q] += g_q;
  acc[EPLA_B] += g_b;
  acc[EPLA_T] += g_t;
  acc[EPLA_F] += g_f;
  acc[EPLA_P0] += gP * W;
}

template <class R> GL_HD void epl_finalize(const R* p, const R* acc, R* g) {
  R theta_E = p[0], e1 = p[2], e2 = p[3];
  Ellip<R> el = ellip_prep(e1, e2, (R)1);
  R q = el.q;
  R b = einstein_b(theta_E, q);
  R opq = (R)1 + q;
  R g_q = acc[EPLA_Q] + acc[EPLA_F] * ((R)-2 / (opq * opq)) - acc[EPLA_P0] * (R)2 * b / (opq * opq);
  R g_b = acc[EPLA_B] + acc[EPLA_P0] * (R)2 / opq;
  R g_te, g_e1, g_e2;
  ellip_chain(theta_E, e1, e2, (R)1, g_b, g_q, acc[EPLA_PHI], g_te, g_e1, g_e2);
  g[0] = g_te;
  g[1] = acc[EPLA_T];
  g[2] = g_e1;
  g[3] = g_e2;
  g[4] = acc[EPLA_CX];
  g[5] = acc[EPLA_CY];
}

// EPL without the table, for plugin-level point evaluation (MassProfile.deriv on arbitrary points)
template <class R> GL_HD void epl_point(const R* p, int cap, R x, R y, R& ax, R& ay) {
  Ellip<R> el = ellip_prep(p[2], p[3], (R)1);
  R q = el.q, b = einstein_b(p[0], q), t = p[1] - (R)1, f = ((R)1 - q) / ((R)1 + q);
  R c = el.cphi, s = el.sphi;
  R dx = x - p[4], dy = y - p[5];
  R xr = dx * c + dy * s, yr = dy * c - dx * s;
  R X = q * xr;
  R R0 = p_sqrt(X * X + yr * yr);
  bool pos = R0 > (R)0;
  R Cs = pos ? X / R0 : (R)1, Ss = pos ? yr / R0 : (R)0;
  R Rc = clamp_(R0, (R)1e-10, (R)1e10);
  R E2x = Cs * Cs - Ss * Ss, E2y = (R)2 * Cs * Ss;
  R lx = Cs, ly = Ss, Ox = Cs, Oy = Ss;
  R niter = p_log((R)1e-12) / p_log(f) + (R)2;
  for (int n = 1; n <= cap; ++n) {
    if (!((R)n < niter)) break;
    R pn = -f * ((R)(2 * n) - ((R)2 - t)) / ((R)(2 * n) + ((R)2 - t));
    R tx = pn * (E2x * lx - E2y * ly);
    ly = pn * (E2y * lx + E2x * ly);
    lx = tx;
    Ox += lx;
    Oy += ly;
  }
  R P = ((R)2 * b) / ((R)1 + q) * p_pow(b / Rc, t - (R)1);
  R arx = P * Ox, ary = P * Oy;
  ax = arx * c - ary * s;
  ay = arx * s + ary * c;
}

// =============================================================================================
// SIE  (tf/profiles/mass/sie.py:13-42; core s == 0 because s_scale is shadowed by a local, :15)
// =============================================================================================
template <class R> GL_HD void sie_prep(const R* p, R* d) {
  Ellip<R> el = ellip_prep(p[1], p[2], (R)0.9999);
  R q = el.q;
  R b = einstein_b(p[0], q);
  R sq = p_sqrt((R)1 - q * q);
  d[SIE_CX] = p[3];
  d[SIE_CY] = p[4];
  d[SIE_C] = el.cphi;
  d[SIE_S] = el.sphi;
  d[SIE_Q] = q;
  d[SIE_SQ] = sq;
  d[SIE_A] = b / sq;  // e == 0 -> b/0: the reference's NaN, zeroed later in the image (Appendix A6)
  d[SIE_ND] = (R)0;
}
template <class R> GL_HD void sie_fwd(const R* d, R x, R y, R& ax, R& ay) {
  R dx = x - d[SIE_CX], dy = y - d[SIE_CY];
  R c = d[SIE_C], s = d[SIE_S], q = d[SIE_Q], sq = d[SIE_SQ], A = d[SIE_A];
  R xr = dx * c + dy * s, yr = dy * c - dx * s;
  R ipsi = rcp(sqrt_(q * q * xr * xr + yr * yr));
  R u = sq * xr * ipsi, v = sq * yr * ipsi;
  R arx = A * atan_(u), ary = A * atanh_(v);
  ax = arx * c - ary * s;
  ay = arx * s + ary * c;
}
template <class R> GL_HD void sie_vjp(const R* d, R x, R y, R gx, R gy, R* acc) {
  R dx = x - d[SIE_CX], dy = y - d[SIE_CY];
  R c = d[SIE_C], s = d[SIE_S], q = d[SIE_Q], sq = d[SIE_SQ], A = d[SIE_A];
  R xr = dx * c + dy * s, yr = dy * c - dx * s;
  R ipsi = rcp(sqrt_(q * q * xr * xr + yr * yr));
  R u = sq * xr * ipsi, v = sq * yr * ipsi;
  R fu = atan_(u), fv = atanh_(v);
  R arx = A * fu, ary = A * fv;
  R ax = arx * c - ary * s, ay = arx * s + ary * c;
  R grx = gx * c + gy * s, gry = gy * c - gx * s;
  R g_phi = gy * ax - gx * ay;
  R gA = grx * fu + gry * fv;
  R gu = grx * A * rcp((R)1 + u * u);
  R gv = gry * A * rcp((R)1 - v * v);
  R g_sq = (gu * xr + gv * yr) * ipsi;
  R gpsi = -(gu * u + gv * v) * ipsi;
  R gxr = gu * sq * ipsi + gpsi * q * q * xr * ipsi;
  R gyr = gv * sq * ipsi + gpsi * yr * ipsi;
  R g_q = gpsi * q * xr * xr * ipsi;
  R gdx = gxr * c - gyr * s, gdy = gxr * s + gyr * c;
  g_phi += gxr * yr - gyr * xr;
  acc[SIEA_CX] -= gdx;
  acc[SIEA_CY] -= gdy;
  acc[SIEA_PHI] += g_phi;
  acc[SIEA_Q] += g_q;
  acc[SIEA_SQ] += g_sq;
  acc[SIEA_A] += gA;
}
template <class R> GL_HD void sie_finalize(const R* p, const R* acc, R* g) {
  Ellip<R> el = ellip_prep(p[1], p[2], (R)0.9999);
  R q = el.q;
  R b = einstein_b(p[0], q);
  R sq = p_sqrt((R)1 - q * q);
  R g_b = acc[SIEA_A] / sq;
  R g_sq = acc[SIEA_SQ] - acc[SIEA_A] * b / (sq * sq);
  R g_q = acc[SIEA_Q] - g_sq * q / sq;
  R g_te, g_e1, g_e2;
  ellip_chain(p[0], p[1], p[2], (R)0.9999, g_b, g_q, acc[SIEA_PHI], g_te, g_e1, g_e2);
  g[0] = g_te;
  g[1] = g_e1;
  g[2] = g_e2;
  g[3] = acc[SIEA_CX];
  g[4] = acc[SIEA_CY];
}

// =============================================================================================
// NFW  (tf/profiles/mass/nfw.py:15-52)
// =============================================================================================
// g(X) = ln(X/2) + w(X),  w = acosh(1/X)/sqrt(1-X^2) (X<1), acos(1/X)/sqrt(X^2-1) (X>1), and the
// reference's g(1) = 1.0 (ones-initialised, nfw.py:38; analytic value is 1-ln2 -- replicated).
// Both branches are ONE analytic function of D = 1-X^2:  w = atanh(sqrt D)/sqrt D = sum D^k/(2k+1),
// so near X = 1 the series replaces the 0/0 cancellation of the reference's formula, and
// g'(X) = X (w-1)/D comes out of the same series without a subtraction.
template <class R> GL_HD void nfw_gw(R X, R& g, R& gp) {
  R iX = rcp(X);
  R D = ((R)1 - X) * ((R)1 + X);
  R aD = fabs_(D);
  R w, w1;  // w1 = (w-1)/D
  if (aD < (R)0.1) {
    w1 = (R)(1.0 / 3) + D * ((R)(1.0 / 5) + D * ((R)(1.0 / 7) + D * ((R)(1.0 / 9) + D * ((R)(1.0 / 11) +
         D * ((R)(1.0 / 13) + D * ((R)(1.0 / 15) + D * (R)(1.0 / 17)))))));
    w = (R)1 + D * w1;
    g = log_((R)0.5 * X) + w;
  } else {
    R sD = sqrt_(aD);
    R isD = rcp(sD);
    if (D > (R)0) {
      w = log_(((R)1 + sD) * iX) * isD;  // atanh(s)/s = ln((1+s)/X)/s
      if (X < (R)0.6) {
        // g = O(X^2 ln X) while ln(X/2) and w are O(ln X): regroup so nothing cancels.
        // With delta = 1 - s = X^2/(1+s):  g = [delta ln(2/X) + log1p(-delta/2)] / s
        R delta = X * X * rcp((R)1 + sD);
        R z = (R)-0.5 * delta;
        R yy = z * rcp((R)2 + z), y2 = yy * yy;  // log1p(z) = 2 atanh(z/(2+z)), |z| <= 0.11
        R l1p = (R)2 * yy * ((R)1 + y2 * ((R)(1.0 / 3) + y2 * ((R)(1.0 / 5) + y2 * (R)(1.0 / 7))));
        g = (delta * log_((R)2 * iX) + l1p) * isD;
      } else {
        g = log_((R)0.5 * X) + w;
      }
    } else {
      w = atan_(sD) * isD;
      g = log_((R)0.5 * X) + w;
    }
    w1 = (w - (R)1) * rcp(D);
  }
  bool one = (X == (R)1);
  g = one ? (R)1 : g;
  gp = one ? (R)0 : X * w1;
}
template <class R> GL_HD void nfw_prep(const R* p, R* d) {
  R Rs = p[0], alpha_Rs = p[1];
  R rho0 = alpha_Rs / ((R)4 * Rs * Rs * ((R)1 - (R)kLn2));  // nfw.py:17, from the UNclamped Rs
  R Rsc = fmax_((R)1e-7, Rs);                              // nfw.py:27
  d[NFW_CX] = p[2];
  d[NFW_CY] = p[3];
  d[NFW_INVRS] = (R)1 / Rsc;
  d[NFW_K0] = (R)4 * rho0 * Rsc;
}
template <class R> GL_HD void nfw_fwd(const R* d, R x, R y, R& ax, R& ay) {
  R dx = x - d[NFW_CX], dy = y - d[NFW_CY];
  R R0 = sqrt_(dx * dx + dy * dy);
  R Rc = fmax_((R)1e-7, R0);                    // nfw.py:26
  R X = fmax_((R)1e-6, Rc * d[NFW_INVRS]);      // nfw.py:37
  R g, gp;
  nfw_gw(X, g, gp);
  R iX = rcp(X);
  R a = d[NFW_K0] * g * iX * iX;                // nfw.py:30
  ax = a * dx;
  ay = a * dy;
}
template <class R> GL_HD void nfw_vjp(const R* d, R x, R y, R gx, R gy, R* acc) {
  R dx = x - d[NFW_CX], dy = y - d[NFW_CY];
  R R0 = sqrt_(dx * dx + dy * dy);
  R Rc = fmax_((R)1e-7, R0);
  R X0 = Rc * d[NFW_INVRS];
  R X = fmax_((R)1e-6, X0);
  R g, gp;
  nfw_gw(X, g, gp);
  R iX = rcp(X);
  R h = g * iX * iX;
  R K0 = d[NFW_K0];
  R a = K0 * h;
  R ga = gx * dx + gy * dy;
  R gdx = gx * a, gdy = gy * a;
  R hp = gp * iX * iX - (R)2 * h * iX;
  R gX0 = (X0 > (R)1e-6) ? ga * K0 * hp : (R)0;
  R gRc = gX0 * d[NFW_INVRS];
  R gR0 = (R0 > (R)1e-7) ? gRc : (R)0;
  R iR0 = (R0 > (R)0) ? rcp(R0) : (R)0;
  gdx += gR0 * dx * iR0;
  gdy += gR0 * dy * iR0;
  acc[NFWA_CX] -= gdx;
  acc[NFWA_CY] -= gdy;
  acc[NFWA_RS] -= gX0 * X0 * d[NFW_INVRS];  // d X0 / d Rs_clamped = -X0/Rs
  acc[NFWA_K0] += ga * h;
}
template <class R> GL_HD void nfw_finalize(const R* p, const R* acc, R* g) {
  R Rs = p[0], alpha_Rs = p[1];
  R k = (R)1 / ((R)1 - (R)kLn2);
  R Rsc = fmax_((R)1e-7, Rs);
  R rho0 = alpha_Rs * k / ((R)4 * Rs * Rs);
  // K0 = 4 rho0 Rsc = alpha_Rs k Rsc / Rs^2
  R gK0 = acc[NFWA_K0];
  g[1] = gK0 * k * Rsc / (Rs * Rs);
  R gRs = gK0 * ((R)-2 * alpha_Rs * k * Rsc / (Rs * Rs * Rs));
  if (Rs > (R)1e-7) gRs += gK0 * (R)4 * rho0 + acc[NFWA_RS];
  g[0] = gRs;
  g[2] = acc[NFWA_CX];
  g[3] = acc[NFWA_CY];
}

// =============================================================================================
// SHEAR (tf/profiles/mass/shear.py:14-16; evaluated at the UN-shifted coordinates) and
// SIS   (tf/profiles/mass/sis.py:12-17)
// =============================================================================================
template <class R> GL_HD void shear_prep(const R* p, R* d) { d[SHR_G1] = p[0]; d[SHR_G2] = p[1]; d[2] = (R)0; d[3] = (R)0; }
template <class R> GL_HD void shear_fwd(const R* d, R x, R y, R& ax, R& ay) {
  ax = d[SHR_G1] * x + d[SHR_G2] * y;
  ay = d[SHR_G2] * x - d[SHR_G1] * y;
}
template <class R> GL_HD void shear_vjp(const R* d, R x, R y, R gx, R gy, R* acc) {
  (void)d;
  acc[0] += gx * x - gy * y;
  acc[1] += gx * y + gy * x;
}
template <class R> GL_HD void shear_finalize(const R* p, const R* acc, R* g) { (void)p; g[0] = acc[0]; g[1] = acc[1]; }

template <class R> GL_HD void sis_prep(const R* p, R* d) { d[SIS_CX] = p[1]; d[SIS_CY] = p[2]; d[SIS_TE] = p[0]; d[3] = (R)0; }
template <class R> GL_HD void sis_fwd(const R* d, R x, R y, R& ax, R& ay) {
  R dx = x - d[SIS_CX], dy = y - d[SIS_CY];
  R R0 = sqrt_(dx * dx + dy * dy);
  R a = (R0 == (R)0) ? (R)0 : d[SIS_TE] * rcp(R0);
  ax = a * dx;
  ay = a * dy;
}
template <class R> GL_HD void sis_vjp(const R* d, R x, R y, R gx, R gy, R* acc) {
  R dx = x - d[SIS_CX], dy = y - d[SIS_CY];
  R R0 = sqrt_(dx * dx + dy * dy);
  bool z = (R0 == (R)0);
  R iR = z ? (R)0 : rcp(R0);
  R a = d[SIS_TE] * iR;
  R ga = gx * dx + gy * dy;
  R gR0 = -ga * a * iR;
  R gdx = gx * a + gR0 * dx * iR, gdy = gy * a + gR0 * dy * iR;
  acc[0] -= gdx;
  acc[1] -= gdy;
  acc[2] += ga * iR;
}
template <class R> GL_HD void sis_finalize(const R* p, const R* acc, R* g) { (void)p; g[0] = acc[2]; g[1] = acc[0]; g[2] = acc[1]; }

// =============================================================================================
// SERSIC / SERSIC_ELLIPSE  (tf/profiles/light/sersic.py:29-80)
// =============================================================================================
// raw rows: SERSIC [R_sersic,n_sersic,center_x,center_y,Ie]; ELLIPSE [R_sersic,n_sersic,e1,e2,center_x,center_y,Ie]
template <class R> GL_HD void sersic_prep(const R* p, bool ellipse, R* d) {
  R e1 = ellipse ? p[2] : (R)0, e2 = ellipse ? p[3] : (R)0;
  const R* rest = ellipse ? p + 4 : p + 2;
  Ellip<R> el = ellip_prep(e1, e2, (R)0.9999);
  R sq = p_sqrt(el.q);
  d[SER_CX] = rest[0];
  d[SER_CY] = rest[1];
  d[SER_C] = el.cphi;
  d[SER_S] = el.sphi;
  d[SER_SQ] = sq;
  d[SER_ISQ] = (R)1 / sq;
  d[SER_INVRS] = (R)1 / p[0];
  d[SER_INVN] = (R)1 / p[1];
  d[SER_BN] = (R)1.9992 * p[1] - (R)0.3271;  // sersic.py:33
  d[SER_IE] = rest[2];
  d[SER_L2IRS] = -p_log(p[0]) * (R)kLog2e;
  // products of per-sample constants the cluster kernel's VJP multiplies pixel values with: wave-uniform, but there is no scalar
  // float unit to form them on, so once here instead of four vector instructions per source and tile there
  d[SER_CG] = d[SER_IE] * d[SER_BN] * d[SER_INVN];
  d[SER_IRS2] = d[SER_INVRS] * d[SER_INVRS];
  d[SER_PAD] = (R)0;
}
template <class R> GL_HD R sersic_fwd(const R* d, R x, R y) {
  R dx = x - d[SER_CX], dy = y - d[SER_CY];
  R c = d[SER_C], s = d[SER_S];
  R xt1 = (c * dx + s * dy) * d[SER_SQ];
  R xt2 = (c * dy - s * dx) * d[SER_ISQ];
  R Rr = sqrt_(xt1 * xt1 + xt2 * xt2);
  R u = exp2_(log2_(Rr * d[SER_INVRS]) * d[SER_INVN]);  // (R/R_sersic)^(1/n); 0 at R == 0
  return d[SER_IE] * exp_(-d[SER_BN] * (u - (R)1));
}
// gI: cotangent of the surface brightness. Returns I; adds the cotangent of (x, y) to (gpx, gpy).
template <class R> GL_HD R sersic_vjp(const R* d, R x, R y, R gI, R* acc, R& gpx, R& gpy) {
  R dx = x - d[SER_CX], dy = y - d[SER_CY];
  R c = d[SER_C], s = d[SER_S], sq = d[SER_SQ], isq = d[SER_ISQ];
  R a1 = c * dx + s * dy, a2 = c * dy - s * dx;
  R xt1 = a1 * sq, xt2 = a2 * isq;
  R r2 = xt1 * xt1 + xt2 * xt2;
  R Rr = sqrt_(r2);
  bool pos = Rr > (R)0;
  R L2 = log2_(Rr * d[SER_INVRS]);
  R invn = d[SER_INVN], bn = d[SER_BN];
  R u = exp2_(L2 * invn);
  R E = exp_(-bn * (u - (R)1));
  R I = d[SER_IE] * E;
  R tI = gI * I;
  R guu = -tI * bn * u;                                   // g_u * u
  R gL = guu * invn;                                      // cotangent of L = ln(R/R_sersic)
  R k = pos ? gL * rcp(r2) : (R)0;                        // g_R / R
  R gxt1 = k * xt1, gxt2 = k * xt2;
  R ga1 = gxt1 * sq, ga2 = gxt2 * isq;
  R gdx = ga1 * c - ga2 * s, gdy = ga1 * s + ga2 * c;
  acc[SERA_CX] -= gdx;
  acc[SERA_CY] -= gdy;
  acc[SERA_PHI] += ga1 * a2 - ga2 * a1;
  acc[SERA_SQ] += gxt1 * a1 - gxt2 * a2 * isq * isq;
  acc[SERA_L] += gL;
  acc[SERA_INVN] += pos ? guu * L2 * (R)kLn2 : (R)0;      // TF's pow gradient uses where(x>0, log x, 0)
  acc[SERA_BN] -= tI * (u - (R)1);
  acc[SERA_IE] += gI * E;
  gpx += gdx;
  gpy += gdy;
  return I;
}
template <class R> GL_HD void sersic_finalize(const R* p, bool ellipse, const R* acc, R* g) {
  R Rs = p[0], n = p[1];
  g[0] = -acc[SERA_L] / Rs;
  g[1] = acc[SERA_BN] * (R)1.9992 - acc[SERA_INVN] / (n * n);
  if (ellipse) {
    Ellip<R> el = ellip_prep(p[2], p[3], (R)0.9999);
    R sq = p_sqrt(el.q);
    R g_q = acc[SERA_SQ] / ((R)2 * sq);
    R g_te, g_e1, g_e2;
    ellip_chain((R)0, p[2], p[3], (R)0.9999, (R)0, g_q, acc[SERA_PHI], g_te, g_e1, g_e2);
    g[2] = g_e1;
    g[3] = g_e2;
    g[4] = acc[SERA_CX];
    g[5] = acc[SERA_CY];
    g[6] = acc[SERA_IE];
  } else {
    g[2] = acc[SERA_CX];
    g[3] = acc[SERA_CY];
    g[4] = acc[SERA_IE];
  }
}

// =============================================================================================
// SHAPELETS  (tf/profiles/light/shapelets.py:20-85)
// =============================================================================================
// raw row [beta, center_x, center_y, amp_0..amp_{L-1}], amplitudes in the reference's (n1,n2) order
// (0,0),(1,0),(0,1),(2,0),(1,1),(0,2),...  i.e. i = n(n+1)/2 + n2 with n = n1+n2 (shapelets.py:41-46).
template <class R> GL_HD void shapelets_prep(const R* p, int n_max, R* d) {
  d[SHP_CX] = p[1];
  d[SHP_CY] = p[2];
  d[SHP_IB] = (R)1 / p[0];
  d[SHP_NMAX] = (R)n_max;
  int L = sh_layers(n_max);
  for (int i = 0; i < L; ++i) d[SHP_AMP + i] = p[3 + i];
  if (n_max > SH_CAP) {  // runtime-order path: the triangle alone (zero-padded to its block)
    for (int i = L; i < ((SH_MAXLB + 3) & ~3); ++i) d[SHP_AMP + i] = (R)0;
    return;
  }
  for (int i = L; i < ((SH_MAXL + 3) & ~3); ++i) d[SHP_AMP + i] = (R)0;  // the separable kernels run the full triangle
  for (int n1 = 0; n1 < SH_SQ; ++n1)
    for (int n2 = 0; n2 < SH_SQ; ++n2) {
      const int n = n1 + n2;
      d[SHP_SQ + n1 * SH_SQ + n2] = n <= n_max ? p[3 + n * (n + 1) / 2 + n2] * (R)(SH_K[n1] * SH_K[n2]) : (R)0;
    }
}

// orthonormal Gauss-Hermite functions without the Gaussian: X_n = H_n / sqrt(2^n sqrt(pi) n!)
// (shapelets.py:48,77-85), by the normalised three-term recurrence; dX_n/du = sqrt(2n) X_{n-1}.
template <class R, int CAP> GL_HD void hermite_basis(R u, int n_max, R* Xv, R* dXv) {
  Xv[0] = (R)0.75112554446494248286;  // pi^(-1/4)
  dXv[0] = (R)0;
#pragma unroll
  for (int n = 1; n <= CAP; ++n) {
    if (n <= n_max) {
      R a = (R)::sqrt(2.0 / n), bcoef = (R)::sqrt((n - 1.0) / n);
      Xv[n] = a * u * Xv[n - 1] - (n >= 2 ? bcoef * Xv[n - 2] : (R)0);
      dXv[n] = (R)::sqrt(2.0 * n) * Xv[n - 1];
    } else {
      Xv[n] = (R)0;
      dXv[n] = (R)0;
    }
  }
}
// table mode (shapelets.py:39-40,55-65): linear interpolation of phi_n on 6000 nodes over [-5,5], zero
// outside (tfp.math.interp_regular_1d_grid, fill 0/0).  `tab` is node-major [6000][stride] so the
// n_max+1 orders of one node are contiguous.  Derivative = slope of the segment (what autodiff of
// the interpolation gives).
template <class R, int CAP> GL_HD void table_basis(const float* tab, int stride, R u, int n_max, R* Xv, R* dXv) {
  const R scale = (R)(SH_NODES - 1) / (R)10;
  R fi = (u + (R)5) * scale;  // == (u - x_min)/(x_max - x_min) * (ny - 1)
  bool inside = (fi >= (R)0) && (fi <= (R)(SH_NODES - 1));
  R fic = clamp_(fi, (R)0, (R)(SH_NODES - 1));
  R fb = floor_(fic);
  R fa = fmin_(fb + (R)1, (R)(SH_NODES - 1));
  fb = fmax_(fa - (R)1, (R)0);
  R tt = fic - fb;
  const float* rb = tab + (int64_t)(int)fb * stride;
  const float* ra = tab + (int64_t)(int)fa * stride;
#pragma unroll
  for (int n = 0; n <= CAP; ++n) {
    if (n <= n_max && inside) {
      R yb = (R)rb[n], ya = (R)ra[n];
      Xv[n] = tt * ya + ((R)1 - tt) * yb;
      dXv[n] = (ya - yb) * scale;
    } else {
      Xv[n] = (R)0;
      dXv[n] = (R)0;
    }
  }
}

// Evaluates S = sum_i amp_i X_{n1}(u) Y_{n2}(v) and (optionally) dS/du, dS/dv and the per-amplitude
// basis products.  GRAD==false keeps it to the forward sum.
template <class R, int CAP, bool GRAD>
GL_HD void shapelets_sum(const R* amp, int n_max, const R* Xv, const R* dXv, const R* Yv, const R* dYv,
                         R& S, R& Su, R& Sv) {
  S = (R)0; Su = (R)0; Sv = (R)0;
#pragma unroll
  for (int n = 0; n <= CAP; ++n) {
    if (n <= n_max) {
#pragma unroll
      for (int n2 = 0; n2 <= n; ++n2) {
        const int n1 = n - n2;
        R a = amp[n * (n + 1) / 2 + n2];
        S += a * Xv[n1] * Yv[n2];
        if (GRAD) {
          Su += a * dXv[n1] * Yv[n2];
          Sv += a * Xv[n1] * dYv[n2];
        }
      }
    }
  }
}

// the L basis images phi_n1(u) phi_n2(v) themselves (use_lstsq=True, shapelets.py:61-62,71-72), in amplitude order
template <class R, int CAP, class F>
GL_HD void shapelets_basis(const R* d, const float* tab, int stride, bool interp, R x, R y, F&& emit) {
  const int n_max = (int)d[SHP_NMAX];
  R ib = d[SHP_IB];
  R u = (x - d[SHP_CX]) * ib, v = (y - d[SHP_CY]) * ib;
  R Xv[CAP + 1], dXv[CAP + 1], Yv[CAP + 1], dYv[CAP + 1];
  R fac = (R)1;
  if (interp) {
    table_basis<R, CAP>(tab, stride, u, n_max, Xv, dXv);
    table_basis<R, CAP>(tab, stride, v, n_max, Yv, dYv);
  } else {
    hermite_basis<R, CAP>(u, n_max, Xv, dXv);
    hermite_basis<R, CAP>(v, n_max, Yv, dYv);
    fac = exp_(-(u * u + v * v) * (R)0.5);
  }
#pragma unroll
  for (int n = 0; n <= CAP; ++n) {
    if (n <= n_max) {
#pragma unroll
      for (int n2 = 0; n2 <= n; ++n2) emit(n * (n + 1) / 2 + n2, fac * Xv[n - n2] * Yv[n2]);
    }
  }
}

template <class R, int CAP>
GL_HD R shapelets_fwd(const R* d, const float* tab, int stride, bool interp, R x, R y) {
  const int n_max = (int)d[SHP_NMAX];
  R ib = d[SHP_IB];
  R u = (x - d[SHP_CX]) * ib, v = (y - d[SHP_CY]) * ib;
  R Xv[CAP + 1], dXv[CAP + 1], Yv[CAP + 1], dYv[CAP + 1];
  R fac = (R)1;
  if (interp) {
    table_basis<R, CAP>(tab, stride, u, n_max, Xv, dXv);
    table_basis<R, CAP>(tab, stride, v, n_max, Yv, dYv);
  } else {
    hermite_basis<R, CAP>(u, n_max, Xv, dXv);
    hermite_basis<R, CAP>(v, n_max, Yv, dYv);
    fac = exp_(-(u * u + v * v) * (R)0.5);  // shapelets.py:70
  }
  R S, Su, Sv;
  shapelets_sum<R, CAP, false>(d + SHP_AMP, n_max, Xv, dXv, Yv, dYv, S, Su, Sv);
  return fac * S;
}

// the same with the amplitude triangle somewhere else than behind the four constants (plugin-level evaluation of orders
// above SH_CAP reads it straight from the parameter row)
template <class R, int CAP>
GL_HD R shapelets_fwd_amp(const R* d, const R* amp, const float* tab, int stride, bool interp, R x, R y) {
  const int n_max = (int)d[SHP_NMAX];
  R ib = d[SHP_IB];
  R u = (x - d[SHP_CX]) * ib, v = (y - d[SHP_CY]) * ib;
  R Xv[CAP + 1], dXv[CAP + 1], Yv[CAP + 1], dYv[CAP + 1];
  R fac = (R)1;
  if (interp) {
    table_basis<R, CAP>(tab, stride, u, n_max, Xv, dXv);
    table_basis<R, CAP>(tab, stride, v, n_max, Yv, dYv);
  } else {
    hermite_basis<R, CAP>(u, n_max, Xv, dXv);
    hermite_basis<R, CAP>(v, n_max, Yv, dYv);
    fac = exp_(-(u * u + v * v) * (R)0.5);
  }
  R S, Su, Sv;
  shapelets_sum<R, CAP, false>(amp, n_max, Xv, dXv, Yv, dYv, S, Su, Sv);
  return fac * S;
}

// acc layout: [cx, cy, ib, amp_0..]; amplitudes accumulate straight into acc (registers)
template <class R, int CAP>
GL_HD R shapelets_vjp(const R* d, const float* tab, int stride, bool interp, R x, R y, R gI, R* acc,
                      R& gpx, R& gpy) {
  const int n_max = (int)d[SHP_NMAX];
  R ib = d[SHP_IB];
  R dx = x - d[SHP_CX], dy = y - d[SHP_CY];
  R u = dx * ib, v = dy * ib;
  R Xv[CAP + 1], dXv[CAP + 1], Yv[CAP + 1], dYv[CAP + 1];
  R fac = (R)1;
  if (interp) {
    table_basis<R, CAP>(tab, stride, u, n_max, Xv, dXv);
    table_basis<R, CAP>(tab, stride, v, n_max, Yv, dYv);
  } else {
    hermite_basis<R, CAP>(u, n_max, Xv, dXv);
    hermite_basis<R, CAP>(v, n_max, Yv, dYv);
    fac = exp_(-(u * u + v * v) * (R)0.5);
  }
  R S, Su, Sv;
  shapelets_sum<R, CAP, true>(d + SHP_AMP, n_max, Xv, dXv, Yv, dYv, S, Su, Sv);
  R I = fac * S;
  R gS = gI * fac;
#pragma unroll
  for (int n = 0; n <= CAP; ++n) {
    if (n <= n_max) {
#pragma unroll
      for (int n2 = 0; n2 <= n; ++n2) acc[SHPA_AMP + n * (n + 1) / 2 + n2] += gS * Xv[n - n2] * Yv[n2];
    }
  }
  R gu = gS * Su, gv = gS * Sv;
  if (!interp) {  // d fac/du = -u fac
    R gIf = gI * I;
    gu -= gIf * u;
    gv -= gIf * v;
  }
  R gdx = gu * ib, gdy = gv * ib;
  acc[SHPA_CX] -= gdx;
  acc[SHPA_CY] -= gdy;
  acc[SHPA_IB] += gu * dx + gv * dy;
  gpx += gdx;
  gpy += gdy;
  return I;
}
// The same VJP for orders that do not fit a register-resident accumulator block (n_max up to SH_CAPB: 231 amplitudes): the
// amplitude gradient leaves shell by shell -- `flush(first amplitude index, count, values)` after every n = n1 + n2 -- so a
// caller keeps at most CAP + 1 values live; `acc3` takes the centre / 1/beta terms.
template <class R, int CAP, class F>
GL_HD R shapelets_vjp_shells(const R* d, const float* tab, int stride, bool interp, R x, R y, R gI, R* acc3, R& gpx, R& gpy,
                             F&& flush) {
  const int n_max = (int)d[SHP_NMAX];
  R ib = d[SHP_IB];
  R dx = x - d[SHP_CX], dy = y - d[SHP_CY];
  R u = dx * ib, v = dy * ib;
  R Xv[CAP + 1], dXv[CAP + 1], Yv[CAP + 1], dYv[CAP + 1];
  R fac = (R)1;
  if (interp) {
    table_basis<R, CAP>(tab, stride, u, n_max, Xv, dXv);
    table_basis<R, CAP>(tab, stride, v, n_max, Yv, dYv);
  } else {
    hermite_basis<R, CAP>(u, n_max, Xv, dXv);
    hermite_basis<R, CAP>(v, n_max, Yv, dYv);
    fac = exp_(-(u * u + v * v) * (R)0.5);
  }
  R S, Su, Sv;
  shapelets_sum<R, CAP, true>(d + SHP_AMP, n_max, Xv, dXv, Yv, dYv, S, Su, Sv);
  R I = fac * S;
  R gS = gI * fac;
#pragma unroll
  for (int n = 0; n <= CAP; ++n) {
    if (n <= n_max) {
      R vals[CAP + 1];
#pragma unroll
      for (int n2 = 0; n2 <= CAP; ++n2) vals[n2] = n2 <= n ? gS * Xv[n2 <= n ? n - n2 : 0] * Yv[n2] : (R)0;
      flush(n * (n + 1) / 2, n + 1, vals);
    }
  }
  R gu = gS * Su, gv = gS * Sv;
  if (!interp) {
    R gIf = gI * I;
    gu -= gIf * u;
    gv -= gIf * v;
  }
  R gdx = gu * ib, gdy = gv * ib;
  acc3[SHPA_CX] -= gdx;
  acc3[SHPA_CY] -= gdy;
  acc3[SHPA_IB] += gu * dx + gv * dy;
  gpx += gdx;
  gpy += gdy;
  return I;
}
template <class R> GL_HD void shapelets_finalize(const R* p, int n_max, const R* acc, R* g) {
  R beta = p[0];
  g[0] = -acc[SHPA_IB] / (beta * beta);
  g[1] = acc[SHPA_CX];
  g[2] = acc[SHPA_CY];
  int L = sh_layers(n_max);
  for (int i = 0; i < L; ++i) g[3 + i] = acc[SHPA_AMP + i];
}

// =============================================================================================
// pixel chi^2 terms  (tf/model.py:89-101) and the cotangent of the model image
// =============================================================================================
// sigma = error_map if given, else sqrt(bg^2 + m/exp_time) with NO clip (NaN if negative, Appendix A10).
// chi2 += ((m-o)/sigma)^2 w ; norm += log(2 pi sigma^2) w ; loglike = -1/2 (chi2 + norm).
// d loglike / d m = w [ -(m-o)/s2 + (m-o)^2/(2 s2^2 t) - 1/(2 s2 t) ]   (last two vanish with error_map)
template <class R>
GL_HD void chi2_terms(R m, R o, R w, bool has_err, R err, R bg2, R inv_t, R& chi2, R& norm) {
  R sig = has_err ? err : sqrt_(bg2 + m * inv_t);
  R r = (m - o) * rcp(sig);
  chi2 = r * r * w;
  norm = log_((R)(2 * kPi) * sig * sig) * w;
}
template <class R> GL_HD R chi2_gm(R m, R o, R w, bool has_err, R err, R bg2, R inv_t) {
  R dmo = m - o;
  if (has_err) return -w * dmo * rcp(err * err);
  R is2 = rcp(bg2 + m * inv_t);
  return w * (-dmo * is2 + (R)0.5 * inv_t * is2 * (dmo * dmo * is2 - (R)1));
}

}  // namespace glp
