// gl_extra.h -- the remaining profile families of the reference, in the structure of gl_profiles.h
// (prep / fwd / vjp / finalize, generic in the real type R):
//   NFW_ELLIPSE  NFW on coordinates stretched by sqrt(1 -+ e)          tf/profiles/mass/nfw.py:97-134
//   TNFW         truncated NFW (Baltz, Marshall & Oguri 2009 form)     tf/profiles/mass/tnfw.py:10-62
//   CORE_SERSIC  core-Sersic light, AS WRITTEN in the reference        tf/profiles/light/sersic.py:83-132
#pragma once
#include "gl_profiles.h"

namespace glp {

// ---------------------------------------------------------------------------------------------
// NFW_ELLIPSE  [Rs, alpha_Rs, e1, e2, center_x, center_y]   (nfw.py:100)
//   e = |1 - q^2| / (1 + q^2) (nfw.py:128-133);  (x', y') rotated by phi;  p = (x' sqrt(1-e), y' sqrt(1+e));
//   alpha' = nfwAlpha(|p|, p) * (sqrt(1-e), sqrt(1+e));  rotate back.
// ---------------------------------------------------------------------------------------------
enum { NFE_CX = 0, NFE_CY, NFE_INVRS, NFE_K0, NFE_C, NFE_S, NFE_SM, NFE_SP, NFE_ND };
enum { NFEA_CX = 0, NFEA_CY, NFEA_RS, NFEA_K0, NFEA_PHI, NFEA_E, NFE_NACC };

template <class R> GL_HD void nfw_ell_prep(const R* p, R* d) {
  R q4[4] = {p[0], p[1], (R)0, (R)0};
  R dn[NFW_ND];
  nfw_prep<R>(q4, dn);
  Ellip<R> el = ellip_prep(p[2], p[3], (R)0.9999);
  R e = fabs_((R)1 - el.q * el.q) / ((R)1 + el.q * el.q);
  d[NFE_CX] = p[4];
  d[NFE_CY] = p[5];
  d[NFE_INVRS] = dn[NFW_INVRS];
  d[NFE_K0] = dn[NFW_K0];
  d[NFE_C] = el.cphi;
  d[NFE_S] = el.sphi;
  d[NFE_SM] = p_sqrt((R)1 - e);
  d[NFE_SP] = p_sqrt((R)1 + e);
}
template <class R> GL_HD void nfw_ell_fwd(const R* d, R x, R y, R& ax, R& ay) {
  R dx = x - d[NFE_CX], dy = y - d[NFE_CY];
  R c = d[NFE_C], s = d[NFE_S], sm = d[NFE_SM], sp = d[NFE_SP];
  R xr = dx * c + dy * s, yr = dy * c - dx * s;
  const R dl[NFW_ND] = {(R)0, (R)0, d[NFE_INVRS], d[NFE_K0]};
  R fx, fy;
  nfw_fwd<R>(dl, xr * sm, yr * sp, fx, fy);
  fx = fx * sm;
  fy = fy * sp;
  ax = fx * c - fy * s;
  ay = fx * s + fy * c;
}
template <class R> GL_HD void nfw_ell_vjp(const R* d, R x, R y, R gx, R gy, R* acc) {
  R dx = x - d[NFE_CX], dy = y - d[NFE_CY];
  R c = d[NFE_C], s = d[NFE_S], sm = d[NFE_SM], sp = d[NFE_SP];
  R xr = dx * c + dy * s, yr = dy * c - dx * s;
  R px = xr * sm, py = yr * sp;
  const R dl[NFW_ND] = {(R)0, (R)0, d[NFE_INVRS], d[NFE_K0]};
  R f0x, f0y;
  nfw_fwd<R>(dl, px, py, f0x, f0y);  // before the outer stretch
  R grx = gx * c + gy * s, gry = gy * c - gx * s;
  R a4[NFW_NACC] = {(R)0, (R)0, (R)0, (R)0};
  nfw_vjp<R>(dl, px, py, grx * sm, gry * sp, a4);
  R gpx = -a4[NFWA_CX], gpy = -a4[NFWA_CY];  // cotangent of the stretched position
  R g_sm = grx * f0x + gpx * xr, g_sp = gry * f0y + gpy * yr;
  R gxr = gpx * sm, gyr = gpy * sp;
  R fx = f0x * sm, fy = f0y * sp;
  R ax = fx * c - fy * s, ay = fx * s + fy * c;
  acc[NFEA_CX] -= gxr * c - gyr * s;
  acc[NFEA_CY] -= gxr * s + gyr * c;
  acc[NFEA_RS] += a4[NFWA_RS];
  acc[NFEA_K0] += a4[NFWA_K0];
  acc[NFEA_PHI] += gy * ax - gx * ay + gxr * yr - gyr * xr;
  acc[NFEA_E] += (R)0.5 * (g_sp / sp - g_sm / sm);
}
template <class R> GL_HD void nfw_ell_finalize(const R* p, const R* acc, R* g) {
  const R q4[4] = {p[0], p[1], (R)0, (R)0};
  const R a4[NFW_NACC] = {(R)0, (R)0, acc[NFEA_RS], acc[NFEA_K0]};
  R g4[4];
  nfw_finalize<R>(q4, a4, g4);
  g[0] = g4[0];
  g[1] = g4[1];
  Ellip<R> el = ellip_prep(p[2], p[3], (R)0.9999);
  R c2 = el.c * el.c;  // e = 2c/(1+c^2)
  R g_c = acc[NFEA_E] * (R)2 * ((R)1 - c2) / (((R)1 + c2) * ((R)1 + c2));
  ellip_chain_c(p[2], p[3], (R)0.9999, g_c, acc[NFEA_PHI], g[2], g[3]);
  g[4] = acc[NFEA_CX];
  g[5] = acc[NFEA_CY];
}

// ---------------------------------------------------------------------------------------------
// TNFW  [Rs, alpha_Rs, r_trunc, center_x, center_y]   (tnfw.py:12)
//   X = max(R, 0.001 Rs)/Rs, tau = r_trunc/Rs, F(X) = atanh(sqrt(1-X^2))/sqrt(1-X^2) | atan(.)/. | 1,
//   L = ln(X/(tau + sqrt(tau^2+X^2))),
//   g = tau^2/(tau^2+1)^2 [ (tau^2+1+2(X^2-1)) F + tau pi + (tau^2-1) ln tau + sqrt(tau^2+X^2)(-pi + L (tau^2-1)/tau) ],
//   alpha = 4 rho0 Rs g / X^2 (x, y),  rho0 = alpha_Rs/(4 Rs^2 (1 + ln 1/2)).
// ---------------------------------------------------------------------------------------------
enum { TNF_CX = 0, TNF_CY, TNF_INVRS, TNF_K, TNF_TAU, TNF_ND = 8 };
enum { TNFA_CX = 0, TNFA_CY, TNFA_K, TNFA_XRS, TNFA_TAU, TNF_NACC };

// F(X) and dF/dX; F = w(D), D = 1 - X^2, through X = 1 with the series of nfw_gw
template <class R> GL_HD void tnfw_F(R X, R& F, R& Fp) {
  R D = ((R)1 - X) * ((R)1 + X);
  R aD = fabs_(D);
  R w, w1;  // w1 = (w - 1)/D
  if (aD < (R)0.1) {
    w1 = (R)(1.0 / 3) + D * ((R)(1.0 / 5) + D * ((R)(1.0 / 7) + D * ((R)(1.0 / 9) + D * ((R)(1.0 / 11) +
         D * ((R)(1.0 / 13) + D * ((R)(1.0 / 15) + D * (R)(1.0 / 17)))))));
    w = (R)1 + D * w1;
  } else {
    R sD = sqrt_(aD);
    w = (D > (R)0) ? atanh_(sD) / sD : atan_(sD) / sD;
    w1 = (w - (R)1) / D;
  }
  F = w;
  // dF/dX = (X^2 F - 1)/(X D);  X^2 F - 1 = X^2 (1 + D w1) - 1 = D (X^2 w1 - 1): no division by D
  Fp = (X * X * w1 - (R)1) / X;
}
template <class R> GL_HD void tnfw_g_impl(R X, R tau, R& g, R& gX, R& gT) {
  const R pi = (R)kPi;
  R T = tau * tau, Tp1 = T + (R)1;
  R S = sqrt_(T + X * X);
  R L = log_(X / (tau + S));
  R F, Fp;
  tnfw_F(X, F, Fp);
  R lt = log_(tau);
  R m = (T - (R)1) / tau;
  R br = -pi + L * m;
  R A = T / (Tp1 * Tp1);
  R B = (Tp1 + (R)2 * (X * X - (R)1)) * F + tau * pi + (T - (R)1) * lt + S * br;
  g = A * B;
  R LX = (R)1 / X - X / (S * (tau + S));
  R BX = (R)4 * X * F + (T + (R)2 * X * X - (R)1) * Fp + (X / S) * br + S * m * LX;
  R BT = (R)2 * tau * F + pi + (R)2 * tau * lt + m + (tau / S) * br + S * (-m / S + L * ((R)1 + (R)1 / T));
  R AT = (R)2 * tau * ((R)1 - T) / (Tp1 * Tp1 * Tp1);
  gX = A * BX;
  gT = AT * B + A * BT;
}
// The bracket cancels from O(ln X) down to O(X^2 ln X) as X -> 0: in fp32 nothing is left below X ~ 0.1 (the
// reference's fp32 graph has the same cancellation).  Those pixels are evaluated in fp64 -- full rate on CDNA4 and
// only the core of the halo -- which keeps the result at fp32 rounding of the exact value everywhere.
template <class R> GL_HD void tnfw_g(R X, R tau, R& g, R& gX, R& gT) {
  using W = typename Wide<R>::type;
  if (sizeof(W) != sizeof(R) && X < (R)0.15) {
    W gw, gxw, gtw;
    tnfw_g_impl<W>(Wide<R>::up(X), Wide<R>::up(tau), gw, gxw, gtw);
    g = Wide<R>::down(gw);
    gX = Wide<R>::down(gxw);
    gT = Wide<R>::down(gtw);
  } else {
    tnfw_g_impl<R>(X, tau, g, gX, gT);
  }
}
template <class R> GL_HD void tnfw_prep(const R* p, R* d) {
  R Rs = p[0];
  for (int i = 0; i < TNF_ND; ++i) d[i] = (R)0;
  d[TNF_CX] = p[3];
  d[TNF_CY] = p[4];
  d[TNF_INVRS] = (R)1 / Rs;
  d[TNF_K] = p[1] / (Rs * ((R)1 - (R)kLn2));  // 4 rho0 Rs
  d[TNF_TAU] = p[2] / Rs;
}
template <class R> GL_HD void tnfw_fwd(const R* d, R x, R y, R& ax, R& ay) {
  R dx = x - d[TNF_CX], dy = y - d[TNF_CY];
  R X = fmax_(sqrt_(dx * dx + dy * dy) * d[TNF_INVRS], (R)0.001);  // R = max(R, 0.001 Rs), tnfw.py:22
  R g, gX, gT;
  tnfw_g(X, d[TNF_TAU], g, gX, gT);
  R a = d[TNF_K] * g / (X * X);
  ax = a * dx;
  ay = a * dy;
}
template <class R> GL_HD void tnfw_vjp(const R* d, R x, R y, R gx, R gy, R* acc) {
  R dx = x - d[TNF_CX], dy = y - d[TNF_CY];
  R R0 = sqrt_(dx * dx + dy * dy);
  R X0 = R0 * d[TNF_INVRS];
  bool free = X0 > (R)0.001;
  R X = free ? X0 : (R)0.001;
  R g, gX, gT;
  tnfw_g(X, d[TNF_TAU], g, gX, gT);
  R iX2 = (R)1 / (X * X);
  R h = g * iX2, K = d[TNF_K];
  R a = K * h;
  R ga = gx * dx + gy * dy;
  R g_h = ga * K;
  R g_X = free ? g_h * (gX * iX2 - (R)2 * h / X) : (R)0;
  R gR0 = g_X * d[TNF_INVRS];
  R iR0 = (R0 > (R)0) ? (R)1 / R0 : (R)0;
  acc[TNFA_CX] -= gx * a + gR0 * dx * iR0;
  acc[TNFA_CY] -= gy * a + gR0 * dy * iR0;
  acc[TNFA_K] += ga * h;
  acc[TNFA_XRS] += g_X * X0;
  acc[TNFA_TAU] += g_h * gT * iX2;
}
template <class R> GL_HD void tnfw_finalize(const R* p, const R* acc, R* g) {
  R Rs = p[0], k1 = (R)1 - (R)kLn2;
  R K = p[1] / (Rs * k1), tau = p[2] / Rs;
  g[0] = -(acc[TNFA_XRS] + acc[TNFA_TAU] * tau + acc[TNFA_K] * K) / Rs;
  g[1] = acc[TNFA_K] / (Rs * k1);
  g[2] = acc[TNFA_TAU] / Rs;
  g[3] = acc[TNFA_CX];
  g[4] = acc[TNFA_CY];
}

// ---------------------------------------------------------------------------------------------
// CORE_SERSIC  [R_sersic, n_sersic, Rb, alpha, gamma, e1, e2, center_x, center_y, Ie]   (sersic.py:85-96)
//   I = Ie (1 + (Rb/R)^alpha)^(gamma/alpha) exp(-bn (R^alpha + Rb^alpha)/(R_sersic^alpha alpha n) - 1)
//   -- as written (sersic.py:121-130: `/ R_sersic ** alpha ** 1.0 / (alpha * n_sersic)` divides, it is not the
//   1/(alpha n) power of the published core-Sersic law), bn = 1.9992 n - 0.3271.
// ---------------------------------------------------------------------------------------------
enum { CSR_CX = 0, CSR_CY, CSR_C, CSR_S, CSR_SQ, CSR_ISQ, CSR_LS, CSR_LB, CSR_AL, CSR_GOA, CSR_BN, CSR_IAN, CSR_IE, CSR_N, CSR_ND = 16 };
enum { CSRA_CX = 0, CSRA_CY, CSRA_PHI, CSRA_SQ, CSRA_LS, CSRA_LB, CSRA_AL, CSRA_GA, CSRA_N, CSRA_IE, CSR_NACC };

template <class R> GL_HD void core_sersic_prep(const R* p, R* d) {
  Ellip<R> el = ellip_prep(p[5], p[6], (R)0.9999);
  R sq = p_sqrt(el.q);
  for (int i = 0; i < CSR_ND; ++i) d[i] = (R)0;
  d[CSR_CX] = p[7];
  d[CSR_CY] = p[8];
  d[CSR_C] = el.cphi;
  d[CSR_S] = el.sphi;
  d[CSR_SQ] = sq;
  d[CSR_ISQ] = (R)1 / sq;
  d[CSR_LS] = p_log(p[0]);
  d[CSR_LB] = p_log(p[2]);
  d[CSR_AL] = p[3];
  d[CSR_GOA] = p[4] / p[3];
  d[CSR_BN] = (R)1.9992 * p[1] - (R)0.3271;
  d[CSR_IAN] = (R)1 / (p[3] * p[1]);
  d[CSR_IE] = p[9];
  d[CSR_N] = p[1];
}
template <class R> struct CoreSersicPix { R a1, a2, xt1, xt2, r2, lR, pp, rr, u, l1u, v, I; };
template <class R> GL_HD void core_sersic_pix(const R* d, R x, R y, CoreSersicPix<R>& o) {
  R dx = x - d[CSR_CX], dy = y - d[CSR_CY];
  R c = d[CSR_C], s = d[CSR_S];
  o.a1 = c * dx + s * dy;
  o.a2 = c * dy - s * dx;
  o.xt1 = o.a1 * d[CSR_SQ];
  o.xt2 = o.a2 * d[CSR_ISQ];
  o.r2 = o.xt1 * o.xt1 + o.xt2 * o.xt2;
  o.lR = (R)0.5 * log_(o.r2);
  R al = d[CSR_AL];
  o.pp = exp_(al * (o.lR - d[CSR_LS]));     // (R/Rs)^alpha
  o.rr = exp_(al * (d[CSR_LB] - d[CSR_LS]));  // (Rb/Rs)^alpha
  o.u = exp_(al * (d[CSR_LB] - o.lR));      // (Rb/R)^alpha
  o.l1u = log_((R)1 + o.u);
  o.v = (o.pp + o.rr) * d[CSR_IAN];
  o.I = d[CSR_IE] * exp_(d[CSR_GOA] * o.l1u - d[CSR_BN] * o.v - (R)1);
}
template <class R> GL_HD R core_sersic_fwd(const R* d, R x, R y) {
  CoreSersicPix<R> o;
  core_sersic_pix(d, x, y, o);
  return o.I;
}
template <class R> GL_HD R core_sersic_vjp(const R* d, R x, R y, R gI, R* acc, R& gpx, R& gpy) {
  CoreSersicPix<R> o;
  core_sersic_pix(d, x, y, o);
  R c = d[CSR_C], s = d[CSR_S], sq = d[CSR_SQ], isq = d[CSR_ISQ];
  R al = d[CSR_AL], goa = d[CSR_GOA], bn = d[CSR_BN], ian = d[CSR_IAN], n = d[CSR_N];
  R tI = gI * o.I;                      // cotangent of ln I
  R uf = o.u / ((R)1 + o.u);
  R gam = goa * al;
  R g_lR = tI * (-gam * uf - bn * o.pp * ian * al);
  R g_lb = tI * (gam * uf - bn * o.rr * ian * al);
  R g_ls = tI * (bn * (o.pp + o.rr) * ian * al);
  R g_ga = tI * o.l1u / al;
  R dlb = d[CSR_LB] - o.lR, dls_p = o.lR - d[CSR_LS], dls_r = d[CSR_LB] - d[CSR_LS];
  R g_al = tI * (-goa / al * o.l1u + goa * uf * dlb - bn * ((o.pp * dls_p + o.rr * dls_r) * ian - o.v / al));
  R g_n = tI * (-(R)1.9992 * o.v + bn * o.v / n);
  bool pos = o.r2 > (R)0;
  R k = pos ? g_lR / o.r2 : (R)0;       // d lR / d xt = xt / r2
  R gxt1 = k * o.xt1, gxt2 = k * o.xt2;
  R ga1 = gxt1 * sq, ga2 = gxt2 * isq;
  R gdx = ga1 * c - ga2 * s, gdy = ga1 * s + ga2 * c;
  acc[CSRA_CX] -= gdx;
  acc[CSRA_CY] -= gdy;
  acc[CSRA_PHI] += ga1 * o.a2 - ga2 * o.a1;
  acc[CSRA_SQ] += gxt1 * o.a1 - gxt2 * o.a2 * isq * isq;
  acc[CSRA_LS] += g_ls;
  acc[CSRA_LB] += g_lb;
  acc[CSRA_AL] += g_al;
  acc[CSRA_GA] += g_ga;
  acc[CSRA_N] += g_n;
  acc[CSRA_IE] += gI * o.I / d[CSR_IE];
  gpx += gdx;
  gpy += gdy;
  return o.I;
}
template <class R> GL_HD void core_sersic_finalize(const R* p, const R* acc, R* g) {
  g[0] = acc[CSRA_LS] / p[0];
  g[1] = acc[CSRA_N];
  g[2] = acc[CSRA_LB] / p[2];
  g[3] = acc[CSRA_AL];
  g[4] = acc[CSRA_GA];
  Ellip<R> el = ellip_prep(p[5], p[6], (R)0.9999);
  R sq = p_sqrt(el.q);
  R g_q = acc[CSRA_SQ] / ((R)2 * sq);
  R g_te;
  ellip_chain((R)0, p[5], p[6], (R)0.9999, (R)0, g_q, acc[CSRA_PHI], g_te, g[5], g[6]);
  g[7] = acc[CSRA_CX];
  g[8] = acc[CSRA_CY];
  g[9] = acc[CSRA_IE];
}

}  // namespace glp
