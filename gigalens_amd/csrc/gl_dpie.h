// gl_dpie.h -- the dPIE family of the cluster-lens workload and the galaxy-population scaling relation:
//   DPIS   dual pseudo-isothermal sphere                         tf/profiles/mass/piemd.py:21-94
//   DPIE   dual pseudo-isothermal elliptical mass distribution   tf/profiles/mass/piemd.py:97-255
//          (Kassiola & Kovner 1993 eq. 4.1.2 evaluated for r_core and r_cut)
//   DPIEP  the dPIS on coordinates stretched by sqrt(1 -+ e)     tf/profiles/mass/piep.py:18-55
//   SCALED ScalingRelation / DPIESubhalo: a catalogue of G galaxies whose (theta_E, r_core, r_cut) follow
//          (L/L*)^power * scale, summed                          tf/profiles/mass/scaling_relation.py:6-70
//
// Same structure as gl_profiles.h (prep / fwd / vjp / finalize, generic in the real type R), with the constants
// of one halo split in two blocks so that a catalogue member and a free-standing halo share the per-pixel code:
//   static  block  DPS_*  -- centre, orientation, ellipticity terms   (catalogue: one per galaxy, shared by samples)
//   dynamic block  DPD_*  -- sorted radii, amplitude, and for catalogue members the 3x3 map W from the per-pixel
//                            cotangents (dS, d rc, d rt) to the three population scales (one per sample x galaxy)
// How the reference's formulas are evaluated is new:  sqrt(r^2+w^2) - w  is taken as  r^2/(sqrt(r^2+w^2)+w)
// (no cancellation in fp32), the complex ratio of piemd.py:236-250 is never divided out (atan2 and log take
// numerator and denominator directly), and |ratio|^2 comes from the four norms the VJP needs anyway.
#pragma once
#include "gl_profiles.h"

namespace glp {

constexpr int DP_NS = 12;  // floats per static block
constexpr int DP_ND = 12;  // floats per dynamic block
enum { DPS_CX = 0, DPS_CY, DPS_CPHI, DPS_SPHI,
       DPS_Q = 4, DPS_IQ, DPS_S2, DPS_IX, DPS_IY, DPS_Z, DPS_ZL /* Z ln2/2 */,  // DPIE
       DPS_M1 = 4, DPS_P1 };                              // DPIS / DPIEP: 1 - e, 1 + e
enum { DPD_RC = 0, DPD_RT, DPD_S, DPD_DR2 /* rc^2 - rt^2 */, DPD_W = 4 /* 6 floats */, DPD_DR = 10 /* rt - rc */ };
// free-standing halo: [static | dynamic | d(static e-terms)/de], see dpie_prep
enum { DPX_DE = DP_NS + DP_ND, DPX_ND = DP_NS + DP_ND + 8 };
enum { DPA_S = 0, DPA_RC, DPA_RT, DPA_CX, DPA_CY, DPA_PHI, DPA_E, DP_NACC };

template <class R> GL_HD R dpie_r_min() { return (R)0.0001; }  // piemd.py:28,100

// piemd.py:52-60 / :191-199.  After the first `where` r_core <= r_cut, so the second never fires; r_cut keeps its
// value unless it is within r_min of r_core.  j_core / j_cut: d rc / d(raw r_core, raw r_cut); d rt/d r_cut = 1.
template <class R> GL_HD void dpie_sort(R r_core, R r_cut, R& rc, R& rt, R& j_core, R& j_cut) {
  const bool lt = r_core < r_cut;
  R m = lt ? r_core : r_cut;
  const bool above = m > dpie_r_min<R>();
  rc = above ? m : dpie_r_min<R>();
  rt = (r_cut > rc + dpie_r_min<R>()) ? r_cut : r_cut + dpie_r_min<R>();
  j_core = (above && lt) ? (R)1 : (R)0;
  j_cut = (above && !lt) ? (R)1 : (R)0;
}

// dynamic block of one halo from its raw (theta_E, r_core, r_cut); u_* = d raw / d population scale (catalogue
// members; 0 where the quantity is a catalogue constant).
template <class R> GL_HD void dpie_dyn(R theta_E, R r_core, R r_cut, R u_te, R u_rc, R u_rt, R* dd) {
  R rc, rt, j_core, j_cut;
  dpie_sort(r_core, r_cut, rc, rt, j_core, j_cut);
  R idr = (R)1 / (rt - rc);
  R S = theta_E * rt * idr;  // piemd.py:38,116
  dd[DPD_RC] = rc;
  dd[DPD_RT] = rt;
  dd[DPD_S] = S;
  dd[DPD_DR2] = (rc - rt) * (rc + rt);
  dd[DPD_DR] = rt - rc;
  // d S/d rc = S/(rt-rc),  d S/d rt = -S rc/(rt (rt-rc))
  R c1 = S * idr, c2 = -S * rc * idr / rt;
  R* w = dd + DPD_W;
  w[0] = u_te * rt * idr;                    // scale_theta_E <- A_S
  w[1] = u_rc * (j_core * c1);               // scale_r_core  <- A_S
  w[2] = u_rc * j_core;                      //               <- A_rc   (rt does not depend on raw r_core)
  w[3] = u_rt * (j_cut * c1 + c2);           // scale_r_cut   <- A_S
  w[4] = u_rt * j_cut;                       //               <- A_rc
  w[5] = u_rt;                               //               <- A_rt
  dd[11] = (R)0;
}

// ---- DPIE static block (piemd.py:183-188, :203-207); de (nullable): d/de of q, 1/q, 2 sqrt e, ix, iy, Z ----------
template <class R> GL_HD void piemd_static(R e1, R e2, R cx, R cy, R* ds, R* de) {
  Ellip<R> el = ellip_prep(e1, e2, (R)0.9999);
  R e = el.c, q = el.q;
  R sqe = p_sqrt(e);
  R ope = (R)1 + e, ome = (R)1 - e;
  ds[DPS_CX] = cx;
  ds[DPS_CY] = cy;
  ds[DPS_CPHI] = el.cphi;
  ds[DPS_SPHI] = el.sphi;
  ds[DPS_Q] = q;
  ds[DPS_IQ] = (R)1 / q;
  ds[DPS_S2] = (R)2 * sqe;
  ds[DPS_IX] = (R)1 / (ope * ope);
  ds[DPS_IY] = (R)1 / (ome * ome);
  ds[DPS_Z] = (R)-0.5 * ((R)1 - e * e) / sqe;  // zci_im; e == 0 -> -inf, the reference's NaN deflection
  ds[DPS_ZL] = ds[DPS_Z] * (R)(0.5 * kLn2);
  ds[11] = (R)0;
  if (de) {
    de[0] = (R)-2 / (ope * ope);
    de[1] = (R)2 / (ome * ome);
    de[2] = (R)1 / sqe;
    de[3] = (R)-2 / (ope * ope * ope);
    de[4] = (R)2 / (ome * ome * ome);
    de[5] = sqe + ((R)1 - e * e) / ((R)4 * e * sqe);
    de[6] = (R)0;
    de[7] = (R)0;
  }
}
// ---- DPIS / DPIEP static block (piep.py:50-55: e = |1-q^2|/(1+q^2) = 2c/(1+c^2)) -------------------------------
template <class R> GL_HD void piep_static(R e1, R e2, R cx, R cy, bool spherical, R* ds) {
  for (int i = 0; i < DP_NS; ++i) ds[i] = (R)0;
  ds[DPS_CX] = cx;
  ds[DPS_CY] = cy;
  if (spherical) {
    ds[DPS_CPHI] = (R)1;
    ds[DPS_M1] = (R)1;
    ds[DPS_P1] = (R)1;
    return;
  }
  Ellip<R> el = ellip_prep(e1, e2, (R)0.9999);
  R q = el.q;
  R e = fabs_((R)1 - q * q) / ((R)1 + q * q);
  ds[DPS_CPHI] = el.cphi;
  ds[DPS_SPHI] = el.sphi;
  ds[DPS_M1] = (R)1 - e;
  ds[DPS_P1] = (R)1 + e;
}

template <class R> GL_HD R dp_nan() { return (R)(0.0f * __builtin_inff()); }

// =============================================================================================
// DPIS / DPIEP per pixel  (piemd.py:33-49, piep.py:31-43)
//   alpha_r/r = S/r^2 (sqrt(r^2+rc^2) - rc - sqrt(r^2+rt^2) + rt) = S [1/(Wc+rc) - 1/(Wt+rt)]
// =============================================================================================
template <class R> GL_HD void piep_fwd(const R* ds, const R* dd, R x, R y, R& ax, R& ay) {
  R dx = x - ds[DPS_CX], dy = y - ds[DPS_CY];
  R c = ds[DPS_CPHI], s = ds[DPS_SPHI], m1 = ds[DPS_M1], p1 = ds[DPS_P1];
  R rc = dd[DPD_RC], rt = dd[DPD_RT], S = dd[DPD_S];
  R xr = dx * c + dy * s, yr = dy * c - dx * s;
  R r2 = xr * xr * m1 + yr * yr * p1;
  R Wc = sqrt_(r2 + rc * rc), Wt = sqrt_(r2 + rt * rt);
  R h = rcp(Wc + rc) - rcp(Wt + rt);
  R ar = (r2 == (R)0) ? dp_nan<R>() : S * h;  // 0/0 in the reference's form (piemd.py:39)
  R arx = ar * xr * m1, ary = ar * yr * p1;
  ax = arx * c - ary * s;
  ay = arx * s + ary * c;
}
// FULL: all seven accumulators; otherwise only (S, rc, rt) -- what a catalogue member needs
template <class R, bool FULL> GL_HD void piep_vjp(const R* ds, const R* dd, R x, R y, R gx, R gy, R* acc) {
  R dx = x - ds[DPS_CX], dy = y - ds[DPS_CY];
  R c = ds[DPS_CPHI], s = ds[DPS_SPHI], m1 = ds[DPS_M1], p1 = ds[DPS_P1];
  R rc = dd[DPD_RC], rt = dd[DPD_RT], S = dd[DPD_S];
  R xr = dx * c + dy * s, yr = dy * c - dx * s;
  R r2 = xr * xr * m1 + yr * yr * p1;
  R iWc = rcp(sqrt_(r2 + rc * rc)), iWt = rcp(sqrt_(r2 + rt * rt));
  R Wc = (r2 + rc * rc) * iWc, Wt = (r2 + rt * rt) * iWt;
  R hc = rcp(Wc + rc), ht = rcp(Wt + rt);
  R h = hc - ht;
  const bool ok = !(r2 == (R)0);
  R grx = gx * c + gy * s, gry = gy * c - gx * s;
  R g_ar = ok ? grx * xr * m1 + gry * yr * p1 : (R)0;
  R g_h = g_ar * S;
  acc[DPA_S] += g_ar * h;
  acc[DPA_RC] -= g_h * hc * iWc;  // d hc/d rc = -1/(Wc (Wc+rc))
  acc[DPA_RT] += g_h * ht * iWt;
  if (FULL) {
    R ar = S * h;
    R g_r2 = g_h * (R)0.5 * (ht * ht * iWt - hc * hc * iWc);
    R arx = ar * xr * m1, ary = ar * yr * p1;
    R ax = arx * c - ary * s, ay = arx * s + ary * c;
    R g_m1 = ok ? grx * ar * xr + g_r2 * xr * xr : (R)0;
    R g_p1 = ok ? gry * ar * yr + g_r2 * yr * yr : (R)0;
    R gxr = ok ? grx * ar * m1 + g_r2 * (R)2 * xr * m1 : (R)0;
    R gyr = ok ? gry * ar * p1 + g_r2 * (R)2 * yr * p1 : (R)0;
    R g_phi = ok ? gy * ax - gx * ay + gxr * yr - gyr * xr : (R)0;
    acc[DPA_CX] -= gxr * c - gyr * s;
    acc[DPA_CY] -= gxr * s + gyr * c;
    acc[DPA_PHI] += g_phi;
    acc[DPA_E] += g_p1 - g_m1;
  }
}

// =============================================================================================
// DPIE per pixel  (piemd.py:201-255)
//   znum_w = q x + i (2 sqrt(e) sqrt(w^2 + rem^2) - y/q),  zden_w = x + i (2 sqrt(e) w - y)
//   alpha' = zci * log[(znum_rc/zden_rc) / (znum_rt/zden_rt)],  zci = i Z
// =============================================================================================
template <class R> struct PiemdPix {
  R xr, yr, Wc, Wt, a, bc, bt, dc, dt, n1, n2, n3, n4, L, arg;
};
template <class R> GL_HD R atan2_(R y, R x) { return p_atan2(y, x); }
template <class R> GL_HD void piemd_pix(const R* ds, const R* dd, R x, R y, PiemdPix<R>& o) {
  R dx = x - ds[DPS_CX], dy = y - ds[DPS_CY];
  R c = ds[DPS_CPHI], s = ds[DPS_SPHI];
  R rc = dd[DPD_RC], rt = dd[DPD_RT], s2 = ds[DPS_S2];
  o.xr = dx * c + dy * s;
  o.yr = dy * c - dx * s;
  R rem2 = o.xr * o.xr * ds[DPS_IX] + o.yr * o.yr * ds[DPS_IY];
  o.Wc = sqrt_(rc * rc + rem2);
  o.Wt = sqrt_(rt * rt + rem2);
  R yq = o.yr * ds[DPS_IQ];
  o.a = ds[DPS_Q] * o.xr;
  o.bc = s2 * o.Wc - yq;
  o.bt = s2 * o.Wt - yq;
  o.dc = s2 * rc - o.yr;
  o.dt = s2 * rt - o.yr;
  o.n1 = o.a * o.a + o.bc * o.bc;      // |znum_rc|^2
  o.n2 = o.xr * o.xr + o.dc * o.dc;    // |zden_rc|^2
  o.n3 = o.a * o.a + o.bt * o.bt;      // |znum_rt|^2
  o.n4 = o.xr * o.xr + o.dt * o.dt;    // |zden_rt|^2
  // ratio (piemd.py:236-250) = N conj(D)/|D|^2 with N conj(D) = [znum_rc conj(znum_rt)] [zden_rt conj(zden_rc)]:
  //   znum_rc conj(znum_rt) = a^2 + bc bt + i a (bc - bt),   bc - bt = 2 sqrt(e) (rc^2 - rt^2)/(Wc + Wt)
  //   zden_rt conj(zden_rc) = x^2 + dc dt + i x (dt - dc),   dt - dc = 2 sqrt(e) (rt - rc)
  // -- the same complex number the reference takes atan2 of (up to the positive 1/norm), with the two imaginary
  // parts free of the fp32 cancellation the expanded products aa..dd suffer far from the halo.
  R p1r = o.a * o.a + o.bc * o.bt, p1i = o.a * (s2 * dd[DPD_DR2] * rcp(o.Wc + o.Wt));
  R p2r = o.xr * o.xr + o.dc * o.dt, p2i = o.xr * (s2 * dd[DPD_DR]);
  o.arg = atan2_(p1r * p2i + p1i * p2r, p1r * p2r - p1i * p2i);
  o.L = (R)(0.5 * kLn2) * log2_((o.n1 * o.n4) * rcp(o.n2 * o.n3));
}
template <class R> GL_HD void piemd_fwd(const R* ds, const R* dd, R x, R y, R& ax, R& ay) {
  PiemdPix<R> o;
  piemd_pix(ds, dd, x, y, o);
  R c = ds[DPS_CPHI], s = ds[DPS_SPHI];
  R SZ = dd[DPD_S] * ds[DPS_Z];
  R arx = -SZ * o.arg, ary = SZ * o.L;
  ax = arx * c - ary * s;
  ay = arx * s + ary * c;
}
template <class R, bool FULL>
GL_HD void piemd_vjp(const R* ds, const R* dd, const R* de, R x, R y, R gx, R gy, R* acc) {
  PiemdPix<R> o;
  piemd_pix(ds, dd, x, y, o);
  R c = ds[DPS_CPHI], s = ds[DPS_SPHI], Z = ds[DPS_Z], s2 = ds[DPS_S2];
  R rc = dd[DPD_RC], rt = dd[DPD_RT], S = dd[DPD_S];
  R grx = gx * c + gy * s, gry = gy * c - gx * s;  // cotangent of S * alpha'
  R arx0 = -Z * o.arg, ary0 = Z * o.L;             // alpha' before the amplitude
  acc[DPA_S] += grx * arx0 + gry * ary0;
  R gA = -S * Z * grx, gL = S * Z * gry;           // cotangents of arg and L
  // d(L + i arg) = dz/z for each of the four factors, sign +,-,-,+ :  cotangent of Im z = (gL Im z + gA Re z)/|z|^2
  R i1 = rcp(o.n1), i2 = rcp(o.n2), i3 = rcp(o.n3), i4 = rcp(o.n4);
  R k1 = (gL * o.bc + gA * o.a) * i1;
  R k2 = -(gL * o.dc + gA * o.xr) * i2;
  R k3 = -(gL * o.bt + gA * o.a) * i3;
  R k4 = (gL * o.dt + gA * o.xr) * i4;
  R iWc = rcp(o.Wc), iWt = rcp(o.Wt);
  acc[DPA_RC] += s2 * (k1 * rc * iWc + k2);
  acc[DPA_RT] += s2 * (k3 * rt * iWt + k4);
  if (FULL) {
    // cotangents of the real parts: (gL Re z - gA Im z)/|z|^2
    R r1 = (gL * o.a - gA * o.bc) * i1;
    R r2 = -(gL * o.xr - gA * o.dc) * i2;
    R r3 = -(gL * o.a - gA * o.bt) * i3;
    R r4 = (gL * o.xr - gA * o.dt) * i4;
    R g_a = r1 + r3;
    R g_rem2 = (R)0.5 * s2 * (k1 * iWc + k3 * iWt);
    R gxr = g_a * ds[DPS_Q] + r2 + r4 + g_rem2 * (R)2 * o.xr * ds[DPS_IX];
    R gyr = -(k1 + k3) * ds[DPS_IQ] - (k2 + k4) + g_rem2 * (R)2 * o.yr * ds[DPS_IY];
    R g_q = g_a * o.xr;
    R g_iq = -(k1 + k3) * o.yr;
    R g_s2 = k1 * o.Wc + k3 * o.Wt + k2 * rc + k4 * rt;
    R g_ix = g_rem2 * o.xr * o.xr, g_iy = g_rem2 * o.yr * o.yr;
    R g_Z = S * (-grx * o.arg + gry * o.L);
    R ax = S * (arx0 * c - ary0 * s), ay = S * (arx0 * s + ary0 * c);
    acc[DPA_CX] -= gxr * c - gyr * s;
    acc[DPA_CY] -= gxr * s + gyr * c;
    acc[DPA_PHI] += gy * ax - gx * ay + gxr * o.yr - gyr * o.xr;
    acc[DPA_E] += g_q * de[0] + g_iq * de[1] + g_s2 * de[2] + g_ix * de[3] + g_iy * de[4] + g_Z * de[5];
  }
}

// =============================================================================================
// free-standing halos: parameter rows  DPIS [theta_E, r_core, r_cut, center_x, center_y]   (piemd.py:27)
//                                      DPIE / DPIEP [theta_E, r_core|Ra, r_cut|Rs, center_x, center_y, e1, e2] (:99, piep.py:23)
// =============================================================================================
template <class R> GL_HD void dpie_prep(int kind, const R* p, R* d) {
  for (int i = 0; i < DPX_ND; ++i) d[i] = (R)0;
  if (kind == K_DPIE) piemd_static<R>(p[5], p[6], p[3], p[4], d, d + DPX_DE);
  else piep_static<R>(kind == K_DPIS ? (R)0 : p[5], kind == K_DPIS ? (R)0 : p[6], p[3], p[4], kind == K_DPIS, d);
  dpie_dyn<R>(p[0], p[1], p[2], (R)1, (R)1, (R)1, d + DP_NS);
}
template <class R> GL_HD void dpie_fwd(int kind, const R* d, R x, R y, R& ax, R& ay) {
  if (kind == K_DPIE) piemd_fwd<R>(d, d + DP_NS, x, y, ax, ay);
  else piep_fwd<R>(d, d + DP_NS, x, y, ax, ay);
}
template <class R> GL_HD void dpie_vjp(int kind, const R* d, R x, R y, R gx, R gy, R* acc) {
  if (kind == K_DPIE) piemd_vjp<R, true>(d, d + DP_NS, d + DPX_DE, x, y, gx, gy, acc);
  else piep_vjp<R, true>(d, d + DP_NS, x, y, gx, gy, acc);
}
template <class R> GL_HD void dpie_finalize(int kind, const R* p, const R* acc, R* g) {
  R dd[DP_ND];
  dpie_dyn<R>(p[0], p[1], p[2], (R)1, (R)1, (R)1, dd);
  const R* w = dd + DPD_W;
  g[0] = w[0] * acc[DPA_S];
  g[1] = w[1] * acc[DPA_S] + w[2] * acc[DPA_RC];
  g[2] = w[3] * acc[DPA_S] + w[4] * acc[DPA_RC] + w[5] * acc[DPA_RT];
  g[3] = acc[DPA_CX];
  g[4] = acc[DPA_CY];
  if (kind == K_DPIS) return;
  R g_c = acc[DPA_E];
  if (kind == K_DPIEP) {  // e = 2c/(1+c^2)
    Ellip<R> el = ellip_prep(p[5], p[6], (R)0.9999);
    R c2 = el.c * el.c;
    g_c *= (R)2 * ((R)1 - c2) / (((R)1 + c2) * ((R)1 + c2));
  }
  ellip_chain_c(p[5], p[6], (R)0.9999, g_c, acc[DPA_PHI], g[5], g[6]);
}

// reference's analytic DPIS Hessian (piemd.py:62-83) differs from the derivative of its own deflection by a factor
// (rc+rt)/rt on the convergence; the image-position likelihood uses that override (tf/simulator.py:83-84), so the
// difference  kappa_ref - kappa = (S/2)(rc/rt)(1/Wc - 1/Wt), r clamped at r_min,  is added to f_xx and f_yy.
template <class R> GL_HD R dpis_kappa_excess(const R* ds, const R* dd, R x, R y) {
  R dx = x - ds[DPS_CX], dy = y - ds[DPS_CY];
  R r = fmax_(p_sqrt(dx * dx + dy * dy), dpie_r_min<R>());
  R rc = dd[DPD_RC], rt = dd[DPD_RT];
  return dd[DPD_S] * (R)0.5 * (rc / rt) * ((R)1 / p_sqrt(rc * rc + r * r) - (R)1 / p_sqrt(rt * rt + r * r));
}

// =============================================================================================
// catalogue member g of a ScalingRelation (scaling_relation.py:44-59)
//   row  = [theta_E, r_core, r_cut, center_x, center_y, e1, e2] : for a scaling parameter the entry is
//          (L_g/L*)^power (the sample's scale multiplies it), otherwise the catalogue constant
//   col  = column of the scale inside the component's parameter row, or -1
// =============================================================================================
struct ScaledDesc {
  int base_kind, n_gal, col[3];
};
template <class R> GL_HD void scaled_static(int base_kind, const float* row, R* ds) {
  if (base_kind == K_DPIE) piemd_static<R>((R)row[5], (R)row[6], (R)row[3], (R)row[4], ds, nullptr);
  else piep_static<R>((R)row[5], (R)row[6], (R)row[3], (R)row[4], base_kind == K_DPIS, ds);
}
template <class R> GL_HD void scaled_dyn(const ScaledDesc& sd, const float* row, const R* scales, R* dd) {
  R v[3], u[3];
  for (int k = 0; k < 3; ++k) {
    const bool sc = sd.col[k] >= 0;
    u[k] = sc ? (R)row[k] : (R)0;
    v[k] = sc ? (R)row[k] * scales[sd.col[k]] : (R)row[k];
  }
  dpie_dyn<R>(v[0], v[1], v[2], u[0], u[1], u[2], dd);
}
// per-pixel cotangents (A_S, A_rc, A_rt) of one member -> the three scale gradients
template <class R> GL_HD void scaled_fold(const R* dd, const R* a, R* gs) {
  const R* w = dd + DPD_W;
  gs[0] += w[0] * a[DPA_S];
  gs[1] += w[1] * a[DPA_S] + w[2] * a[DPA_RC];
  gs[2] += w[3] * a[DPA_S] + w[4] * a[DPA_RC] + w[5] * a[DPA_RT];
}

}  // namespace glp
