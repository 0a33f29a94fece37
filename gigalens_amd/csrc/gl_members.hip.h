// gl_members.hip.h -- the inner loop of the cluster workload: one catalogue member (ScalingRelation galaxy,
// scaling_relation.py:61-70) evaluated on a pixel pair, with the tangents of the deflection along the three
// population scales carried forward in the same pass.
//
// Why forward mode here: a member's own parameters are catalogue constants except (theta_E, r_core, r_cut), which are
// (L/L*)^power x scale.  Reverse mode would need the pixel's cotangent (known only after the light profiles) and so a
// second evaluation of every member; instead the ray-shooting pass accumulates  d alpha / d scale_k  (3 x 2 values
// per pixel, independent of the number of galaxies) and the gradient is a dot product with the cotangent afterwards.
// Written over V in {float, v2f}: v2f is a pixel pair -> packed fp32 (see gl_vec.hip.h).  Member constants come
// from wave-uniform addresses (scalar loads, SGPR operands).
#pragma once
#include "gl_dpie.h"
#include "gl_vec.hip.h"

namespace glk {

// per (sample, galaxy) block written by gl_galprep_kernel
enum { GM_RC = 0, GM_RT, GM_RC2, GM_RT2, GM_S, GM_S2RC, GM_S2RT, GM_S2DR, GM_S2DR2, GM_pad,
       GM_WA = 10, GM_WB, GM_WC, GM_WD, GM_WE, GM_WF, GM_ND = 16 };

__device__ inline void member_dyn(const ScaledDesc& sd, const float* row, const float* gs, const float* scales,
                                  float* gm) {
  using R = float;
  R dd[DP_ND];
  scaled_dyn<R>(sd, row, scales, dd);
  const R* w = dd + DPD_W;
  const R rc = dd[DPD_RC], rt = dd[DPD_RT], S = dd[DPD_S];
  gm[GM_RC] = rc;
  gm[GM_RT] = rt;
  gm[GM_RC2] = rc * rc;
  gm[GM_RT2] = rt * rt;
  gm[GM_S] = S;
  gm[GM_pad] = 0.f;
  if (sd.base_kind == K_DPIE) {
    const R s2 = gs[DPS_S2];
    const R K = S * gs[DPS_Z] * s2;  // d alpha'/d rc = K (-eA, eL),  d alpha'/d rt = -K (-eA, eL)
    gm[GM_S2RC] = s2 * rc;
    gm[GM_S2RT] = s2 * rt;
    gm[GM_S2DR] = s2 * dd[DPD_DR];
    gm[GM_S2DR2] = s2 * dd[DPD_DR2];
    gm[GM_WA] = w[0];
    gm[GM_WB] = w[1];
    gm[GM_WC] = w[2] * K;
    gm[GM_WD] = w[3];
    gm[GM_WE] = w[4] * K;
    gm[GM_WF] = -w[5] * K;
  } else {
    gm[GM_S2RC] = gm[GM_S2RT] = gm[GM_S2DR] = gm[GM_S2DR2] = 0.f;
    gm[GM_WA] = w[0];
    gm[GM_WB] = w[1];
    gm[GM_WC] = -w[2] * S;  // d h/d rc = -hc/Wc
    gm[GM_WD] = w[3];
    gm[GM_WE] = -w[4] * S;
    gm[GM_WF] = w[5] * S;   // d h/d rt = +ht/Wt
  }
}

// ---- atan2 on lanes / pairs: one v_rcp, Cephes' atanf polynomial on |t| <= tan(pi/8) ------------------------------
template <class V> __device__ __forceinline__ V vabs(V a) { return __builtin_elementwise_abs(a); }
template <class V> __device__ __forceinline__ V vatan2(V y, V x) {
  const V ax = vabs(x), ay = vabs(y);
  const V mx = vmax(ax, ay), mn = vmin(ax, ay);
  const auto red = mn > mx * 0.41421356237f;                 // t > tan(pi/8): atan t = pi/4 + atan((t-1)/(t+1))
  const V num = red ? mn - mx : mn, den = red ? mn + mx : mx;
  V t = num * rcp(den);
  t = (mx == V(0.f)) ? V(0.f) : t;                           // atan2(0, 0) = 0
  const V z = t * t;
  V p = V(8.05374449538e-2f);
  p = __builtin_elementwise_fma(p, z, V(-1.38776856032e-1f));
  p = __builtin_elementwise_fma(p, z, V(1.99777106478e-1f));
  p = __builtin_elementwise_fma(p, z, V(-3.33329491539e-1f));
  V r = __builtin_elementwise_fma(p * z, t, t);
  r = red ? r + 0.78539816339744831f : r;
  r = (ay > ax) ? 1.57079632679489662f - r : r;
  r = (x < V(0.f)) ? 3.14159265358979324f - r : r;
  return __builtin_elementwise_copysign(r, y);
}

// accumulated tangents of the deflection along (scale_theta_E, scale_r_core, scale_r_cut)
template <class V> struct ScaleTan { V tx, ty, cx, cy, ux, uy; };

// ---- dPIE member (piemd.py:201-255; same maths as piemd_fwd / piemd_vjp in gl_dpie.h) -----------------------------
// (P: the pointer type of the two constant blocks -- plain, or address space 4 where the caller's kernel holds an opaque asm, next
// to which only constant-address-space loads stay scalar: gl_clusterw.hip.h)
template <class V, bool GRAD, class P = const float*>
__device__ __forceinline__ void piemd_member_v(P gs, P gm, V x, V y,
                                               V& bx, V& by, ScaleTan<V>& tn) {
  const float c = gs[DPS_CPHI], s = gs[DPS_SPHI], s2 = gs[DPS_S2];
  const V dx = x - gs[DPS_CX], dy = y - gs[DPS_CY];
  const V xr = dx * c + dy * s, yr = dy * c - dx * s;
  const V x2 = xr * xr;
  const V rem2 = x2 * gs[DPS_IX] + (yr * yr) * gs[DPS_IY];
  const V Wc2 = rem2 + gm[GM_RC2], Wt2 = rem2 + gm[GM_RT2];
  const V iWc = rsq_(Wc2), iWt = rsq_(Wt2);
  const V Wc = Wc2 * iWc, Wt = Wt2 * iWt;
  const V yq = yr * gs[DPS_IQ];
  const V a = xr * gs[DPS_Q];
  const V a2 = a * a;
  const V bc = Wc * s2 - yq, bt = Wt * s2 - yq;
  const V dc = gm[GM_S2RC] - yr, dt = gm[GM_S2RT] - yr;
  const V n1 = __builtin_elementwise_fma(bc, bc, a2), n2 = __builtin_elementwise_fma(dc, dc, x2);
  const V n3 = __builtin_elementwise_fma(bt, bt, a2), n4 = __builtin_elementwise_fma(dt, dt, x2);
  // atan2 is scale invariant: both parts carry the positive factor (Wc + Wt) instead of dividing p1i by it
  const V p1r = __builtin_elementwise_fma(bc, bt, a2) * (Wc + Wt), p1i = a * gm[GM_S2DR2];
  const V p2r = __builtin_elementwise_fma(dc, dt, x2), p2i = xr * gm[GM_S2DR];
  const V arg = vatan2<V>(p1r * p2i + p1i * p2r, p1r * p2r - p1i * p2i);
  V i2, i3, L2;
  if (GRAD) {
    i2 = rcp(n2);
    i3 = rcp(n3);
    L2 = log2_((n1 * n4) * (i2 * i3));
  } else {
    L2 = log2_((n1 * n4) * rcp(n2 * n3));
  }
  const V upx = arg * (-gs[DPS_Z]), upy = L2 * gs[DPS_ZL];  // alpha'/S
  const V ux = upx * c - upy * s, uy = upx * s + upy * c;
  const float S = gm[GM_S];
  bx -= ux * S;
  by -= uy * S;
  if (GRAD) {
    const V i1 = rcp(n1), i4 = rcp(n4);
    const V kc = iWc * gm[GM_RC], kt = iWt * gm[GM_RT];
    const V h1 = kc * i1, h3 = kt * i3;
    const V eAc = a * h1 - xr * i2, eLc = bc * h1 - dc * i2;   // d(arg, L)/d rc / s2
    const V eAt = a * h3 - xr * i4, eLt = bt * h3 - dt * i4;   // -d(arg, L)/d rt / s2
    const V vcx = -eAc * c - eLc * s, vcy = eLc * c - eAc * s;  // rotate (-eA, eL) back by -phi
    const V vtx = -eAt * c - eLt * s, vty = eLt * c - eAt * s;
    tn.tx += ux * gm[GM_WA];
    tn.ty += uy * gm[GM_WA];
    tn.cx += ux * gm[GM_WB] + vcx * gm[GM_WC];
    tn.cy += uy * gm[GM_WB] + vcy * gm[GM_WC];
    tn.ux += ux * gm[GM_WD] + vcx * gm[GM_WE] + vtx * gm[GM_WF];
    tn.uy += uy * gm[GM_WD] + vcy * gm[GM_WE] + vty * gm[GM_WF];
  }
}

// ---- dPIS / dPIEP member (piemd.py:33-49, piep.py:31-43): alpha' = S h (x' m1, y' p1), h = 1/(Wc+rc) - 1/(Wt+rt) ----
template <class V, bool GRAD, class P = const float*>
__device__ __forceinline__ void piep_member_v(P gs, P gm, V x, V y,
                                              V& bx, V& by, ScaleTan<V>& tn) {
  const float c = gs[DPS_CPHI], s = gs[DPS_SPHI];
  const V dx = x - gs[DPS_CX], dy = y - gs[DPS_CY];
  const V xr = dx * c + dy * s, yr = dy * c - dx * s;
  const V px = xr * gs[DPS_M1], py = yr * gs[DPS_P1];
  const V r2 = xr * px + yr * py;
  const V Wc2 = r2 + gm[GM_RC2], Wt2 = r2 + gm[GM_RT2];
  const V iWc = rsq_(Wc2), iWt = rsq_(Wt2);
  const V hc = rcp(Wc2 * iWc + gm[GM_RC]), ht = rcp(Wt2 * iWt + gm[GM_RT]);
  const auto zero = r2 == V(0.f);
  const V h = zero ? V(__builtin_nanf("")) : hc - ht;  // 0/0 in the reference's form (piemd.py:39)
  const V Px = px * c - py * s, Py = px * s + py * c;
  const V sh = h * gm[GM_S];
  bx -= Px * sh;
  by -= Py * sh;
  if (GRAD) {
    const V hh = zero ? V(0.f) : hc - ht;
    const V gc = zero ? V(0.f) : hc * iWc, gt = zero ? V(0.f) : ht * iWt;
    const V ka = hh * gm[GM_WA];
    const V kb = hh * gm[GM_WB] + gc * gm[GM_WC];
    const V kc = hh * gm[GM_WD] + gc * gm[GM_WE] + gt * gm[GM_WF];
    tn.tx += Px * ka; tn.ty += Py * ka;
    tn.cx += Px * kb; tn.cy += Py * kb;
    tn.ux += Px * kc; tn.uy += Py * kc;
  }
}

// the whole catalogue on one pixel (pair)
template <class V, bool GRAD>
__device__ __forceinline__ void members_v(int base_kind, int n_gal, const float* __restrict__ gs,
                                          const float* __restrict__ gm, V x, V y, V& bx, V& by, ScaleTan<V>& tn) {
  if (base_kind == K_DPIE) {
    for (int g = 0; g < n_gal; ++g) piemd_member_v<V, GRAD>(gs + g * DP_NS, gm + g * GM_ND, x, y, bx, by, tn);
  } else {
    for (int g = 0; g < n_gal; ++g) piep_member_v<V, GRAD>(gs + g * DP_NS, gm + g * GM_ND, x, y, bx, by, tn);
  }
}

}  // namespace glk
