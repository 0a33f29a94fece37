// gl_shapelets.hip.h -- the shapelet render / VJP of the specialised kernels in SEPARABLE form
// (tf/profiles/light/shapelets.py:53-85).
//
// The reference contracts the (n1, n2) amplitude triangle with the outer product of the two 1-D bases
// (einsum 'ij,i...j', shapelets.py:63-64,73-74): 2 multiply-adds per amplitude for the surface brightness and again for
// each derivative.  Here the triangle is contracted one axis at a time,
//     t_n1 = sum_n2 a(n1,n2) Y_n2 ,   S = sum_n1 X_n1 t_n1 ,   dS/du = sum_n1 X'_n1 t_n1 ,
//     s_n2 = sum_n1 a(n1,n2) X_n1 ,   dS/dv = sum_n2 Y'_n2 s_n2 ,   dS/da(n1,n2) = X_n1 Y_n2 ,
// which is 66 + 11 FMAs where the outer-product form spends 132, keeps the bases of the forward pass for the VJP
// (no second table gather / recurrence), takes the amplitudes through wave-uniform scalar loads (SGPR operands)
// instead of 66 LDS broadcasts per contraction, and fetches the two table rows of a coordinate as six float4.
// The amplitude block of the derived constants is zero-padded to the full n_max = 10 triangle (shapelets_prep), so the
// loops carry no n_max guards.
#pragma once
#include <hip/hip_runtime.h>

#include "gl_profiles.h"

namespace glk {
using namespace glp;

template <int CAP> struct ShpState {
  float X[CAP + 1], Y[CAP + 1], dX[CAP + 1], dY[CAP + 1], t[CAP + 1];
  float fac, u, v, dx, dy, S;
  float ds;  // table mode: dX / dY hold node differences, the derivative is ds times them
};

// linear interpolation on the 6000-node table (tfp.math.interp_regular_1d_grid, fill 0 outside, shapelets.py:58-60);
// rows fb and fb + 1 are contiguous: 2 x 12 floats = six aligned float4 (the table is built with stride 12)
template <int CAP>
__device__ __forceinline__ void shp_table_basis(const float* __restrict__ tab, float u, float* Xv, float* dXv) {
  static_assert(CAP + 1 <= 12, "table stride");
  const float scale = (float)(SH_NODES - 1) / 10.f;
  const float fi = (u + 5.f) * scale;
  const bool inside = (fi >= 0.f) && (fi <= (float)(SH_NODES - 1));
  const float fic = clamp_(fi, 0.f, (float)(SH_NODES - 1));
  float fb = floor_(fic);
  const float fa = fmin_(fb + 1.f, (float)(SH_NODES - 1));
  fb = fmax_(fa - 1.f, 0.f);
  const float tt = fic - fb;
  // outside the table: the two zero rows appended to it (value and slope 0 without a select per order)
  const float4* __restrict__ r = reinterpret_cast<const float4*>(tab + (size_t)(inside ? (int)fb : SH_NODES) * 12);
  float lo[12], hi[12];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const float4 a = r[q], b = r[3 + q];
    lo[4 * q] = a.x; lo[4 * q + 1] = a.y; lo[4 * q + 2] = a.z; lo[4 * q + 3] = a.w;
    hi[4 * q] = b.x; hi[4 * q + 1] = b.y; hi[4 * q + 2] = b.z; hi[4 * q + 3] = b.w;
  }
#pragma unroll
  for (int n = 0; n <= CAP; ++n) {  // three instructions per order: the node difference serves the value and the slope
    const float dn = hi[n] - lo[n];
    Xv[n] = fmaf(tt, dn, lo[n]);
    dXv[n] = dn;  // x (nodes per unit) once, on the contracted derivative (ShpState::ds)
  }
}

__device__ __forceinline__ constexpr int shp_idx(int n1, int n2) { return (n1 + n2) * (n1 + n2 + 1) / 2 + n2; }

// `gamp`: this sample's amplitude block in GLOBAL memory (wave-uniform address -> scalar loads)
template <int CAP>
__device__ __forceinline__ float shp_fwd_state(const float* d, const float* __restrict__ gamp, const float* __restrict__ tab,
                                               bool interp, float x, float y, ShpState<CAP>& st) {
  const int n_max = (int)d[SHP_NMAX];
  const float ib = d[SHP_IB];
  st.dx = x - d[SHP_CX];
  st.dy = y - d[SHP_CY];
  st.u = st.dx * ib;
  st.v = st.dy * ib;
  st.fac = 1.f;
  st.ds = interp ? (float)(SH_NODES - 1) / 10.f : 1.f;
  if (interp) {
    shp_table_basis<CAP>(tab, st.u, st.X, st.dX);
    shp_table_basis<CAP>(tab, st.v, st.Y, st.dY);
  } else {
    hermite_basis<float, CAP>(st.u, n_max, st.X, st.dX);
    hermite_basis<float, CAP>(st.v, n_max, st.Y, st.dY);
    st.fac = exp_(-(st.u * st.u + st.v * st.v) * 0.5f);  // shapelets.py:70
  }
  float S = 0.f;
#pragma unroll
  for (int n1 = 0; n1 <= CAP; ++n1) {
    float t = 0.f;
#pragma unroll
    for (int n2 = 0; n2 <= CAP - n1; ++n2) t = fmaf(gamp[shp_idx(n1, n2)], st.Y[n2], t);
    st.t[n1] = t;
    S = fmaf(st.X[n1], t, S);
  }
  st.S = S;
  return st.fac * S;
}

// acc layout as shapelets_vjp: [cx, cy, 1/beta, amp_0 ..]
template <int CAP>
__device__ __forceinline__ void shp_vjp_state(const float* d, const float* __restrict__ gamp, bool interp,
                                              const ShpState<CAP>& st, float gI, float* acc, float& gpx, float& gpy) {
  const float ib = d[SHP_IB];
  const float gS = gI * st.fac;
  float Su = 0.f, Sv = 0.f;
#pragma unroll
  for (int n1 = 0; n1 <= CAP; ++n1) Su = fmaf(st.dX[n1], st.t[n1], Su);
#pragma unroll
  for (int n2 = 0; n2 <= CAP; ++n2) {
    float s = 0.f;
#pragma unroll
    for (int n1 = 0; n1 <= CAP - n2; ++n1) s = fmaf(gamp[shp_idx(n1, n2)], st.X[n1], s);
    Sv = fmaf(st.dY[n2], s, Sv);
  }
#pragma unroll
  for (int n1 = 0; n1 <= CAP; ++n1) {
    const float gx = gS * st.X[n1];
#pragma unroll
    for (int n2 = 0; n2 <= CAP - n1; ++n2)
      acc[SHPA_AMP + shp_idx(n1, n2)] = fmaf(gx, st.Y[n2], acc[SHPA_AMP + shp_idx(n1, n2)]);
  }
  float gu = gS * (Su * st.ds), gv = gS * (Sv * st.ds);
  if (!interp) {  // d fac/du = -u fac
    const float gIf = gI * (st.fac * st.S);
    gu -= gIf * st.u;
    gv -= gIf * st.v;
  }
  const float gdx = gu * ib, gdy = gv * ib;
  acc[SHPA_CX] -= gdx;
  acc[SHPA_CY] -= gdy;
  acc[SHPA_IB] += gu * st.dx + gv * st.dy;
  gpx += gdx;
  gpy += gdy;
}

}  // namespace glk
