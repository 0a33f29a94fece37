// gl_series.h -- series-expansion accelerator of the scaled dPIE population (tf/series/series_profile.py,
// tf/profiles/mass/{dpie_series,scaling_series,dpie_subhalo_series}.py).
//
// The population's deflection per unit amplitude is expanded in the population cut-radius scale r around r0:
//     alpha(pixel; theta_E, r) = theta_E * sum_n C_n(pixel) (r - r0)^n ,
//     C_n(pixel) = sum_g (L_g/L*)^p_theta * [h^n] alpha_g(pixel; r_cut_g = u_g (r0 + h))
// which is scaling_series.py:19-35 with  pre_factor = amplitude_factor * series_factor^n  and  f^(n)/n!  folded into
// Taylor coefficients.  C_n comes from evaluating the ordinary member templates on Jet<F,N> (gl_jet.h) -- the
// reference's generated deriv_0..deriv_5 are exactly these derivatives of piemd.py's deflection
// (series_codegen/profiles/dpie.py:18-58), without the radius sort / clamps, which are inactive for r_core < r_cut.
// Precompute is a one-off per (grid, constants); at run time a pixel reads 2 (N+1) coefficients.
#pragma once
#include "gl_dpie.h"
#include "gl_dual.h"
#include "gl_jet.h"

namespace glp {

constexpr int SERIES_MAX_ORDER = 5;  // tf/series/profiles/dpie.py ships deriv_0 .. deriv_5

// scales: the component's scale row (theta_E entry ignored: amplitude 1, scaling_series.py:20; r_cut entry = r0)
template <int N, class F>
GL_HD void series_point(const ScaledDesc& sd, const float* table, const F* scales, F x, F y, F* cx, F* cy) {
  using R = glj::Jet<F, N>;
  for (int n = 0; n <= N; ++n) { cx[n] = F(0); cy[n] = F(0); }
  R sc[3];
  for (int k = 0; k < 3; ++k) sc[k] = R(F(1));
  for (int k = 0; k < 3; ++k)
    if (sd.col[k] >= 0) sc[sd.col[k]] = R(scales[sd.col[k]]);
  if (sd.col[0] >= 0) sc[sd.col[0]] = R(F(1));
  if (sd.col[2] >= 0) sc[sd.col[2]].c[1] = F(1);  // the expansion variable
  for (int g = 0; g < sd.n_gal; ++g) {
    const float* row = table + (size_t)7 * g;
    F dsf[DP_NS];
    scaled_static<F>(sd.base_kind, row, dsf);
    R ds[DP_NS], dd[DP_ND], ax, ay;
    for (int i = 0; i < DP_NS; ++i) ds[i] = R(dsf[i]);
    scaled_dyn<R>(sd, row, sc, dd);
    if (sd.base_kind == K_DPIE) piemd_fwd<R>(ds, dd, R(x), R(y), ax, ay);
    else piep_fwd<R>(ds, dd, R(x), R(y), ax, ay);
    for (int n = 0; n <= N; ++n) { cx[n] += ax.c[n]; cy[n] += ay.c[n]; }
  }
}

// The Hessian half (MassSeries.set_hessian, series_profile.py:64-65; dpie_series.py:35-49): Taylor coefficients in r
// of d alpha / d(x, y).  The reference expands Lenstool's closed-form dPIE Hessian (series_codegen/profiles/dpie.py:
// 60-105), which IS the space derivative of the deflection it expands in `deriv`; here the same member templates run
// on Dual<Jet<F,N>,2> -- two space tangents, each a series in r.  hxx, hxy, hyy: [N+1] each.
template <int N, class F>
GL_HD void series_point_hessian(const ScaledDesc& sd, const float* table, const F* scales, F x, F y, F* hxx, F* hxy,
                                F* hyy) {
  using J = glj::Jet<F, N>;
  using R = gld::Dual<J, 2>;
  for (int n = 0; n <= N; ++n) { hxx[n] = F(0); hxy[n] = F(0); hyy[n] = F(0); }
  J scj[3];
  for (int k = 0; k < 3; ++k) scj[k] = J(F(1));
  for (int k = 0; k < 3; ++k)
    if (sd.col[k] >= 0) scj[sd.col[k]] = J(scales[sd.col[k]]);
  if (sd.col[0] >= 0) scj[sd.col[0]] = J(F(1));
  if (sd.col[2] >= 0) scj[sd.col[2]].c[1] = F(1);
  R sc[3];
  for (int k = 0; k < 3; ++k) sc[k] = R(scj[k]);
  R xd{J(x)}, yd{J(y)};
  xd.d[0] = J(F(1));
  yd.d[1] = J(F(1));
  for (int g = 0; g < sd.n_gal; ++g) {
    const float* row = table + (size_t)7 * g;
    F dsf[DP_NS];
    scaled_static<F>(sd.base_kind, row, dsf);
    R ds[DP_NS], dd[DP_ND], ax, ay;
    for (int i = 0; i < DP_NS; ++i) ds[i] = R(J(dsf[i]));
    scaled_dyn<R>(sd, row, sc, dd);
    if (sd.base_kind == K_DPIE) piemd_fwd<R>(ds, dd, xd, yd, ax, ay);
    else piep_fwd<R>(ds, dd, xd, yd, ax, ay);
    for (int n = 0; n <= N; ++n) { hxx[n] += ax.d[0].c[n]; hxy[n] += ax.d[1].c[n]; hyy[n] += ay.d[1].c[n]; }
  }
}

}  // namespace glp
