// gl_positions.hip.h -- image-position likelihood, ForwardProbModel.stats_positions (tf/model.py:103-124),
// with LensSimulator.beta (tf/simulator.py:72-78) and .magnification (:80-91) on the observed image positions.
//
// For every family of multiple images the reference ray-shoots the positions to the source plane, compares them
// with their barycentre and weights by the magnification:  err = sigma_pos / mu,  mu = 1 / det(1 - Hessian),
//   chi2 = sum ((beta - mean beta) / err)^2 ,  norm = sum log(2 pi err^2) ,  loglike = -1/2 (chi2 + norm).
// Hessians come from differentiating `deriv` (tf/profile.py:9-43); the gradient w.r.t. the lens parameters
// therefore needs mixed second derivatives.  Tensors are tiny (images x batch), so this is four small kernels:
//   P1  (sample, image):        beta and Hessian           -- profile templates on Dual<float,2>
//   P2  (sample, family):       statistics + adjoints d loglike / d(beta, det A)
//   P3  (sample, image, lens parameter):  gradient         -- profile templates on Dual<Dual<float,1>,2>
//   P4  (sample, parameter):    sum over images
#pragma once
#include <hip/hip_runtime.h>

#include "gl_dual.h"
#include "gl_kernels.hip.h"

namespace glk {

struct PosArgs {
  const CompDesc* comps;
  int n_lens, P, B, J, F;
  const float* params;  // [B,P]
  const float* px;      // [J] image positions of all families, concatenated
  const float* py;
  const float* ex;      // [J] position errors
  const float* ey;
  const int* fam_off;   // [F+1]
  float* w_pos;         // [B][J][6]  beta_x, beta_y, f_xx, f_xy, f_yx, f_yy
  float* w_adj;         // [B][J][3]  d ll/d beta_x, d ll/d beta_y, d ll/d det
  float* w_g;           // [B][J][P]
  float* w_fam;         // [B][F][2]  loglike, chi2 per family
  float* ll;            // [B]
  float* chi2;          // [B]
  float* grad;          // [B][P] or null
  // galaxy catalogues of K_SCALED lenses
  const CatDev* cats;
  const float* gal_table;   // [G][7]
  const float* gal_static;  // [G][DP_NS]
  // series-expansion lenses (gl_lens_maps on the model's own grid only)
  const SeriesDev* series;
};

// parameters of one lens as the point kernels hold them: 7 for the built-in kinds; a user-written body (the run-time compiled
// build of this header, gl_user.hip compile_user_points: GL_HAVE_USER_POINT) takes up to 16
#ifdef GL_HAVE_USER_POINT
constexpr int POS_MAXP = 16;
#else
constexpr int POS_MAXP = 7;
#endif

// deflection of one lens at (x, y) with raw parameters p, generic in the real type (catalogues are summed)
template <class R> __device__ void lens_point(const PosArgs& a, const CompDesc& cd, const R* p, R x, R y, R& ax, R& ay) {
  using namespace glp;
  switch (cd.kind) {
#ifdef GL_HAVE_USER_POINT
    case K_USER_MASS: glu::mass_point<R>((int)cd.flags, p, x, y, ax, ay); break;  // the body on this number type (Hessian and mixed derivatives from the nested duals)
#endif
    case K_EPL: epl_point<R>(p, cd.iparam, x, y, ax, ay); break;
    case K_SIE: { R d[SIE_ND + 1]; sie_prep<R>(p, d); sie_fwd<R>(d, x, y, ax, ay); } break;
    case K_NFW: { R d[NFW_ND]; nfw_prep<R>(p, d); nfw_fwd<R>(d, x, y, ax, ay); } break;
    case K_SHEAR: { R d[4]; shear_prep<R>(p, d); shear_fwd<R>(d, x, y, ax, ay); } break;
    case K_DPIS:
    case K_DPIE:
    case K_DPIEP: { R d[DPX_ND]; dpie_prep<R>(cd.kind, p, d); dpie_fwd<R>(cd.kind, d, x, y, ax, ay); } break;
    case K_NFW_ELLIPSE: { R d[NFE_ND]; nfw_ell_prep<R>(p, d); nfw_ell_fwd<R>(d, x, y, ax, ay); } break;
    case K_TNFW: { R d[TNF_ND]; tnfw_prep<R>(p, d); tnfw_fwd<R>(d, x, y, ax, ay); } break;
    case K_SCALED: {
      const CatDev cat = a.cats[cd.iparam];
      const ScaledDesc sd{cat.base_kind, cat.n_gal, {cat.col[0], cat.col[1], cat.col[2]}};
      ax = R(0.f);
      ay = R(0.f);
      for (int g = 0; g < cat.n_gal; ++g) {
        const float* gs = a.gal_static + (size_t)(cat.g_off + g) * DP_NS;
        R ds[DP_NS], dd[DP_ND], fx, fy;
        for (int i = 0; i < DP_NS; ++i) ds[i] = R(gs[i]);
        scaled_dyn<R>(sd, a.gal_table + (size_t)(cat.g_off + g) * 7, p, dd);
        if (cat.base_kind == K_DPIE) piemd_fwd<R>(ds, dd, x, y, fx, fy);
        else piep_fwd<R>(ds, dd, x, y, fx, fy);
        ax += fx;
        ay += fy;
      }
    } break;
    default: { R d[4]; sis_prep<R>(p, d); sis_fwd<R>(d, x, y, ax, ay); } break;
  }
}
// what the reference's analytic dPIS Hessian adds to f_xx and f_yy on top of the derivative of the deflection
// (piemd.py:62-83, see dpis_kappa_excess); zero for every other profile
template <class R1> __device__ R1 lens_kappa_excess(const PosArgs& a, const CompDesc& cd, const R1* p, float x, float y) {
  using namespace glp;
  if (cd.kind == K_DPIS) {
    R1 d[DPX_ND];
    dpie_prep<R1>(cd.kind, p, d);
    return dpis_kappa_excess<R1>(d, d + DP_NS, R1(x), R1(y));
  }
  if (cd.kind == K_SCALED) {
    const CatDev cat = a.cats[cd.iparam];
    if (cat.base_kind != K_DPIS) return R1(0.f);
    const ScaledDesc sd{cat.base_kind, cat.n_gal, {cat.col[0], cat.col[1], cat.col[2]}};
    R1 ex(0.f);
    for (int g = 0; g < cat.n_gal; ++g) {
      const float* gs = a.gal_static + (size_t)(cat.g_off + g) * DP_NS;
      R1 ds[DP_NS], dd[DP_ND];
      for (int i = 0; i < DP_NS; ++i) ds[i] = R1(gs[i]);
      scaled_dyn<R1>(sd, a.gal_table + (size_t)(cat.g_off + g) * 7, p, dd);
      ex += dpis_kappa_excess<R1>(ds, dd, R1(x), R1(y));
    }
    return ex;
  }
  return R1(0.f);
}

__global__ void __launch_bounds__(64) gl_pos_p1_kernel(PosArgs a) {
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.B * a.J) return;
  int b = i / a.J, j = i - b * a.J;
  using R = gld::Dual<float, 2>;
  R x(a.px[j]), y(a.py[j]);
  x.d[0] = 1.f;
  y.d[1] = 1.f;
  float bx = a.px[j], by = a.py[j], fxx = 0.f, fxy = 0.f, fyx = 0.f, fyy = 0.f;
  for (int l = 0; l < a.n_lens; ++l) {
    CompDesc cd = a.comps[l];
    R p[POS_MAXP];
    float pf[POS_MAXP];
    for (int k = 0; k < cd.n_par; ++k) { pf[k] = a.params[(size_t)b * a.P + cd.p_off + k]; p[k] = R(pf[k]); }
    R ax, ay;
    lens_point<R>(a, cd, p, x, y, ax, ay);
    const float ex = lens_kappa_excess<float>(a, cd, pf, a.px[j], a.py[j]);
    bx -= ax.v; by -= ay.v;
    fxx += ax.d[0] + ex; fxy += ax.d[1]; fyx += ay.d[0]; fyy += ay.d[1] + ex;
  }
  float* o = a.w_pos + (size_t)i * 6;
  o[0] = bx; o[1] = by; o[2] = fxx; o[3] = fxy; o[4] = fyx; o[5] = fyy;
}

// LensSimulator.beta / .magnification / .convergence / .shear on arbitrary points (tf/simulator.py:72-107):
// out[6][n_pts][B] = beta_x, beta_y, f_xx, f_xy, f_yx, f_yy summed over the lenses (Hessians as `lens.hessian`
// resolves them in the reference: derivative of the deflection, plus the dPIS override's convergence excess)
__global__ void __launch_bounds__(64) gl_lens_maps_kernel(PosArgs a, const float* __restrict__ x,
                                                         const float* __restrict__ y, long long n_pts, int xy_batched,
                                                         float* __restrict__ out) {
  long long i = (long long)blockIdx.x * 64 + threadIdx.x;
  if (i >= n_pts * a.B) return;
  const long long pt = i / a.B;
  const int b = (int)(i - pt * a.B);
  const float px = xy_batched ? x[i] : x[pt], py = xy_batched ? y[i] : y[pt];
  using R = gld::Dual<float, 2>;
  R xd(px), yd(py);
  xd.d[0] = 1.f;
  yd.d[1] = 1.f;
  float bx = px, by = py, fxx = 0.f, fxy = 0.f, fyx = 0.f, fyy = 0.f;
  for (int l = 0; l < a.n_lens; ++l) {
    CompDesc cd = a.comps[l];
    R p[POS_MAXP];
    float pf[POS_MAXP];
    for (int k = 0; k < cd.n_par; ++k) { pf[k] = a.params[(size_t)b * a.P + cd.p_off + k]; p[k] = R(pf[k]); }
    if (cd.kind == glp::K_SERIES) {  // host guarantees: points = the model grid, both fields attached
      const SeriesDev sv = a.series[cd.flags];
      const float dl = pf[1] - sv.r0;
      float acc[5];
      for (int f = 0; f < 5; ++f) {
        const float* c = (f < 2 ? sv.coef + (size_t)f * (sv.order + 1) * n_pts
                                : sv.hcoef + (size_t)(f - 2) * (sv.order + 1) * n_pts) + pt;
        float v = c[(size_t)sv.order * n_pts];
        for (int n = sv.order - 1; n >= 0; --n) v = v * dl + c[(size_t)n * n_pts];
        acc[f] = pf[0] * v;
      }
      bx -= acc[0]; by -= acc[1];
      fxx += acc[2]; fxy += acc[3]; fyx += acc[3]; fyy += acc[4];
      continue;
    }
    R ax, ay;
    lens_point<R>(a, cd, p, xd, yd, ax, ay);
    const float ex = lens_kappa_excess<float>(a, cd, pf, px, py);
    bx -= ax.v; by -= ay.v;
    fxx += ax.d[0] + ex; fxy += ax.d[1]; fyx += ay.d[0]; fyy += ay.d[1] + ex;
  }
  const long long st = n_pts * a.B;
  out[i] = bx; out[st + i] = by; out[2 * st + i] = fxx; out[3 * st + i] = fxy; out[4 * st + i] = fyx; out[5 * st + i] = fyy;
}

// MassProfile.hessian at plugin level (tf/profile.py:9-27 and the analytic overrides): out[4][n_pts][B]
__global__ void __launch_bounds__(64) gl_profile_hessian_kernel(CompDesc cd, const float* __restrict__ x,
                                                               const float* __restrict__ y, long long n_pts, int B,
                                                               int xy_batched, const float* __restrict__ params,
                                                               float* __restrict__ out) {
  long long i = (long long)blockIdx.x * 64 + threadIdx.x;
  if (i >= n_pts * B) return;
  const long long pt = i / B;
  const int b = (int)(i - pt * B);
  const float px = xy_batched ? x[i] : x[pt], py = xy_batched ? y[i] : y[pt];
  using R = gld::Dual<float, 2>;
  R xd(px), yd(py);
  xd.d[0] = 1.f;
  yd.d[1] = 1.f;
  R p[7];
  float pf[7];
  for (int k = 0; k < cd.n_par; ++k) { pf[k] = params[(size_t)b * cd.n_par + k]; p[k] = R(pf[k]); }
  PosArgs none{};
  R ax, ay;
  lens_point<R>(none, cd, p, xd, yd, ax, ay);
  const float ex = lens_kappa_excess<float>(none, cd, pf, px, py);
  const long long st = n_pts * B;
  out[i] = ax.d[0] + ex; out[st + i] = ax.d[1]; out[2 * st + i] = ay.d[0]; out[3 * st + i] = ay.d[1] + ex;
}

// ScalingRelation.hessian at plugin level (scaling_relation.py:72-83): sum of the member Hessians, out[4][n_pts][B]
__global__ void __launch_bounds__(64) gl_scaled_hessian_kernel(ScaledDesc sd, const float* __restrict__ table,
                                                              const float* __restrict__ x, const float* __restrict__ y,
                                                              long long n_pts, int B, int xy_batched,
                                                              const float* __restrict__ scales, int n_scales,
                                                              float* __restrict__ out) {
  long long i = (long long)blockIdx.x * 64 + threadIdx.x;
  if (i >= n_pts * B) return;
  const long long pt = i / B;
  const int b = (int)(i - pt * B);
  const float px = xy_batched ? x[i] : x[pt], py = xy_batched ? y[i] : y[pt];
  using R = gld::Dual<float, 2>;
  R xd(px), yd(py);
  xd.d[0] = 1.f;
  yd.d[1] = 1.f;
  R sc[3];
  float scf[3] = {1.f, 1.f, 1.f};
  for (int k = 0; k < n_scales; ++k) scf[k] = scales[(size_t)b * n_scales + k];
  for (int k = 0; k < 3; ++k) sc[k] = R(scf[k]);
  float fxx = 0.f, fxy = 0.f, fyx = 0.f, fyy = 0.f;
  for (int g = 0; g < sd.n_gal; ++g) {
    const float* row = table + (size_t)7 * g;
    float dsf[DP_NS], ddf[DP_ND];
    scaled_static<float>(sd.base_kind, row, dsf);
    R ds[DP_NS], dd[DP_ND], ax, ay;
    for (int k = 0; k < DP_NS; ++k) ds[k] = R(dsf[k]);
    scaled_dyn<R>(sd, row, sc, dd);
    if (sd.base_kind == K_DPIE) piemd_fwd<R>(ds, dd, xd, yd, ax, ay);
    else piep_fwd<R>(ds, dd, xd, yd, ax, ay);
    float ex = 0.f;
    if (sd.base_kind == K_DPIS) {  // the analytic dPIS override (piemd.py:62-83)
      scaled_dyn<float>(sd, row, scf, ddf);
      ex = dpis_kappa_excess<float>(dsf, ddf, px, py);
    }
    fxx += ax.d[0] + ex; fxy += ax.d[1]; fyx += ay.d[0]; fyy += ay.d[1] + ex;
  }
  const long long st = n_pts * B;
  out[i] = fxx; out[st + i] = fxy; out[2 * st + i] = fyx; out[3 * st + i] = fyy;
}

__global__ void __launch_bounds__(64) gl_pos_p2_kernel(PosArgs a) {
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.B * a.F) return;
  int b = i / a.F, f = i - b * a.F;
  const int j0 = a.fam_off[f], j1 = a.fam_off[f + 1], n = j1 - j0;
  const float* pos = a.w_pos + ((size_t)b * a.J) * 6;
  float mbx = 0.f, mby = 0.f;
  for (int j = j0; j < j1; ++j) { mbx += pos[j * 6]; mby += pos[j * 6 + 1]; }
  mbx /= (float)n;  // tf.reduce_mean over the images of the family
  mby /= (float)n;
  float chi2 = 0.f, norm = 0.f, sx = 0.f, sy = 0.f;
  const float two_pi = 6.283185307179586f;
  for (int j = j0; j < j1; ++j) {
    const float* q = pos + j * 6;
    float det = (1.f - q[2]) * (1.f - q[5]) - q[3] * q[4];
    float mu = 1.f / det;  // tf/simulator.py:89-91
    float errx = a.ex[j] / mu, erry = a.ey[j] / mu;
    float rx = (q[0] - mbx) / errx, ry = (q[1] - mby) / erry;
    chi2 += rx * rx + ry * ry;
    norm += logf(two_pi * errx * errx) + logf(two_pi * erry * erry);
    sx += rx / errx;
    sy += ry / erry;
  }
  a.w_fam[((size_t)b * a.F + f) * 2] = -0.5f * (chi2 + norm);
  a.w_fam[((size_t)b * a.F + f) * 2 + 1] = chi2;
  if (!a.grad) return;
  // adjoints:  d ll/d beta_kc = -r_kc/err_kc + (1/n) sum_j r_jc/err_jc ;  d ll/d mu_j = (2 - sum_c r_jc^2)/mu_j ;
  //            mu = 1/det  =>  d ll/d det_j = -mu_j^2 d ll/d mu_j
  for (int j = j0; j < j1; ++j) {
    const float* q = pos + j * 6;
    float det = (1.f - q[2]) * (1.f - q[5]) - q[3] * q[4];
    float mu = 1.f / det;
    float errx = a.ex[j] / mu, erry = a.ey[j] / mu;
    float rx = (q[0] - mbx) / errx, ry = (q[1] - mby) / erry;
    float* o = a.w_adj + ((size_t)b * a.J + j) * 3;
    o[0] = -rx / errx + sx / (float)n;
    o[1] = -ry / erry + sy / (float)n;
    o[2] = -mu * (2.f - rx * rx - ry * ry);
  }
}

// one thread per (sample, image, lens-parameter column): the jets carry ONE parameter direction
// (Dual<Dual<float,1>,2> = 6 floats per value), which keeps every profile's nested-dual code small enough to stay
// in registers; the (n_params x) repeated evaluation is irrelevant at this size.
__global__ void __launch_bounds__(64) gl_pos_p3_kernel(PosArgs a, int lens_params) {
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.B * a.J * lens_params) return;
  const int col = i % lens_params, bj = i / lens_params;
  const int b = bj / a.J, j = bj - b * a.J;
  int l = 0;
  while (l + 1 < a.n_lens && col >= a.comps[l + 1].p_off) ++l;
  const CompDesc cd = a.comps[l];
  const int k = col - cd.p_off;
  using R1 = gld::Dual<float, 1>;
  using R = gld::Dual<R1, 2>;
  R x(R1(a.px[j])), y(R1(a.py[j]));
  x.d[0] = R1(1.f);
  y.d[1] = R1(1.f);
  R p[POS_MAXP];
  R1 p1[POS_MAXP];
  for (int m = 0; m < POS_MAXP; ++m) {
    R1 v(m < cd.n_par ? a.params[(size_t)b * a.P + cd.p_off + m] : 0.f);
    if (m == k) v.d[0] = 1.f;
    p1[m] = v;
    p[m] = R(v);
  }
  R ax, ay;
  lens_point<R>(a, cd, p, x, y, ax, ay);
  const R1 ex = lens_kappa_excess<R1>(a, cd, p1, a.px[j], a.py[j]);
  const float* q = a.w_pos + ((size_t)b * a.J + j) * 6;
  const float* adj = a.w_adj + ((size_t)b * a.J + j) * 3;
  // beta = x - sum alpha ;  det = (1-fxx)(1-fyy) - fxy fyx
  const float dbx = -ax.v.d[0], dby = -ay.v.d[0];
  const float ddet = -(1.f - q[5]) * (ax.d[0].d[0] + ex.d[0]) - (1.f - q[2]) * (ay.d[1].d[0] + ex.d[0]) -
                     q[4] * ax.d[1].d[0] - q[3] * ay.d[0].d[0];
  a.w_g[((size_t)b * a.J + j) * a.P + col] = adj[0] * dbx + adj[1] * dby + adj[2] * ddet;
}

__global__ void __launch_bounds__(64) gl_pos_p4_kernel(PosArgs a, int lens_params) {
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.B * (a.P + 1)) return;
  int b = i / (a.P + 1), p = i - b * (a.P + 1);
  if (p == a.P) {
    float ll = 0.f, c2 = 0.f;
    for (int f = 0; f < a.F; ++f) { ll += a.w_fam[((size_t)b * a.F + f) * 2]; c2 += a.w_fam[((size_t)b * a.F + f) * 2 + 1]; }
    a.ll[b] = ll;
    a.chi2[b] = c2;
    return;
  }
  if (!a.grad) return;
  float g = 0.f;
  if (p < lens_params)
    for (int j = 0; j < a.J; ++j) g += a.w_g[((size_t)b * a.J + j) * a.P + p];
  a.grad[(size_t)b * a.P + p] = g;
}

}  // namespace glk
