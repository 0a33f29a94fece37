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
//   P3  (sample, image, lens):  parameter gradient         -- profile templates on Dual<Dual<float,P>,2>
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
};

// deflection of one lens at (x, y) with raw parameters p, generic in the real type
template <class R> __device__ void lens_point(int kind, int iparam, const R* p, R x, R y, R& ax, R& ay) {
  using namespace glp;
  switch (kind) {
    case K_EPL: epl_point<R>(p, iparam, x, y, ax, ay); break;
    case K_SIE: { R d[SIE_ND + 1]; sie_prep<R>(p, d); sie_fwd<R>(d, x, y, ax, ay); } break;
    case K_NFW: { R d[NFW_ND]; nfw_prep<R>(p, d); nfw_fwd<R>(d, x, y, ax, ay); } break;
    case K_SHEAR: { R d[4]; shear_prep<R>(p, d); shear_fwd<R>(d, x, y, ax, ay); } break;
    default: { R d[4]; sis_prep<R>(p, d); sis_fwd<R>(d, x, y, ax, ay); } break;
  }
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
    R p[6];
    for (int k = 0; k < cd.n_par; ++k) p[k] = R(a.params[(size_t)b * a.P + cd.p_off + k]);
    R ax, ay;
    lens_point<R>(cd.kind, cd.iparam, p, x, y, ax, ay);
    bx -= ax.v; by -= ay.v;
    fxx += ax.d[0]; fxy += ax.d[1]; fyx += ay.d[0]; fyy += ay.d[1];
  }
  float* o = a.w_pos + (size_t)i * 6;
  o[0] = bx; o[1] = by; o[2] = fxx; o[3] = fxy; o[4] = fyx; o[5] = fyy;
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

template <int PL> __device__ void pos_grad_lens(const PosArgs& a, const CompDesc& cd, int b, int j) {
  using R1 = gld::Dual<float, PL>;
  using R = gld::Dual<R1, 2>;
  R x(R1(a.px[j])), y(R1(a.py[j]));
  x.d[0] = R1(1.f);
  y.d[1] = R1(1.f);
  R p[PL];
  for (int k = 0; k < PL; ++k) {
    R1 v(a.params[(size_t)b * a.P + cd.p_off + k]);
    v.d[k] = 1.f;
    p[k] = R(v);
  }
  R ax, ay;
  lens_point<R>(cd.kind, cd.iparam, p, x, y, ax, ay);
  const float* q = a.w_pos + ((size_t)b * a.J + j) * 6;
  const float* adj = a.w_adj + ((size_t)b * a.J + j) * 3;
  float* g = a.w_g + ((size_t)b * a.J + j) * a.P + cd.p_off;
  for (int k = 0; k < PL; ++k) {
    // beta = x - sum alpha ;  det = (1-fxx)(1-fyy) - fxy fyx
    float dbx = -ax.v.d[k], dby = -ay.v.d[k];
    float ddet = -(1.f - q[5]) * ax.d[0].d[k] - (1.f - q[2]) * ay.d[1].d[k] - q[4] * ax.d[1].d[k] - q[3] * ay.d[0].d[k];
    g[k] = adj[0] * dbx + adj[1] * dby + adj[2] * ddet;
  }
}

__global__ void __launch_bounds__(64) gl_pos_p3_kernel(PosArgs a) {
  int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= a.B * a.J * a.n_lens) return;
  int l = i % a.n_lens, bj = i / a.n_lens;
  int b = bj / a.J, j = bj - b * a.J;
  CompDesc cd = a.comps[l];
  switch (cd.kind) {
    case K_EPL: pos_grad_lens<6>(a, cd, b, j); break;
    case K_SIE: pos_grad_lens<5>(a, cd, b, j); break;
    case K_NFW: pos_grad_lens<4>(a, cd, b, j); break;
    case K_SHEAR: pos_grad_lens<2>(a, cd, b, j); break;
    default: pos_grad_lens<3>(a, cd, b, j); break;
  }
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
