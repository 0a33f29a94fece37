// TEST HARNESS ONLY -- never linked into the product library.
// Instantiates the per-profile templates of gigalens_amd/csrc/gl_profiles.h on the host (double and
// float) so that the CPU test-suite can check the hand-written prep / fwd / vjp / finalize chain
// against autograd of the oracle without a GPU.  The product evaluates these same templates only
// inside its HIP kernels.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../gigalens_amd/csrc/gl_host_tables.h"
#include "../../gigalens_amd/csrc/gl_profiles.h"
#include "../../gigalens_amd/csrc/gl_dpie.h"
#include "../../gigalens_amd/csrc/gl_series.h"
#include "../../gigalens_amd/csrc/gl_extra.h"
#include "../../gigalens_amd/csrc/gl_dual.h"
#include "../../gigalens_amd/csrc/gl_eigh.h"

using namespace glp;

namespace {

template <class R> struct Tab {
  std::vector<float> tab;
  int stride = 0;
};

template <class R>
void run_mass(int kind, int iparam, const R* p, int n, const R* x, const R* y, const R* gx, const R* gy, R* ax,
              R* ay, R* grad) {
  std::vector<R> d(kind_num_derived(kind, iparam) + 8, (R)0);
  R acc[16] = {0};
  switch (kind) {
    case K_EPL: epl_prep<R>(p, iparam, d.data()); break;
    case K_SIE: sie_prep<R>(p, d.data()); break;
    case K_NFW: nfw_prep<R>(p, d.data()); break;
    case K_SHEAR: shear_prep<R>(p, d.data()); break;
    case K_SIS: sis_prep<R>(p, d.data()); break;
    case K_DPIS: case K_DPIE: case K_DPIEP: dpie_prep<R>(kind, p, d.data()); break;
    case K_NFW_ELLIPSE: nfw_ell_prep<R>(p, d.data()); break;
    case K_TNFW: tnfw_prep<R>(p, d.data()); break;
  }
  for (int i = 0; i < n; ++i) {
    switch (kind) {
      case K_EPL: epl_fwd<R>(d.data(), x[i], y[i], ax[i], ay[i]); epl_vjp<R>(d.data(), x[i], y[i], gx[i], gy[i], acc); break;
      case K_SIE: sie_fwd<R>(d.data(), x[i], y[i], ax[i], ay[i]); sie_vjp<R>(d.data(), x[i], y[i], gx[i], gy[i], acc); break;
      case K_NFW: nfw_fwd<R>(d.data(), x[i], y[i], ax[i], ay[i]); nfw_vjp<R>(d.data(), x[i], y[i], gx[i], gy[i], acc); break;
      case K_SHEAR: shear_fwd<R>(d.data(), x[i], y[i], ax[i], ay[i]); shear_vjp<R>(d.data(), x[i], y[i], gx[i], gy[i], acc); break;
      case K_SIS: sis_fwd<R>(d.data(), x[i], y[i], ax[i], ay[i]); sis_vjp<R>(d.data(), x[i], y[i], gx[i], gy[i], acc); break;
      case K_DPIS: case K_DPIE: case K_DPIEP:
        dpie_fwd<R>(kind, d.data(), x[i], y[i], ax[i], ay[i]); dpie_vjp<R>(kind, d.data(), x[i], y[i], gx[i], gy[i], acc); break;
      case K_NFW_ELLIPSE: nfw_ell_fwd<R>(d.data(), x[i], y[i], ax[i], ay[i]); nfw_ell_vjp<R>(d.data(), x[i], y[i], gx[i], gy[i], acc); break;
      case K_TNFW: tnfw_fwd<R>(d.data(), x[i], y[i], ax[i], ay[i]); tnfw_vjp<R>(d.data(), x[i], y[i], gx[i], gy[i], acc); break;
    }
  }
  switch (kind) {
    case K_EPL: epl_finalize<R>(p, acc, grad); break;
    case K_SIE: sie_finalize<R>(p, acc, grad); break;
    case K_NFW: nfw_finalize<R>(p, acc, grad); break;
    case K_SHEAR: shear_finalize<R>(p, acc, grad); break;
    case K_SIS: sis_finalize<R>(p, acc, grad); break;
    case K_DPIS: case K_DPIE: case K_DPIEP: dpie_finalize<R>(kind, p, acc, grad); break;
    case K_NFW_ELLIPSE: nfw_ell_finalize<R>(p, acc, grad); break;
    case K_TNFW: tnfw_finalize<R>(p, acc, grad); break;
  }
}

// ScalingRelation over a catalogue: deflection and the gradient w.r.t. the population scales
template <class R>
void run_scaled(int base_kind, int n_gal, const float* table, const int* cols, const R* scales, int n, const R* x,
                const R* y, const R* gx, const R* gy, R* ax, R* ay, R* gscales) {
  ScaledDesc sd{base_kind, n_gal, {cols[0], cols[1], cols[2]}};
  R gs[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) { ax[i] = 0; ay[i] = 0; }
  for (int g = 0; g < n_gal; ++g) {
    R ds[DP_NS], dd[DP_ND];
    scaled_static<R>(base_kind, table + 7 * g, ds);
    scaled_dyn<R>(sd, table + 7 * g, scales, dd);
    R a[DP_NACC] = {0};
    for (int i = 0; i < n; ++i) {
      R fx, fy;
      if (base_kind == K_DPIE) {
        piemd_fwd<R>(ds, dd, x[i], y[i], fx, fy);
        piemd_vjp<R, false>(ds, dd, nullptr, x[i], y[i], gx[i], gy[i], a);
      } else {
        piep_fwd<R>(ds, dd, x[i], y[i], fx, fy);
        piep_vjp<R, false>(ds, dd, x[i], y[i], gx[i], gy[i], a);
      }
      ax[i] += fx;
      ay[i] += fy;
    }
    scaled_fold<R>(dd, a, gs);
  }
  for (int k = 0; k < 3; ++k) gscales[k] = gs[k];
}

template <class R>
void run_light(int kind, int iparam, unsigned flags, const R* p, int n, const R* x, const R* y, const R* gI, R* I,
               R* grad, R* gpx, R* gpy) {
  std::vector<R> d(kind_num_derived(kind, iparam) + 8, (R)0);
  std::vector<R> acc(kind_num_acc(kind, iparam) + 8, (R)0);
  static std::vector<float> tab;
  static int stride = 0;
  if (kind == K_SHAPELETS && tab.empty()) glh::build_shapelet_table(SH_CAP, tab, &stride);
  const bool interp = flags & 1u;
  switch (kind) {
    case K_SERSIC: sersic_prep<R>(p, false, d.data()); break;
    case K_SERSIC_ELLIPSE: sersic_prep<R>(p, true, d.data()); break;
    case K_SHAPELETS: shapelets_prep<R>(p, iparam, d.data()); break;
    case K_CORE_SERSIC: core_sersic_prep<R>(p, d.data()); break;
  }
  for (int i = 0; i < n; ++i) {
    gpx[i] = 0;
    gpy[i] = 0;
    if (kind == K_CORE_SERSIC) {
      I[i] = core_sersic_fwd<R>(d.data(), x[i], y[i]);
      core_sersic_vjp<R>(d.data(), x[i], y[i], gI[i], acc.data(), gpx[i], gpy[i]);
    } else if (kind == K_SHAPELETS) {
      I[i] = shapelets_fwd<R, SH_CAP>(d.data(), tab.data(), stride, interp, x[i], y[i]);
      shapelets_vjp<R, SH_CAP>(d.data(), tab.data(), stride, interp, x[i], y[i], gI[i], acc.data(), gpx[i], gpy[i]);
    } else {
      I[i] = sersic_fwd<R>(d.data(), x[i], y[i]);
      sersic_vjp<R>(d.data(), x[i], y[i], gI[i], acc.data(), gpx[i], gpy[i]);
    }
  }
  switch (kind) {
    case K_SERSIC: sersic_finalize<R>(p, false, acc.data(), grad); break;
    case K_SERSIC_ELLIPSE: sersic_finalize<R>(p, true, acc.data(), grad); break;
    case K_SHAPELETS: shapelets_finalize<R>(p, iparam, acc.data(), grad); break;
    case K_CORE_SERSIC: core_sersic_finalize<R>(p, acc.data(), grad); break;
  }
}

}  // namespace

// Nested-dual evaluation used by the image-position kernels (gl_positions.hip.h), on the host in float64:
// out = [ax, ay, fxx, fxy, fyx, fyy] followed by d(those 6)/d(param k) for k < P  ->  6 * (1 + P) doubles.
template <int PL> static void lens_jet(int kind, int iparam, const double* p0, double x0, double y0, double* out) {
  using R1 = gld::Dual<double, PL>;
  using R = gld::Dual<R1, 2>;
  R x{R1(x0)}, y{R1(y0)};
  x.d[0] = R1(1.0);
  y.d[1] = R1(1.0);
  R p[PL];
  for (int k = 0; k < PL; ++k) { R1 v(p0[k]); v.d[k] = 1.0; p[k] = R(v); }
  R ax, ay;
  switch (kind) {
    case K_EPL: epl_point<R>(p, iparam, x, y, ax, ay); break;
    case K_SIE: { R d[SIE_ND + 1]; sie_prep<R>(p, d); sie_fwd<R>(d, x, y, ax, ay); } break;
    case K_NFW: { R d[NFW_ND]; nfw_prep<R>(p, d); nfw_fwd<R>(d, x, y, ax, ay); } break;
    case K_SHEAR: { R d[4]; shear_prep<R>(p, d); shear_fwd<R>(d, x, y, ax, ay); } break;
    case K_DPIS: case K_DPIE: case K_DPIEP: { R d[DPX_ND]; dpie_prep<R>(kind, p, d); dpie_fwd<R>(kind, d, x, y, ax, ay); } break;
    case K_NFW_ELLIPSE: { R d[NFE_ND]; nfw_ell_prep<R>(p, d); nfw_ell_fwd<R>(d, x, y, ax, ay); } break;
    case K_TNFW: { R d[TNF_ND]; tnfw_prep<R>(p, d); tnfw_fwd<R>(d, x, y, ax, ay); } break;
    default: { R d[4]; sis_prep<R>(p, d); sis_fwd<R>(d, x, y, ax, ay); } break;
  }
  if (kind == K_DPIS) {  // the reference's analytic override (piemd.py:62-83), see dpis_kappa_excess
    R1 d1[DPX_ND], p1[PL];
    for (int k = 0; k < PL; ++k) p1[k] = p[k].v;
    dpie_prep<R1>(kind, p1, d1);
    R1 ex = dpis_kappa_excess<R1>(d1, d1 + DP_NS, R1(x0), R1(y0));
    ax.d[0] += ex;
    ay.d[1] += ex;
  }
  const R1 q[6] = {ax.v, ay.v, ax.d[0], ax.d[1], ay.d[0], ay.d[1]};
  for (int i = 0; i < 6; ++i) {
    out[i] = q[i].v;
    for (int k = 0; k < PL; ++k) out[6 * (1 + k) + i] = q[i].d[k];
  }
}

extern "C" {
void hm_mass_f64(int kind, int iparam, const double* p, int n, const double* x, const double* y, const double* gx,
                 const double* gy, double* ax, double* ay, double* grad) {
  run_mass<double>(kind, iparam, p, n, x, y, gx, gy, ax, ay, grad);
}
void hm_mass_f32(int kind, int iparam, const float* p, int n, const float* x, const float* y, const float* gx,
                 const float* gy, float* ax, float* ay, float* grad) {
  run_mass<float>(kind, iparam, p, n, x, y, gx, gy, ax, ay, grad);
}
void hm_light_f64(int kind, int iparam, unsigned flags, const double* p, int n, const double* x, const double* y,
                  const double* gI, double* I, double* grad, double* gpx, double* gpy) {
  run_light<double>(kind, iparam, flags, p, n, x, y, gI, I, grad, gpx, gpy);
}
void hm_light_f32(int kind, int iparam, unsigned flags, const float* p, int n, const float* x, const float* y,
                  const float* gI, float* I, float* grad, float* gpx, float* gpy) {
  run_light<float>(kind, iparam, flags, p, n, x, y, gI, I, grad, gpx, gpy);
}
void hm_chi2_f64(int n, const double* m, const double* o, const double* w, int has_err, const double* err, double bg2,
                 double inv_t, double* chi2, double* norm, double* gm) {
  double c = 0, nm = 0;
  for (int i = 0; i < n; ++i) {
    double c2, n2;
    chi2_terms<double>(m[i], o[i], w[i], has_err, has_err ? err[i] : 1.0, bg2, inv_t, c2, n2);
    c += c2;
    nm += n2;
    gm[i] = chi2_gm<double>(m[i], o[i], w[i], has_err, has_err ? err[i] : 1.0, bg2, inv_t);
  }
  *chi2 = c;
  *norm = nm;
}
int hm_num_params(int kind, int iparam) { return kind_num_params(kind, iparam); }

// The NFW table in s = X^2 (gl_host_tables.h::build_nfw_table_s) read the way gl_clusterw.hip.h::nfw_fwd_s reads it, in float32
// with fused multiply-adds: out_h = H(s), out_dhds = dH/ds for every s inside the table (NaN outside).
void hm_nfw_table_s_f32(int n, const float* s, float* out_h, float* out_dhds) {
  static std::vector<float> tab;
  if (tab.empty()) glh::build_nfw_table_s([](double X, double& g, double& gp) { nfw_gw<double>(X, g, gp); }, tab);
  const int N = glh::kNfwSIntervals;
  for (int k = 0; k < n; ++k) {
    uint32_t b;
    std::memcpy(&b, &s[k], 4);
    const uint32_t i = (b >> 17) - ((uint32_t)(127 + glh::kNfwSLog2Lo) << 6);
    if (!(i < (uint32_t)N)) { out_h[k] = out_dhds[k] = NAN; continue; }
    const float c0 = tab[i], c1 = tab[N + i], c2 = tab[2 * N + i], c3 = tab[3 * N + i], tau = (float)(b & 0x1FFFFu);
    const float p1 = std::fmaf(c3, tau, c2), p2 = std::fmaf(p1, tau, c1), q2 = std::fmaf(c3, tau, p1);
    out_h[k] = std::fmaf(p2, tau, c0);
    const uint32_t sb = 0x8A800000u - (b & 0x7F800000u);
    float dtds;
    std::memcpy(&dtds, &sb, 4);
    out_dhds[k] = std::fmaf(q2, tau, p2) * dtds;
  }
}
// h(X) = g(X) / X^2 and h'(X) of the closed form in float64 (nfw_gw: the function both tables are built from)
void hm_nfw_h_f64(int n, const double* X, double* h, double* hp) {
  for (int k = 0; k < n; ++k) {
    double g, gp;
    nfw_gw<double>(X[k], g, gp);
    const double iX = 1.0 / X[k];
    h[k] = g * iX * iX;
    hp[k] = gp * iX * iX - 2.0 * h[k] * iX;
  }
}

void hm_lens_jet_f64(int kind, int iparam, const double* p, double x, double y, double* out) {
  switch (kind) {
    case K_EPL: lens_jet<6>(kind, iparam, p, x, y, out); break;
    case K_SIE: lens_jet<5>(kind, iparam, p, x, y, out); break;
    case K_NFW: lens_jet<4>(kind, iparam, p, x, y, out); break;
    case K_SHEAR: lens_jet<2>(kind, iparam, p, x, y, out); break;
    case K_DPIS: lens_jet<5>(kind, iparam, p, x, y, out); break;
    case K_DPIE: case K_DPIEP: lens_jet<7>(kind, iparam, p, x, y, out); break;
    case K_NFW_ELLIPSE: lens_jet<6>(kind, iparam, p, x, y, out); break;
    case K_TNFW: lens_jet<5>(kind, iparam, p, x, y, out); break;
    default: lens_jet<3>(kind, iparam, p, x, y, out); break;
  }
}
void hm_scaled_f64(int base_kind, int n_gal, const float* table, const int* cols, const double* scales, int n,
                   const double* x, const double* y, const double* gx, const double* gy, double* ax, double* ay,
                   double* gscales) {
  run_scaled<double>(base_kind, n_gal, table, cols, scales, n, x, y, gx, gy, ax, ay, gscales);
}
// Taylor coefficients C_n (n <= 5) of the population deflection in the cut-radius scale: out [n][2][6]
void hm_series_f64(int base_kind, int n_gal, const float* table, const int* cols, const double* scales, int n,
                   const double* x, const double* y, double* out) {
  ScaledDesc sd{base_kind, n_gal, {cols[0], cols[1], cols[2]}};
  for (int i = 0; i < n; ++i) series_point<5, double>(sd, table, scales, x[i], y[i], out + 12 * i, out + 12 * i + 6);
}
// ... and of its Hessian (Dual<Jet<double,5>,2>): out [n][3][6] = f_xx, f_xy, f_yy
void hm_series_hessian_f64(int base_kind, int n_gal, const float* table, const int* cols, const double* scales, int n,
                           const double* x, const double* y, double* out) {
  ScaledDesc sd{base_kind, n_gal, {cols[0], cols[1], cols[2]}};
  for (int i = 0; i < n; ++i)
    series_point_hessian<5, double>(sd, table, scales, x[i], y[i], out + 18 * i, out + 18 * i + 6, out + 18 * i + 12);
}
// the pseudo-inverse solve of the linear-amplitude normal equations (gl_eigh.h), serial context:
// A [n][n] symmetric float32, rhs [n] -> coeffs [n], eigenvalues [n] (of A / max|diag|, unsorted)
int hm_eigh_pinv_f32(const float* A_in, int n, const float* rhs, float rcond, int allow_shortcut, float* coeffs, float* eigs) {
  const int ld = n | 1;
  std::vector<float> A((size_t)n * ld, 0.f), Z((size_t)n * ld, 0.f), d(n + 1), e(n + 2), v(n + 1), p(n + 1), bet(n + 1, 0.f),
      g(n + 1), y(n + 1);
  float scale = 0.f;
  for (int i = 0; i < n; ++i) scale = std::max(scale, std::fabs(A_in[(size_t)i * n + i]));
  const float inv = scale > 0.f ? 1.0f / scale : 0.f;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) A[(size_t)i * ld + j] = A_in[(size_t)i * n + j] * inv;
  static gle::SerialCtx cx;
  const int took = gle::pinv_solve(cx, A.data(), Z.data(), n, ld, rhs, rcond, inv, d.data(), e.data(), v.data(), p.data(),
                                   bet.data(), g.data(), y.data(), coeffs, allow_shortcut != 0);
  for (int i = 0; i < n; ++i) eigs[i] = d[i];  // eigenvalues on the general path, diag(T) on the short cut
  return took;
}
void hm_scaled_f32(int base_kind, int n_gal, const float* table, const int* cols, const float* scales, int n,
                   const float* x, const float* y, const float* gx, const float* gy, float* ax, float* ay,
                   float* gscales) {
  run_scaled<float>(base_kind, n_gal, table, cols, scales, n, x, y, gx, gy, ax, ay, gscales);
}
}
