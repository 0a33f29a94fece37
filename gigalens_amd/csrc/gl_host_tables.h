// gl_host_tables.h -- host-side construction of the shapelet interpolation table.
#pragma once
#include <cmath>
#include <vector>

namespace glh {
constexpr int kShapeletNodes = 6000;

inline void build_shapelet_table(int n_max, std::vector<float>& tab, int* stride) {
  // shapelets.py:39-40,50-51: phi_n(linspace(-5,5,6000)) in f64 stored as f32; node-major layout.
  const int st = (n_max + 1 + 3) & ~3;
  tab.assign((size_t)(kShapeletNodes + 2) * st, 0.f);  // + two zero rows: where coordinates outside [-5, 5] are sent (fill 0, shapelets.py:58-60)
  for (int i = 0; i < kShapeletNodes; ++i) {
    double x = -5.0 + 10.0 * (double)i / (double)(kShapeletNodes - 1);
    double hm2 = 0.0, hm1 = 0.75112554446494248286 * exp(-0.5 * x * x);  // n = 0
    tab[(size_t)i * st + 0] = (float)hm1;
    for (int n = 1; n <= n_max; ++n) {
      double h = sqrt(2.0 / n) * x * hm1 - (n >= 2 ? sqrt((n - 1.0) / n) * hm2 : 0.0);
      tab[(size_t)i * st + n] = (float)h;
      hm2 = hm1;
      hm1 = h;
    }
  }
  *stride = st;
}

// NFW:  h(X) = g(X) / X^2  -- the radial deflection of a spherical NFW halo is  alpha(R) = 4 rho0 Rs^2 g(X) / X * (d / R)
// = K0 h(X) d  with X = R / Rs (tf/profiles/mass/nfw.py:26-52).  h has no parameters, so the cluster kernel reads it from
// ONE table shared by every halo of every sample.  Nodes follow the float format itself: octave e in [kNfwLog2Lo,
// kNfwLog2Hi), kNfwPerOctave equal steps of the mantissa inside it, X(e, j) = 2^e (1 + j / kNfwPerOctave), j = 0..kNfwPerOctave
// (the last node of an octave repeats the first of the next with the slope scaled to its own interval width) -- so a
// lane finds its interval and its position t inside it from the exponent and mantissa BITS of X, exactly, with no
// logarithm.  Node = (h, dh/dX * dX_e), dX_e = 2^e / kNfwPerOctave, for cubic Hermite interpolation in X: interpolation
// error ~1e-11 relative, so what is left is the rounding of a dozen fp32 operations, like any closed-form evaluation.
// Built in float64 from the same nfw_gw the kernels instantiate.
constexpr int kNfwPerOctave = 128, kNfwLog2Lo = -6, kNfwLog2Hi = 6;
constexpr int kNfwOctaveStride = kNfwPerOctave + 1;
constexpr int kNfwNodes = (kNfwLog2Hi - kNfwLog2Lo) * kNfwOctaveStride;

template <class GW> inline void build_nfw_table(GW gw, std::vector<float>& tab) {
  tab.assign((size_t)2 * kNfwNodes, 0.f);
  for (int e = kNfwLog2Lo; e < kNfwLog2Hi; ++e)
    for (int j = 0; j <= kNfwPerOctave; ++j) {
      const double X = std::ldexp(1.0 + (double)j / kNfwPerOctave, e), dX = std::ldexp(1.0 / kNfwPerOctave, e);
      double g, gp;
      gw(X, g, gp);
      if (X == 1.0) {  // the node AT X = 1 must carry the analytic value 1 - ln 2 (g' = 1/3), not the reference's g(1) = 1
        g = 1.0 - 0.693147180559945309417232121458;  // (that quirk applies to the single float X == 1 and is handled in the kernel)
        gp = 1.0 / 3.0;
      }
      const double iX = 1.0 / X, h = g * iX * iX, hp = gp * iX * iX - 2.0 * h * iX;
      const size_t i = (size_t)(e - kNfwLog2Lo) * kNfwOctaveStride + j;
      tab[2 * i] = (float)h;
      tab[2 * i + 1] = (float)(hp * dX);
    }
}

// ---- the same function tabulated in s = X^2 (gl_clusterw_kernel, round 4) -----------------------------------------------
// H(s) = h(sqrt s).  The deflection and its VJP need  h  and  h'(X) / X = 2 dH/ds  only (gl_clusterw.hip.h::nfw_fwd_s), so with
// s = r^2 / Rs^2 as the table variable the pixel loop takes no square root and no reciprocal.  Intervals follow the float format
// of s: octaves [2^kNfwSLog2Lo, 2^kNfwSLog2Hi), kNfwSPerOctave equal mantissa steps each (64 steps in s = the 128 steps in X of
// the table above), interval index = (bits >> 17) - base, position tau = the low 17 mantissa bits as an integer.  Per interval
// the four coefficients of the cubic Hermite interpolant in tau (SoA planes [4][kNfwSIntervals]: a lane's two pixels load into
// adjacent registers, no shuffles):  H = c0 + tau (c1 + tau (c2 + tau c3)),  dH/dtau by the same Horner steps.
constexpr int kNfwSLog2Lo = -12, kNfwSLog2Hi = 12, kNfwSPerOctave = 64, kNfwSTauBits = 17;
constexpr int kNfwSIntervals = (kNfwSLog2Hi - kNfwSLog2Lo) * kNfwSPerOctave;

template <class GW> inline void build_nfw_table_s(GW gw, std::vector<float>& tab) {
  tab.assign((size_t)4 * kNfwSIntervals, 0.f);
  auto HdH = [&](double s, double& H, double& dH) {  // H(s), dH/ds
    const double X = std::sqrt(s);
    double g, gp;
    gw(X, g, gp);
    if (X == 1.0) {  // analytic value at X = 1 (the reference's g(1) = 1 applies to the single float X == 1: kernel fallback)
      g = 1.0 - 0.693147180559945309417232121458;
      gp = 1.0 / 3.0;
    }
    const double iX = 1.0 / X, h = g * iX * iX, hp = gp * iX * iX - 2.0 * h * iX;
    H = h;
    dH = hp * 0.5 * iX;
  };
  const double tau1 = std::ldexp(1.0, kNfwSTauBits);  // tau runs over [0, 2^17)
  for (int e = kNfwSLog2Lo; e < kNfwSLog2Hi; ++e)
    for (int j = 0; j < kNfwSPerOctave; ++j) {
      const double ds = std::ldexp(1.0 / kNfwSPerOctave, e), s0 = std::ldexp(1.0 + (double)j / kNfwSPerOctave, e);
      double H0, D0, H1, D1;
      HdH(s0, H0, D0);
      HdH(s0 + ds, H1, D1);
      const double c1 = ds * D0, dd = H1 - H0, c2 = 3.0 * dd - ds * (2.0 * D0 + D1), c3 = -2.0 * dd + ds * (D0 + D1);
      const size_t i = (size_t)(e - kNfwSLog2Lo) * kNfwSPerOctave + j;
      tab[i] = (float)H0;
      tab[(size_t)kNfwSIntervals + i] = (float)(c1 / tau1);
      tab[(size_t)2 * kNfwSIntervals + i] = (float)(c2 / (tau1 * tau1));
      tab[(size_t)3 * kNfwSIntervals + i] = (float)(c3 / (tau1 * tau1 * tau1));
    }
}

}  // namespace glh
