// Development probe (round 4): how far is beta = (x, y) - alpha_EPL - alpha_shear of the PAIR kernels' device code (gl_vec.hip.h:
// Clenshaw series, hardware rsq / log2 / exp2) from float64, and is the error a per-sample BIAS or pixel noise?  The gradient
// w.r.t. a source centre that sits near a caustic has a condition number ~1e4 in a uniform offset of beta (tools/dev/psf_grad_probe.py).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/dev/epl_beta_check.hip -o /tmp/epl_beta_check && /tmp/epl_beta_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../gigalens_amd/csrc/gl_kernels.hip.h"
using namespace glk;

__global__ void k_pair(const float* der_e, const float* der_s, const float* X, const float* Y, float* bx, float* by, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  v2f x{X[2 * i], X[2 * i + 1]}, y{Y[2 * i], Y[2 * i + 1]}, ox = x, oy = y;
  EplStateV<v2f> st;
  epl_fwd_v<v2f, true>(der_e, der_e, x, y, ox, oy, st);
  shear_fwd_v<v2f>(der_s, x, y, ox, oy);
  bx[2 * i] = ox.x; bx[2 * i + 1] = ox.y; by[2 * i] = oy.x; by[2 * i + 1] = oy.y;
}
__global__ void k_scalar(const float* der_e, const float* der_s, const float* X, const float* Y, float* bx, float* by, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float ax, ay, sx, sy;
  glp::epl_fwd<float>(der_e, X[i], Y[i], ax, ay);
  glp::shear_fwd<float>(der_s, X[i], Y[i], sx, sy);
  bx[i] = X[i] - ax - sx; by[i] = Y[i] - ay - sy;
}
int main() {
  const double rows[4][8] = {{0, 0, 0, 0, 0, 0, 0, 0},
                             {2.2602527, 1.9558101, -9.4454125e-02, -4.8153888e-02, 4.2441826e-02, -3.3878796e-02, 1.4147206e-02, -1.1726976e-02},
                             {1.3, 2.1, 0.15, -0.2, -0.03, 0.02, 0.03, 0.01}, {0.9, 1.7, -0.3, 0.1, 0.05, 0.06, -0.02, 0.04}};
  const int npix = 60, n = npix * npix;
  std::vector<float> X(n), Y(n);
  for (int r = 0; r < npix; ++r) for (int c = 0; c < npix; ++c) { X[r * npix + c] = (float)((c - (npix - 1) / 2.0) * 0.08); Y[r * npix + c] = (float)((r - (npix - 1) / 2.0) * 0.08); }
  float *dX, *dY, *dbx, *dby, *de, *ds;
  hipMalloc(&dX, n * 4); hipMalloc(&dY, n * 4); hipMalloc(&dbx, n * 4); hipMalloc(&dby, n * 4); hipMalloc(&de, 4096); hipMalloc(&ds, 64);
  hipMemcpy(dX, X.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dY, Y.data(), n * 4, hipMemcpyHostToDevice);
  for (int b = 1; b < 4; ++b) {
    float pe[6], ps[2]; double pe64[6], ps64[2];
    for (int k = 0; k < 6; ++k) { pe[k] = (float)rows[b][k]; pe64[k] = pe[k]; }
    for (int k = 0; k < 2; ++k) { ps[k] = (float)rows[b][6 + k]; ps64[k] = ps[k]; }
    std::vector<float> d_e(1024, 0.f), d_s(8, 0.f); std::vector<double> d_e64(1024, 0.0), d_s64(8, 0.0);
    glp::epl_prep<float>(pe, 50, d_e.data()); glp::shear_prep<float>(ps, d_s.data());
    glp::epl_prep<double>(pe64, 50, d_e64.data()); glp::shear_prep<double>(ps64, d_s64.data());
    // variant: the float32 block rounded ONCE from the float64 block (what a float64 front end would hand the kernels)
    std::vector<float> d_e_r(1024, 0.f);
    for (int k = 0; k < 1024; ++k) d_e_r[k] = (float)d_e64[k];
    reinterpret_cast<int*>(d_e_r.data())[glp::EPL_KI] = reinterpret_cast<int*>(d_e.data())[glp::EPL_KI];
    std::vector<double> rx(n), ry(n);
    for (int i = 0; i < n; ++i) {
      double ax, ay, sx, sy;
      glp::epl_fwd<double>(d_e64.data(), (double)X[i], (double)Y[i], ax, ay);
      glp::shear_fwd<double>(d_s64.data(), (double)X[i], (double)Y[i], sx, sy);
      rx[i] = X[i] - ax - sx; ry[i] = Y[i] - ay - sy;
    }
    hipMemcpy(ds, d_s.data(), 32, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 3; ++variant) {
      hipMemcpy(de, variant == 2 ? d_e_r.data() : d_e.data(), 4096, hipMemcpyHostToDevice);
      if (variant == 1) hipLaunchKernelGGL(k_scalar, dim3((n + 255) / 256), dim3(256), 0, 0, de, ds, dX, dY, dbx, dby, n);
      else hipLaunchKernelGGL(k_pair, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, de, ds, dX, dY, dbx, dby, n);
      std::vector<float> bx(n), by(n);
      hipMemcpy(bx.data(), dbx, n * 4, hipMemcpyDeviceToHost); hipMemcpy(by.data(), dby, n * 4, hipMemcpyDeviceToHost);
      double mx = 0, my = 0, sx2 = 0, sy2 = 0;
      for (int i = 0; i < n; ++i) { mx += bx[i] - rx[i]; my += by[i] - ry[i]; }
      mx /= n; my /= n;
      for (int i = 0; i < n; ++i) { sx2 += std::pow(bx[i] - rx[i] - mx, 2); sy2 += std::pow(by[i] - ry[i] - my, 2); }
      printf("row %d %-34s beta err: x mean %+.2e std %.2e | y mean %+.2e std %.2e\n", b,
             variant == 0 ? "pair kernel code" : variant == 1 ? "scalar device code" : "pair code, block rounded from f64", mx, std::sqrt(sx2 / n), my, std::sqrt(sy2 / n));
    }
  }
  return 0;
}
