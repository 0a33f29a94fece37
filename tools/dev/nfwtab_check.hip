#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../../gigalens_amd/csrc/gl_kernels.hip.h"
#include "../../gigalens_amd/csrc/gl_static.hip.h"
#include "../../gigalens_amd/csrc/gl_pair.hip.h"
#include "../../gigalens_amd/csrc/gl_cluster.hip.h"
#include "../../gigalens_amd/csrc/gl_host_tables.h"
using namespace glk;
__global__ void k(const float* tab, const float* X, float* h, float* hp, int n) {
  extern __shared__ float s_tab[];
  for (int i = threadIdx.x; i < 2 * NFW_TAB_NODES; i += blockDim.x) s_tab[i] = tab[i];
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  v2f x{X[2 * i], X[2 * i + 1]};
  v2f ix = rcp(x), hh, hpp;
  nfw_h_pair(s_tab, x, ix, hh, hpp);
  h[2 * i] = hh.x; h[2 * i + 1] = hh.y; hp[2 * i] = hpp.x; hp[2 * i + 1] = hpp.y;
}
int main() {
  std::vector<float> tab;
  glh::build_nfw_table([](double X, double& g, double& gp) { glp::nfw_gw<double>(X, g, gp); }, tab);
  const int n = 1 << 16;
  std::vector<float> X(n), h(n), hp(n);
  for (int i = 0; i < n; ++i) X[i] = std::exp2(-9.0 + 17.0 * (i + 0.37) / n);
  X[100] = 1.0f;
  for (int i = 2000; i < 60000; i += 1001) X[i] = (i & 2) ? 3e-4f : 200.f;  // mixed pairs: one lane outside the table
  float *dt, *dX, *dh, *dhp;
  hipMalloc(&dt, tab.size() * 4); hipMalloc(&dX, n * 4); hipMalloc(&dh, n * 4); hipMalloc(&dhp, n * 4);
  hipMemcpy(dt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), sizeof(float) * 2 * NFW_TAB_NODES, 0, dt, dX, dh, dhp, n);
  hipMemcpy(h.data(), dh, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hp.data(), dhp, n * 4, hipMemcpyDeviceToHost);
  double worst = 0, worstp = 0; int wi = 0;
  for (int i = 0; i < n; ++i) {
    double g, gp; glp::nfw_gw<double>((double)X[i], g, gp);
    double hx = g / ((double)X[i] * X[i]), hpx = gp / ((double)X[i] * X[i]) - 2 * hx / X[i];
    double e = X[i] == 1.0f ? 0.0 : fabs(h[i] - hx) / fabs(hx);
    if (e > worst) { worst = e; wi = i; }
    if (X[i] == 1.0f) continue;
    worstp = fmax(worstp, fabs(hp[i] - hpx) / fabs(hpx));
  }
  printf("worst rel h %g at X=%g (got %g) ; hp %g ; h(1)=%g\n", worst, X[wi], h[wi], worstp, h[100]);
  for (int i : {10, 5000, 20000, 40000, 60000}) printf("X=%g h=%g hp=%g\n", X[i], h[i], hp[i]);
}
