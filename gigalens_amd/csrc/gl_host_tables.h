// gl_host_tables.h -- host-side construction of the shapelet interpolation table.
#pragma once
#include <cmath>
#include <vector>

namespace glh {
constexpr int kShapeletNodes = 6000;

inline void build_shapelet_table(int n_max, std::vector<float>& tab, int* stride) {
  // shapelets.py:39-40,50-51: phi_n(linspace(-5,5,6000)) in f64 stored as f32; node-major layout.
  const int st = (n_max + 1 + 3) & ~3;
  tab.assign((size_t)kShapeletNodes * st, 0.f);
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

}  // namespace glh
