// Micro-benchmark: what rate does v_mfma_f32_16x16x4_f32 reach in the shape of gl_shp_normal_kernel's inner loop?
//   variant 0: 15 independent accumulators, operands in registers (pure MFMA)
//   variant 1: + per 15 MFMAs, 10 ds_read_b32 of operand factors and 5 v_mul (the product loop of the kernel)
// Build:  hipcc --offload-arch=gfx950 -O3 -o mfma_f32_rate tools/dev/micro/mfma_f32_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ void __launch_bounds__(256, 3) k(float* out, int trips) {
  extern __shared__ float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* xw = sm + wave * 13 * 132;
  for (int i = lane; i < 13 * 132; i += 64) xw[i] = 1.0f + 1e-3f * (float)(i % 7);
  __syncthreads();
  const int c = lane & 15, kq = lane >> 4;
  const float* ra[5];
  const float* rb[5];
  for (int t = 0; t < 5; ++t) {
    ra[t] = xw + ((c + 3 * t) % 6) * 132 + (c & 1) + 2 * kq;
    rb[t] = xw + (6 + (c + t) % 6) * 132 + ((c >> 1) & 1) + 2 * kq;
  }
  v4f acc[15];
  for (int q = 0; q < 15; ++q) acc[q] = v4f{0.f, 0.f, 0.f, 0.f};
  float v[5] = {1.f + lane, 2.f, 3.f, 4.f, 5.f};
  for (int it = 0; it < trips; ++it) {
#pragma unroll 2
    for (int kb = 0; kb < 16; ++kb) {
      if (VAR == 1) {
#pragma unroll
        for (int t = 0; t < 5; ++t) v[t] = ra[t][8 * kb] * rb[t][8 * kb];
      }
      int q = 0;
#pragma unroll
      for (int ti = 0; ti < 5; ++ti)
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj, ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[ti], v[tj], acc[q], 0, 0, 0);
    }
    if (VAR == 1) { __builtin_amdgcn_wave_barrier(); }
  }
  float s = 0.f;
  for (int q = 0; q < 15; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int trips = 64;
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * 8192);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int var = 0; var < 2; ++var)
    for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu) {
      const int grid = 256 * wgs_per_cu;
      const size_t lds = 4 * 13 * 132 * sizeof(float);
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (var == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), lds, 0, out, trips);
        else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), lds, 0, out, trips);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) {
          const double mfma = (double)grid * 4 * trips * 16 * 15;
          const double flops = mfma * 2048.0;
          printf("variant %d, %d workgroup(s)/CU (%d wave(s)/SIMD): %.3f ms, %.1f TFLOP/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", var,
                 wgs_per_cu, wgs_per_cu, ms, flops / ms * 1e-9, ms * 1e-3 * 2.4e9 / (mfma / 1024.0));
        }
      }
    }
  return 0;
}
