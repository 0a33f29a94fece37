// gl_post.hip.h -- image-plane post-processing of LensSimulator.simulate (tf/simulator.py:142-156):
// PSF convolution (SAME, true convolution: the reference flips the kernel and cross-correlates, :62-70,145-147),
// average pooling by `supersample` (:149-155) and the det(T) scale (:156), plus the pixel statistics on a
// materialised image (tf/model.py:89-101) and the transposes needed for the gradient.
//
// Pooling is linear, so convolution + pooling collapse into ONE strided correlation with the effective kernel
//   Keff = flip(psf) (*) box(ss x ss) / ss^2          (size (kh+ss-1) x (kw+ss-1), built once on the host)
//   out[I,J] = scale * sum_{u,v} S[I*ss + u - pt, J*ss + v - pl] * Keff[u,v]
// evaluated only at the pooled pixels: ss^2 fewer outputs than convolving the supersampled image first
// (3.7x fewer MACs for a 27x27 PSF at ss = 2).  The transpose gathers, for every supersampled pixel, the
// pooled cotangents it contributed to.  Input tiles are staged in LDS; the kernel taps are read through
// wave-uniform (scalar) loads.
#pragma once
#include <hip/hip_runtime.h>

namespace glk {

struct PostArgs {
  const float* keff;  // [KH*KW]
  int KH, KW, pt, pl;
  int Hs, Ws, H, W, ss;
  float scale;
};

constexpr int PT = 16;  // output tile edge

// S [B,Hs,Ws] -> out [B,H,W]
__global__ void __launch_bounds__(256) gl_psf_pool_fwd_kernel(const float* __restrict__ S, float* __restrict__ out,
                                                              PostArgs p) {
  extern __shared__ float tile[];
  const int TR = (PT - 1) * p.ss + p.KH, TC = (PT - 1) * p.ss + p.KW, TCp = TC | 1;
  const int b = blockIdx.z, I0 = blockIdx.y * PT, J0 = blockIdx.x * PT;
  const int r0 = I0 * p.ss - p.pt, c0 = J0 * p.ss - p.pl;
  const float* Sb = S + (size_t)b * p.Hs * p.Ws;
  for (int k = threadIdx.x; k < TR * TC; k += 256) {
    int r = k / TC, c = k - r * TC;
    int gr = r0 + r, gc = c0 + c;
    tile[r * TCp + c] = (gr >= 0 && gr < p.Hs && gc >= 0 && gc < p.Ws) ? Sb[(size_t)gr * p.Ws + gc] : 0.f;
  }
  __syncthreads();
  const int ti = threadIdx.x / PT, tj = threadIdx.x % PT;
  const int I = I0 + ti, J = J0 + tj;
  const float* __restrict__ K = p.keff;
  float acc = 0.f;
  const float* base = tile + (ti * p.ss) * TCp + tj * p.ss;
  for (int u = 0; u < p.KH; ++u) {
    const float* row = base + u * TCp;
    const float* kr = K + u * p.KW;
    int v = 0;
    for (; v + 4 <= p.KW; v += 4) {
      acc = fmaf(row[v], kr[v], acc);
      acc = fmaf(row[v + 1], kr[v + 1], acc);
      acc = fmaf(row[v + 2], kr[v + 2], acc);
      acc = fmaf(row[v + 3], kr[v + 3], acc);
    }
    for (; v < p.KW; ++v) acc = fmaf(row[v], kr[v], acc);
  }
  if (I < p.H && J < p.W) out[((size_t)b * p.H + I) * p.W + J] = acc * p.scale;
}

// gP [B,H,W] -> gS [B,Hs,Ws]:  gS[i,j] = scale * sum_{I,J} gP[I,J] * Keff[i + pt - I*ss, j + pl - J*ss]
__global__ void __launch_bounds__(256) gl_psf_pool_bwd_kernel(const float* __restrict__ gP, float* __restrict__ gS,
                                                              PostArgs p) {
  extern __shared__ float tile[];
  const int b = blockIdx.z, i0 = blockIdx.y * PT, j0 = blockIdx.x * PT;
  // pooled rows/cols that can reach this tile: u = i + pt - I*ss in [0, KH)
  auto fdiv = [](int a, int d) { return (a >= 0) ? a / d : -((-a + d - 1) / d); };
  const int Ilo = fdiv(i0 + p.pt - p.KH + 1 + p.ss - 1, p.ss), Ihi = fdiv(i0 + PT - 1 + p.pt, p.ss);
  const int Jlo = fdiv(j0 + p.pl - p.KW + 1 + p.ss - 1, p.ss), Jhi = fdiv(j0 + PT - 1 + p.pl, p.ss);
  const int TR = Ihi - Ilo + 1, TC = Jhi - Jlo + 1, TCp = TC | 1;
  const float* gb = gP + (size_t)b * p.H * p.W;
  for (int k = threadIdx.x; k < TR * TC; k += 256) {
    int r = k / TC, c = k - r * TC;
    int gr = Ilo + r, gc = Jlo + c;
    tile[r * TCp + c] = (gr >= 0 && gr < p.H && gc >= 0 && gc < p.W) ? gb[(size_t)gr * p.W + gc] : 0.f;
  }
  __syncthreads();
  const int ti = threadIdx.x / PT, tj = threadIdx.x % PT;
  const int i = i0 + ti, j = j0 + tj;
  const float* __restrict__ K = p.keff;
  float acc = 0.f;
  // I runs over ceil((i+pt-KH+1)/ss) .. floor((i+pt)/ss)
  const int Ia = fdiv(i + p.pt - p.KH + 1 + p.ss - 1, p.ss), Ib = fdiv(i + p.pt, p.ss);
  const int Ja = fdiv(j + p.pl - p.KW + 1 + p.ss - 1, p.ss), Jb = fdiv(j + p.pl, p.ss);
  for (int I = Ia; I <= Ib; ++I) {
    const int u = i + p.pt - I * p.ss;
    const float* row = tile + (I - Ilo) * TCp - Jlo;
    const float* kr = K + u * p.KW + (j + p.pl);
    for (int J = Ja; J <= Jb; ++J) acc = fmaf(row[J], kr[-J * p.ss], acc);
  }
  if (i < p.Hs && j < p.Ws) gS[((size_t)b * p.Hs + i) * p.Ws + j] = acc * p.scale;
}

// pixel statistics of a materialised image (tf/model.py:89-101) and d loglike / d image
__global__ void __launch_bounds__(256) gl_imgstats_kernel(const float* __restrict__ img, const float* __restrict__ obs,
                                                          const float* __restrict__ err, const float* __restrict__ mask,
                                                          float bg2, float inv_t, int HW, float* __restrict__ stats,
                                                          float* __restrict__ gimg) {
  __shared__ float red[2][4];
  const int b = blockIdx.x;
  const float* im = img + (size_t)b * HW;
  float c2 = 0.f, nm = 0.f;
  for (int k = threadIdx.x; k < HW; k += 256) {
    float m = im[k], o = obs[k];
    float w = mask ? mask[k] : 1.f;
    float e = err ? err[k] : 1.f;
    float tc, tn;
    glp::chi2_terms<float>(m, o, w, err != nullptr, e, bg2, inv_t, tc, tn);
    c2 += tc;
    nm += tn;
    if (gimg) gimg[(size_t)b * HW + k] = glp::chi2_gm<float>(m, o, w, err != nullptr, e, bg2, inv_t);
  }
  c2 = wave_sum63(c2);
  nm = wave_sum63(nm);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 63) { red[0][wave] = c2; red[1][wave] = nm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    stats[2 * b] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    stats[2 * b + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

}  // namespace glk
