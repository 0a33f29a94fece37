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

// ---- register-blocked, sample-pair-packed correlation (the fast path of both directions) -----------------------------
// The kernels above spend one LDS read and one scalar-loaded tap per multiply-add: 17 (forward) and 8 (transpose) TFLOP/s on
// the reference's demo set-up, where they are 80 % of a log-prob step.  Here
//   * a thread owns EIGHT consecutive outputs of a row and keeps the input window they share ((8 - 1) ST + KWP columns) in
//     registers: one LDS read per 5 multiply-adds instead of one per one (with four outputs the LDS pipe, shared by the four
//     SIMDs of a CU, was still the bound), every tap an SGPR operand;
//   * the two halves of every packed fp32 instruction are TWO SAMPLES (b, b + 1): the LDS tile interleaves them, a tap
//     multiplies both (v_pk_fma_f32 with a scalar operand) -- the only pairing whose operands are adjacent registers;
//   * both directions are the same kernel.  Forward: stride ST = ss over the supersampled image with Keff.  Transpose: for
//     each of the ss^2 residue classes (i + pt, j + pl) mod ss the gather over the pooled cotangents is a stride-1
//     correlation with the class's decimated, flipped sub-kernel (~KH / ss taps a side), written back with stride ss.
// Kernels are stored row-padded to KWP (a multiple of 4, zero taps) so the tap loop unrolls at compile time.
// (CorrClass / CorrArgs: gl_model.h, the model keeps one plan per direction)
constexpr int CORR_TR = 16, CORR_TCG = 4, CORR_OX = 8;  // output tile: 16 rows x (4 threads x 8 outputs) columns
constexpr int CORR_GT = CORR_TR * CORR_TCG;             // threads of one group (one wavefront)
// columns of the LDS tile for a window of TC input columns: up to three more on the left (the tile starts at a 16-byte boundary
// of the image row), rounded up to whole groups of four
__host__ __device__ constexpr int corr_tile_width(int TC) { return ((TC + 3 + 3) / 4) * 4; }

// KS: the kernel's rows are dealt to KS wavefronts (u = g, g + KS, ..) that share the LDS tile and add their sums at the end --
// the stride-2 tile is 41 KB, so one wavefront per tile would leave a CU with three.
// NCJ: column classes per thread (transpose: ss -- a thread then writes ss * 8 CONSECUTIVE floats of a row; one class per
// launch unit wrote every ss-th float of a 29 MB buffer from different workgroups, and the partial-line writes bound the kernel)
// OX: consecutive outputs of a thread (8; 16 for the wide stride-2 kernels, whose main loop is otherwise bound by the LDS pipe:
// 42 window reads per 224 multiply-adds at 8, 58 per 448 at 16)
template <int KWP, int ST, int KS, int NCJ, int OX>
__global__ void __launch_bounds__(CORR_GT * KS) gl_corr_pair_kernel(const float* __restrict__ in, float* __restrict__ out, CorrArgs p) {
  typedef float v2 __attribute__((ext_vector_type(2)));
  extern __shared__ float2 ctile[];
  constexpr int NT = CORR_GT * KS;
  // workgroup -> (tile, class, sample pair).  Consecutive workgroups go to consecutive XCDs, each with an L2 of its own: in launch
  // order the tiles of ONE sample pair -- which share their halos, 2.9 x the image in all on the demo set-up -- would land on eight
  // different L2s and every halo would come from memory again.  Instead the k-th workgroup of XCD x walks the tiles and classes of
  // pair 8 j + x before moving to pair 8 (j + 1) + x (the pairs beyond a multiple of eight are dealt the same way among themselves).
  int ci, bp, bx, by;
  {
    const int gx = gridDim.x, gy = gridDim.y, per_pair = gx * gy * p.n_class, n_pairs = (p.B + 1) / 2;
    const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int full = (n_pairs / 8) * 8, lin_full = full * per_pair;
    int t;
    if (lin < lin_full) {
      const int q = lin >> 3;
      t = q % per_pair;
      bp = (q / per_pair) * 8 + (lin & 7);
    } else {
      const int rest = n_pairs - full, l2 = lin - lin_full;
      t = l2 / rest;
      bp = full + l2 % rest;
    }
    bx = t % gx;
    by = (t / gx) % gy;
    ci = t / (gx * gy);
  }
  const CorrClass c = p.cls[ci];
  const int I0 = by * CORR_TR, J0 = bx * (CORR_TCG * OX);
  if (I0 >= c.Ho || J0 >= c.Wo[0]) return;  // classes differ in size: whole workgroups leave together (Wo[0] is the largest)
  const int TR = (CORR_TR - 1) * ST + c.KH;
  constexpr int TC = (CORR_TCG * OX - 1) * ST + KWP, TCW = corr_tile_width(TC), TCp = TCW | 1;
  const int b0 = 2 * bp, b1 = min(b0 + 1, p.B - 1);
  const bool has1 = b0 + 1 < p.B;
  const float* in0 = in + (size_t)b0 * p.Hi * p.Wi;
  const float* in1 = in + (size_t)b1 * p.Hi * p.Wi;
  // the tile starts at the 16-byte boundary at or before its first column (sh columns earlier): rows are then filled by float4 loads
  const int r0 = I0 * ST - c.pt, c_first = J0 * ST - c.pl, sh = p.vec ? (c_first & 3) : 0, c0 = c_first - sh;
  // fill: one wavefront per tile row, lanes along the row (no index division, row-contiguous global reads); all of a wavefront's
  // loads (up to 16 per lane) are issued before the first LDS write: one global round trip per wavefront instead of one per row
  // (round 4: the fill was 11 of the forward kernel's 51 us on the demo set-up and neither latency -- 32 loads in flight, a
  // staggered first round and a register-prefetching resident-workgroup form measured no faster or slower -- nor memory: it
  // is INSTRUCTIONS, ~20 per float2 with its two dword loads against 5 multiply-adds per LDS read in the main loop.  With the image
  // width a multiple of four a lane now moves four columns of both samples per step: two 16-byte loads, four 8-byte LDS writes.)
  constexpr int NW = NT / 64;
  if (p.vec) {
    constexpr int NG = TCW / 4, LPR = NG <= 8 ? 8 : NG <= 16 ? 16 : NG <= 32 ? 32 : 64, RPI = 64 / LPR, RSTEP = NW * RPI, FB = 4;
    static_assert(NG <= 64, "a tile row is at most 64 column groups");
    const int lane = threadIdx.x & 63, gl = lane % LPR, rl = lane / LPR + RPI * (int)(threadIdx.x >> 6);
    const int gc = c0 + 4 * gl;
    const bool col_in = gl < NG && gc >= 0 && gc < p.Wi;  // c0 and Wi are multiples of four: a group is inside or outside as a whole
    for (int rb = GL_DBG(p.dbg, 16) ? TR : 0; rb < TR; rb += RSTEP * FB) {
      float4 a0[FB], a1[FB];
#pragma unroll
      for (int f = 0; f < FB; ++f) {
        const int r = rb + rl + f * RSTEP, gr = r0 + r;
        const bool inside = col_in && r < TR && gr >= 0 && gr < p.Hi;
        const unsigned off = inside ? (unsigned)(gr * p.Wi + gc) : 0u;
        a0[f] = *reinterpret_cast<const float4*>(in0 + off);
        a1[f] = *reinterpret_cast<const float4*>(in1 + off);
      }
#pragma unroll
      for (int f = 0; f < FB; ++f) {
        const int r = rb + rl + f * RSTEP, gr = r0 + r;
        const bool inside = col_in && gr >= 0 && gr < p.Hi;
        if (r < TR && gl < NG) {
          float2* d = ctile + r * TCp + 4 * gl;
          d[0] = inside ? float2{a0[f].x, a1[f].x} : float2{0.f, 0.f};
          d[1] = inside ? float2{a0[f].y, a1[f].y} : float2{0.f, 0.f};
          d[2] = inside ? float2{a0[f].z, a1[f].z} : float2{0.f, 0.f};
          d[3] = inside ? float2{a0[f].w, a1[f].w} : float2{0.f, 0.f};
        }
      }
    }
  } else {
  constexpr int CPASS = (TC + 63) / 64, FR = 16 / CPASS;  // 16 float2 of loads in flight per lane
  for (int rb = GL_DBG(p.dbg, 16) ? TR : (int)(threadIdx.x >> 6); rb < TR; rb += NW * FR) {
    float2 v[FR][CPASS];
#pragma unroll
    for (int f = 0; f < FR; ++f) {
      const int r = rb + f * NW, gr = r0 + r;
      const bool rin = r < TR && gr >= 0 && gr < p.Hi;
      const size_t roff = (size_t)(rin ? gr : 0) * p.Wi;
#pragma unroll
      for (int q = 0; q < CPASS; ++q) {
        const int gc = c0 + (int)(threadIdx.x & 63) + 64 * q;
        const bool inside = rin && gc >= 0 && gc < p.Wi;
        const size_t off = roff + (inside ? gc : 0);
        const float a0 = in0[off], a1 = in1[off];
        v[f][q] = inside ? float2{a0, a1} : float2{0.f, 0.f};
      }
    }
#pragma unroll
    for (int f = 0; f < FR; ++f) {
      const int r = rb + f * NW;
#pragma unroll
      for (int q = 0; q < CPASS; ++q) {
        const int cc = (int)(threadIdx.x & 63) + 64 * q;
        if (r < TR && cc < TC) ctile[r * TCp + cc] = v[f][q];
      }
    }
  }
  }
  __syncthreads();
  // lane -> (row ti fastest, column group tg): the 16 lanes of an LDS pass read 16 different rows, ST * TCp float2 apart with
  // TCp odd -> distinct banks (column groups are 16 banks apart: four of them would collide)
  static_assert(CORR_GT == 64, "one group = one wavefront: its kernel row is wave-uniform");
  const int g = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / CORR_GT)), t128 = threadIdx.x % CORR_GT;  // scalar: taps come through s_load
  const int ti = t128 % CORR_TR, tg = t128 / CORR_TR;
  constexpr int WIN = (OX - 1) * ST + KWP;
  v2 acc[NCJ][OX];
#pragma unroll
  for (int j = 0; j < NCJ; ++j)
#pragma unroll
    for (int o = 0; o < OX; ++o) acc[j][o] = v2{0.f, 0.f};
  const float2* base = ctile + (ti * ST) * TCp + tg * OX * ST + sh;
  const float* __restrict__ kc = p.k + c.koff;
  for (int u = GL_DBG(p.dbg, 32) ? c.KH : g; u < c.KH; u += KS) {
    const float2* row = base + u * TCp;
    v2 w[WIN];
#pragma unroll
    for (int q = 0; q < WIN; ++q) { const float2 t = row[q]; w[q] = v2{t.x, t.y}; }
    const float* __restrict__ kr = kc + u * (NCJ * KWP);  // wave-uniform: scalar loads
#pragma unroll
    for (int j = 0; j < NCJ; ++j)
#pragma unroll
      for (int v = 0; v < KWP; ++v) {
        const float kv = kr[j * KWP + v];
#pragma unroll
        for (int o = 0; o < OX; ++o) acc[j][o] = __builtin_elementwise_fma(w[o * ST + v], v2{kv, kv}, acc[j][o]);
      }
  }
  // ---- epilogue: sums of the row groups, then the tile leaves through LDS so that the global writes are row-contiguous (a lane
  // owns eight outputs of ONE row and neighbouring lanes different rows: written from registers every store instruction touched
  // 64 cache lines -- 42 of the transpose's 64 us were those stores)
  float2* sred = ctile;
  float* otile = reinterpret_cast<float*>(ctile + (KS - 1) * NCJ * OX * CORR_GT);  // [2 samples][16 rows][32 NCJ columns]
  constexpr int WT = CORR_TCG * OX * NCJ;
  // every wavefront OWNS NA / KS of the thread's NA sums: it receives the other row groups' partial sums of those and sends its
  // partial sums of theirs (one wavefront adding up everything left the others waiting at the barrier).  Added in row-group
  // order, whoever owns the sum.
  constexpr int NA = NCJ * OX, OWN = NA / KS;
  static_assert(KS > 1 && NA % KS == 0, "the sums are dealt evenly to the row groups");
  __syncthreads();  // every wavefront is done with the input tile
#pragma unroll
  for (int gg = 0; gg < KS; ++gg)
    if (g == gg) {
#pragma unroll
      for (int ia = 0; ia < NA; ++ia) {
        const int owner = ia / OWN;
        if (owner != gg) sred[(((gg < owner ? gg : gg - 1) * NA + ia)) * CORR_GT + t128] = float2{acc[ia / OX][ia % OX].x, acc[ia / OX][ia % OX].y};
      }
    }
  __syncthreads();
#pragma unroll
  for (int gg = 0; gg < KS; ++gg)
    if (g == gg) {
#pragma unroll
      for (int ia = gg * OWN; ia < (gg + 1) * OWN; ++ia) {
        const int j = ia / OX, o = ia % OX;
        v2 sum = v2{0.f, 0.f};
#pragma unroll
        for (int src = 0; src < KS; ++src) {
          v2 part = acc[j][o];
          if (src != gg) {
            const float2 t = sred[(((src < gg ? src : src - 1) * NA + ia)) * CORR_GT + t128];
            part = v2{t.x, t.y};
          }
          sum = src == 0 ? part : sum + part;
        }
        const int col = (tg * OX + o) * NCJ + c.oo_c[j];  // NCJ == the placement stride of the plan
        otile[ti * WT + col] = sum.x * p.scale;
        otile[(CORR_TR + ti) * WT + col] = sum.y * p.scale;
      }
    }
  __syncthreads();
  if (p.vec) {  // four consecutive outputs per lane (the output width is a multiple of four: a group is inside or outside as a whole)
    for (int e = GL_DBG(p.dbg, 64) ? 2 * CORR_TR * WT : 4 * (int)threadIdx.x; e < 2 * CORR_TR * WT; e += 4 * NT) {
      const int sidx = e / (CORR_TR * WT), rem = e - sidx * (CORR_TR * WT);
      const int row = rem / WT, col = rem - row * WT;
      const int gi = (I0 + row) * p.os + c.oo_r, gj = J0 * p.os + col;
      if (gi < p.Hout && I0 + row < c.Ho && gj < p.Wout && (sidx == 0 || has1))
        *reinterpret_cast<float4*>(out + ((size_t)(sidx ? b1 : b0) * p.Hout + gi) * p.Wout + gj) = *reinterpret_cast<const float4*>(otile + e);
    }
    return;
  }
  for (int idx = GL_DBG(p.dbg, 64) ? 2 * CORR_TR * WT : (int)threadIdx.x; idx < 2 * CORR_TR * WT; idx += NT) {
    const int sidx = idx / (CORR_TR * WT), rem = idx - sidx * (CORR_TR * WT);
    const int row = rem / WT, col = rem - row * WT;
    const int gi = (I0 + row) * p.os + c.oo_r, gj = J0 * p.os + col;
    if (gi < p.Hout && I0 + row < c.Ho && gj < p.Wout && (sidx == 0 || has1))
      out[((size_t)(sidx ? b1 : b0) * p.Hout + gi) * p.Wout + gj] = otile[idx];
  }
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
