// gl_kernels.hip.h -- CDNA4 (gfx950) kernels of the lens hot path.
//
// Launch geometry (all kernels): one 256-thread workgroup (4 wave64) works on ONE sample's chunk of
// pixels; grid = (n_chunks, B).  The sample's derived constants (gl_profiles.h) are staged in LDS
// once per workgroup and read by broadcast; pixels are thread-strided so that every global access
// (grid x/y, observed image, image rows) is a contiguous 256-float run per wave-instruction group.
// Reductions (chi^2, normalisation, parameter gradients) are wave64 DPP reductions into per-wave LDS
// slots, then one deterministic pass over the 4 waves, then one row of partials per (sample, chunk)
// in the caller's workspace; gl_finalize_kernel sums chunks in a fixed order (bitwise reproducible,
// no float atomics) and applies the per-sample chain rule to the raw parameters.
// MFMA is deliberately unused: the path is elementwise + transcendental (VALU-bound).
#pragma once
#include <hip/hip_runtime.h>
#include "gl_profiles.h"
#include "gl_dpie.h"
#include "gl_extra.h"
#include "gl_vec.hip.h"
#include "gl_members.hip.h"
#include "gl_series.h"
#include "gl_shapelets.hip.h"

namespace glk {
using namespace glp;

struct CompDesc {
  int kind, iparam;
  unsigned flags;
  int p_off;  // first column in the packed parameter row
  int d_off;  // offset of the derived block (floats, multiple of 4)
  int a_off;  // offset of the accumulators inside the per-sample accumulator row
  int n_acc;
  int n_par;
  int lin_off;  // light profiles: channel of the first linear basis image in the lstsq stack
};

// one galaxy catalogue of a K_SCALED component (gl_model_set_catalogue)
struct CatDev {
  int base_kind, n_gal;
  int g_off;   // first galaxy inside the model-wide galaxy arrays
  int comp;    // component index
  int col[3];  // theta_E, r_core, r_cut: column of the scale inside the component's parameter row, or -1
  int pad;
};

// coefficient field of one K_SERIES lens (gl_model_set_series): coef[2][order+1][N]
struct SeriesDev {
  const float* coef;   // [2][order+1][N]  deflection series
  const float* hcoef;  // [3][order+1][N]  Hessian series (f_xx, f_xy, f_yy) or null: lens maps only
  float r0;
  int order;
};

struct ZCol;
struct FinArgs {
  int n_comp, P, A, d_z;
  const float* params;
  float *loglike, *chi2, *grad;
  const float* z;
  const ZCol* zcols;
  float *logprob, *grad_z;
  float chi2_scale;
  const float* extra_stats;
  int use_partial;
  const float *pos_ll, *pos_chi2, *pos_grad;
  float pos_chi2_scale;
  const CatDev* cats;
};

enum Mode : int { IMG_FWD = 0, IMG_BWD = 1, LL_FWD = 2, LL_GRAD = 3, IMG_BASIS = 4 };
constexpr int WG = 256;
constexpr int NSTAT = 4;  // accumulator row starts with [chi2, norm, pad, pad]

struct MainArgs {
  const CompDesc* comps;
  int n_lens, n_ll, n_src;
  const float* derived;  // [B][D]
  int D, A, Apad, ncols;
  const float* gx;
  const float* gy;
  const int* pix;  // may be null
  int N, chunk;
  float* img;         // IMG_FWD out  [B][img_stride]
  const float* gimg;  // IMG_BWD in   [B][img_stride]
  long long img_stride;
  float out_scale;  // multiplies the image (conversion factor when no PSF / pooling follows)
  const float* obs;
  const float* err;
  const float* mask;
  float bg2, inv_t;
  float* partial;  // [B][n_chunks][A]
  const float* shp_tab;
  int shp_stride;
  const int* order;  // cost-ordered dispatch: blockIdx.y -> sample index (heaviest first), or null
  // tapered end of the cost-ordered dispatch (pair kernels): the samples from rank tail_from on run as tail_rows workgroups each
  // (whole 512-pixel tiles dealt evenly) instead of gridDim.x chunks; every sample owns n_rows partial rows (tail_rows = 0: off)
  int tail_rows, tail_from, n_rows, n_samples;
  unsigned parts;    // forward-only partial renders (tf/simulator.py:242-328): bit0 deflect, bit1 lens light, bit2 sources
  // galaxy catalogues (K_SCALED): per-galaxy static blocks [G][DP_NS], per-(sample, galaxy) blocks [B][G][GM_ND]
  const CatDev* cats;
  const float* gal_static;
  const float* gal_dyn;
  int G;
  int scaled_first;  // the K_SCALED lens whose scale tangents ride along the ray-shooting pass (-1: none)
  int n_lin;         // IMG_BASIS: channels of the stack  img[B][n_lin][img_stride]
  const SeriesDev* series;
  const float* nfw_tab;  // models with NFW lenses: the shared h(X) table (gl_host_tables.h), [kNfwNodes][2]; else null
  const float* neutral;  // gl_clusterw_kernel: constant blocks of an unused component slot, [NFW (4) | Sersic (16)]; else null
  float grid_rmax;       // largest |(x, y)| of the pixel grid (gl_shp.hip.h: the bound on the shear's deflection)
  int blk_w;             // table-mode shapelet kernel: image width when a wave-tile is an 8-row x 16-column BLOCK of the image (0: 128 consecutive pixels)
  int dbg;  // -DGL_EXPERIMENTS builds only (GIGALENS_HIP_DBGFLAGS): 1 skip the pixel tiles, 2 skip the epilogue reductions, 4 skip the constant staging
};

// The work-skipping dissection knobs exist only in builds made with -DGL_EXPERIMENTS (never by __graft_entry__.build()):
// in the shipped library GL_DBG is a compile-time false and gl_model_create refuses the environment variables.
#ifdef GL_EXPERIMENTS
#define GL_DBG(flags, bit) (((flags) & (bit)) != 0)
#else
#define GL_DBG(flags, bit) false
#endif

// ---- wave64 sum, result valid in lane 63 (DPP row shifts + row broadcasts, no LDS) -------------
__device__ __forceinline__ float dpp_add(float v, int ctrl, int row_mask) {
  int r;
  switch (ctrl) {  // ctrl must be an immediate
    case 0x111: r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true); break;
    case 0x112: r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xF, 0xF, true); break;
    case 0x114: r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xF, 0xF, true); break;
    case 0x118: r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xF, 0xF, true); break;
    case 0x142: r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false); break;
    default: r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false); break;
  }
  (void)row_mask;
  return v + __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float wave_sum63(float v) {
  v = dpp_add(v, 0x111, 0xF);  // row_shr:1
  v = dpp_add(v, 0x112, 0xF);  // row_shr:2
  v = dpp_add(v, 0x114, 0xF);  // row_shr:4
  v = dpp_add(v, 0x118, 0xF);  // row_shr:8   -> lane 15 of each row holds the row sum
  v = dpp_add(v, 0x142, 0xA);  // row_bcast:15 into rows 1,3
  v = dpp_add(v, 0x143, 0xC);  // row_bcast:31 into rows 2,3 -> lane 63 holds the wave sum
  return v;
}
// ---- per-tile accumulation of parameter-gradient partials ------------------------------------------
// A full wave64 reduction costs 12 VALU per value; doing it per (component, tile) dominated the
// instruction count.  Instead each value is summed over a QUAD (2 DPP adds, or over a 16-lane row with
// 4 when the model has too many accumulators for 64 LDS columns) and the quad/row leader adds it into
// its own LDS column: s_acc[col][k].  Columns are private to one lane group of one wave, so the
// read-modify-write needs no atomics and the result is order-deterministic.  The columns are summed
// once per workgroup at the end.  Apad is odd, so the 16 leader lanes of a wave hit 16 different banks.
__device__ __forceinline__ float quad_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));  // quad_perm:[1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));  // quad_perm:[2,3,0,1]
  return v;
}
// 4 x 4 transpose-reduction inside a quad: lane q of every quad receives  sum over the quad's lanes of v_q.
// Step 1 exchanges with lane ^ 1 (even lanes keep v0 / v2, odd lanes v1 / v3), step 2 with lane ^ 2.
__device__ __forceinline__ float dpp_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));  // quad_perm:[1,0,3,2]
}
__device__ __forceinline__ float dpp_xor2(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));  // quad_perm:[2,3,0,1]
}
__device__ __forceinline__ float quad_transpose_sum(float v0, float v1, float v2, float v3, bool odd, bool hi) {
  const float r01 = (odd ? v1 : v0) + dpp_xor1(odd ? v0 : v1);
  const float r23 = (odd ? v3 : v2) + dpp_xor1(odd ? v2 : v3);
  return (hi ? r23 : r01) + dpp_xor2(hi ? r01 : r23);
}

__device__ __forceinline__ float row_sum15(float v) {  // lane 15 of each 16-lane row holds the row sum
  v = dpp_add(v, 0x111, 0xF);
  v = dpp_add(v, 0x112, 0xF);
  v = dpp_add(v, 0x114, 0xF);
  v = dpp_add(v, 0x118, 0xF);
  return v;
}
struct AccCol {
  float* col;   // this lane group's LDS column
  bool leader;  // lane that owns the column
  bool quad;    // 64 columns (quad groups) or 16 columns (row groups)
};
__device__ __forceinline__ AccCol acc_col(float* s_acc, int Apad, int ncols, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  AccCol c;
  c.quad = ncols == 64;
  const int col = c.quad ? (wave * 16 + (lane >> 2)) : (wave * 4 + (lane >> 4));
  c.leader = c.quad ? ((lane & 3) == 0) : ((lane & 15) == 15);
  c.col = s_acc + col * Apad;
  return c;
}
// n <= G live accumulators (n is wave-uniform)
template <int G> __device__ __forceinline__ void wave_acc(float (&acc)[G], const AccCol& c, int off, int n = G) {
  if (c.quad) {
#pragma unroll
    for (int k = 0; k < G; ++k)
      if (k < n) acc[k] = quad_sum(acc[k]);
  } else {
#pragma unroll
    for (int k = 0; k < G; ++k)
      if (k < n) acc[k] = row_sum15(acc[k]);
  }
  if (c.leader) {
    float* dst = c.col + off;
#pragma unroll
    for (int k = 0; k < G; ++k)
      if (k < n) dst[k] += acc[k];
  }
}

#ifdef GL_AUX_KERNELS  // the non-template kernels around the main one: compiled into the main translation unit only
// ---- per-sample prep: raw parameter rows -> derived constants --------------------------------
// `cost` (optional): per-sample dispatch cost = the EPL trip count, written by the thread of component `cost_comp`
// (models with exactly one EPL), so that gl_order_kernel reads one coalesced int array
__global__ void __launch_bounds__(128) gl_prep_kernel(const CompDesc* __restrict__ comps, int n_comp,
                                                      const float* __restrict__ params, int P, int B,
                                                      float* __restrict__ derived, int D, int* __restrict__ cost,
                                                      int cost_comp) {
  int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= B * n_comp) return;
  int b = i / n_comp, c = i - b * n_comp;
  CompDesc cd = comps[c];
  const float* p = params + (size_t)b * P + cd.p_off;
  float* d = derived + (size_t)b * D + cd.d_off;
  switch (cd.kind) {
    case K_EPL: epl_prep<float>(p, cd.iparam, d); break;
    case K_SIE: sie_prep<float>(p, d); break;
    case K_NFW: nfw_prep<float>(p, d); break;
    case K_SHEAR: shear_prep<float>(p, d); break;
    case K_SIS: sis_prep<float>(p, d); break;
    case K_DPIS: case K_DPIE: case K_DPIEP: dpie_prep<float>(cd.kind, p, d); break;
    case K_SCALED: d[0] = d[1] = d[2] = d[3] = 0.f; break;
    case K_SERIES: d[0] = p[0]; d[1] = p[1]; d[2] = d[3] = 0.f; break;
    case K_NFW_ELLIPSE: nfw_ell_prep<float>(p, d); break;
    case K_TNFW: tnfw_prep<float>(p, d); break;
    case K_CORE_SERSIC: core_sersic_prep<float>(p, d); break;
    case K_SERSIC: sersic_prep<float>(p, false, d); break;
    case K_SERSIC_ELLIPSE: sersic_prep<float>(p, true, d); break;
    case K_SHAPELETS: shapelets_prep<float>(p, cd.iparam, d); break;
    case K_USER_MASS: case K_USER_LIGHT: for (int k = 0; k < cd.iparam; ++k) d[k] = p[k]; break;  // the body reads its parameters
  }
  if (cost && c == cost_comp) cost[b] = reinterpret_cast<const int*>(d)[EPL_KI];
}

// ---- unconstrained-space front/back end: bijector + prior fused into prep / finalize -------------
// One ZCol per column k of z (the k-th leaf of the prior in tf.nest.flatten order, tf/model.py:76-87).
// Default event-space bijectors and log-densities restate TFP's (Identity / Exp / Sigmoid(lo,hi);
// Normal / LogNormal / Uniform / TruncatedNormal) -- see gigalens_amd/prior.py for the same maths in torch.
struct ZCol {
  int param_col;  // destination column in the packed [B,P] row
  int bijector;   // 0 identity, 1 exp, 2 sigmoid(lo,hi)
  int prior;      // 0 normal, 1 lognormal, 2 uniform, 3 truncated normal
  float a, b, lo, hi, log_norm;
};

struct ZEval { float x, dxdz, logp_plus_fldj, dlogp_dx, dfldj_dz; };

__device__ __forceinline__ ZEval z_eval(const ZCol& c, float z) {
  ZEval o;
  float fldj;
  float lnx = 0.f;
  if (c.bijector == 0) {
    o.x = z; o.dxdz = 1.f; fldj = 0.f; o.dfldj_dz = 0.f;
  } else if (c.bijector == 1) {
    o.x = expf(z); o.dxdz = o.x; fldj = z; o.dfldj_dz = 1.f; lnx = z;
  } else {
    float sg = 1.f / (1.f + expf(-z));
    float w = c.hi - c.lo;
    o.x = c.lo + w * sg;
    o.dxdz = w * sg * (1.f - sg);
    // log(hi-lo) - softplus(-z) - softplus(z)
    float az = fabsf(z);
    fldj = logf(w) - az - 2.f * log1pf(expf(-az));
    o.dfldj_dz = 1.f - 2.f * sg;
  }
  const float half_log_2pi = 0.91893853320467274178f;
  float logp;
  if (c.prior == 0 || c.prior == 3) {
    float u = (o.x - c.a) / c.b;
    logp = -0.5f * u * u - logf(c.b) - half_log_2pi - (c.prior == 3 ? c.log_norm : 0.f);
    o.dlogp_dx = -u / c.b;
    if (c.prior == 3 && !(o.x >= c.lo && o.x <= c.hi)) { logp = -INFINITY; o.dlogp_dx = 0.f; }
  } else if (c.prior == 1) {
    if (c.bijector != 1) lnx = logf(o.x);
    float u = (lnx - c.a) / c.b;
    logp = -0.5f * u * u - logf(c.b) - half_log_2pi - lnx;
    o.dlogp_dx = (-u / c.b - 1.f) / o.x;
  } else {
    bool in = (o.x >= c.lo && o.x <= c.hi);
    logp = in ? -logf(c.hi - c.lo) : -INFINITY;
    o.dlogp_dx = 0.f;
  }
  o.logp_plus_fldj = logp + fldj;
  return o;
}

// the constrained value alone (the front end needs nothing else of z_eval)
__device__ __forceinline__ float z_eval_x(const ZCol& c, float z) {
  if (c.bijector == 0) return z;
  if (c.bijector == 1) return expf(z);
  return c.lo + (c.hi - c.lo) * (1.f / (1.f + expf(-z)));
}

// z [B,d] -> packed constrained rows [B,P] (also kept for finalize) -> derived constants
__global__ void __launch_bounds__(128) gl_zprep_kernel(const CompDesc* __restrict__ comps, int n_comp,
                                                       const float* __restrict__ z, int d_z,
                                                       const ZCol* __restrict__ zcols, const int* __restrict__ src,
                                                       const float* __restrict__ const_row, int P, int B,
                                                       float* __restrict__ params, float* __restrict__ derived, int D,
                                                       int* __restrict__ cost, int cost_comp) {
  int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= B * n_comp) return;
  int b = i / n_comp, c = i - b * n_comp;
  CompDesc cd = comps[c];
  float* p = params + (size_t)b * P + cd.p_off;
  for (int j = 0; j < cd.n_par; ++j) {
    int col = cd.p_off + j;
    int k = src[col];
    p[j] = (k >= 0) ? z_eval(zcols[k], z[(size_t)b * d_z + k]).x : const_row[col];
  }
  float* dd = derived + (size_t)b * D + cd.d_off;
  switch (cd.kind) {
    case K_EPL: epl_prep<float>(p, cd.iparam, dd); break;
    case K_SIE: sie_prep<float>(p, dd); break;
    case K_NFW: nfw_prep<float>(p, dd); break;
    case K_SHEAR: shear_prep<float>(p, dd); break;
    case K_SIS: sis_prep<float>(p, dd); break;
    case K_DPIS: case K_DPIE: case K_DPIEP: dpie_prep<float>(cd.kind, p, dd); break;
    case K_SCALED: dd[0] = dd[1] = dd[2] = dd[3] = 0.f; break;
    case K_SERIES: dd[0] = p[0]; dd[1] = p[1]; dd[2] = dd[3] = 0.f; break;
    case K_NFW_ELLIPSE: nfw_ell_prep<float>(p, dd); break;
    case K_TNFW: tnfw_prep<float>(p, dd); break;
    case K_CORE_SERSIC: core_sersic_prep<float>(p, dd); break;
    case K_SERSIC: sersic_prep<float>(p, false, dd); break;
    case K_SERSIC_ELLIPSE: sersic_prep<float>(p, true, dd); break;
    case K_SHAPELETS: shapelets_prep<float>(p, cd.iparam, dd); break;
    case K_USER_MASS: case K_USER_LIGHT: for (int k = 0; k < cd.iparam; ++k) dd[k] = p[k]; break;
  }
  if (cost && c == cost_comp) cost[b] = reinterpret_cast<const int*>(dd)[EPL_KI];
}

// ---- wave-per-sample front end (models with EPL lenses) --------------------------------------------------------------
// The thread-per-component kernels above leave the EPL coefficient table to ONE thread: ~15-50 dependent iterations with a
// division each, 7.5 us of latency in front of a 90 us main kernel.  Here one wavefront owns a sample: lane c does what the
// thread of component c does above except the table, then all 64 lanes build the table of each EPL lens -- lane n takes
// row n: its factors (one division), and the running products by an inclusive scan over the lanes.  The recurrence
//   (c, cf, ct) <- (c p, cf p + c r, ct p + c dp/dt)      [r = p / f = dp/df]
// is the product of matrices [[p,0,0],[r,p,0],[dpdt,0,p]], closed under (P, Qf, Qt) o (P', Qf', Qt') =
// (P P', Qf P' + P Qf', Qt P' + P Qt'): associative, so six shuffle steps replace the chain (no division by p or f:
// f = 0 and gamma = 1 stay finite exactly like the sequential form).
struct EplScan { float P, Qf, Qt; };
__device__ __forceinline__ EplScan epl_scan_combine(const EplScan& lo, const EplScan& hi) {  // rows of `lo` come first
  return EplScan{lo.P * hi.P, lo.Qf * hi.P + lo.P * hi.Qf, lo.Qt * hi.P + lo.P * hi.Qt};
}
__device__ __forceinline__ void epl_table_wave(float f, float two_mt, int K, float* __restrict__ tab, int lane) {
  EplScan carry{1.f, 0.f, 0.f};
  for (int base = 0; base <= K + 3; base += 64) {
    const int n = base + lane;
    EplScan v{1.f, 0.f, 0.f};  // row 0: c_0 = 1, derivatives 0
    if (n >= 1 && n <= K) {
      float r, pn, dpdt;
      epl_row_factors<float>(n, f, two_mt, r, pn, dpdt);
      v = EplScan{pn, r, dpdt};
    }
#pragma unroll
    for (int delta = 1; delta < 64; delta <<= 1) {
      EplScan u{__shfl_up(v.P, delta), __shfl_up(v.Qf, delta), __shfl_up(v.Qt, delta)};
      if (lane >= delta) v = epl_scan_combine(u, v);
    }
    v = epl_scan_combine(carry, v);
    if (n <= K) {
      reinterpret_cast<float4*>(tab)[n] = float4{v.P, (float)(2 * n + 1) * v.P, v.Qf, v.Qt};
    } else if (n <= K + 3) {
      reinterpret_cast<float4*>(tab)[n] = float4{0.f, 0.f, 0.f, 0.f};  // the four-row trips of the Clenshaw loop may start above K
    }
    carry = EplScan{__shfl(v.P, 63), __shfl(v.Qf, 63), __shfl(v.Qt, 63)};
  }
}

// counting sort of the samples on their cost (<= 255), heaviest first, by ONE workgroup of NT threads: LDS histogram, one
// wavefront's scan over the 256 bins in descending order (four bins per lane + a shuffle scan), scatter through the bins'
// running offsets.  cost_of(b) is evaluated twice per sample (no staging array).
// split_rank >= 0 (B <= 4 NT, NT = 256): the order inside the cost bin that straddles that rank is made the sample order instead
// of the order of arrival of the atomics, so WHICH samples have a rank below split_rank is the same on every launch
// (tail_plan: they are summed over other pixel chunks than the rest, and results stay bitwise reproducible).
template <int NT, class F>
__device__ __forceinline__ void gl_order_sort(F&& cost_of, int B, int* __restrict__ order, int split_rank = -1) {
  __shared__ int hist[256];
  __shared__ int offs[256];
  __shared__ int s_split, s_group[4 * NT / 64];
  const int tid = threadIdx.x;
  for (int i = tid; i < 256; i += NT) hist[i] = 0;
  if (tid == 0) s_split = -1;
  __syncthreads();
  // the first four samples of a thread stay in registers between the two passes (B <= 4 NT: all of them), their loads in flight together
  int mine[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = tid + i * NT;
    mine[i] = b < B ? max(0, min(cost_of(b), 255)) : 0;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (tid + i * NT < B) atomicAdd(&hist[mine[i]], 1);
  for (int b = tid + 4 * NT; b < B; b += NT) atomicAdd(&hist[max(0, min(cost_of(b), 255))], 1);
  __syncthreads();
  if (tid < 64) {
    const int top = 255 - 4 * tid;  // this lane's bins, heaviest first: top, top - 1, top - 2, top - 3
    const int h0 = hist[top], h1 = hist[top - 1], h2 = hist[top - 2], h3 = hist[top - 3];
    const int sum = h0 + h1 + h2 + h3;
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (tid >= d) incl += t;
    }
    const int excl = incl - sum;  // samples in strictly heavier bins of other lanes
    offs[top] = excl;
    offs[top - 1] = excl + h0;
    offs[top - 2] = excl + h0 + h1;
    offs[top - 3] = excl + h0 + h1 + h2;
  }
  __syncthreads();
  int sb = -1;
  if (split_rank >= 0 && NT == 256) {  // wave-uniform
    if (offs[tid] < split_rank && split_rank < offs[tid] + hist[tid]) s_split = tid;  // at most one bin
    __syncthreads();
    sb = s_split;
  }
  if (sb >= 0) {
    // stable ranks inside bin sb: samples 64 g .. 64 g + 63 are group g = 4 i + wavefront; per-group counts, then a prefix
    const int lane = tid & 63, wv = tid >> 6;
    int rk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned long long mk = __ballot(tid + i * NT < B && mine[i] == sb);
      rk[i] = __popcll(mk & ((1ull << lane) - 1ull));
      if (lane == 0) s_group[i * (NT / 64) + wv] = __popcll(mk);
    }
    __syncthreads();
    const int base = offs[sb];  // no atomic touches this bin's counter below
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (tid + i * NT < B && mine[i] == sb) {
        int pre = 0;
        for (int g = 0; g < i * (NT / 64) + wv; ++g) pre += s_group[g];
        order[base + pre + rk[i]] = tid + i * NT;
      }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (tid + i * NT < B && mine[i] != sb) order[atomicAdd(&offs[mine[i]], 1)] = tid + i * NT;
  for (int b = tid + 4 * NT; b < B; b += NT) order[atomicAdd(&offs[max(0, min(cost_of(b), 255))], 1)] = b;
}

// params != null: packed constrained rows in (gl_prep_kernel's job); else z -> params_out through the bijectors
// (gl_zprep_kernel's job).  n_comp <= 64.
__global__ void __launch_bounds__(256) gl_prep_wave_kernel(const CompDesc* __restrict__ comps, int n_comp,
                                                           const float* __restrict__ params_in, const float* __restrict__ z,
                                                           int d_z, const ZCol* __restrict__ zcols,
                                                           const int* __restrict__ src, const float* __restrict__ const_row,
                                                           int P, int B, float* __restrict__ params_out,
                                                           float* __restrict__ derived, int D, int* __restrict__ cost,
                                                           int cost_comp, int* __restrict__ order, int row_lds, int split_rank) {
  // Cost-ordered dispatch without a launch of its own: with `order` the grid carries ONE extra workgroup that sorts the samples
  // by the trip count of their EPL series while the others build the samples' constants.  It needs no result of theirs: the count
  // depends on (e1, e2) alone (epl_cost), which it takes from the parameter rows -- or, on the z path, through the two columns'
  // bijectors.  (A "last workgroup to arrive sorts" scheme was measured first: 256 device-scope atomics on one counter, 0.5 ms.)
  if (order && blockIdx.x == gridDim.x - 1) {
    const CompDesc ce = comps[cost_comp];
    const int c1 = ce.p_off + 2, c2 = ce.p_off + 3, cap = ce.iparam;
    // what is the same for every sample is fetched once: where e1 and e2 come from (a z column and its bijector, or a constant)
    int k1 = -1, k2 = -1;
    ZCol z1{}, z2{};
    float k1c = 0.f, k2c = 0.f;
    if (!params_in) {
      k1 = src[c1];
      k2 = src[c2];
      if (k1 >= 0) z1 = zcols[k1]; else k1c = const_row[c1];
      if (k2 >= 0) z2 = zcols[k2]; else k2c = const_row[c2];
    }
    gl_order_sort<256>([&](int b) {
      float e1, e2;
      if (params_in) {
        e1 = params_in[(size_t)b * P + c1];
        e2 = params_in[(size_t)b * P + c2];
      } else {
        e1 = k1 >= 0 ? z_eval_x(z1, z[(size_t)b * d_z + k1]) : k1c;
        e2 = k2 >= 0 ? z_eval_x(z2, z[(size_t)b * d_z + k2]) : k2c;
      }
      return epl_cost<float>(e1, e2, cap);
    }, B, order, split_rank);
    return;
  }
  const int b = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (b >= B) return;  // whole wavefronts leave together
  float f = 0.f, two_mt = 0.f;
  int K = 0;
  // The constrained row of the sample goes to params_out (the finalize kernel reads it) AND, with row_lds, into the wavefront's own
  // LDS row: the component lanes take their parameters from there instead of reading the global row back (a round trip through
  // the L2 behind a store), and the component descriptors are requested before the bijector loads, not after them -- the
  // front end is a chain of dependent memory round trips and nothing else (5.7 us at C2 with four of them in a row).
  extern __shared__ float s_rows[];  // [4 wavefronts][P], or nothing
  float* row = row_lds ? s_rows + (threadIdx.x >> 6) * P : nullptr;
  CompDesc cd{};
  if (lane < n_comp) cd = comps[lane];
  if (!params_in) {
    // z -> constrained row, one COLUMN per lane: the bijectors of a sample's columns are independent, so their loads
    // (column descriptor, z) and transcendentals overlap instead of forming one lane's chain of n_par dependent round trips
    float* po = params_out + (size_t)b * P;
    for (int k = lane; k < d_z; k += 64) {
      const ZCol zc = zcols[k];
      const float v = z_eval_x(zc, z[(size_t)b * d_z + k]);
      po[zc.param_col] = v;
      if (row) row[zc.param_col] = v;
    }
    for (int col = lane; col < P; col += 64)
      if (src[col] < 0) {
        const float v = const_row[col];
        po[col] = v;
        if (row) row[col] = v;
      }
    if (!row) __threadfence_block();  // the component lanes below read the global row back (same wavefront, same L1)
  } else if (row) {
    for (int col = lane; col < P; col += 64) row[col] = params_in[(size_t)b * P + col];
  }
  // a wavefront's LDS accesses execute in order: the fence only keeps the compiler from moving the reads above the writes
  if (row) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  const float* prow = row ? row : (params_in ? params_in : params_out) + (size_t)b * P;
  if (lane < n_comp) {
    const float* p = prow + cd.p_off;
    float* d = derived + (size_t)b * D + cd.d_off;
    switch (cd.kind) {
      case K_EPL: K = epl_prep_head<float>(p, cd.iparam, d, f, two_mt); break;
      case K_SIE: sie_prep<float>(p, d); break;
      case K_NFW: nfw_prep<float>(p, d); break;
      case K_SHEAR: shear_prep<float>(p, d); break;
      case K_SIS: sis_prep<float>(p, d); break;
      case K_DPIS: case K_DPIE: case K_DPIEP: dpie_prep<float>(cd.kind, p, d); break;
      case K_SCALED: d[0] = d[1] = d[2] = d[3] = 0.f; break;
      case K_SERIES: d[0] = p[0]; d[1] = p[1]; d[2] = d[3] = 0.f; break;
      case K_NFW_ELLIPSE: nfw_ell_prep<float>(p, d); break;
      case K_TNFW: tnfw_prep<float>(p, d); break;
      case K_CORE_SERSIC: core_sersic_prep<float>(p, d); break;
      case K_SERSIC: sersic_prep<float>(p, false, d); break;
      case K_SERSIC_ELLIPSE: sersic_prep<float>(p, true, d); break;
      case K_USER_MASS: case K_USER_LIGHT: for (int k = 0; k < cd.iparam; ++k) d[k] = p[k]; break;
      case K_SHAPELETS:  // the four constants here; the amplitude blocks below, by the whole wavefront
        d[SHP_CX] = p[1]; d[SHP_CY] = p[2]; d[SHP_IB] = 1.f / p[0]; d[SHP_NMAX] = (float)cd.iparam;
        break;
    }
    if (cost && lane == cost_comp) cost[b] = K;
  }
  for (int c = 0; c < n_comp; ++c) {  // wave-uniform: every lane joins the amplitude blocks of every shapelet component
    if (comps[c].kind != K_SHAPELETS) continue;  // (one lane copying 66 + 144 values one by one: 18.6 us of prep at C3)
    const CompDesc cs = comps[c];
    const float* p = prow + cs.p_off;
    float* d = derived + (size_t)b * D + cs.d_off;
    const int n_max = cs.iparam, L = sh_layers(n_max);
    const int tri = n_max > SH_CAP ? ((SH_MAXLB + 3) & ~3) : ((SH_MAXL + 3) & ~3);
    for (int i = lane; i < tri; i += 64) d[SHP_AMP + i] = i < L ? p[3 + i] : 0.f;
    if (n_max <= SH_CAP)
      for (int e = lane; e < SH_SQ * SH_SQ; e += 64) {
        const int n1 = e / SH_SQ, n2 = e - n1 * SH_SQ, n = n1 + n2;
        d[SHP_SQ + e] = n <= n_max ? p[3 + n * (n + 1) / 2 + n2] * (SH_K[n1] * SH_K[n2]) : 0.f;  // scaled for the monic basis of gl_shp.hip.h
      }
  }
  for (int c = 0; c < n_comp; ++c) {  // wave-uniform: every lane joins the table of every EPL lens
    if (comps[c].kind != K_EPL) continue;
    const float fc = __shfl(f, c), tc = __shfl(two_mt, c);
    const int Kc = __shfl(K, c);
    epl_table_wave(fc, tc, Kc, derived + (size_t)b * D + comps[c].d_off + EPL_TAB, lane);
  }
}

// per (sample, galaxy) constants of the catalogue members: radii, amplitude and the map to the scale gradients
__global__ void __launch_bounds__(128) gl_galprep_kernel(const CompDesc* __restrict__ comps,
                                                         const CatDev* __restrict__ cats, int n_cats,
                                                         const float* __restrict__ params, int P, int B,
                                                         const float* __restrict__ table,
                                                         const float* __restrict__ gal_static,
                                                         float* __restrict__ gal_dyn, int G) {
  int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= B * G) return;
  int b = i / G, g = i - b * G;
  int c = 0;
  while (c + 1 < n_cats && g >= cats[c + 1].g_off) ++c;
  const CatDev cat = cats[c];
  ScaledDesc sd{cat.base_kind, cat.n_gal, {cat.col[0], cat.col[1], cat.col[2]}};
  member_dyn(sd, table + (size_t)7 * g, gal_static + (size_t)g * DP_NS, params + (size_t)b * P + comps[cat.comp].p_off,
             gal_dyn + ((size_t)b * G + g) * GM_ND);
}

// cost-ordered dispatch: samples sorted by descending EPL trip count (the only data-dependent cost on the
// path), so the heaviest workgroups start first and the tail of the launch is filled with light ones.
constexpr int ORDER_WG = 1024;
__global__ void __launch_bounds__(ORDER_WG) gl_order_kernel(const CompDesc* __restrict__ comps, int n_lens,
                                                            const float* __restrict__ derived, int D, int B,
                                                            int* __restrict__ order, const int* __restrict__ cost_in) {
  // counting sort on the cost (<= 255), three barriers in all: LDS histogram, one wavefront's scan over the 256 bins in
  // descending order (four bins per lane + a shuffle scan), scatter through the bins' running offsets
  __shared__ int hist[256];
  __shared__ int offs[256];
  const int tid = threadIdx.x;
  if (tid < 256) hist[tid] = 0;
  __syncthreads();
  auto cost = [&](int b) {
    if (cost_in) return min(cost_in[b], 255);
    int k = 0;
    for (int l = 0; l < n_lens; ++l)
      if (comps[l].kind == K_EPL) k += reinterpret_cast<const int*>(derived + (size_t)b * D + comps[l].d_off)[EPL_KI];
    return min(k, 255);
  };
  const int c0 = tid < B ? cost(tid) : 0;  // the first sample of a thread stays in a register (B <= 1024: the only one)
  if (tid < B) atomicAdd(&hist[c0], 1);
  for (int b = tid + ORDER_WG; b < B; b += ORDER_WG) atomicAdd(&hist[cost(b)], 1);
  __syncthreads();
  if (tid < 64) {
    const int top = 255 - 4 * tid;  // this lane's bins, heaviest first: top, top - 1, top - 2, top - 3
    const int h0 = hist[top], h1 = hist[top - 1], h2 = hist[top - 2], h3 = hist[top - 3];
    const int sum = h0 + h1 + h2 + h3;
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (tid >= d) incl += t;
    }
    const int excl = incl - sum;  // samples in strictly heavier bins of other lanes
    offs[top] = excl;
    offs[top - 1] = excl + h0;
    offs[top - 2] = excl + h0 + h1;
    offs[top - 3] = excl + h0 + h1 + h2;
  }
  __syncthreads();
  if (tid < B) order[atomicAdd(&offs[c0], 1)] = tid;
  for (int b = tid + ORDER_WG; b < B; b += ORDER_WG) order[atomicAdd(&offs[cost(b)], 1)] = b;
}

#endif  // GL_AUX_KERNELS

// ---- T-pixel EPL: series loop outermost so one LDS table read serves T pixels -------------------
template <int T> __device__ __forceinline__ void epl_fwd_T(const float* d, const float (&x)[T], const float (&y)[T],
                                                            float (&bx)[T], float (&by)[T]) {
  float Cs[T], Ss[T], twoc[T], Ex[T], Ey[T], Px[T], Py[T], Ox[T], Oy[T], Rc[T];
  const float c = d[EPL_C], s = d[EPL_S], q = d[EPL_Q];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float dx = x[t] - d[EPL_CX], dy = y[t] - d[EPL_CY];
    float xr = dx * c + dy * s, yr = dy * c - dx * s;
    float X = q * xr;
    float R0 = sqrt_(X * X + yr * yr);
    bool pos = R0 > 0.f;
    float inv = pos ? rcp(R0) : 0.f;
    Cs[t] = pos ? X * inv : 1.f;
    Ss[t] = yr * inv;
    Rc[t] = clamp_(R0, 1e-10f, 1e10f);
    twoc[t] = 2.f * (Cs[t] * Cs[t] - Ss[t] * Ss[t]);  // E_{n+1} = 2 cos(2 theta) E_n - E_{n-1}, see epl_fwd_v
    Ex[t] = Cs[t]; Ey[t] = Ss[t]; Px[t] = Cs[t]; Py[t] = -Ss[t]; Ox[t] = Cs[t]; Oy[t] = Ss[t];
  }
  const int K = (int)d[EPL_K];
  const float* tab = d + EPL_TAB;
  int n = 1;
  for (; n + 1 <= K; n += 2) {
    const float ca = tab[4 * n], cb = tab[4 * n + 4];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      Px[t] = twoc[t] * Ex[t] - Px[t];
      Py[t] = twoc[t] * Ey[t] - Py[t];
      Ox[t] += ca * Px[t];
      Oy[t] += ca * Py[t];
      Ex[t] = twoc[t] * Px[t] - Ex[t];
      Ey[t] = twoc[t] * Py[t] - Ey[t];
      Ox[t] += cb * Ex[t];
      Oy[t] += cb * Ey[t];
    }
  }
  if (n <= K) {
    const float ca = tab[4 * n];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      Ox[t] += ca * (twoc[t] * Ex[t] - Px[t]);
      Oy[t] += ca * (twoc[t] * Ey[t] - Py[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float L2 = log2_(d[EPL_B] * rcp(Rc[t]));
    float P = d[EPL_P0] * exp2_(d[EPL_TM1] * L2);
    float arx = P * Ox[t], ary = P * Oy[t];
    bx[t] -= arx * c - ary * s;
    by[t] -= arx * s + ary * c;
  }
}

template <int T> __device__ __forceinline__ void epl_vjp_T(const float* d, const float (&x)[T], const float (&y)[T],
                                                            const float (&gx)[T], const float (&gy)[T], float* acc) {
  float Cs[T], Ss[T], twoc[T], Ex[T], Ey[T], Px[T], Py[T], Ox[T], Oy[T], Fx[T], Fy[T], Tx[T], Ty[T];
  float xr[T], yr[T], inv[T], R0[T];
  const float c = d[EPL_C], s = d[EPL_S], q = d[EPL_Q];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    float dx = x[t] - d[EPL_CX], dy = y[t] - d[EPL_CY];
    xr[t] = dx * c + dy * s;
    yr[t] = dy * c - dx * s;
    float X = q * xr[t];
    R0[t] = sqrt_(X * X + yr[t] * yr[t]);
    bool pos = R0[t] > 0.f;
    inv[t] = pos ? rcp(R0[t]) : 0.f;
    Cs[t] = pos ? X * inv[t] : 1.f;
    Ss[t] = yr[t] * inv[t];
    twoc[t] = 2.f * (Cs[t] * Cs[t] - Ss[t] * Ss[t]);
    Ex[t] = Cs[t]; Ey[t] = Ss[t]; Px[t] = Cs[t]; Py[t] = -Ss[t];
    Ox[t] = Cs[t]; Oy[t] = Ss[t];
    Fx[t] = 0.f; Fy[t] = 0.f; Tx[t] = 0.f; Ty[t] = 0.f;
  }
  const int K = (int)d[EPL_K];
  const float4* tab = reinterpret_cast<const float4*>(d + EPL_TAB);
  auto add = [&](const float4 cc, int t, float ex, float ey) {
    Ox[t] += cc.x * ex; Oy[t] += cc.x * ey;
    Fx[t] += cc.z * ex; Fy[t] += cc.z * ey;
    Tx[t] += cc.w * ex; Ty[t] += cc.w * ey;
  };
  int n = 1;
  for (; n + 1 <= K; n += 2) {
    const float4 ca = tab[n], cb = tab[n + 1];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      Px[t] = twoc[t] * Ex[t] - Px[t];
      Py[t] = twoc[t] * Ey[t] - Py[t];
      add(ca, t, Px[t], Py[t]);
      Ex[t] = twoc[t] * Px[t] - Ex[t];
      Ey[t] = twoc[t] * Py[t] - Ey[t];
      add(cb, t, Ex[t], Ey[t]);
    }
  }
  if (n <= K) {
    const float4 ca = tab[n];
#pragma unroll
    for (int t = 0; t < T; ++t) add(ca, t, twoc[t] * Ex[t] - Px[t], twoc[t] * Ey[t] - Py[t]);
  }
  const float tm1 = d[EPL_TM1], P0 = d[EPL_P0];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    bool inclamp = (R0[t] >= 1e-10f) && (R0[t] <= 1e10f);
    float iRc = rcp(clamp_(R0[t], 1e-10f, 1e10f));
    float L2 = log2_(d[EPL_B] * iRc);
    float W = exp2_(tm1 * L2);
    float P = P0 * W;
    float arx = P * Ox[t], ary = P * Oy[t];
    float ax = arx * c - ary * s, ay = arx * s + ary * c;
    float grx = gx[t] * c + gy[t] * s, gry = gy[t] * c - gx[t] * s;
    float g_phi = gy[t] * ax - gx[t] * ay;
    float gP = grx * Ox[t] + gry * Oy[t];
    float gOx = P * grx, gOy = P * gry;
    // S = sum (2n+1) c_n E_n = Omega + 2 f dOmega/df
    float g_ang = (gOy * Ox[t] - gOx * Oy[t]) + d[EPL_F2] * (gOy * Fx[t] - gOx * Fy[t]);
    float g_t = gOx * Tx[t] + gOy * Ty[t];
    float g_f = gOx * Fx[t] + gOy * Fy[t];
    float gW_W = gP * P;
    g_t += gW_W * (L2 * (float)kLn2);
    float g_b = gW_W * tm1 * d[EPL_INVB];
    float gR0 = inclamp ? -gW_W * tm1 * iRc : 0.f;
    float gX = gR0 * Cs[t] - g_ang * Ss[t] * inv[t];
    float gyr = gR0 * Ss[t] + g_ang * Cs[t] * inv[t];
    float g_q = gX * xr[t];
    float gxr = gX * q;
    float gdx = gxr * c - gyr * s, gdy = gxr * s + gyr * c;
    g_phi += gxr * yr[t] - gyr * xr[t];
    acc[EPLA_CX] -= gdx;
    acc[EPLA_CY] -= gdy;
    acc[EPLA_PHI] += g_phi;
    acc[EPLA_Q] += g_q;
    acc[EPLA_B] += g_b;
    acc[EPLA_T] += g_t;
    acc[EPLA_F] += g_f;
    acc[EPLA_P0] += gP * W;
  }
}

// ---- the main kernel ----------------------------------------------------------------------------
// SHP / FAM: the model contains shapelets / lenses of the extended families -- their code (and register budget) is
// compiled only into the variants that need it, so the common compositions keep their occupancy.
//   FAM 0: EPL, SIE, NFW, Shear, SIS | Sersic[Ellipse]     FAM 1: + dPIS / dPIE / dPIEP, catalogues, series lenses
//   FAM 2: + NFW_ELLIPSE, TNFW, CoreSersic (gl_extra.h; the fp64 core of TNFW alone costs ~80 VGPRs)
//   BIG: some shapelet component has n_max above SH_CAP (up to SH_CAPB = 20): every shapelet component of the model runs the
//   runtime-order code of gl_profiles.h on the model's wide table, amplitude gradients leave shell by shell into the LDS columns
template <int MODE, int T, bool SHP, int FAM, bool BIG = false>
__global__ void __launch_bounds__(WG, (SHP || FAM) ? 2 : 4) gl_main_kernel(MainArgs a) {
  constexpr bool DP = FAM >= 1, XF = FAM >= 2;
  static_assert(!BIG || SHP, "BIG is a shapelet variant");
  extern __shared__ float smem[];
  float* s_d = smem;
  float* s_acc = smem + ((a.D + 3) & ~3);
  float* s_nfw = s_acc + a.ncols * a.Apad;  // models with NFW lenses: the shared h(X) table (8-byte aligned: ncols is even)
  const int tid = threadIdx.x;
  const int b = a.order ? a.order[blockIdx.y] : blockIdx.y, chunk = blockIdx.x;
  const CompDesc* __restrict__ comps = a.comps;
  {
    const float* src = a.derived + (size_t)b * a.D;
    for (int i = tid; i < a.D; i += WG) s_d[i] = src[i];
    if (a.nfw_tab)
      for (int i = tid; i < 2 * NFW_TAB_NODES; i += WG) s_nfw[i] = a.nfw_tab[i];
    if (MODE != IMG_FWD && MODE != IMG_BASIS)
      for (int i = tid; i < a.ncols * a.Apad; i += WG) s_acc[i] = 0.f;
  }
  __syncthreads();
  const AccCol ac = acc_col(s_acc, a.Apad, a.ncols, tid);
  const int n_lens = a.n_lens, n_light = a.n_ll + a.n_src, n_ll = a.n_ll;
  const int p0 = chunk * a.chunk;
  const int p1 = min(p0 + a.chunk, a.N);
  float st[2] = {0.f, 0.f};  // chi2, norm

  for (int base = p0; base < p1; base += WG * T) {
    float x[T], y[T], bx[T], by[T], m[T];
    int pidx[T];
    bool valid[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      int j = base + t * WG + tid;
      valid[t] = j < p1;
      int jj = valid[t] ? j : p1 - 1;
      x[t] = a.gx[jj];
      y[t] = a.gy[jj];
      pidx[t] = a.pix ? a.pix[jj] : jj;
      bx[t] = x[t]; by[t] = y[t]; m[t] = 0.f;
    }
    constexpr bool GRADMODE = (MODE == IMG_BWD || MODE == LL_GRAD);
    ScaleTan<v2f> tans[DP ? (T + 1) / 2 : 1];
    if constexpr (DP && GRADMODE) {
#pragma unroll
      for (int t = 0; t < (T + 1) / 2; ++t) tans[t] = ScaleTan<v2f>{v2f(0.f), v2f(0.f), v2f(0.f), v2f(0.f), v2f(0.f), v2f(0.f)};
    }
    // ---- phase 1: ray-shoot  beta = (x,y) - sum_i alpha_i(x,y)   (tf/simulator.py:72-78) ----
    for (int l = 0; l < ((a.parts & 1u) ? n_lens : 0); ++l) {
      const int kind = comps[l].kind;
      const float* d = s_d + comps[l].d_off;
      switch (kind) {
        case K_EPL: epl_fwd_T<T>(d, x, y, bx, by); break;
        case K_SIE:
#pragma unroll
          for (int t = 0; t < T; ++t) { float ax, ay; sie_fwd(d, x[t], y[t], ax, ay); bx[t] -= ax; by[t] -= ay; }
          break;
        case K_NFW:
          if constexpr (T % 2 == 0) {  // pixel pairs -> packed fp32
#pragma unroll
            for (int t = 0; t < T; t += 2) {
              v2f vbx{bx[t], bx[t + 1]}, vby{by[t], by[t + 1]};
              nfw_fwd_v(d, s_nfw, v2f{x[t], x[t + 1]}, v2f{y[t], y[t + 1]}, vbx, vby);
              bx[t] = vbx.x; bx[t + 1] = vbx.y; by[t] = vby.x; by[t + 1] = vby.y;
            }
          } else {
#pragma unroll
            for (int t = 0; t < T; ++t) { float ax, ay; nfw_fwd(d, x[t], y[t], ax, ay); bx[t] -= ax; by[t] -= ay; }
          }
          break;
        case K_SHEAR:
#pragma unroll
          for (int t = 0; t < T; ++t) { float ax, ay; shear_fwd(d, x[t], y[t], ax, ay); bx[t] -= ax; by[t] -= ay; }
          break;
        case K_SIS:
#pragma unroll
          for (int t = 0; t < T; ++t) { float ax, ay; sis_fwd(d, x[t], y[t], ax, ay); bx[t] -= ax; by[t] -= ay; }
          break;
#ifdef GL_HAVE_USER  // run-time compiled variant of this kernel (gl_user.hip): the bodies of the model's user-written profiles
        case K_USER_MASS:
#pragma unroll
          for (int t = 0; t < T; ++t) { float ax, ay; glu::mass_fwd(comps[l].flags, d, x[t], y[t], ax, ay); bx[t] -= ax; by[t] -= ay; }
          break;
#endif
        case K_DPIE:
          if constexpr (DP) {
#pragma unroll
            for (int t = 0; t < T; ++t) { float ax, ay; piemd_fwd<float>(d, d + DP_NS, x[t], y[t], ax, ay); bx[t] -= ax; by[t] -= ay; }
          }
          break;
        case K_DPIS:
        case K_DPIEP:
          if constexpr (DP) {
#pragma unroll
            for (int t = 0; t < T; ++t) { float ax, ay; piep_fwd<float>(d, d + DP_NS, x[t], y[t], ax, ay); bx[t] -= ax; by[t] -= ay; }
          }
          break;
        case K_NFW_ELLIPSE:
          if constexpr (XF) {
#pragma unroll
            for (int t = 0; t < T; ++t) { float ax, ay; nfw_ell_fwd<float>(d, x[t], y[t], ax, ay); bx[t] -= ax; by[t] -= ay; }
          }
          break;
        case K_TNFW:
          if constexpr (XF) {
#pragma unroll
            for (int t = 0; t < T; ++t) { float ax, ay; tnfw_fwd<float>(d, x[t], y[t], ax, ay); bx[t] -= ax; by[t] -= ay; }
          }
          break;
        case K_SERIES: if constexpr (DP) {  // alpha = theta_E sum_n C_n(pixel) (r_cut - r0)^n   (series_profile.py:76-95)
          const SeriesDev sv = a.series[comps[l].flags];
          const float te = d[0], dl = d[1] - sv.r0;
          const size_t stn = (size_t)a.N;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const int j = base + t * WG + tid;
            const float* c = sv.coef + (j < p1 ? j : p1 - 1);
            float ax = c[(size_t)sv.order * stn], ay = c[(size_t)(2 * sv.order + 1) * stn];
            for (int n = sv.order - 1; n >= 0; --n) {
              ax = ax * dl + c[(size_t)n * stn];
              ay = ay * dl + c[(size_t)(sv.order + 1 + n) * stn];
            }
            bx[t] -= te * ax;
            by[t] -= te * ay;
          }
        } break;
        case K_SCALED: if constexpr (DP) {  // sum over the catalogue (scaling_relation.py:61-70), gl_members.hip.h
          const CatDev cat = a.cats[comps[l].iparam];
          const float* __restrict__ gs = a.gal_static + (size_t)cat.g_off * DP_NS;
          const float* __restrict__ gm = a.gal_dyn + ((size_t)b * a.G + cat.g_off) * GM_ND;
          const bool carry = GRADMODE && l == a.scaled_first;
#pragma unroll
          for (int t = 0; t < T; t += 2) {
            v2f vbx{bx[t], bx[t + 1]}, vby{by[t], by[t + 1]};
            const v2f vx{x[t], x[t + 1]}, vy{y[t], y[t + 1]};
            if (carry) members_v<v2f, true>(cat.base_kind, cat.n_gal, gs, gm, vx, vy, vbx, vby, tans[t / 2]);
            else members_v<v2f, false>(cat.base_kind, cat.n_gal, gs, gm, vx, vy, vbx, vby, tans[t / 2]);
            bx[t] = vbx.x; bx[t + 1] = vbx.y; by[t] = vby.x; by[t + 1] = vby.y;
          }
        } break;
      }
    }
    if constexpr (MODE == IMG_BASIS) {
      // lstsq_simulate (tf/simulator.py:183-201): every linear component as its own image, amplitude 1, NaN -> 0
      for (int ci = 0; ci < n_light; ++ci) {
        const CompDesc& cd = comps[n_lens + ci];
        const float* d = s_d + cd.d_off;
        const bool src = ci >= n_ll;
        float* row = a.img + ((size_t)b * a.n_lin + cd.lin_off) * a.img_stride;
        if (cd.kind == K_SHAPELETS) {
          if constexpr (SHP) {
#pragma unroll 1
            for (int t = 0; t < T; ++t) {
              if (!valid[t]) continue;
              const int pi = pidx[t];
              const long long st = a.img_stride;
              shapelets_basis<float, BIG ? SH_CAPB : SH_CAP>(d, a.shp_tab, a.shp_stride, cd.flags & 1u, src ? bx[t] : x[t],
                                             src ? by[t] : y[t], [&](int k, float v) { row[(size_t)k * st + pi] = isnan_(v) ? 0.f : v; });
            }
          }
        } else {
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const float px_ = src ? bx[t] : x[t], py_ = src ? by[t] : y[t];
            float v;
            if (XF && cd.kind == K_CORE_SERSIC) v = core_sersic_fwd<float>(d, px_, py_);
#ifdef GL_HAVE_USER
            else if (cd.kind == K_USER_LIGHT) v = glu::light_fwd(cd.flags, d, px_, py_);  // (its amplitude column holds 1, like a Sersic's)
#endif
            else v = sersic_fwd(d, px_, py_);
            if (valid[t]) row[pidx[t]] = isnan_(v) ? 0.f : v;
          }
        }
      }
      continue;
    }
    // ---- phase 2: render lens light at the grid, sources at beta (tf/simulator.py:128-138) ----
    for (int ci = 0; ci < n_light; ++ci) {
      const CompDesc& cd = comps[n_lens + ci];
      const float* d = s_d + cd.d_off;
      const bool src = ci >= n_ll;
      if (!(a.parts & (src ? 4u : 2u))) continue;
      if (cd.kind == K_SHAPELETS) {
        if (SHP) {
          const bool interp = cd.flags & 1u;
          const float* __restrict__ gamp = a.derived + (size_t)b * a.D + cd.d_off + SHP_AMP;
#pragma unroll 1
          for (int t = 0; t < T; ++t) {
            if constexpr (BIG) {
              m[t] += shapelets_fwd<float, SH_CAPB>(d, a.shp_tab, a.shp_stride, interp, src ? bx[t] : x[t], src ? by[t] : y[t]);
            } else {
              ShpState<SH_CAP> hs;
              m[t] += shp_fwd_state<SH_CAP>(d, gamp, a.shp_tab, interp, src ? bx[t] : x[t], src ? by[t] : y[t], hs);
            }
          }
        }
      } else if (cd.kind == K_CORE_SERSIC) {
        if constexpr (XF) {
#pragma unroll
          for (int t = 0; t < T; ++t) m[t] += core_sersic_fwd<float>(d, src ? bx[t] : x[t], src ? by[t] : y[t]);
        }
#ifdef GL_HAVE_USER
      } else if (cd.kind == K_USER_LIGHT) {
#pragma unroll
        for (int t = 0; t < T; ++t) m[t] += glu::light_fwd(cd.flags, d, src ? bx[t] : x[t], src ? by[t] : y[t]);
#endif
      } else if constexpr (T % 2 == 0) {
#pragma unroll
        for (int t = 0; t < T; t += 2) {
          SerStateV<v2f> stv;
          v2f I = sersic_fwd_v<v2f>(d, src ? v2f{bx[t], bx[t + 1]} : v2f{x[t], x[t + 1]},
                                    src ? v2f{by[t], by[t + 1]} : v2f{y[t], y[t + 1]}, stv);
          m[t] += I.x;
          m[t + 1] += I.y;
        }
      } else {
#pragma unroll
        for (int t = 0; t < T; ++t) m[t] += sersic_fwd(d, src ? bx[t] : x[t], src ? by[t] : y[t]);
      }
    }
    bool nanp[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      nanp[t] = isnan_(m[t]);
      m[t] = (nanp[t] ? 0.f : m[t]) * a.out_scale;  // NaN -> 0 (tf/simulator.py:140), then x det(T) (:156)
    }
    if (MODE == IMG_FWD) {
      float* row = a.img + (size_t)b * a.img_stride;
#pragma unroll
      for (int t = 0; t < T; ++t)
        if (valid[t]) row[pidx[t]] = m[t];
      continue;
    }
    float gm[T];
    if (MODE == IMG_BWD) {
      const float* row = a.gimg + (size_t)b * a.img_stride;
#pragma unroll
      for (int t = 0; t < T; ++t) gm[t] = (valid[t] && !nanp[t]) ? row[pidx[t]] * a.out_scale : 0.f;
    } else {
      const bool has_err = a.err != nullptr;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        float o = a.obs[pidx[t]];
        float w = a.mask ? a.mask[pidx[t]] : 1.f;
        float e = has_err ? a.err[pidx[t]] : 1.f;
        float c2, nm;
        chi2_terms(m[t], o, w, has_err, e, a.bg2, a.inv_t, c2, nm);
        if (valid[t]) { st[0] += c2; st[1] += nm; }
        if (MODE == LL_GRAD)
          gm[t] = (valid[t] && !nanp[t]) ? chi2_gm(m[t], o, w, has_err, e, a.bg2, a.inv_t) * a.out_scale : 0.f;
      }
    }
    if (MODE == IMG_BWD || MODE == LL_GRAD) {
      float gbx[T], gby[T];
#pragma unroll
      for (int t = 0; t < T; ++t) { gbx[t] = 0.f; gby[t] = 0.f; }
      // ---- phase 3: light VJPs -> parameter gradients and the cotangent of beta ----
      for (int ci = 0; ci < n_light; ++ci) {
        const CompDesc& cd = comps[n_lens + ci];
        const float* d = s_d + cd.d_off;
        const bool src = ci >= n_ll;
        if (cd.kind == K_SHAPELETS) {
          if constexpr (SHP && BIG) {
            // runtime-order path: one pixel at a time, every shell n = n1 + n2 of the amplitude gradient reduced over the lane
            // group and added to the LDS columns as soon as it is formed (at most SH_CAPB + 1 values live)
            const bool interp = cd.flags & 1u;
            float acc3[SHPA_AMP] = {0.f, 0.f, 0.f};
#pragma unroll 1
            for (int t = 0; t < T; ++t) {
              float dgx = 0.f, dgy = 0.f;
              (void)shapelets_vjp_shells<float, SH_CAPB>(d, a.shp_tab, a.shp_stride, interp, src ? bx[t] : x[t], src ? by[t] : y[t],
                                                         gm[t], acc3, dgx, dgy, [&](int first, int count, float* vals) {
                float tmp[SH_CAPB + 1];
#pragma unroll
                for (int k = 0; k <= SH_CAPB; ++k) tmp[k] = vals[k];
                wave_acc<SH_CAPB + 1>(tmp, ac, cd.a_off + SHPA_AMP + first, count);
              });
              if (src) { gbx[t] += dgx; gby[t] += dgy; }
            }
            wave_acc<SHPA_AMP>(acc3, ac, cd.a_off, SHPA_AMP);
          } else if (SHP) {
            const bool interp = cd.flags & 1u;
            float acc[SHPA_AMP + SH_MAXL];
#pragma unroll
            for (int k = 0; k < SHPA_AMP + SH_MAXL; ++k) acc[k] = 0.f;
            const float* __restrict__ gamp = a.derived + (size_t)b * a.D + cd.d_off + SHP_AMP;
#pragma unroll 1
            for (int t = 0; t < T; ++t) {
              float dgx = 0.f, dgy = 0.f;
              ShpState<SH_CAP> hs;  // bases re-evaluated (the interpreter keeps no per-component state), contractions separable
              (void)shp_fwd_state<SH_CAP>(d, gamp, a.shp_tab, interp, src ? bx[t] : x[t], src ? by[t] : y[t], hs);
              shp_vjp_state<SH_CAP>(d, gamp, interp, hs, gm[t], acc, dgx, dgy);
              if (src) { gbx[t] += dgx; gby[t] += dgy; }
            }
            wave_acc<SHPA_AMP + SH_MAXL>(acc, ac, cd.a_off, cd.n_acc);
          }
#ifdef GL_HAVE_USER
        } else if (cd.kind == K_USER_LIGHT) {  // d I / d (x, y, p) from forward-mode duals of the user's body
          float acc[USER_MAXP];
#pragma unroll
          for (int k = 0; k < USER_MAXP; ++k) acc[k] = 0.f;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            float dgx = 0.f, dgy = 0.f;
            glu::light_vjp(cd.flags, d, src ? bx[t] : x[t], src ? by[t] : y[t], gm[t], acc, dgx, dgy);
            if (src) { gbx[t] += dgx; gby[t] += dgy; }
          }
          wave_acc<USER_MAXP>(acc, ac, cd.a_off, cd.n_acc);
#endif
        } else if (cd.kind == K_CORE_SERSIC) {
          if constexpr (XF) {
            float acc[CSR_NACC];
#pragma unroll
            for (int k = 0; k < CSR_NACC; ++k) acc[k] = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
              float dgx = 0.f, dgy = 0.f;
              core_sersic_vjp<float>(d, src ? bx[t] : x[t], src ? by[t] : y[t], gm[t], acc, dgx, dgy);
              if (src) { gbx[t] += dgx; gby[t] += dgy; }
            }
            wave_acc<CSR_NACC>(acc, ac, cd.a_off);
          }
        } else {
          float acc[SER_NACC];
          if constexpr (T % 2 == 0) {
            v2f va[SER_NACC];
#pragma unroll
            for (int k = 0; k < SER_NACC; ++k) va[k] = v2f(0.f);
#pragma unroll
            for (int t = 0; t < T; t += 2) {
              SerStateV<v2f> stv;
              (void)sersic_fwd_v<v2f>(d, src ? v2f{bx[t], bx[t + 1]} : v2f{x[t], x[t + 1]},
                                      src ? v2f{by[t], by[t + 1]} : v2f{y[t], y[t + 1]}, stv);
              v2f dgx(0.f), dgy(0.f);
              sersic_vjp_v<v2f, true>(d, stv, v2f{gm[t], gm[t + 1]}, va, dgx, dgy);
              if (src) { gbx[t] += dgx.x; gbx[t + 1] += dgx.y; gby[t] += dgy.x; gby[t + 1] += dgy.y; }
            }
#pragma unroll
            for (int k = 0; k < SER_NACC; ++k) acc[k] = va[k].x + va[k].y;
            acc[SERA_INVN] *= (float)kLn2;  // sersic_vjp_v defers this factor
          } else {
#pragma unroll
            for (int k = 0; k < SER_NACC; ++k) acc[k] = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
              float dgx = 0.f, dgy = 0.f;
              sersic_vjp(d, src ? bx[t] : x[t], src ? by[t] : y[t], gm[t], acc, dgx, dgy);
              if (src) { gbx[t] += dgx; gby[t] += dgy; }
            }
          }
          wave_acc<SER_NACC>(acc, ac, cd.a_off);
        }
      }
      // ---- phase 4: lens VJPs with cotangent -g_beta (beta = x - sum alpha) ----
#pragma unroll
      for (int t = 0; t < T; ++t) { gbx[t] = -gbx[t]; gby[t] = -gby[t]; }
      for (int l = 0; l < n_lens; ++l) {
        const CompDesc& cd = comps[l];
        const float* d = s_d + cd.d_off;
        switch (cd.kind) {
          case K_EPL: {
            float acc[EPL_NACC];
#pragma unroll
            for (int k = 0; k < EPL_NACC; ++k) acc[k] = 0.f;
            epl_vjp_T<T>(d, x, y, gbx, gby, acc);
            wave_acc<EPL_NACC>(acc, ac, cd.a_off);
          } break;
          case K_SIE: {
            float acc[SIE_NACC];
#pragma unroll
            for (int k = 0; k < SIE_NACC; ++k) acc[k] = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) sie_vjp(d, x[t], y[t], gbx[t], gby[t], acc);
            wave_acc<SIE_NACC>(acc, ac, cd.a_off);
          } break;
          case K_NFW: {
            float acc[NFW_NACC];
            if constexpr (T % 2 == 0) {
              v2f va[NFW_NACC];
#pragma unroll
              for (int k = 0; k < NFW_NACC; ++k) va[k] = v2f(0.f);
#pragma unroll
              for (int t = 0; t < T; t += 2)
                nfw_vjp_v(d, s_nfw, v2f{x[t], x[t + 1]}, v2f{y[t], y[t + 1]}, v2f{gbx[t], gbx[t + 1]}, v2f{gby[t], gby[t + 1]}, va);
#pragma unroll
              for (int k = 0; k < NFW_NACC; ++k) acc[k] = va[k].x + va[k].y;
            } else {
#pragma unroll
              for (int k = 0; k < NFW_NACC; ++k) acc[k] = 0.f;
#pragma unroll
              for (int t = 0; t < T; ++t) nfw_vjp(d, x[t], y[t], gbx[t], gby[t], acc);
            }
            wave_acc<NFW_NACC>(acc, ac, cd.a_off);
          } break;
          case K_SHEAR: {
            float acc[SHR_NACC];
            acc[0] = 0.f; acc[1] = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) shear_vjp(d, x[t], y[t], gbx[t], gby[t], acc);
            wave_acc<SHR_NACC>(acc, ac, cd.a_off);
          } break;
          case K_SIS: {
            float acc[SIS_NACC];
            acc[0] = 0.f; acc[1] = 0.f; acc[2] = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) sis_vjp(d, x[t], y[t], gbx[t], gby[t], acc);
            wave_acc<SIS_NACC>(acc, ac, cd.a_off);
          } break;
#ifdef GL_HAVE_USER
          case K_USER_MASS: {
            float acc[USER_MAXP];
#pragma unroll
            for (int k = 0; k < USER_MAXP; ++k) acc[k] = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) glu::mass_vjp(cd.flags, d, x[t], y[t], gbx[t], gby[t], acc);
            wave_acc<USER_MAXP>(acc, ac, cd.a_off, cd.n_acc);
          } break;
#endif
          case K_DPIS:
          case K_DPIE:
          case K_DPIEP: if constexpr (DP) {
            float acc[DP_NACC];
#pragma unroll
            for (int k = 0; k < DP_NACC; ++k) acc[k] = 0.f;
            if (cd.kind == K_DPIE) {
#pragma unroll
              for (int t = 0; t < T; ++t) piemd_vjp<float, true>(d, d + DP_NS, d + DPX_DE, x[t], y[t], gbx[t], gby[t], acc);
            } else {
#pragma unroll
              for (int t = 0; t < T; ++t) piep_vjp<float, true>(d, d + DP_NS, x[t], y[t], gbx[t], gby[t], acc);
            }
            wave_acc<DP_NACC>(acc, ac, cd.a_off);
          } break;
          case K_NFW_ELLIPSE: if constexpr (XF) {
            float acc[NFE_NACC];
#pragma unroll
            for (int k = 0; k < NFE_NACC; ++k) acc[k] = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) nfw_ell_vjp<float>(d, x[t], y[t], gbx[t], gby[t], acc);
            wave_acc<NFE_NACC>(acc, ac, cd.a_off);
          } break;
          case K_TNFW: if constexpr (XF) {
            float acc[TNF_NACC];
#pragma unroll
            for (int k = 0; k < TNF_NACC; ++k) acc[k] = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) tnfw_vjp<float>(d, x[t], y[t], gbx[t], gby[t], acc);
            wave_acc<TNF_NACC>(acc, ac, cd.a_off);
          } break;
          case K_SERIES: if constexpr (DP) {
            const SeriesDev sv = a.series[cd.flags];
            const float te = d[0], dl = d[1] - sv.r0;
            const size_t stn = (size_t)a.N;
            float acc[2] = {0.f, 0.f};
#pragma unroll
            for (int t = 0; t < T; ++t) {
              const int j = base + t * WG + tid;
              const float* c = sv.coef + (j < p1 ? j : p1 - 1);
              float ax = c[(size_t)sv.order * stn], ay = c[(size_t)(2 * sv.order + 1) * stn], dx = 0.f, dy = 0.f;
              for (int n = sv.order - 1; n >= 0; --n) {  // Horner with derivative
                dx = dx * dl + ax;
                dy = dy * dl + ay;
                ax = ax * dl + c[(size_t)n * stn];
                ay = ay * dl + c[(size_t)(sv.order + 1 + n) * stn];
              }
              acc[0] += gbx[t] * ax + gby[t] * ay;
              acc[1] += te * (gbx[t] * dx + gby[t] * dy);
            }
            wave_acc<2>(acc, ac, cd.a_off);
          } break;
          case K_SCALED: if constexpr (DP) {
            // gradient of the scales = cotangent . (tangents carried by the ray-shooting pass); a second catalogue in
            // the same model re-evaluates its members here
            float sc[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < T; t += 2) {
              ScaleTan<v2f> tn = tans[t / 2];
              if (l != a.scaled_first) {
                const CatDev cat = a.cats[cd.iparam];
                tn = ScaleTan<v2f>{v2f(0.f), v2f(0.f), v2f(0.f), v2f(0.f), v2f(0.f), v2f(0.f)};
                v2f dbx(0.f), dby(0.f);
                members_v<v2f, true>(cat.base_kind, cat.n_gal, a.gal_static + (size_t)cat.g_off * DP_NS,
                                     a.gal_dyn + ((size_t)b * a.G + cat.g_off) * GM_ND, v2f{x[t], x[t + 1]},
                                     v2f{y[t], y[t + 1]}, dbx, dby, tn);
              }
              const v2f gx{gbx[t], gbx[t + 1]}, gy{gby[t], gby[t + 1]};
              sc[0] += hsum(gx * tn.tx + gy * tn.ty);
              sc[1] += hsum(gx * tn.cx + gy * tn.cy);
              sc[2] += hsum(gx * tn.ux + gy * tn.uy);
            }
            wave_acc<3>(sc, ac, cd.a_off);
          } break;
        }
      }
    }
  }
  if (MODE == IMG_FWD || MODE == IMG_BASIS) return;
  if (MODE == LL_FWD || MODE == LL_GRAD) wave_acc<2>(st, ac, 0);
  __syncthreads();
  float* out = a.partial + ((size_t)b * gridDim.x + chunk) * a.A;
  for (int k = tid; k < a.A; k += WG) {
    float v = 0.f;
    for (int j = 0; j < a.ncols; ++j) v += s_acc[j * a.Apad + k];
    out[k] = v;
  }
}

#ifdef GL_AUX_KERNELS
// ---- finalize: sum chunk partials (fixed order), chain rule to raw parameters -------------------
// With zcols != null the gradient is carried on to the unconstrained vector z and the log-prior
// + log|J| is added:  log_prob = loglike + sum_k [log p_k(x_k) + fldj_k(z_k)]   (tf/model.py:164-167).

// one sample, executed by NT threads of one workgroup; `s`: LDS scratch of A + P + d_z (+12) floats
// BASIC: only EPL / SIE / Shear / SIS / Sersic in the model -- the other families' chain rules (TNFW's float64 core, shapelets,
// dPIE, ...) stay out of the kernel: a third of the code, and the finalize launch is instruction-fetch bound
template <int NT, bool BASIC = false>
__device__ __forceinline__ void finalize_sample(const CompDesc* __restrict__ comps, const FinArgs& f,
                                                const float* __restrict__ partial, int n_chunks, int b, int tid,
                                                float* s) {
  const int A = f.A, P = f.P, d_z = f.d_z;
  float* s_g = s + ((A + 3) & ~3);
  float* s_t = s_g + ((P + 3) & ~3);
  float* s_e = s_t + ((d_z + 3) & ~3);  // [4][d_z]: dlogp/dx, dx/dz, dfldj/dz, parameter column of every column of z
  const float* src = partial + (size_t)b * n_chunks * A;
  // Phase 0 -- three independent jobs on three groups of threads, so their global round trips and transcendentals overlap
  // instead of queueing behind two barriers: (a) sum the chunk partials, (b) bijector / prior terms of z (they do not depend
  // on the accumulators), (c) fetch the component descriptors
  for (int k = tid; k < A; k += NT) {
    float v = 0.f;
    if (f.use_partial)
      for (int ch = 0; ch < n_chunks; ++ch) v += src[(size_t)ch * A + k];
    if (f.extra_stats && k < 2) v += f.extra_stats[2 * b + k];  // chi2 / normalisation of a materialised image (PSF path)
    s[k] = v;
  }
  constexpr int ZT0 = NT / 2;  // the upper half of the workgroup serves the columns of z
  if (f.zcols && tid >= ZT0) {
    for (int k = tid - ZT0; k < d_z; k += NT - ZT0) {
      const ZCol c = f.zcols[k];
      const ZEval e = z_eval(c, f.z[(size_t)b * d_z + k]);
      s_t[k] = e.logp_plus_fldj;
      s_e[k] = e.dlogp_dx;
      s_e[d_z + k] = e.dxdz;
      s_e[2 * d_z + k] = e.dfldj_dz;
      s_e[3 * d_z + k] = __int_as_float(c.param_col);
    }
  }
  const bool want_grad = f.grad != nullptr || f.grad_z != nullptr;
  CompDesc cd{};
  // the sample's parameter row goes to LDS whole, requested together with everything else (the components' own slices would
  // be a round trip that can only start once their descriptors have arrived)
  float* s_p = s_e + 4 * d_z + 4;
  if (want_grad) {
    for (int k = tid; k < P; k += NT) s_p[k] = f.params[(size_t)b * P + k];
    if (tid < f.n_comp) cd = comps[tid];
  }
  __syncthreads();
  if (want_grad) {
    for (int c = tid; c < f.n_comp; c += NT) {
      if (c != tid) cd = comps[c];
      const float* p = s_p + cd.p_off;
      float* g = s_g + cd.p_off;
      const float* acc = s + cd.a_off;
      switch (cd.kind) {
        case K_EPL: epl_finalize<float>(p, acc, g); break;
        case K_SIE: sie_finalize<float>(p, acc, g); break;
        case K_NFW: if constexpr (!BASIC) nfw_finalize<float>(p, acc, g); break;
        case K_SHEAR: shear_finalize<float>(p, acc, g); break;
        case K_SIS: sis_finalize<float>(p, acc, g); break;
        case K_DPIS: case K_DPIE: case K_DPIEP: if constexpr (!BASIC) dpie_finalize<float>(cd.kind, p, acc, g); break;
        case K_SCALED:
          if constexpr (!BASIC) {
            const CatDev cat = f.cats[cd.iparam];
            for (int k = 0; k < 3; ++k)
              if (cat.col[k] >= 0) g[cat.col[k]] = acc[k];
          }
          break;
        case K_SERIES: if constexpr (!BASIC) { g[0] = acc[0]; g[1] = acc[1]; } break;
        case K_NFW_ELLIPSE: if constexpr (!BASIC) nfw_ell_finalize<float>(p, acc, g); break;
        case K_TNFW: if constexpr (!BASIC) tnfw_finalize<float>(p, acc, g); break;
        case K_CORE_SERSIC: if constexpr (!BASIC) core_sersic_finalize<float>(p, acc, g); break;
        case K_SERSIC: sersic_finalize<float>(p, false, acc, g); break;
        case K_SERSIC_ELLIPSE: sersic_finalize<float>(p, true, acc, g); break;
        case K_SHAPELETS: if constexpr (!BASIC) shapelets_finalize<float>(p, cd.iparam, acc, g); break;
        case K_USER_MASS: case K_USER_LIGHT: if constexpr (!BASIC) { for (int k = 0; k < cd.iparam; ++k) g[k] = acc[k]; } break;  // d/dp_k as summed
      }
    }
    __syncthreads();
    if (f.pos_grad) {  // image-position likelihood: its parameter gradient joins before the chain to z
      for (int k = tid; k < P; k += NT) s_g[k] += f.pos_grad[(size_t)b * P + k];
      __syncthreads();
    }
    // A NaN log-likelihood (sigma^2 = bg^2 + model / t < 0 somewhere: tf/model.py:96 takes its square root) has a NaN gradient
    // in the reference -- the square root's derivative is NaN there and NaN x 0 stays NaN through every pixel sum -- while the
    // cotangent the kernels form (from 1 / sigma^2) stays finite: the whole row follows the reference.
    const float poison = (s[0] + s[1]) != (s[0] + s[1]) ? __int_as_float(0x7fc00000) : 0.f;
    if (f.grad)
      for (int k = tid; k < P; k += NT) f.grad[(size_t)b * P + k] = s_g[k] + poison;
    if (f.zcols && f.grad_z && tid >= ZT0)
      for (int k = tid - ZT0; k < d_z; k += NT - ZT0)
        f.grad_z[(size_t)b * d_z + k] = (s_g[__float_as_int(s_e[3 * d_z + k])] + s_e[k]) * s_e[d_z + k] + s_e[2 * d_z + k] + poison;
  }
  if (tid == 0 && f.loglike) {
    float ll = -0.5f * (s[0] + s[1]);  // tf/model.py:99
    float c2 = s[0] * f.chi2_scale;
    if (f.pos_ll) {  // tf/model.py:157-162
      ll += f.pos_ll[b];
      c2 += f.pos_chi2[b] * f.pos_chi2_scale;
    }
    f.loglike[b] = ll;
    f.chi2[b] = c2;
    if (f.zcols && f.logprob) {
      float lp = 0.f;
      for (int k = 0; k < d_z; ++k) lp += s_t[k];
      f.logprob[b] = ll + lp;
    }
  }
}

template <bool BASIC>
__global__ void __launch_bounds__(128) gl_finalize_kernel(const CompDesc* __restrict__ comps, FinArgs f,
                                                          const float* __restrict__ partial, int n_chunks) {
  extern __shared__ float s[];  // [A] accumulators, [P] parameter gradients, [d_z] prior terms, [4][d_z] bijector / prior derivatives, [P] parameters
#ifdef GL_EXPERIMENTS
  if (n_chunks < 0) return;  // GIGALENS_HIP_DBGFLAGS & 8: the cost of the bare launch (results undefined)
#endif
  finalize_sample<128, BASIC>(comps, f, partial, n_chunks, blockIdx.x, threadIdx.x, s);
}

// Fused form (specialised kernels, likelihood modes): the LAST workgroup of a sample to publish its partial row runs
// the sample's finalize in its own tail -- one kernel launch less on the critical path of every step.
// (Fusing this into the tail of each sample's last main-kernel workgroup was measured and dropped: with one L2 per XCD
// the hand-over of the partial rows needs agent-scope release / L2-bypassing traffic per workgroup, which cost more
// (0.172 ms per step) than the separate 5 us launch (0.137 ms).)

// ---- the optimiser update of the MAP / SVI loops (tf/inference.py:33-39 hands the gradient to a Keras Adam) ----------
// One launch instead of ~10 elementwise ones: x -= lr * (m / c1) / (sqrt(v / c2) + eps) with m, v updated in place,
// grad scaled by grad_scale first; c1 = 1 - b1^t, c2 = 1 - b2^t with t from the host or, inside a captured graph,
// from a device counter that thread 0 of block 0 advances AFTER every block has read it (it is read at kernel start
// and written only by the last block to finish, see the ticket).
__global__ void __launch_bounds__(256) gl_adam_kernel(float* __restrict__ x, const float* __restrict__ grad,
                                                      float* __restrict__ m, float* __restrict__ v, long long n,
                                                      float grad_scale, float lr, float b1, float b2, float eps,
                                                      double t_host, double* __restrict__ t_dev,
                                                      unsigned* __restrict__ ticket, float c1_host, float c2_host) {
  // the bias corrections 1 - beta^t: from the host when it knows the step count (two double-precision pow per THREAD were most
  // of this kernel's 4.5 us), on the device only under graph replay, where the count lives in t_dev
  const double t = (t_dev ? t_dev[0] : t_host) + (t_dev ? 1.0 : 0.0);
  float c1 = c1_host, c2 = c2_host;
  if (t_dev) {
    c1 = (float)(1.0 - ::pow((double)b1, t));
    c2 = (float)(1.0 - ::pow((double)b2, t));
  }
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float g = grad[i] * grad_scale;
    const float mi = m[i] * b1 + g * (1.0f - b1);
    const float vi = v[i] * b2 + (g * g) * (1.0f - b2);
    m[i] = mi;
    v[i] = vi;
    x[i] -= lr * (mi / c1) / (sqrtf(vi / c2) + eps);
  }
  if (t_dev) {  // advance the device counter once per launch, after the last reader
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned done = atomicAdd(ticket, 1u);
      if (done == gridDim.x - 1) {
        t_dev[0] = t;
        *ticket = 0u;
      }
    }
  }
}

// ---- the Gaussian surrogate of the SVI loop (tf/inference.py:64-91; jax/inference.py:98-128) ---------------------------
// q(z) = N(mu, L L^T) with L = FillScaleTriL(diag_bijector=Exp, diag_shift) over the row-major lower-triangle packing
// (full rank) or L = diag(exp(p)) (mean field).  Two small launches bracket the native forward+gradient call:
//   gl_svi_sample_kernel  z_i = mu + L eps_i
//   gl_svi_grad_kernel    the fused collective buffer  [ELBO, dELBO/dmu (d), dELBO/dp (packed)]  from eps, log p(z_i) and
//                         G_i = d log p / d z_i:  dELBO/dmu = -mean G,  dELBO/dL_jk = -mean G_ij eps_ik (k <= j), Exp diagonal
//                         and -log det L of log q in closed form.  One workgroup per output, fixed-order reduction over i.
__device__ __forceinline__ void tril_jk(int t, int& j, int& k) {
  j = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
  while ((j + 1) * (j + 2) / 2 <= t) ++j;
  while (j * (j + 1) / 2 > t) --j;
  k = t - j * (j + 1) / 2;
}

__global__ void __launch_bounds__(256) gl_svi_sample_kernel(const float* __restrict__ mu, const float* __restrict__ lp,
                                                            int d, int full_rank, const float* __restrict__ eps, int n,
                                                            float diag_shift, float* __restrict__ z) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n * d) return;
  const int r = (int)(i / d), j = (int)(i - (long long)r * d);
  const float* e = eps + (size_t)r * d;
  float v = mu[j];
  if (full_rank) {
    const float* row = lp + j * (j + 1) / 2;
    for (int k = 0; k < j; ++k) v += row[k] * e[k];
    v += (expf(row[j]) + diag_shift) * e[j];
  } else {
    v += expf(lp[j]) * e[j];
  }
  z[i] = v;
}

__global__ void __launch_bounds__(256) gl_svi_grad_kernel(const float* __restrict__ lp, int d, int full_rank,
                                                          const float* __restrict__ eps, const float* __restrict__ logp,
                                                          const float* __restrict__ G, int n, float diag_shift,
                                                          float* __restrict__ buf) {
  __shared__ float red[4];
  const int o = blockIdx.x, tid = threadIdx.x;
  int j = 0, k = 0;
  const int kind = o == 0 ? 0 : (o <= d ? 1 : 2);  // ELBO, d/dmu_j, d/dp_t
  if (kind == 1) j = o - 1;
  if (kind == 2) {
    if (full_rank) tril_jk(o - 1 - d, j, k);
    else j = k = o - 1 - d;
  }
  float acc = 0.f;
  for (int i = tid; i < n; i += 256) {
    if (kind == 0) {
      const float* e = eps + (size_t)i * d;
      float q = 0.f;
      for (int c = 0; c < d; ++c) q += e[c] * e[c];
      acc += -0.5f * q - logp[i];
    } else if (kind == 1) {
      acc -= G[(size_t)i * d + j];
    } else {
      acc -= G[(size_t)i * d + j] * eps[(size_t)i * d + k];
    }
  }
  acc = wave_sum63(acc);
  if ((tid & 63) == 63) red[tid >> 6] = acc;
  __syncthreads();
  if (tid != 0) return;
  float v = (red[0] + red[1] + red[2] + red[3]) / (float)n;
  if (kind == 0) {
    float log_det = 0.f;
    for (int c = 0; c < d; ++c) log_det += full_rank ? logf(expf(lp[c * (c + 1) / 2 + c]) + diag_shift) : lp[c];
    v += -log_det - 0.5f * (float)d * 1.8378770664093453f;  // log 2 pi
  } else if (kind == 2 && j == k) {
    const float p = lp[full_rank ? j * (j + 1) / 2 + j : j];
    const float e = expf(p);
    v = full_rank ? v * e - e / (e + diag_shift) : v * e - 1.f;
  }
  buf[o] = v;
}

// ---- the leapfrog of the preconditioned HMC loop (tf/inference.py:95-182) ----------------------------------------------
// Momentum precision = the surrogate covariance Sigma = L L^T, so a drift is z += eps * (p Sigma).  One launch does the
// momentum kick that precedes a drift and the drift itself; one launch closes a transition: last half kick, kinetic
// energies 1/2 |p L|^2, Metropolis test against the supplied uniforms, and the in-place selection of the state.
// One workgroup per chain, any d (cluster models: d = 132): the chain's momentum row is staged in LDS, thread j owns
// column j, so the rows of Sigma / L are read coalesced and every output element is read and written by the same
// thread (the calls may run in place: p_out == p_in, z_out == z_in).
constexpr int HMC_WG = 128;

__global__ void __launch_bounds__(HMC_WG) gl_hmc_kick_drift_kernel(const float* p_in, const float* __restrict__ grad,
                                                                   float kick, const float* z_in,
                                                                   const float* __restrict__ sigma, float eps, int n, int d,
                                                                   float* p_out, float* z_out) {
  extern __shared__ float s_p[];  // [d]
  const int i = blockIdx.x, tid = threadIdx.x;
  const size_t row = (size_t)i * d;
  for (int j = tid; j < d; j += HMC_WG) {
    const float v = p_in[row + j] + kick * grad[row + j];
    s_p[j] = v;
    p_out[row + j] = v;
  }
  __syncthreads();
  for (int j = tid; j < d; j += HMC_WG) {
    float s = 0.f;
    for (int k = 0; k < d; ++k) s += s_p[k] * sigma[(size_t)k * d + j];
    z_out[row + j] = z_in[row + j] + eps * s;
  }
}

__global__ void __launch_bounds__(HMC_WG) gl_hmc_accept_kernel(float* z, float* g, float* lp, const float* __restrict__ zn,
                                                               const float* __restrict__ gn, const float* __restrict__ lpn,
                                                               const float* __restrict__ p0, const float* __restrict__ pn,
                                                               float kick, const float* __restrict__ L,
                                                               const float* __restrict__ u, int n, int d,
                                                               float* __restrict__ acc_prob) {
  extern __shared__ float s_ab[];  // [2][d] momenta at both ends, then [4] reduction slots
  float* s_a = s_ab;
  float* s_b = s_ab + d;
  __shared__ float red[2][HMC_WG / 64];
  __shared__ int s_take;
  const int i = blockIdx.x, tid = threadIdx.x;
  const size_t row = (size_t)i * d;
  for (int j = tid; j < d; j += HMC_WG) {
    s_a[j] = p0[row + j];
    s_b[j] = pn[row + j] + kick * gn[row + j];
  }
  __syncthreads();
  float ke0 = 0.f, ke1 = 0.f;
  for (int k = tid; k < d; k += HMC_WG) {  // (p L)_k = sum_{j >= k} p_j L_jk, L lower triangular
    float s0 = 0.f, s1 = 0.f;
    for (int j = k; j < d; ++j) {
      const float l = L[(size_t)j * d + k];
      s0 += s_a[j] * l;
      s1 += s_b[j] * l;
    }
    ke0 += s0 * s0;
    ke1 += s1 * s1;
  }
  ke0 = wave_sum63(ke0);
  ke1 = wave_sum63(ke1);
  if ((tid & 63) == 63) { red[0][tid >> 6] = ke0; red[1][tid >> 6] = ke1; }
  __syncthreads();
  if (tid == 0) {
    float k0 = 0.f, k1 = 0.f;
    for (int w = 0; w < HMC_WG / 64; ++w) { k0 += red[0][w]; k1 += red[1][w]; }
    float log_acc = (lpn[i] - 0.5f * k1) - (lp[i] - 0.5f * k0);
    if (!(fabsf(log_acc) <= 3.0e38f)) log_acc = -INFINITY;  // NaN / inf proposals are rejected
    acc_prob[i] = expf(fminf(log_acc, 0.f));
    const int take = logf(u[i]) < log_acc;
    s_take = take;
    if (take) lp[i] = lpn[i];
  }
  __syncthreads();
  if (s_take)
    for (int j = tid; j < d; j += HMC_WG) {
      z[row + j] = zn[row + j];
      g[row + j] = gn[row + j];
    }
}

// ---- plugin-level point evaluation (MassProfile.deriv / LightProfile.light on arbitrary points) ----
__global__ void __launch_bounds__(256) gl_point_kernel(CompDesc cd, const float* __restrict__ x,
                                                       const float* __restrict__ y, long long n_pts, int B,
                                                       int xy_batched, const float* __restrict__ params,
                                                       float* __restrict__ out0, float* __restrict__ out1,
                                                       const float* __restrict__ shp_tab, int shp_stride) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pts * B) return;
  long long pt = i / B;
  int b = (int)(i - pt * B);
  float px = xy_batched ? x[i] : x[pt], py = xy_batched ? y[i] : y[pt];
  const float* p = params + (size_t)b * cd.n_par;
  float o0 = 0.f, o1 = 0.f;
  switch (cd.kind) {
    case K_EPL: epl_point<float>(p, cd.iparam, px, py, o0, o1); break;
    case K_SIE: { float d[SIE_ND + 1]; sie_prep<float>(p, d); sie_fwd(d, px, py, o0, o1); } break;
    case K_NFW: { float d[NFW_ND]; nfw_prep<float>(p, d); nfw_fwd(d, px, py, o0, o1); } break;
    case K_SHEAR: { float d[4]; shear_prep<float>(p, d); shear_fwd(d, px, py, o0, o1); } break;
    case K_SIS: { float d[4]; sis_prep<float>(p, d); sis_fwd(d, px, py, o0, o1); } break;
    case K_DPIS: case K_DPIE: case K_DPIEP: { float d[DPX_ND]; dpie_prep<float>(cd.kind, p, d); dpie_fwd<float>(cd.kind, d, px, py, o0, o1); } break;
    case K_NFW_ELLIPSE: { float d[NFE_ND]; nfw_ell_prep<float>(p, d); nfw_ell_fwd<float>(d, px, py, o0, o1); } break;
    case K_TNFW: { float d[TNF_ND]; tnfw_prep<float>(p, d); tnfw_fwd<float>(d, px, py, o0, o1); } break;
    case K_CORE_SERSIC: { float d[CSR_ND]; core_sersic_prep<float>(p, d); o0 = core_sersic_fwd<float>(d, px, py); } break;
    case K_SERSIC: { float d[SER_NDX]; sersic_prep<float>(p, false, d); o0 = sersic_fwd(d, px, py); } break;
    case K_SERSIC_ELLIPSE: { float d[SER_NDX]; sersic_prep<float>(p, true, d); o0 = sersic_fwd(d, px, py); } break;
    case K_SHAPELETS: {
      if (cd.iparam > SH_CAP) {  // runtime-order path; amplitudes straight from the parameter row (same triangle order)
        float d[SHP_AMP];
        d[SHP_CX] = p[1]; d[SHP_CY] = p[2]; d[SHP_IB] = 1.f / p[0]; d[SHP_NMAX] = (float)cd.iparam;
        o0 = shapelets_fwd_amp<float, SH_CAPB>(d, p + 3, shp_tab, shp_stride, cd.flags & 1u, px, py);
      } else {
        float d[SHP_SQ + SH_SQ * SH_SQ];
        shapelets_prep<float>(p, cd.iparam, d);
        o0 = shapelets_fwd<float, SH_CAP>(d, shp_tab, shp_stride, cd.flags & 1u, px, py);
      }
    } break;
  }
  out0[i] = o0;
  if (out1) out1[i] = o1;
}

// LightProfile.light of a use_lstsq profile at plugin level: the unit-amplitude basis images (sersic.py:30-34
// `Ie = ones`, `ret[tf.newaxis]`; shapelets.py:61-62,71-72), out[depth][n_pts][B]; amplitude columns are not read
__global__ void __launch_bounds__(256) gl_basis_point_kernel(CompDesc cd, const float* __restrict__ x,
                                                             const float* __restrict__ y, long long n_pts, int B,
                                                             int xy_batched, const float* __restrict__ params,
                                                             float* __restrict__ out,
                                                             const float* __restrict__ shp_tab, int shp_stride) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = n_pts * B;
  if (i >= total) return;
  long long pt = i / B;
  int b = (int)(i - pt * B);
  float px = xy_batched ? x[i] : x[pt], py = xy_batched ? y[i] : y[pt];
  const float* p = params + (size_t)b * cd.n_par;
  if (cd.kind == K_SHAPELETS) {
    float d[SHP_AMP];
    d[SHP_CX] = p[1];
    d[SHP_CY] = p[2];
    d[SHP_IB] = 1.f / p[0];
    d[SHP_NMAX] = (float)cd.iparam;
    if (cd.iparam > SH_CAP)
      shapelets_basis<float, SH_CAPB>(d, shp_tab, shp_stride, cd.flags & 1u, px, py,
                                      [&](int k, float v) { out[(size_t)k * total + i] = v; });
    else
      shapelets_basis<float, SH_CAP>(d, shp_tab, shp_stride, cd.flags & 1u, px, py,
                                     [&](int k, float v) { out[(size_t)k * total + i] = v; });
    return;
  }
  float q[10];
  for (int k = 0; k < cd.n_par; ++k) q[k] = p[k];
  q[kind_linear_col(cd.kind, cd.iparam)] = 1.f;
  float v = 0.f;
  switch (cd.kind) {
    case K_CORE_SERSIC: { float d[CSR_ND]; core_sersic_prep<float>(q, d); v = core_sersic_fwd<float>(d, px, py); } break;
    case K_SERSIC: { float d[SER_NDX]; sersic_prep<float>(q, false, d); v = sersic_fwd(d, px, py); } break;
    case K_SERSIC_ELLIPSE: { float d[SER_NDX]; sersic_prep<float>(q, true, d); v = sersic_fwd(d, px, py); } break;
  }
  out[i] = v;
}

// ScalingRelation.deriv on arbitrary points (scaling_relation.py:61-70)
__global__ void __launch_bounds__(256) gl_scaled_point_kernel(ScaledDesc sd, const float* __restrict__ table,
                                                              const float* __restrict__ x, const float* __restrict__ y,
                                                              long long n_pts, int B, int xy_batched,
                                                              const float* __restrict__ scales, int n_scales,
                                                              float* __restrict__ out0, float* __restrict__ out1) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pts * B) return;
  long long pt = i / B;
  int b = (int)(i - pt * B);
  float px = xy_batched ? x[i] : x[pt], py = xy_batched ? y[i] : y[pt];
  const float* sc = scales + (size_t)b * n_scales;
  float sx = 0.f, sy = 0.f;
  for (int g = 0; g < sd.n_gal; ++g) {
    float ds[DP_NS], dd[DP_ND], ax, ay;
    scaled_static<float>(sd.base_kind, table + (size_t)7 * g, ds);
    scaled_dyn<float>(sd, table + (size_t)7 * g, sc, dd);
    if (sd.base_kind == K_DPIE) piemd_fwd<float>(ds, dd, px, py, ax, ay);
    else piep_fwd<float>(ds, dd, px, py, ax, ay);
    sx += ax;
    sy += ay;
  }
  out0[i] = sx;
  out1[i] = sy;
}

// Taylor coefficients of the population deflection at arbitrary points: coeffs[2][order+1][n_pts]
template <int N>
__global__ void __launch_bounds__(64) gl_series_precompute_kernel(ScaledDesc sd, const float* __restrict__ table,
                                                                 float s0, float s1, float s2, int order,
                                                                 const float* __restrict__ x, const float* __restrict__ y,
                                                                 long long n_pts, float* __restrict__ coeffs) {
  const long long i = (long long)blockIdx.x * 64 + threadIdx.x;
  if (i >= n_pts) return;
  // float64 jets: near a member's foci (removable 0/0 of the Kassiola-Kovner form) the k-th coefficient loses
  // ~(1/distance)^k digits -- in fp32 orders >= 2 are noise at ~1 % of the pixels; the one-off precompute can afford
  // CDNA4's full-rate fp64, the stored field is fp32 like every other operand of the path
  const double scales[3] = {(double)s0, (double)s1, (double)s2};
  double cx[N + 1], cy[N + 1];
  series_point<N, double>(sd, table, scales, (double)x[i], (double)y[i], cx, cy);
  for (int n = 0; n <= order; ++n) {
    coeffs[(size_t)n * n_pts + i] = (float)cx[n];
    coeffs[(size_t)(order + 1 + n) * n_pts + i] = (float)cy[n];
  }
}

// Taylor coefficients of the population Hessian: coeffs[3][order+1][n_pts] = f_xx, f_xy, f_yy (one-off, fp64 jets)
template <int N>
__global__ void __launch_bounds__(64) gl_series_hessian_precompute_kernel(ScaledDesc sd, const float* __restrict__ table,
                                                                         float s0, float s1, float s2, int order,
                                                                         const float* __restrict__ x,
                                                                         const float* __restrict__ y, long long n_pts,
                                                                         float* __restrict__ coeffs) {
  const long long i = (long long)blockIdx.x * 64 + threadIdx.x;
  if (i >= n_pts) return;
  const double scales[3] = {(double)s0, (double)s1, (double)s2};
  double hxx[N + 1], hxy[N + 1], hyy[N + 1];
  series_point_hessian<N, double>(sd, table, scales, (double)x[i], (double)y[i], hxx, hxy, hyy);
  for (int n = 0; n <= order; ++n) {
    coeffs[(size_t)n * n_pts + i] = (float)hxx[n];
    coeffs[(size_t)(order + 1 + n) * n_pts + i] = (float)hxy[n];
    coeffs[(size_t)(2 * (order + 1) + n) * n_pts + i] = (float)hyy[n];
  }
}

// theta_E[b] * sum_n coeffs[f][n][pt] (r_cut[b] - r0)^n for n_fields fields: out[n_fields][n_pts][B]
__global__ void __launch_bounds__(256) gl_series_fields_kernel(const float* __restrict__ coeffs, int n_fields, int order,
                                                               long long n_pts, int B,
                                                               const float* __restrict__ theta_E,
                                                               const float* __restrict__ r_cut, float r0,
                                                               float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pts * B) return;
  const long long pt = i / B;
  const int b = (int)(i - pt * B);
  const float dl = r_cut[b] - r0, te = theta_E[b];
  for (int f = 0; f < n_fields; ++f) {
    const float* c = coeffs + (size_t)f * (order + 1) * n_pts + pt;
    float v = c[(size_t)order * n_pts];
    for (int n = order - 1; n >= 0; --n) v = v * dl + c[(size_t)n * n_pts];
    out[(size_t)f * n_pts * B + i] = te * v;
  }
}

__global__ void __launch_bounds__(256) gl_series_eval_kernel(const float* __restrict__ coeffs, int order, long long n_pts,
                                                             int B, const float* __restrict__ theta_E,
                                                             const float* __restrict__ r_cut, float r0,
                                                             float* __restrict__ out0, float* __restrict__ out1) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pts * B) return;
  const long long pt = i / B;
  const int b = (int)(i - pt * B);
  const float dl = r_cut[b] - r0;
  float ax = coeffs[(size_t)order * n_pts + pt], ay = coeffs[(size_t)(2 * order + 1) * n_pts + pt];
  for (int n = order - 1; n >= 0; --n) {
    ax = ax * dl + coeffs[(size_t)n * n_pts + pt];
    ay = ay * dl + coeffs[(size_t)(order + 1 + n) * n_pts + pt];
  }
  out0[i] = theta_E[b] * ax;
  out1[i] = theta_E[b] * ay;
}

#endif  // GL_AUX_KERNELS

}  // namespace glk
