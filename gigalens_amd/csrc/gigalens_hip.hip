// gigalens_hip.hip -- C ABI (include/gigalens_hip.h) over the kernels in gl_kernels.hip.h.
// gfx950 only.  No per-call allocation, no host synchronisation: every entry point enqueues on
// the caller's stream and returns (hipGraph-capturable).
#include "../../include/gigalens_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#define GL_AUX_KERNELS 1
#include "gl_host_tables.h"
#include "gl_model.h"
#include "gl_static.hip.h"
#include "gl_post.hip.h"
#include "gl_positions.hip.h"
#include "gl_lstsq.hip.h"
#include "gl_shp.hip.h"

using namespace glk;

namespace {
thread_local char g_err[512] = "";
}

namespace glk {
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace glk

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

}  // namespace


namespace {

// number of pixel chunks per sample: enough workgroups to fill 256 CUs several times over
// (dynamic load balance: EPL trip counts differ per sample), but whole tiles per chunk.
void chunking(const gl_model* m, int B, int* chunk, int* n_chunks) {
  const long long tile_px = (long long)WG * 4;  // whole tiles for every T in {1, 2, 4}
  // chunks per sample: enough workgroups for one resident round of the chip (768 = 256 CUs x 3), and at most ~8192 pixels
  // (16 pair tiles) per workgroup so the tail of the launch stays short -- measured at B = 64 .. 1024, 60^2 .. 256^2 px:
  // never behind the older "2048 workgroups" rule, 2-5 % ahead of it at small batches.  GIGALENS_HIP_TARGET_WGS restores that rule.
  long long want = std::max<long long>((768 + B - 1) / B, ((long long)m->N + 8191) / 8192);
  // (the specialised compositions only: with hundreds to thousands of instructions per pixel -- interpreter and cluster
  // kernels -- a workgroup's fixed costs vanish and more, smaller workgroups balance better: C6 4.83 vs 4.98 ms)
  if (m->target_wgs_set || !m->static_id) want = std::max<long long>(1, (m->target_wgs + B - 1) / B);
  long long per = ((long long)m->N + want - 1) / want;
  per = std::max(tile_px, (per + tile_px - 1) / tile_px * tile_px);
  if (m->chunk_px_override > 0) per = m->chunk_px_override;  // experiments: GIGALENS_HIP_CHUNK_PX (a multiple of the kernel's tile)
  *chunk = (int)per;
  *n_chunks = (int)(((long long)m->N + per - 1) / per);
}

// the wavefront-per-sample front end (gl_prep_wave_kernel) carries the sort of the cost-ordered dispatch as one extra workgroup:
// no launch of gl_order_kernel (one launch boundary less on every step)
bool wave_front_end(const gl_model* m) { return m->has_epl && (int)m->comps.size() <= 64 && m->wave_prep; }
bool order_in_front_end(const gl_model* m, int B) {
  return wave_front_end(m) && m->use_order && m->order_fused && m->epl_comp >= 0 && B >= 2;
}

// Tapered end of the cost-ordered dispatch (pair kernels, fused likelihood).  A launch is whole resident rounds of the chip
// (256 CUs x the workgroups a CU holds) plus a remainder; the samples of the remainder -- the cheapest ones, dispatched last --
// run as twice as many workgroups of half the pixels, so the last round is made of shorter workgroups (C2 at 1024 samples: two
// rounds of 768 + 512 workgroups -> two rounds + 1024 half-size ones: -0.9 us of 87).  Splitting samples that are NOT the
// remainder puts the boundary inside a round and costs 2-3 %, hence the exact count.  Returns the workgroups per tail sample
// (0 = does not apply), the first tail rank and the partial rows per sample.
int tail_plan(const gl_model* m, int B, int n_chunks, int* tail_from, int* n_rows) {
  *tail_from = B;
  *n_rows = n_chunks;
  const bool pair_epl = m->pair && (m->static_id == ST_EPLSHEAR_SERSIC || m->static_id == ST_EPLSHEAR_SERSIC_SERSIC);
  if (m->tail_rows == 0 || !pair_epl || m->has_post || !order_in_front_end(m, B) || B > 1024) return 0;
  const int tiles = (m->N + 2 * WG - 1) / (2 * WG);
  const int slots = 256 * (m->static_id == ST_EPLSHEAR_SERSIC ? 3 : 2);  // launch_static: waves per SIMD the kernel is budgeted for
  int n_tail = m->tail_n, rows = m->tail_rows;
  if (n_tail < 0) {  // the remainder beyond whole rounds, when it is a substantial part of a round
    const long long wgs = (long long)B * n_chunks, rem = wgs % slots;
    if (wgs < slots || rem * 4 < slots) return 0;
    n_tail = (int)(rem / n_chunks);
  }
  if (rows < 0) rows = 2 * n_chunks;
  if (n_tail <= 0 || n_tail > B || rows > tiles || rows == n_chunks) return 0;
  *tail_from = B - n_tail;
  *n_rows = std::max(n_chunks, rows);
  return rows;
}

struct Workspace {
  float* derived;
  float* partial;
  float* params;  // [B,P] constrained rows produced from z (gl_logprob_fwd_bwd)
  int* order;     // [B] cost-ordered dispatch
  int* cost;      // [B] per-sample dispatch cost written by prep (single-EPL models)
  float* gal_dyn;  // [B][G][DP_ND] catalogue members' per-sample constants
  float *pos_w, *pos_adj, *pos_g, *pos_fam, *pos_ll, *pos_chi2, *pos_grad;  // image-position likelihood
  float* img_ss;   // supersampled / pre-PSF image or its cotangent (PSF path only)
  float* img_tmp;  // final-resolution image / its cotangent (PSF path only)
  float* stats;    // [B][2] chi2, normalisation of the materialised image (PSF path only)
  size_t bytes;
};

Workspace carve(const gl_model* m, int B, void* base) {
  int chunk, n_chunks;
  chunking(m, B, &chunk, &n_chunks);
  Workspace w{};
  size_t off = 0;
  char* p = (char*)base;
  w.derived = (float*)(p + off);
  off += align_up((size_t)B * m->D * sizeof(float), 256);
  w.partial = (float*)(p + off);
  int tail_from, n_rows;
  tail_plan(m, B, n_chunks, &tail_from, &n_rows);
  off += align_up((size_t)B * n_rows * m->A * sizeof(float), 256);
  w.params = (float*)(p + off);
  off += align_up((size_t)B * std::max(m->P, 1) * sizeof(float), 256);
  w.order = (int*)(p + off);
  off += align_up((size_t)B * sizeof(int), 256);
  w.cost = (int*)(p + off);
  off += align_up((size_t)B * sizeof(int), 256);
  if (m->G) {
    w.gal_dyn = (float*)(p + off);
    off += align_up((size_t)B * m->G * GM_ND * sizeof(float), 256);
  }
  if (m->pos_J) {
    auto take = [&](size_t n) { float* q = (float*)(p + off); off += align_up(n * sizeof(float), 256); return q; };
    w.pos_w = take((size_t)B * m->pos_J * 6);
    w.pos_adj = take((size_t)B * m->pos_J * 3);
    w.pos_g = take((size_t)B * m->pos_J * std::max(m->P, 1));
    w.pos_fam = take((size_t)B * m->pos_F * 2);
    w.pos_ll = take(B);
    w.pos_chi2 = take(B);
    w.pos_grad = take((size_t)B * std::max(m->P, 1));
  }
  if (m->has_post) {
    w.img_ss = (float*)(p + off);
    off += align_up((size_t)B * m->height * m->width * sizeof(float), 256);
    w.img_tmp = (float*)(p + off);  // final-resolution image / its cotangent
    off += align_up((size_t)B * (m->height / m->supersample) * (m->width / m->supersample) * sizeof(float), 256);
    w.stats = (float*)(p + off);
    off += align_up((size_t)B * 2 * sizeof(float), 256);
  }
  w.bytes = off;
  return w;
}


MainArgs base_args(const gl_model* m, const Workspace& w, int chunk) {
  MainArgs a{};
  a.comps = m->d_comps;
  a.n_lens = m->n_lens;
  a.n_ll = m->n_ll;
  a.n_src = m->n_src;
  a.derived = w.derived;
  a.D = m->D;
  a.A = m->A;
  a.Apad = m->Apad;
  a.ncols = m->ncols;
  a.gx = m->d_gx;
  a.gy = m->d_gy;
  a.pix = m->d_pix;
  a.N = m->N;
  a.chunk = chunk;
  a.img_stride = (long long)m->height * m->width;
  a.out_scale = m->conversion_factor;
  a.partial = w.partial;
  a.shp_tab = m->d_shp_tab;
  a.nfw_tab = m->d_nfw_tab;
  a.neutral = m->d_nfw_tab ? m->d_nfw_tab + 2 * glh::kNfwNodes : nullptr;
  a.grid_rmax = m->shp_cull ? m->grid_rmax : -1.f;  // (negative: the culling test of the table-mode shapelet kernels is off)
  a.dbg = m->dbg_flags;
  a.shp_stride = m->shp_stride;
  a.parts = 7u;
  a.cats = m->d_cats;
  a.gal_static = m->d_gal_static;
  a.gal_dyn = w.gal_dyn;
  a.G = m->G;
  a.scaled_first = m->cats.empty() ? -1 : m->cats[0].dev.comp;
  a.series = m->d_series;
  return a;
}

int check_call(const gl_model* m, const void* params, int B, void* ws, size_t ws_bytes) {
  if (!m) return fail(GL_EINVAL, "model is null");
  if (!params) return fail(GL_EINVAL, "params is null");
  if (B <= 0 || B > 65535) return fail(GL_EINVAL, "batch size %d outside [1, 65535]", B);
  if ((int)m->cats.size() != m->n_scaled)
    return fail(GL_EINVAL, "%d GL_SCALED component(s) without a catalogue (gl_model_set_catalogue)",
                m->n_scaled - (int)m->cats.size());
  if (m->n_series_set != m->n_series)
    return fail(GL_EINVAL, "%d GL_SERIES component(s) without a coefficient field (gl_model_set_series)",
                m->n_series - m->n_series_set);
  if (!ws) return fail(GL_EINVAL, "workspace is null");
  size_t need = gl_workspace_bytes(m, B);
  if (ws_bytes < need) return fail(GL_ENOMEM, "workspace too small: %zu < %zu bytes", ws_bytes, need);
  return GL_OK;
}

// per (sample, galaxy) constants of the catalogue members, from the constrained parameter rows
int run_galprep(const gl_model* m, const float* params, int B, const Workspace& w, hipStream_t stream) {
  if (!m->G) return GL_OK;
  long long total = (long long)B * m->G;
  hipLaunchKernelGGL(gl_galprep_kernel, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, stream, m->d_comps,
                     m->d_cats, (int)m->cats.size(), params, m->P, B, m->d_gal_table, m->d_gal_static, w.gal_dyn, m->G);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

// LDS of the wavefront front end: one parameter row per wavefront of the workgroup (0: rows too long, read back from global memory)
size_t prep_row_bytes(const gl_model* m) {
  const size_t bytes = (size_t)4 * m->P * sizeof(float);
  return (m->prep_lds && bytes <= 48 * 1024) ? bytes : 0;
}

// the rank the sort must split deterministically (tail_plan): which samples run the tapered end may not depend on the order
// the atomics of the counting sort leave inside a cost bin
int prep_tail_from(const gl_model* m, int B) {
  int chunk, n_chunks, tail_from, n_rows;
  chunking(m, B, &chunk, &n_chunks);
  return tail_plan(m, B, n_chunks, &tail_from, &n_rows) ? tail_from : -1;
}

int run_prep(const gl_model* m, const float* params, int B, const Workspace& w, hipStream_t stream) {
  int n_comp = (int)m->comps.size();
  int total = B * n_comp;
  if (wave_front_end(m)) {  // one wavefront per sample: the EPL coefficient tables are built by a scan over its lanes
    const bool ord = order_in_front_end(m, B);
    const size_t rows = prep_row_bytes(m);
    hipLaunchKernelGGL(gl_prep_wave_kernel, dim3((B + 3) / 4 + (ord ? 1 : 0)), dim3(256), rows, stream, m->d_comps, n_comp, params, nullptr,
                       0, (const ZCol*)nullptr, (const int*)nullptr, (const float*)nullptr, m->P, B, (float*)nullptr, w.derived,
                       m->D, m->epl_comp >= 0 ? w.cost : nullptr, m->epl_comp, ord ? w.order : nullptr, rows ? 1 : 0, prep_tail_from(m, B));
  } else
    hipLaunchKernelGGL(gl_prep_kernel, dim3((total + 127) / 128), dim3(128), 0, stream, m->d_comps, n_comp, params,
                       m->P, B, w.derived, m->D, m->epl_comp >= 0 ? w.cost : nullptr, m->epl_comp);
  GL_HIP(hipGetLastError());
  return run_galprep(m, params, B, w, stream);
}

FinArgs fin_args(const gl_model* m, const float* params, const Workspace& w, float* loglike, float* chi2, float* grad,
                 const float* z, float* logprob, float* grad_z, float chi2_scale, const float* extra_stats,
                 int use_partial, bool with_positions, float pos_chi2_scale) {
  FinArgs f{};
  f.n_comp = (int)m->comps.size();
  f.P = m->P;
  f.A = m->A;
  f.d_z = m->d_z;
  f.params = params;
  f.loglike = loglike;
  f.chi2 = chi2;
  f.grad = grad;
  f.z = z;
  f.zcols = z ? (const ZCol*)m->d_zcols : nullptr;
  f.logprob = logprob;
  f.grad_z = grad_z;
  f.chi2_scale = chi2_scale;
  f.extra_stats = extra_stats;
  f.use_partial = use_partial;
  f.pos_ll = with_positions ? w.pos_ll : nullptr;
  f.pos_chi2 = with_positions ? w.pos_chi2 : nullptr;
  f.pos_grad = with_positions && (grad || grad_z) ? w.pos_grad : nullptr;
  f.pos_chi2_scale = pos_chi2_scale;
  f.cats = m->d_cats;
  return f;
}

int run_finalize(const gl_model* m, const float* params, int B, int n_chunks, const Workspace& w, float* loglike,
                 float* chi2, float* grad, hipStream_t stream, const float* z = nullptr, float* logprob = nullptr,
                 float* grad_z = nullptr, float chi2_scale = 1.f, const float* extra_stats = nullptr,
                 int use_partial = 1, bool with_positions = false, float pos_chi2_scale = 0.f) {
  size_t shmem = (size_t)(((m->A + 3) & ~3) + ((m->P + 3) & ~3) + ((m->d_z + 3) & ~3) + 4 * m->d_z + 4 + m->P) * sizeof(float);
  FinArgs f = fin_args(m, params, w, loglike, chi2, grad, z, logprob, grad_z, chi2_scale, extra_stats, use_partial,
                       with_positions, pos_chi2_scale);
  bool basic = true;
  for (const CompDesc& c : m->comps)
    basic = basic && (c.kind == K_EPL || c.kind == K_SIE || c.kind == K_SHEAR || c.kind == K_SIS || c.kind == K_SERSIC || c.kind == K_SERSIC_ELLIPSE);
  const int nc = GL_DBG(m->dbg_flags, 8) ? -1 : n_chunks;
  if (basic) hipLaunchKernelGGL(gl_finalize_kernel<true>, dim3(B), dim3(128), shmem, stream, m->d_comps, f, w.partial, nc);
  else hipLaunchKernelGGL(gl_finalize_kernel<false>, dim3(B), dim3(128), shmem, stream, m->d_comps, f, w.partial, nc);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

// heaviest samples first (only EPL has a data-dependent cost)
int run_order(const gl_model* m, int B, const Workspace& w, MainArgs* a, hipStream_t stream) {
  a->order = nullptr;
  if (!m->has_epl || !m->use_order || B < 2) return GL_OK;
  if (order_in_front_end(m, B)) {  // the front end's extra workgroup has written it
    a->order = w.order;
    return GL_OK;
  }
  hipLaunchKernelGGL(gl_order_kernel, dim3(1), dim3(ORDER_WG), 0, stream, m->d_comps, m->n_lens, w.derived, m->D, B,
                     w.order, m->epl_comp >= 0 ? w.cost : nullptr);
  GL_HIP(hipGetLastError());
  a->order = w.order;
  return GL_OK;
}


// image-position likelihood on the packed parameter rows `params` [B,P] (already on the device)
int run_positions(const gl_model* m, const float* params, int B, const Workspace& w, bool want_grad, hipStream_t stream) {
  if (m->n_series) return fail(GL_EUNSUPPORTED, "a series-expansion lens lives on the pixel grid only (series_profile.py:76-81): no image-position likelihood");
  if (m->has_user)  // the four kernels below, compiled at run time with the user's bodies on the nested duals (once per model text)
    if (int rc = compile_user_points(m)) return rc;
  PosArgs a{};
  a.comps = m->d_comps;
  a.n_lens = m->n_lens;
  a.P = m->P;
  a.B = B;
  a.J = m->pos_J;
  a.F = m->pos_F;
  a.params = params;
  a.px = m->d_pos;
  a.py = m->d_pos + m->pos_J;
  a.ex = m->d_pos + 2 * m->pos_J;
  a.ey = m->d_pos + 3 * m->pos_J;
  a.fam_off = m->d_fam;
  a.w_pos = w.pos_w;
  a.w_adj = w.pos_adj;
  a.w_g = w.pos_g;
  a.w_fam = w.pos_fam;
  a.ll = w.pos_ll;
  a.chi2 = w.pos_chi2;
  a.grad = want_grad ? w.pos_grad : nullptr;
  a.cats = m->d_cats;
  a.gal_table = m->d_gal_table;
  a.gal_static = m->d_gal_static;
  auto blocks = [](long long n) { return dim3((unsigned)((n + 63) / 64)); };
  if (m->has_user) {
    int lens_params = m->lens_params;
    void* args1[] = {&a};
    void* args2[] = {&a, &lens_params};
    auto go = [&](int k, long long n, void** args) {
      return hipModuleLaunchKernel(m->user_point_fn[k], blocks(n).x, 1, 1, 64, 1, 1, 0, stream, args, nullptr);
    };
    GL_HIP(go(0, (long long)B * a.J, args1));
    GL_HIP(go(1, (long long)B * a.F, args1));
    if (want_grad && m->lens_params) GL_HIP(go(2, (long long)B * a.J * m->lens_params, args2));
    GL_HIP(go(3, (long long)B * (a.P + 1), args2));
    return GL_OK;
  }
  hipLaunchKernelGGL(gl_pos_p1_kernel, blocks((long long)B * a.J), dim3(64), 0, stream, a);
  hipLaunchKernelGGL(gl_pos_p2_kernel, blocks((long long)B * a.F), dim3(64), 0, stream, a);
  if (want_grad && m->lens_params)
    hipLaunchKernelGGL(gl_pos_p3_kernel, blocks((long long)B * a.J * m->lens_params), dim3(64), 0, stream, a,
                       m->lens_params);
  hipLaunchKernelGGL(gl_pos_p4_kernel, blocks((long long)B * (a.P + 1)), dim3(64), 0, stream, a, m->lens_params);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

// ---- PSF / supersampling path (gl_post.hip.h) -------------------------------------------------------------
PostArgs post_args(const gl_model* m, float scale) {
  PostArgs p{};
  p.keff = m->d_psf;
  p.KH = m->KH; p.KW = m->KW; p.pt = m->pad_t; p.pl = m->pad_l;
  p.Hs = m->height; p.Ws = m->width; p.ss = m->supersample;
  p.H = m->height / m->supersample; p.W = m->width / m->supersample;
  p.scale = scale;
  return p;
}
// supersampled pre-PSF image S [B,Hs,Ws] -> final image [B,H,W] (x conversion factor)
// the register-blocked pair kernel on one plan (gl_post.hip.h); false: no instantiation for this kernel width / stride
bool launch_corr(const gl_model::CorrPlan& pl, int B, const float* in, float* out, float scale, hipStream_t stream, int dbg = 0,
                 int max_pairs_env = 0, int corr_wide = 1) {
  if (!pl.ok) return false;
  CorrArgs a = pl.args;
  a.B = B;
  a.scale = scale;
  a.dbg = dbg;
  a.vec = (a.Wi % 4 == 0 && a.Wout % 4 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) ? 1 : 0;
  // forward at supersample 2 with kernels up to 28 taps wide: 16 outputs per thread and 8 wavefronts per tile (two 75 KB tiles per
  // CU); everything else 8 outputs per thread
  const bool wide = pl.ST == 2 && a.ncj == 1 && pl.KWP <= 28 && pl.max_KH <= 28 && corr_wide;
  // (16 outputs per thread in the transpose at supersample 2 as well: 62.7 us against 47.6 -- half the workgroups, 1.56 rounds)
  const int ox = wide ? 16 : CORR_OX, ks = wide ? 8 : pl.ST == 2 ? 4 : 2;
  const int TR = (CORR_TR - 1) * pl.ST + pl.max_KH, TC = corr_tile_width((CORR_TCG * ox - 1) * pl.ST + pl.KWP) | 1;
  const size_t sh = std::max((size_t)TR * TC * sizeof(float2), (size_t)(ks - 1) * a.ncj * ox * CORR_GT * sizeof(float2) +
                                                                   (size_t)2 * CORR_TR * CORR_TCG * ox * a.ncj * sizeof(float));
  if (sh > (wide ? 80 : 64) * 1024) return false;
  // grid.z carries (row class, sample pair): at most 65535 per launch -- larger batches (the basis stack of lstsq_simulate is
  // B x D images) go out in slices
  const int max_pairs = max_pairs_env > 0 ? max_pairs_env : 65535 / a.n_class;
  if ((B + 1) / 2 > max_pairs) {
    for (int b_lo = 0; b_lo < B; b_lo += 2 * max_pairs) {
      const int nb = std::min(B - b_lo, 2 * max_pairs);
      if (!launch_corr(pl, nb, in + (size_t)b_lo * a.Hi * a.Wi, out + (size_t)b_lo * a.Hout * a.Wout, scale, stream, dbg, max_pairs_env, corr_wide)) return false;
    }
    return true;
  }
  const dim3 grid((pl.max_Wo + CORR_TCG * ox - 1) / (CORR_TCG * ox), (pl.max_Ho + CORR_TR - 1) / CORR_TR,
                  (unsigned)(a.n_class * ((B + 1) / 2)));
#define GL_CORR(KWP_, ST_, KS_, NCJ_, OX_)                                                                             \
  {                                                                                                                    \
    auto* fn = gl_corr_pair_kernel<KWP_, ST_, KS_, NCJ_, OX_>;                                                          \
    if (sh > 64 * 1024) {                                                                                              \
      static const hipError_t big = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
      if (big != hipSuccess) return false;                                                                             \
    }                                                                                                                  \
    hipLaunchKernelGGL(fn, grid, dim3(CORR_GT * KS_), sh, stream, in, out, a);                                         \
    return true;                                                                                                       \
  }
#define GL_CORR_W(ST_, KS_, NCJ_, OX_)                                                                                          \
  switch (pl.KWP) {                                                                                                             \
    case 4: GL_CORR(4, ST_, KS_, NCJ_, OX_) case 8: GL_CORR(8, ST_, KS_, NCJ_, OX_) case 12: GL_CORR(12, ST_, KS_, NCJ_, OX_)     \
    case 16: GL_CORR(16, ST_, KS_, NCJ_, OX_) case 20: GL_CORR(20, ST_, KS_, NCJ_, OX_) case 24: GL_CORR(24, ST_, KS_, NCJ_, OX_) \
    case 28: GL_CORR(28, ST_, KS_, NCJ_, OX_) case 32: GL_CORR(32, ST_, KS_, NCJ_, OX_)                                          \
    default: return false;                                                                                                      \
  }
  if (wide) { GL_CORR_W(2, 8, 1, 16) }
  if (pl.ST == 2 && a.ncj == 1) { GL_CORR_W(2, 4, 1, CORR_OX) }  // forward at supersample 2
  if (pl.ST == 1 && a.ncj == 1) { GL_CORR_W(1, 2, 1, CORR_OX) }  // forward / transpose at supersample 1
  if (pl.ST == 1 && a.ncj == 2) { GL_CORR_W(1, 2, 2, CORR_OX) }  // transpose at supersample 2
#undef GL_CORR_W
#undef GL_CORR
  return false;
}

int post_fwd(const gl_model* m, int B, const float* S, float* out, hipStream_t stream, float scale = -1.f) {
  if (launch_corr(m->corr_fwd, B, S, out, scale < 0.f ? m->conversion_factor : scale, stream, m->dbg_flags, m->corr_max_pairs, m->corr_wide)) {
    GL_HIP(hipGetLastError());
    return GL_OK;
  }
  PostArgs p = post_args(m, scale < 0.f ? m->conversion_factor : scale);
  const int TR = (PT - 1) * p.ss + p.KH, TC = ((PT - 1) * p.ss + p.KW) | 1;
  size_t shmem = (size_t)TR * TC * sizeof(float);
  if (shmem > 64 * 1024) return fail(GL_EUNSUPPORTED, "PSF too large for the LDS-tiled convolution (%zu B)", shmem);
  dim3 grid((p.W + PT - 1) / PT, (p.H + PT - 1) / PT, B);
  hipLaunchKernelGGL(gl_psf_pool_fwd_kernel, grid, dim3(256), shmem, stream, S, out, p);
  GL_HIP(hipGetLastError());
  return GL_OK;
}
// cotangent of the final image [B,H,W] -> cotangent of S [B,Hs,Ws]
int post_bwd(const gl_model* m, int B, const float* gP, float* gS, hipStream_t stream) {
  if (launch_corr(m->corr_bwd, B, gP, gS, m->conversion_factor, stream, m->dbg_flags, m->corr_max_pairs, m->corr_wide)) {
    GL_HIP(hipGetLastError());
    return GL_OK;
  }
  PostArgs p = post_args(m, m->conversion_factor);
  const int TR = (PT - 1 + p.KH) / p.ss + 3, TC = ((PT - 1 + p.KW) / p.ss + 3) | 1;
  size_t shmem = (size_t)TR * TC * sizeof(float);
  dim3 grid((p.Ws + PT - 1) / PT, (p.Hs + PT - 1) / PT, B);
  hipLaunchKernelGGL(gl_psf_pool_bwd_kernel, grid, dim3(256), shmem, stream, gP, gS, p);
  GL_HIP(hipGetLastError());
  return GL_OK;
}
int render_ss(const gl_model* m, MainArgs a, int B, int n_chunks, const Workspace& w, hipStream_t stream) {
  if (m->d_pix) GL_HIP(hipMemsetAsync(w.img_ss, 0, sizeof(float) * (size_t)B * m->height * m->width, stream));
  a.img = w.img_ss;
  a.out_scale = 1.f;  // NaN -> 0 happens in the kernel; the det(T) scale is applied after pooling (tf/simulator.py:156)
  return launch_main<IMG_FWD>(m, a, B, n_chunks, stream);
}

// likelihood after prep: fused kernel when the image never has to exist, else render -> PSF/pool -> pixel
// statistics (-> transposes -> VJP).  Tells finalize where chi2 / normalisation come from.
int run_likelihood(const gl_model* m, int B, const Workspace& w, int chunk, int n_chunks, const float* obs,
                   const float* err, const float* mask, float bg_rms, float exp_time, bool want_grad,
                   hipStream_t stream, const float** extra_stats, int* use_partial, int* n_rows) {
  int rc;
  MainArgs a = base_args(m, w, chunk);
  a.obs = obs;
  a.err = err;
  a.mask = mask;
  a.bg2 = bg_rms * bg_rms;
  a.inv_t = 1.0f / exp_time;
  if ((rc = run_order(m, B, w, &a, stream))) return rc;
  *extra_stats = nullptr;
  *use_partial = 1;
  *n_rows = n_chunks;
  if (!m->has_post && a.order && want_grad) {  // (the rounds are counted for the gradient kernels' occupancy; forward-only calls keep the plain grid)
    a.tail_rows = tail_plan(m, B, n_chunks, &a.tail_from, &a.n_rows);
    a.n_samples = B;
    *n_rows = a.n_rows;
  }
  if (!m->has_post) return want_grad ? launch_main<LL_GRAD>(m, a, B, n_chunks, stream) : launch_main<LL_FWD>(m, a, B, n_chunks, stream);
  if ((rc = render_ss(m, a, B, n_chunks, w, stream))) return rc;
  if ((rc = post_fwd(m, B, w.img_ss, w.img_tmp, stream))) return rc;
  const int HW = (m->height / m->supersample) * (m->width / m->supersample);
  hipLaunchKernelGGL(gl_imgstats_kernel, dim3(B), dim3(256), 0, stream, w.img_tmp, obs, err, mask, a.bg2, a.inv_t, HW,
                     w.stats, want_grad ? w.img_tmp : nullptr);
  GL_HIP(hipGetLastError());
  *extra_stats = w.stats;
  *use_partial = want_grad ? 1 : 0;
  if (!want_grad) return GL_OK;
  if ((rc = post_bwd(m, B, w.img_tmp, w.img_ss, stream))) return rc;
  a.gimg = w.img_ss;
  a.out_scale = 1.f;
  return launch_main<IMG_BWD>(m, a, B, n_chunks, stream);
}

}  // namespace

namespace {
struct LstsqWs {
  float *stack_ss, *stack, *partial, *coeffs;
  float* mats;  // [B][2][D][D | 1]: A and V of the eigen solve for systems above LS_LDS_MAXN unknowns (else null)
  int* todo;    // [B]: 1 = the Cholesky attempt left this system to the eigenvalue solve
  int chunk, n_chunks, Dp;
  int n_chunks_f;  // workgroups per sample of the stack-free kernel (gl_shp_normal_kernel: 512-pixel tiles dealt round-robin)
  size_t bytes;
};
LstsqWs carve_lstsq(const gl_model* m, int B, void* base, size_t off) {
  LstsqWs w{};
  const int D = (int)m->lin_cols.size();
  const size_t HWs = (size_t)m->height * m->width, HW = HWs / ((size_t)m->supersample * m->supersample);
  char* p = (char*)base;
  auto take = [&](size_t n) { float* q = (float*)(p + off); off += align_up(n * sizeof(float), 256); return q; };
  w.Dp = (D + 1 + 3) & ~3;
  // pixel chunks per sample: ~2048 workgroups in flight, whole LDS tiles per chunk
  long long want = std::max<long long>(1, (m->lstsq_wgs + B - 1) / B);
  long long per = ((long long)HW + want - 1) / want;
  per = std::max<long long>(2 * LS_TPP, (per + 2 * LS_TPP - 1) / (2 * LS_TPP) * (2 * LS_TPP));
  w.chunk = (int)per;
  w.n_chunks = (int)(((long long)HW + per - 1) / per);
  w.stack_ss = m->has_post ? take((size_t)B * D * HWs) : nullptr;
  w.stack = take((size_t)B * D * HW);
  // three workgroups per CU there: 1.5 x the workgroup target = four full rounds of the chip at the default (measured: 2048 ->
  // 0.708, 3072 -> 0.694, 4096 -> 0.697, 6144 -> 0.715 ms per C3L solve)
  w.n_chunks_f = (int)std::min<long long>(((long long)HW + 511) / 512, std::max<long long>(1, (3LL * m->lstsq_wgs / 2 + B - 1) / B));
  w.partial = take((size_t)B * std::max(w.n_chunks, w.n_chunks_f) * w.Dp * w.Dp);
  w.coeffs = take((size_t)B * D);
  w.mats = D > LS_LDS_MAXN ? take((size_t)B * 2 * D * (D | 1)) : nullptr;
  w.todo = (int*)take((size_t)B);
  w.bytes = off;
  return w;
}
}  // namespace

extern "C" {

const char* gl_last_error(void) { return g_err; }
const char* gl_version(void) { return "gigalens_hip 0.1 (gfx950)"; }

int gl_kind_num_params(const gl_component* comp) {
  if (!comp) return fail(GL_EINVAL, "component is null");
  int n = kind_num_params(comp->kind, comp->iparam);
  if (n < 0) return fail(GL_EINVAL, "unknown profile kind %d (iparam %d)", comp->kind, comp->iparam);
  return n;
}

int gl_model_create(const gl_component* comps, int n_lens, int n_lens_light, int n_src, const gl_grid* grid,
                    gl_model** out) {
  return gl_model_create_user(comps, n_lens, n_lens_light, n_src, grid, nullptr, 0, out);
}

int gl_model_create_user(const gl_component* comps, int n_lens, int n_lens_light, int n_src, const gl_grid* grid,
                         const char* const* bodies, int n_bodies, gl_model** out) {
  if (!out) return fail(GL_EINVAL, "out is null");
  if (n_bodies < 0 || (n_bodies > 0 && !bodies)) return fail(GL_EINVAL, "bad user bodies");
  *out = nullptr;
  if (!grid) return fail(GL_EINVAL, "grid is null");
  if (n_lens < 0 || n_lens_light < 0 || n_src < 0) return fail(GL_EINVAL, "negative component count");
  int n_comp = n_lens + n_lens_light + n_src;
  if (n_comp > 0 && !comps) return fail(GL_EINVAL, "comps is null");
  if (grid->height <= 0 || grid->width <= 0 || grid->n_region <= 0) return fail(GL_EINVAL, "empty grid");
  if (!grid->grid_x || !grid->grid_y) return fail(GL_EINVAL, "grid_x / grid_y is null");
  if (grid->supersample < 1) return fail(GL_EINVAL, "supersample must be >= 1");
  if (grid->height % grid->supersample || grid->width % grid->supersample)
    return fail(GL_EINVAL, "grid size not a multiple of supersample");
  if ((long long)grid->n_region > (long long)grid->height * grid->width)
    return fail(GL_EINVAL, "n_region exceeds height*width");
  if (!grid->pix_index && (long long)grid->n_region != (long long)grid->height * grid->width)
    return fail(GL_EINVAL, "pix_index is required when n_region != height*width");
  if (grid->psf && (grid->psf_h <= 0 || grid->psf_w <= 0)) return fail(GL_EINVAL, "bad PSF shape");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(GL_ENODEVICE, "no HIP device available");

  gl_model* m = new (std::nothrow) gl_model();
  if (!m) return fail(GL_ENOMEM, "host allocation failed");
  m->n_lens = n_lens;
  m->n_ll = n_lens_light;
  m->n_src = n_src;
  int p_off = 0, d_off = 0, a_off = NSTAT, sh_nmax = -1;
  for (int i = 0; i < n_comp; ++i) {
    const gl_component& c = comps[i];
    const bool mass = i < n_lens;
    const bool is_mass_kind = (c.kind >= GL_EPL && c.kind <= GL_TNFW) || c.kind == GL_USER_MASS;
    const bool is_light_kind = (c.kind >= GL_SERSIC && c.kind <= GL_CORE_SERSIC) || c.kind == GL_USER_LIGHT;
    if ((mass && !is_mass_kind) || (!mass && !is_light_kind)) {
      delete m;
      return fail(GL_EINVAL, "component %d: kind %d is not a %s profile", i, c.kind, mass ? "mass" : "light");
    }
    int iparam = c.iparam;
    if (c.kind == GL_USER_MASS || c.kind == GL_USER_LIGHT) {
      if (iparam < 0 || iparam > USER_MAXP || (int)c.flags >= n_bodies || !bodies[c.flags]) {
        delete m;
        return fail(GL_EINVAL, "component %d: a user-written profile takes 0..%d parameters and the index of its body (got %d parameters, "
                               "body %u of %d)", i, USER_MAXP, iparam, c.flags, n_bodies);
      }
      m->has_user = true;
    }
    if (c.kind == GL_EPL) {
      if (iparam <= 0) iparam = 50;  // epl.py:15
      if (iparam > 1000) { delete m; return fail(GL_EINVAL, "EPL niter %d too large", iparam); }
    }
    if (c.kind == GL_EPL) { m->epl_comp = m->has_epl ? -2 : i; m->has_epl = true; }
    if (c.kind >= GL_DPIS && c.kind <= GL_SERIES) m->fam = std::max(m->fam, 1);
    if (c.kind == GL_NFW_ELLIPSE || c.kind == GL_TNFW || c.kind == GL_CORE_SERSIC) m->fam = 2;
    if (c.kind == GL_SERIES && (iparam < 0 || iparam > SERIES_MAX_ORDER)) {
      delete m;
      return fail(GL_EINVAL, "component %d: series order %d outside [0, %d]", i, iparam, SERIES_MAX_ORDER);
    }
    if (c.kind == GL_SCALED) {
      if (iparam < 1 || iparam > 3) { delete m; return fail(GL_EINVAL, "component %d: GL_SCALED takes 1..3 scales, got %d", i, iparam); }
      ++m->n_scaled;
    }
    if (c.kind == GL_SHAPELETS) {
      if (iparam < 0 || iparam > GL_SHAPELETS_NMAX_CAP) {
        delete m;
        return fail(GL_EUNSUPPORTED, "shapelets n_max=%d outside [0,%d]", iparam, GL_SHAPELETS_NMAX_CAP);
      }
      m->has_shapelets = true;
      if (iparam > SH_CAP) m->shp_big = true;
      if (c.flags & GL_FLAG_SHAPELETS_INTERPOLATE) { m->has_table = true; sh_nmax = std::max(sh_nmax, iparam); }
    }
    CompDesc cd{};
    cd.kind = c.kind;
    cd.iparam = iparam;
    cd.flags = c.flags;
    cd.p_off = p_off;
    cd.d_off = d_off;
    cd.a_off = a_off;
    cd.n_par = kind_num_params(c.kind, iparam);
    cd.n_acc = kind_num_acc(c.kind, iparam);
    if (c.kind == GL_SCALED) cd.iparam = -1;  // catalogue slot, set by gl_model_set_catalogue
    if (c.kind == GL_SERIES) {
      cd.flags = (unsigned)m->n_series++;
      m->series.push_back(SeriesDev{nullptr, nullptr, 0.f, iparam});
      m->series_comp.push_back(i);
    }
    cd.lin_off = (int)m->lin_cols.size();
    for (int k = 0; k < kind_num_linear(c.kind, iparam); ++k) m->lin_cols.push_back(p_off + kind_linear_col(c.kind, iparam) + k);
    // a user-written light whose last parameter is declared the linear amplitude (gl_component::reserved): one basis image
    if (c.kind == GL_USER_LIGHT && c.reserved == 1 && iparam >= 1) m->lin_cols.push_back(p_off + iparam - 1);
    p_off += cd.n_par;
    d_off += (kind_num_derived(c.kind, iparam) + 3) & ~3;
    a_off += cd.n_acc;
    m->comps.push_back(cd);
  }
  m->P = p_off;
  for (int i = 0; i < n_lens; ++i) m->lens_params += m->comps[i].n_par;
  m->D = std::max(d_off, 4);
  m->A = a_off;
  m->Apad = a_off | 1;  // odd: the 16 leader lanes of a wave land on 16 different LDS banks
  for (int i = 0; i < n_lens; ++i) m->has_nfw = m->has_nfw || m->comps[i].kind == K_NFW;
  m->nfw_lds = m->has_nfw ? sizeof(float) * 2 * glh::kNfwNodes : 0;  // the h(X) table rides in every main kernel's LDS
  m->ncols = ((size_t)(((m->D + 3) & ~3) + 64 * m->Apad) * sizeof(float) + m->nfw_lds <= 60 * 1024) ? 64 : 16;
  m->height = grid->height;
  m->width = grid->width;
  m->supersample = grid->supersample;
  m->N = grid->n_region;
  m->conversion_factor = grid->conversion_factor;
  {
    int t = env_int("GIGALENS_HIP_TILE", 0);
    m->tile = (t == 4 || t == 1) ? t : 2;
    int tg = env_int("GIGALENS_HIP_TILE_GRAD", t ? t : 0);
    m->tile_grad = (tg == 4 || tg == 1 || tg == 2) ? tg : 0;
  }
  m->static_id = env_int("GIGALENS_HIP_STATIC", 1) ? match_static(m) : 0;
  if (m->has_user) {  // the run-time compiled interpreter serves the whole model
    m->static_id = 0;
    if (m->shp_big) { delete m; return fail(GL_EUNSUPPORTED, "user-written profiles beside shapelets with n_max > %d", SH_CAP); }
  }
  if (m->shp_big) {  // orders above SH_CAP: the runtime-order interpreter variant only (compiled for the basic profile families)
    m->static_id = 0;
    if (m->fam) {
      delete m;
      return fail(GL_EUNSUPPORTED, "shapelets with n_max > %d are served together with EPL / SIE / NFW / Shear / SIS lenses and Sersic "
                                   "lights only (this model also holds dPIE-family, catalogue, series or extended profiles)", SH_CAP);
    }
  }
  m->static_variant = env_int("GIGALENS_HIP_STATIC_VARIANT", 0);
  m->pair = env_int("GIGALENS_HIP_PAIR", 1);
  if (m->pair && m->static_id) {
    // the pair kernels' epilogue addresses the accumulator row in closed form: [NSTAT | components in order, static_nacc each]
    int off = NSTAT;
    bool ok_row = true;
    for (int i = 0; i < n_comp; ++i) {
      ok_row = ok_row && m->comps[i].a_off == off;
      off += static_nacc(m->comps[i].kind);
    }
    if (!ok_row || off != m->A) m->pair = 0;
  }
  // gl_shp.hip.h addresses the row as [NSTAT | lenses | lens lights | shapelet] in closed form and needs a table-mode model's pair table
  {
    int off = NSTAT;
    bool ok_row = true;
    for (int i = 0; i < n_comp; ++i) {
      ok_row = ok_row && m->comps[i].a_off == off;
      off += static_nacc(m->comps[i].kind);
    }
    m->shp_kernel = env_int("GIGALENS_HIP_SHP", 1) && env_int("GIGALENS_HIP_PAIR", 1) && m->static_id && ok_row && n_src == 1 &&
                    m->comps.back().kind == K_SHAPELETS;
  }
  m->light_spherical = n_comp > n_lens;
  for (int i = n_lens; i < n_comp; ++i) m->light_spherical = m->light_spherical && m->comps[i].kind == K_SERSIC;
  if (!m->tile_grad) m->tile_grad = m->static_id ? 1 : 2;  // measured: T=1 wins once the VJP state lives in registers
  if (!m->static_id) {  // the interpreter kernel is built for T = 2 and 4
    const bool env_tile = env_int("GIGALENS_HIP_TILE", 0) != 0;
    if (!env_tile && !m->has_epl && !m->has_shapelets && !m->fam) m->tile = 4;  // cheap profiles, forward modes: amortise the per-tile work
    if (m->tile == 1) m->tile = 2;
    if (m->tile_grad == 1) m->tile_grad = 2;
  }
  if (env_int("GIGALENS_HIP_CLUSTER", 1) && !m->static_id && !m->has_user && n_lens_light == 0 && n_lens >= 1 && n_lens <= 8 && n_src >= 1 &&
      n_src <= 20 && (size_t)64 * m->Apad * sizeof(float) <= 64 * 1024) {
    bool ok_c = true, ell = false;
    for (int i = 0; i < n_lens; ++i) ok_c = ok_c && m->comps[i].kind == K_NFW;
    for (int i = n_lens; i < n_comp; ++i) {
      ok_c = ok_c && (m->comps[i].kind == K_SERSIC || m->comps[i].kind == K_SERSIC_ELLIPSE);
      ell = ell || m->comps[i].kind == K_SERSIC_ELLIPSE;
    }
    // the kernel addresses the derived / accumulator blocks in closed form: component-major, fixed block sizes
    constexpr int NFWP = (NFW_ND + 3) & ~3, SERP = (SER_NDX + 3) & ~3;  // the strides gl_cluster_kernel walks the derived row with
    for (int i = 0; i < n_lens && ok_c; ++i) ok_c = m->comps[i].d_off == NFWP * i && m->comps[i].a_off == NSTAT + NFW_NACC * i;
    for (int i = 0; i < n_src && ok_c; ++i)
      ok_c = m->comps[n_lens + i].d_off == NFWP * n_lens + SERP * i && m->comps[n_lens + i].a_off == NSTAT + NFW_NACC * n_lens + SER_NACC * i;
    ok_c = ok_c && (size_t)64 * m->Apad * sizeof(float) + sizeof(float) * 2 * glh::kNfwNodes <= 64 * 1024;
    if (ok_c) m->cluster = ell ? 2 : 1;
    // ... in its component-per-wave form (gl_clusterw.hip.h) when the model fills at least 60 % of the component slots of the
    // instantiation that holds it (4 waves x (1 + 2), (2 + 3) or (2 + 5) halos + sources): an unused slot is evaluated all the same.
    // GIGALENS_HIP_CLUSTER: 1 = the pixel-split kernel for every cluster model, 2 = the component-per-wave kernel for every one
    if (m->cluster) {
      const int cap = (n_lens <= 4 && n_src <= 8) ? 12 : (n_src <= 12 ? 20 : 28);
      const int mode = env_int("GIGALENS_HIP_CLUSTER", -1);
      m->cluster_w = mode == 2 || (mode != 1 && 10 * (n_lens + n_src) >= 6 * cap);
    }
  }
  m->target_wgs = std::max(1, env_int("GIGALENS_HIP_TARGET_WGS", 2048));
  m->target_wgs_set = getenv("GIGALENS_HIP_TARGET_WGS") != nullptr;
  m->use_order = env_int("GIGALENS_HIP_ORDER", 1) != 0;
  m->order_fused = env_int("GIGALENS_HIP_ORDER_FUSED", 1) != 0;  // tests: 0 = the sort as a launch of its own (gl_order_kernel)
  m->prep_lds = env_int("GIGALENS_HIP_PREP_LDS", 1) != 0;  // tests: 0 = the front end reads the parameter row back from global memory
  m->tail_rows = env_int("GIGALENS_HIP_TAIL_ROWS", -1);  // -1: twice the chunks; 0: no tapered end; n: n workgroups per tail sample
  m->tail_n = env_int("GIGALENS_HIP_TAIL_N", -1);        // -1: the remainder beyond whole rounds; n: the last n samples
#ifdef GL_EXPERIMENTS
  // dissection builds only (hipcc -DGL_EXPERIMENTS; never __graft_entry__.build()): work-skipping flags and a raw chunk size
  m->chunk_px_override = env_int("GIGALENS_HIP_CHUNK_PX", 0);
  m->dbg_flags = env_int("GIGALENS_HIP_DBGFLAGS", 0);
  if (m->chunk_px_override < 0 || m->chunk_px_override % (WG * 4) != 0) {
    const int bad = m->chunk_px_override;
    delete m;
    return fail(GL_EINVAL, "GIGALENS_HIP_CHUNK_PX=%d is not a positive multiple of the tile (%d pixels)", bad, WG * 4);
  }
  if (m->dbg_flags || m->chunk_px_override)
    fprintf(stderr, "libgigalens_hip: EXPERIMENT BUILD with GIGALENS_HIP_DBGFLAGS=%d GIGALENS_HIP_CHUNK_PX=%d -- results are not valid\n",
            m->dbg_flags, m->chunk_px_override);
#else
  // the shipped library has no work-skipping paths: a stray dissection variable is an error, not a silently ignored hint
  for (const char* name : {"GIGALENS_HIP_DBGFLAGS", "GIGALENS_HIP_CHUNK_PX"}) {
    const char* v = getenv(name);
    if (v && *v && atoi(v) != 0) {
      delete m;
      return fail(GL_EINVAL, "%s is set but this library was built without -DGL_EXPERIMENTS (the dissection knobs do not exist in it)", name);
    }
  }
#endif
  m->shp_cull = env_int("GIGALENS_HIP_SHP_CULL", 1);
  m->shp_blocked = env_int("GIGALENS_HIP_SHP_BLOCKED", 1);
  m->corr_max_pairs = env_int("GIGALENS_HIP_CORR_MAXPAIRS", 0);  // tests: force the slicing of the PSF launches (read once)
  m->corr_wide = env_int("GIGALENS_HIP_CORR_WIDE", 1);           // 0: 8 outputs per thread in the stride-2 forward correlation as well
  m->wave_prep = env_int("GIGALENS_HIP_WAVE_PREP", 1) != 0;
  m->lstsq_wgs = std::max(1, env_int("GIGALENS_HIP_LSTSQ_WGS", 2048));
  m->lstsq_chol = env_int("GIGALENS_HIP_LSTSQ_CHOL", 1) != 0;    // tests: 0 = every system through the eigenvalue solve
  m->lstsq_fused = env_int("GIGALENS_HIP_LSTSQ_FUSED", 1) != 0;  // tests: 0 = the linear solve through the basis stack (read once)
  size_t shmem = (size_t)(((m->D + 3) & ~3) + m->ncols * m->Apad) * sizeof(float) + m->nfw_lds;
  if (shmem > 64 * 1024) { delete m; return fail(GL_EUNSUPPORTED, "model needs %zu B of LDS per workgroup (> 64 KiB)", shmem); }

  auto up = [&](void** dst, const void* src, size_t bytes) -> bool {
    if (hipMalloc(dst, bytes) != hipSuccess) return false;
    return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  bool ok = true;
  if (n_comp) ok = ok && up((void**)&m->d_comps, m->comps.data(), sizeof(CompDesc) * n_comp);
  else ok = ok && (hipMalloc((void**)&m->d_comps, sizeof(CompDesc)) == hipSuccess);
  ok = ok && up((void**)&m->d_gx, grid->grid_x, sizeof(float) * m->N);
  for (int i = 0; i < m->N; ++i) m->grid_rmax = std::max(m->grid_rmax, std::hypot(grid->grid_x[i], grid->grid_y[i]));
  ok = ok && up((void**)&m->d_gy, grid->grid_y, sizeof(float) * m->N);
  if (!m->lin_cols.empty()) ok = ok && up((void**)&m->d_lin_cols, m->lin_cols.data(), sizeof(int) * m->lin_cols.size());
  if (grid->pix_index) {
    for (int i = 0; i < m->N && ok; ++i)
      if (grid->pix_index[i] < 0 || grid->pix_index[i] >= m->height * m->width) {
        gl_model_destroy(m);
        return fail(GL_EINVAL, "pix_index[%d]=%d out of range", i, grid->pix_index[i]);
      }
    ok = ok && up((void**)&m->d_pix, grid->pix_index, sizeof(int) * m->N);
  }
  if (m->has_nfw) {
    // [h(X) node table | neutral blocks | H(s) cubics]: the layout gl_clusterw_kernel addresses (CW_NEUTRAL_OFF, CW_TABS_OFF)
    std::vector<float> tab;
    glh::build_nfw_table([](double X, double& g, double& gp) { glp::nfw_gw<double>(X, g, gp); }, tab);
    // behind the table: the constant blocks of an unused component slot of gl_clusterw_kernel (zero amplitude, all else finite)
    const float neutral_nfw[4] = {0.f, 0.f, 1.f, 0.f};  // NFW_CX, NFW_CY, NFW_INVRS, NFW_K0
    float neutral_ser[16] = {0.f};
    neutral_ser[glp::SER_C] = neutral_ser[glp::SER_SQ] = neutral_ser[glp::SER_ISQ] = neutral_ser[glp::SER_INVRS] = 1.f;
    neutral_ser[glp::SER_INVN] = neutral_ser[glp::SER_IRS2] = 1.f;
    neutral_ser[glp::SER_BN] = 1.6721f;  // n = 1; SER_IE = SER_CG = 0
    tab.insert(tab.end(), neutral_nfw, neutral_nfw + 4);
    tab.insert(tab.end(), neutral_ser, neutral_ser + 16);
    // ... and the table of the same function in s = X^2 (gl_host_tables.h::build_nfw_table_s), [4][kNfwSIntervals]
    std::vector<float> tab_s;
    glh::build_nfw_table_s([](double X, double& g, double& gp) { glp::nfw_gw<double>(X, g, gp); }, tab_s);
    tab.insert(tab.end(), tab_s.begin(), tab_s.end());
    ok = ok && up((void**)&m->d_nfw_tab, tab.data(), tab.size() * sizeof(float));
  }
  if (m->has_table) {
    std::vector<float> tab;
    (void)sh_nmax;  // the full n_max = 10 table (stride 12, two rows of a node pair = six aligned float4), or -- for a model with
                    // orders above 10, whose shapelet components all run the runtime-order path -- the n_max = 20 one (stride 24)
    glh::build_shapelet_table(m->shp_big ? SH_CAPB : SH_CAP, tab, &m->shp_stride);
    ok = ok && up((void**)&m->d_shp_tab, tab.data(), tab.size() * sizeof(float));
  }
  m->has_post = grid->psf != nullptr || grid->supersample != 1;
  if (m->has_post) {
    // flat = flip(psf) cross-correlated with SAME padding (tf/simulator.py:62-70,145-147), then box(ss)/ss^2 pooling
    const int kh = grid->psf ? grid->psf_h : 1, kw = grid->psf ? grid->psf_w : 1, ss = grid->supersample;
    m->psf_h = kh;
    m->psf_w = kw;
    m->KH = kh + ss - 1;
    m->KW = kw + ss - 1;
    m->pad_t = (kh - 1) / 2;
    m->pad_l = (kw - 1) / 2;
    std::vector<double> keff((size_t)m->KH * m->KW, 0.0);
    for (int u = 0; u < kh; ++u)
      for (int v = 0; v < kw; ++v) {
        double f = grid->psf ? (double)grid->psf[(size_t)(kh - 1 - u) * kw + (kw - 1 - v)] : 1.0;
        for (int a2 = 0; a2 < ss; ++a2)
          for (int c2 = 0; c2 < ss; ++c2) keff[(size_t)(u + a2) * m->KW + (v + c2)] += f / (double)(ss * ss);
      }
    std::vector<float> kf(keff.begin(), keff.end());
    ok = ok && up((void**)&m->d_psf, kf.data(), sizeof(float) * kf.size());
    // plans of the register-blocked pair kernel: forward = one class (stride ss, Keff), transpose = ss^2 residue classes
    // (stride 1, the class's decimated and flipped sub-kernel); rows padded to a multiple of four taps
    if (env_int("GIGALENS_HIP_CORR_PAIR", 1) && ss <= 2 && m->KW <= 32 && m->KH <= 64) {
      const int Hs = m->height, Ws = m->width, H = Hs / ss, W = Ws / ss;
      std::vector<float> kbuf;
      auto pad4 = [](int n) { return std::max(4, (n + 3) & ~3); };
      {
        gl_model::CorrPlan& f = m->corr_fwd;
        f.KWP = pad4(m->KW);
        f.ST = ss;
        CorrClass c{};
        c.koff = 0; c.KH = m->KH; c.pt = m->pad_t; c.pl = m->pad_l; c.Ho = H; c.Wo[0] = W; c.oo_r = 0; c.oo_c[0] = 0;
        kbuf.assign((size_t)m->KH * f.KWP, 0.f);
        for (int u = 0; u < m->KH; ++u)
          for (int v = 0; v < m->KW; ++v) kbuf[(size_t)u * f.KWP + v] = (float)keff[(size_t)u * m->KW + v];
        f.args.n_class = 1; f.args.ncj = 1; f.args.Hi = Hs; f.args.Wi = Ws; f.args.Hout = H; f.args.Wout = W; f.args.os = 1;
        f.args.cls[0] = c;
        f.max_Ho = H; f.max_Wo = W; f.max_KH = m->KH; f.ok = true;
      }
      {
        // transpose: row class pi = (i + pt) mod ss -> one workgroup family; its ss column classes pj share a thread.  Class
        // (pi, pj): outputs i = ss n + ri, j = ss q + rj;  gS = sum_{t, s} gP[n + t - padT][q + s - padL] Kf[t][s] with the
        // flipped decimated kernel Kf[t][s] = Keff[pi + ss (A - 1 - t)][pj + ss (C - 1 - s)].  The column classes' left
        // paddings differ by at most one: they are levelled to the largest by shifting the kernel right.
        gl_model::CorrPlan& g = m->corr_bwd;
        g.ST = 1;
        g.args.n_class = ss; g.args.ncj = ss; g.args.Hi = H; g.args.Wi = W; g.args.Hout = Hs; g.args.Wout = Ws; g.args.os = ss;
        int Cn[4], rjn[4], pln[4], max_pl = -(1 << 30), width = 1;
        for (int pj = 0; pj < ss; ++pj) {
          Cn[pj] = m->KW > pj ? (m->KW - pj + ss - 1) / ss : 0;
          rjn[pj] = ((pj - m->pad_l) % ss + ss) % ss;
          pln[pj] = (Cn[pj] - 1) - (rjn[pj] + m->pad_l - pj) / ss;
          max_pl = std::max(max_pl, pln[pj]);
        }
        for (int pj = 0; pj < ss; ++pj) width = std::max(width, Cn[pj] + (max_pl - pln[pj]));
        g.KWP = pad4(width);
        // column classes ordered by their output offset, so Wo[0] is the largest
        int order[4];
        for (int pj = 0; pj < ss; ++pj) order[rjn[pj]] = pj;
        for (int pi = 0; pi < ss; ++pi) {
          const int A = m->KH > pi ? (m->KH - pi + ss - 1) / ss : 0;
          const int ri = ((pi - m->pad_t) % ss + ss) % ss;
          CorrClass c{};
          c.koff = (int)kbuf.size();
          c.KH = A;
          c.pt = (A - 1) - (ri + m->pad_t - pi) / ss;
          c.pl = max_pl;
          c.Ho = ri < Hs ? (Hs - ri + ss - 1) / ss : 0;
          c.oo_r = ri;
          kbuf.resize(kbuf.size() + (size_t)A * ss * g.KWP, 0.f);
          for (int jj = 0; jj < ss; ++jj) {
            const int pj = order[jj], C = Cn[pj], sh = max_pl - pln[pj];
            c.Wo[jj] = rjn[pj] < Ws ? (Ws - rjn[pj] + ss - 1) / ss : 0;
            c.oo_c[jj] = rjn[pj];
            for (int t = 0; t < A; ++t)
              for (int q = 0; q < C; ++q)
                kbuf[(size_t)c.koff + ((size_t)t * ss + jj) * g.KWP + sh + q] =
                    (float)keff[(size_t)(pi + ss * (A - 1 - t)) * m->KW + (pj + ss * (C - 1 - q))];
          }
          g.args.cls[pi] = c;
          g.max_Ho = std::max(g.max_Ho, c.Ho); g.max_Wo = std::max(g.max_Wo, c.Wo[0]); g.max_KH = std::max(g.max_KH, c.KH);
        }
        g.ok = true;
      }
      ok = ok && up((void**)&m->d_corr_k, kbuf.data(), sizeof(float) * kbuf.size());
      m->corr_fwd.args.k = m->corr_bwd.args.k = m->d_corr_k;
    }
  }
  if (!ok) {
    gl_model_destroy(m);
    return fail(GL_ENOMEM, "device allocation / upload failed in gl_model_create");
  }
  if (m->has_user) {  // the interpreter kernel with the user's bodies inside, compiled now (a few seconds, once per model)
    m->tile = 2;
    m->tile_grad = 2;
    if (int rc = compile_user_model(m, bodies, n_bodies)) {
      gl_model_destroy(m);
      return rc;
    }
  }
  *out = m;
  return GL_OK;
}

int gl_model_set_timing(gl_model* m, int slots) {
  if (!m) return fail(GL_EINVAL, "model is null");
  if (slots < 0 || slots > 65536) return fail(GL_EINVAL, "timing slots %d outside [0, 65536]", slots);
  for (hipEvent_t e : m->evs) (void)hipEventDestroy(e);
  m->evs.clear();
  m->timing_slots = 0;
  m->timing_count = 0;
  m->timing_calls = 0;
  m->evs.reserve((size_t)2 * slots);
  for (int i = 0; i < 2 * slots; ++i) {
    hipEvent_t e;
    GL_HIP(hipEventCreate(&e));
    m->evs.push_back(e);
  }
  m->timing_slots = slots;
  return GL_OK;
}

int gl_model_set_timing_stride(gl_model* m, int stride) {
  if (!m) return fail(GL_EINVAL, "model is null");
  if (stride < 1) return fail(GL_EINVAL, "stride must be >= 1");
  m->timing_stride = stride;
  m->timing_calls = 0;
  return GL_OK;
}

int gl_model_last_main_ms(gl_model* m, float* ms) {
  if (!m || !ms) return fail(GL_EINVAL, "null argument");
  if (!m->timing_slots || !m->timing_count) return fail(GL_EINVAL, "no timed main launch on this model");
  const int slot = (int)((m->timing_count.load() - 1) % m->timing_slots);
  GL_HIP(hipEventSynchronize(m->evs[2 * slot + 1]));
  GL_HIP(hipEventElapsedTime(ms, m->evs[2 * slot], m->evs[2 * slot + 1]));
  return GL_OK;
}

int gl_model_timing_drain(gl_model* m, float* ms, int cap, int* n_out) {
  if (!m || !ms || !n_out || cap < 0) return fail(GL_EINVAL, "bad argument");
  *n_out = 0;
  if (!m->timing_slots) return fail(GL_EINVAL, "timing is not enabled on this model");
  const long long count = m->timing_count.load();
  const long long have = std::min<long long>(count, m->timing_slots);
  const long long first = count - have;  // oldest launch still in the ring
  int n = 0;
  for (long long k = first; k < count && n < cap; ++k, ++n) {
    const int slot = (int)(k % m->timing_slots);
    GL_HIP(hipEventSynchronize(m->evs[2 * slot + 1]));
    GL_HIP(hipEventElapsedTime(&ms[n], m->evs[2 * slot], m->evs[2 * slot + 1]));
  }
  *n_out = n;
  m->timing_count = 0;
  m->timing_calls = 0;
  return GL_OK;
}

int gl_model_last_main_kernel(const gl_model* m, char* buf, size_t cap) {
  if (!m || !buf || cap == 0) return fail(GL_EINVAL, "bad argument");
  if (m->last_main_user >= 0) {  // a model with user-written profiles: a run-time compiled kernel (gl_user.hip)
    const int u = m->last_main_user.load();
    if (u >= 16)
      snprintf(buf, cap, "gl_pair_kernel<%d, v2f, 2, KindList<the model's own component list>...> [run-time compiled with the model's user-written profile bodies]", u - 16);
    else
      snprintf(buf, cap, "gl_main_kernel<%d, 2, %s, %d> [run-time compiled with the model's user-written profile bodies]", u,
               m->has_shapelets ? "true" : "false", m->fam);
    return GL_OK;
  }
  if (!m->last_main_fn) return fail(GL_EINVAL, "no main kernel has been launched on this model yet");
  const char* name = hipKernelNameRefByPtr(m->last_main_fn.load(), nullptr);
  if (!name) return fail(GL_ELAUNCH, "hipKernelNameRefByPtr returned no name");
  snprintf(buf, cap, "%s", name);
  return GL_OK;
}

int gl_model_launch_shape(const gl_model* m, int B, int* chunk_px, int* n_chunks, int* row_floats, size_t* partial_offset_bytes) {
  if (!m || B < 1) return fail(GL_EINVAL, "bad argument");
  int chunk = 0, nc = 0;
  chunking(m, B, &chunk, &nc);
  const Workspace w = carve(m, B, nullptr);
  int tail_from, n_rows;
  tail_plan(m, B, nc, &tail_from, &n_rows);
  if (chunk_px) *chunk_px = chunk;
  if (n_chunks) *n_chunks = n_rows;  // partial rows a sample owns (the chunks, or the workgroups of a tail sample if more: tail_plan)
  if (row_floats) *row_floats = m->A;
  if (partial_offset_bytes) *partial_offset_bytes = (size_t)((const char*)w.partial - (const char*)nullptr);
  return GL_OK;
}

void gl_model_destroy(gl_model* m) {
  if (!m) return;
  if (m->user_module) (void)hipModuleUnload(m->user_module);
  if (m->user_point_module) (void)hipModuleUnload(m->user_point_module);
  for (hipEvent_t e : m->evs) (void)hipEventDestroy(e);
  if (m->d_comps) (void)hipFree(m->d_comps);
  if (m->d_gx) (void)hipFree(m->d_gx);
  if (m->d_gy) (void)hipFree(m->d_gy);
  if (m->d_pix) (void)hipFree(m->d_pix);
  if (m->d_shp_tab) (void)hipFree(m->d_shp_tab);
  if (m->d_nfw_tab) (void)hipFree(m->d_nfw_tab);
  if (m->d_corr_k) (void)hipFree(m->d_corr_k);
  if (m->d_psf) (void)hipFree(m->d_psf);
  if (m->d_pos) (void)hipFree(m->d_pos);
  if (m->d_fam) (void)hipFree(m->d_fam);
  if (m->d_zcols) (void)hipFree(m->d_zcols);
  if (m->d_src) (void)hipFree(m->d_src);
  if (m->d_const) (void)hipFree(m->d_const);
  if (m->d_lin_cols) (void)hipFree(m->d_lin_cols);
  for (auto& sv : m->series) {
    if (sv.coef) (void)hipFree((void*)sv.coef);
    if (sv.hcoef) (void)hipFree((void*)sv.hcoef);
  }
  if (m->d_series) (void)hipFree(m->d_series);
  if (m->d_cats) (void)hipFree(m->d_cats);
  if (m->d_gal_table) (void)hipFree(m->d_gal_table);
  if (m->d_gal_static) (void)hipFree(m->d_gal_static);
  delete m;
}

int gl_model_num_params(const gl_model* m) { return m ? m->P : fail(GL_EINVAL, "model is null"); }
int gl_model_param_offset(const gl_model* m, int component) {
  if (!m) return fail(GL_EINVAL, "model is null");
  if (component < 0 || component >= (int)m->comps.size()) return fail(GL_EINVAL, "component index out of range");
  return m->comps[component].p_off;
}
int64_t gl_model_num_pixels(const gl_model* m) { return m ? m->N : fail(GL_EINVAL, "model is null"); }

size_t gl_workspace_bytes(const gl_model* m, int B) {
  if (!m || B <= 0) return 0;
  return carve(m, B, nullptr).bytes;
}

int gl_simulate_fwd(const gl_model* m, const float* params, int B, float* img, void* workspace,
                    size_t workspace_bytes, void* hip_stream) {
  int rc = check_call(m, params, B, workspace, workspace_bytes);
  if (rc) return rc;
  if (!img) return fail(GL_EINVAL, "img is null");
  hipStream_t stream = (hipStream_t)hip_stream;
  Workspace w = carve(m, B, workspace);
  int chunk, n_chunks;
  chunking(m, B, &chunk, &n_chunks);
  if ((rc = run_prep(m, params, B, w, stream))) return rc;
  MainArgs a = base_args(m, w, chunk);
  if ((rc = run_order(m, B, w, &a, stream))) return rc;
  if (m->has_post) {
    if ((rc = render_ss(m, a, B, n_chunks, w, stream))) return rc;
    return post_fwd(m, B, w.img_ss, img, stream);
  }
  if (m->d_pix) GL_HIP(hipMemsetAsync(img, 0, sizeof(float) * (size_t)B * m->height * m->width, stream));
  a.img = img;
  return launch_main<IMG_FWD>(m, a, B, n_chunks, stream);
}

int gl_simulate_parts_fwd(const gl_model* m, const float* params, int B, unsigned parts, float* img, void* workspace,
                          size_t workspace_bytes, void* hip_stream) {
  int rc = check_call(m, params, B, workspace, workspace_bytes);
  if (rc) return rc;
  if (!img) return fail(GL_EINVAL, "img is null");
  if (parts == 0 || parts > 7u) return fail(GL_EINVAL, "parts must be a non-empty subset of {1,2,4}");
  hipStream_t stream = (hipStream_t)hip_stream;
  Workspace w = carve(m, B, workspace);
  int chunk, n_chunks;
  chunking(m, B, &chunk, &n_chunks);
  if ((rc = run_prep(m, params, B, w, stream))) return rc;
  MainArgs a = base_args(m, w, chunk);
  a.parts = parts;
  if ((rc = run_order(m, B, w, &a, stream))) return rc;
  if (m->has_post) {
    if ((rc = render_ss(m, a, B, n_chunks, w, stream))) return rc;
    return post_fwd(m, B, w.img_ss, img, stream);
  }
  if (m->d_pix) GL_HIP(hipMemsetAsync(img, 0, sizeof(float) * (size_t)B * m->height * m->width, stream));
  a.img = img;
  return launch_main<IMG_FWD>(m, a, B, n_chunks, stream);
}

int gl_simulate_bwd(const gl_model* m, const float* params, const float* grad_img, int B, float* grad_params,
                    void* workspace, size_t workspace_bytes, void* hip_stream) {
  int rc = check_call(m, params, B, workspace, workspace_bytes);
  if (rc) return rc;
  if (!grad_img || !grad_params) return fail(GL_EINVAL, "grad_img / grad_params is null");
  hipStream_t stream = (hipStream_t)hip_stream;
  Workspace w = carve(m, B, workspace);
  int chunk, n_chunks;
  chunking(m, B, &chunk, &n_chunks);
  if ((rc = run_prep(m, params, B, w, stream))) return rc;
  MainArgs a = base_args(m, w, chunk);
  a.gimg = grad_img;
  if (m->has_post) {
    if ((rc = post_bwd(m, B, grad_img, w.img_ss, stream))) return rc;
    a.gimg = w.img_ss;
    a.out_scale = 1.f;
  }
  if ((rc = run_order(m, B, w, &a, stream))) return rc;
  if ((rc = launch_main<IMG_BWD>(m, a, B, n_chunks, stream))) return rc;
  return run_finalize(m, params, B, n_chunks, w, nullptr, nullptr, grad_params, stream);
}

int gl_loglike_fwd_bwd(const gl_model* m, const float* params, const float* obs, const float* err_or_null,
                       const float* mask_or_null, float bg_rms, float exp_time, int B, float* loglike, float* chi2,
                       float* grad_params_or_null, void* workspace, size_t workspace_bytes, void* hip_stream) {
  int rc = check_call(m, params, B, workspace, workspace_bytes);
  if (rc) return rc;
  if (!obs || !loglike || !chi2) return fail(GL_EINVAL, "obs / loglike / chi2 is null");
  hipStream_t stream = (hipStream_t)hip_stream;
  Workspace w = carve(m, B, workspace);
  int chunk, n_chunks;
  chunking(m, B, &chunk, &n_chunks);
  if ((rc = run_prep(m, params, B, w, stream))) return rc;
  const float* extra = nullptr;
  int use_partial = 1;
  if ((rc = run_likelihood(m, B, w, chunk, n_chunks, obs, err_or_null, mask_or_null, bg_rms, exp_time,
                           grad_params_or_null != nullptr, stream, &extra, &use_partial, &n_chunks)))
    return rc;
  return run_finalize(m, params, B, n_chunks, w, loglike, chi2, grad_params_or_null, stream, nullptr, nullptr, nullptr,
                      1.f, extra, use_partial);
}

int gl_model_num_linear(const gl_model* m) { return m ? (int)m->lin_cols.size() : fail(GL_EINVAL, "model is null"); }
int gl_model_linear_column(const gl_model* m, int k) {
  if (!m) return fail(GL_EINVAL, "model is null");
  if (k < 0 || k >= (int)m->lin_cols.size()) return fail(GL_EINVAL, "linear coefficient index out of range");
  return m->lin_cols[k];
}

size_t gl_lstsq_workspace_bytes(const gl_model* m, int B) {
  if (!m || B <= 0) return 0;
  return carve_lstsq(m, B, nullptr, align_up(carve(m, B, nullptr).bytes, 256)).bytes;
}

int gl_lstsq_solve_flags(const gl_model* m, int B, size_t* offset_bytes) {
  if (!m || B <= 0 || !offset_bytes) return fail(GL_EINVAL, "bad argument");
  if ((int)m->lin_cols.size() > LS_LDS_MAXN || !m->lstsq_chol)
    return fail(GL_EUNSUPPORTED, "no Cholesky attempt for this model: every system goes through the eigenvalue solve");
  const LstsqWs lw = carve_lstsq(m, B, nullptr, align_up(carve(m, B, nullptr).bytes, 256));
  *offset_bytes = (size_t)((const char*)lw.todo - (const char*)nullptr);
  return GL_OK;
}

int gl_lstsq_fwd(const gl_model* m, const float* params, const float* obs, const float* err, int B, unsigned parts,
                 float* coeffs_or_null, float* stacked_or_null, float* image_or_null, void* workspace,
                 size_t workspace_bytes, void* hip_stream) {
  if (!m) return fail(GL_EINVAL, "model is null");
  if (m->has_user && !m->user_fn[IMG_BASIS]) return fail(GL_EUNSUPPORTED, "the basis-stack kernel of this model with user-written profiles was not built");
  const int D = (int)m->lin_cols.size();
  if (D == 0) return fail(GL_EINVAL, "the model has no linear (light amplitude) coefficients");
  if (!params || !workspace) return fail(GL_EINVAL, "params / workspace is null");
  if (B <= 0 || B > 65535) return fail(GL_EINVAL, "batch size %d outside [1, 65535]", B);
  if ((int)m->cats.size() != m->n_scaled) return fail(GL_EINVAL, "GL_SCALED component without a catalogue");
  if (m->n_series_set != m->n_series) return fail(GL_EINVAL, "GL_SERIES component without a coefficient field");
  const bool solve = coeffs_or_null || image_or_null;
  if (solve && D > LS_MAXN)  // the basis stack alone (return_stacked) is served at any depth
    return fail(GL_EUNSUPPORTED, "%d linear coefficients exceed the %d the solve serves", D, LS_MAXN);
  if (solve && (!obs || !err)) return fail(GL_EINVAL, "obs / err_map are required to solve for the coefficients");
  if (!solve && !stacked_or_null) return fail(GL_EINVAL, "nothing to compute");
  if (!(parts & (GL_PART_LENS_LIGHT | GL_PART_SOURCE_LIGHT)) || parts > 7u) return fail(GL_EINVAL, "bad parts");
  const size_t need = gl_lstsq_workspace_bytes(m, B);
  if (workspace_bytes < need) return fail(GL_ENOMEM, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
  hipStream_t stream = (hipStream_t)hip_stream;
  Workspace w = carve(m, B, workspace);
  LstsqWs lw = carve_lstsq(m, B, workspace, align_up(w.bytes, 256));
  int chunk, n_chunks, rc;
  chunking(m, B, &chunk, &n_chunks);
  const int HW = (m->height / m->supersample) * (m->width / m->supersample);
  // unit amplitudes -> derived constants -> basis stack
  hipLaunchKernelGGL(gl_unit_amplitudes_kernel, dim3((unsigned)(((long long)B * m->P + 255) / 256)), dim3(256), 0,
                     stream, params, m->P, B, m->d_lin_cols, D, w.params);
  GL_HIP(hipGetLastError());
  if ((rc = run_prep(m, w.params, B, w, stream))) return rc;
  MainArgs a = base_args(m, w, chunk);
  a.parts = parts | GL_PART_LENS_LIGHT | GL_PART_SOURCE_LIGHT;
  a.n_lin = D;
  if ((rc = run_order(m, B, w, &a, stream))) return rc;
  // One shapelet source as the only light component, no PSF / supersampling / pixel list, the stack not asked for: the normal
  // matrix straight from the bases (gl_shp_normal_kernel), no stack in HBM; a fitted image is rendered from the solved amplitudes
  const bool fused = solve && !stacked_or_null && !m->has_post && !m->d_pix && m->shp_kernel && m->static_id == ST_EPLSHEAR_SHAPELETS &&
                     m->n_ll == 0 && m->comps.back().iparam <= SH_CAP && D == sh_layers(m->comps.back().iparam) &&
                     (a.parts & (GL_PART_DEFLECT | GL_PART_SOURCE_LIGHT)) == (GL_PART_DEFLECT | GL_PART_SOURCE_LIGHT) &&
                     m->lstsq_fused;
  if (fused) {
    lw.n_chunks = lw.n_chunks_f;
    const bool interp = (m->comps.back().flags & GL_FLAG_SHAPELETS_INTERPOLATE) != 0;
    constexpr int NPS = SH_SQ / 2;
    ShpNormalArgs sn{obs, err, lw.partial, D, lw.Dp};
    MainArgs fa = a;
    // (table mode on a whole image: 8 x 16 blocks of the image as wave-tiles, like gl_shp_kernel)
    fa.blk_w = (interp && m->shp_blocked && m->width % 16 == 0 && m->height % 8 == 0 && (long long)a.N == (long long)m->width * m->height) ? m->width : 0;
    const int nt = (D + 1 + 15) / 16;
    const size_t red = (size_t)(nt * (nt + 1) / 2 * 256 + 8) * sizeof(float);
    const size_t sh = (size_t)((m->D + 3) & ~3) * sizeof(float) +
                      std::max((size_t)4 * shn_wave_floats(NPS) * sizeof(float), red);
    const dim3 grid(lw.n_chunks, B), block(WG);
#define GL_SHPN(NT_, I_)                                                                                        \
  do {                                                                                                          \
    /* (table mode, five tile rows: 17 spilled VGPRs under the 128-register budget of four waves per SIMD since the live-pixel \
       list of round 4 -- three waves there) */                                                                  \
    m->last_main_fn = (const void*)&gl_shp_normal_kernel<NT_, ((I_ && NT_ < 5) ? 4 : 3), L_EplShear, NPS, I_>;              \
    hipLaunchKernelGGL((gl_shp_normal_kernel<NT_, ((I_ && NT_ < 5) ? 4 : 3), L_EplShear, NPS, I_>), grid, block, sh, stream, fa, sn);  \
  } while (0)
    if (nt == 1) { if (interp) GL_SHPN(1, true); else GL_SHPN(1, false); }
    else if (nt == 2) { if (interp) GL_SHPN(2, true); else GL_SHPN(2, false); }
    else if (nt == 3) { if (interp) GL_SHPN(3, true); else GL_SHPN(3, false); }
    else if (nt == 4) { if (interp) GL_SHPN(4, true); else GL_SHPN(4, false); }
    else { if (interp) GL_SHPN(5, true); else GL_SHPN(5, false); }
#undef GL_SHPN
    GL_HIP(hipGetLastError());
  }
  float* target = m->has_post ? lw.stack_ss : lw.stack;
  if (!fused) {
  if (m->d_pix) GL_HIP(hipMemsetAsync(target, 0, sizeof(float) * (size_t)B * D * m->height * m->width, stream));
  a.img = target;
  if ((rc = launch_main<IMG_BASIS>(m, a, B, n_chunks, stream))) return rc;
  if (m->has_post && (rc = post_fwd(m, B * D, lw.stack_ss, lw.stack, stream, 1.f))) return rc;  // no det(T) here (:226-240)
  if (stacked_or_null)
    GL_HIP(hipMemcpyAsync(stacked_or_null, lw.stack, sizeof(float) * (size_t)B * D * HW, hipMemcpyDeviceToDevice, stream));
  }  // !fused
  if (!solve) return GL_OK;
  NormalArgs na{};
  na.stack = lw.stack;
  na.obs = obs;
  na.err = err;
  na.D = D;
  na.Dp = lw.Dp;
  na.HW = HW;
  na.chunk = lw.chunk;
  na.n_chunks = lw.n_chunks;
  na.partial = lw.partial;
  if (fused) {
    // the partials are already there
  } else if (D + 1 <= LS_SMALL)
    hipLaunchKernelGGL((gl_normal_small_kernel<LS_SMALL>), dim3(lw.n_chunks, B), dim3(256), 0, stream, na);
  else if (D + 1 > LS_MAXD) {  // more than five tile rows: super-block pairs (gl_normal_pair_kernel)
    const int vec_ok = (HW % 4 == 0) && ((uintptr_t)obs % 16 == 0) && ((uintptr_t)err % 16 == 0) && ((uintptr_t)lw.stack % 16 == 0);
    const int n_sb = (D + 1 + 16 * LS_SB - 1) / (16 * LS_SB);
    const dim3 grid(lw.n_chunks, B, n_sb * (n_sb + 1) / 2), block(256);
    if (vec_ok) hipLaunchKernelGGL((gl_normal_pair_kernel<true>), grid, block, 0, stream, na);
    else hipLaunchKernelGGL((gl_normal_pair_kernel<false>), grid, block, 0, stream, na);
  } else {
    // 16-byte loads need every channel row, obs and err on a 16-byte pitch
    const int vec_ok = (HW % 4 == 0) && ((uintptr_t)obs % 16 == 0) && ((uintptr_t)err % 16 == 0) && ((uintptr_t)lw.stack % 16 == 0);
    const dim3 grid(lw.n_chunks, B), block(256);
#define GL_NORMAL_MFMA(NT_)                                                                          \
  if (vec_ok) hipLaunchKernelGGL((gl_normal_mfma_kernel<NT_, true>), grid, block, 0, stream, na);   \
  else hipLaunchKernelGGL((gl_normal_mfma_kernel<NT_, false>), grid, block, 0, stream, na)
    switch ((D + 1 + 15) / 16) {
      case 1: GL_NORMAL_MFMA(1); break;
      case 2: GL_NORMAL_MFMA(2); break;
      case 3: GL_NORMAL_MFMA(3); break;
      case 4: GL_NORMAL_MFMA(4); break;
      default: GL_NORMAL_MFMA(5); break;
    }
#undef GL_NORMAL_MFMA
  }
  GL_HIP(hipGetLastError());
  float* coeffs = coeffs_or_null ? coeffs_or_null : lw.coeffs;
  int n_sum = lw.n_chunks;
  if (n_sum > 8) {  // many chunks (small batches): reduce them with the whole chip first
    hipLaunchKernelGGL(gl_partial_sum_kernel, dim3((lw.Dp * lw.Dp + 255) / 256, B), dim3(256), 0, stream, lw.partial,
                       lw.n_chunks, lw.Dp * lw.Dp);
    GL_HIP(hipGetLastError());
    n_sum = 1;
  }
  const int* todo = nullptr;
  if (D <= LS_LDS_MAXN && m->lstsq_chol) {  // the inverse when the pseudo-inverse's cut is provably idle (gl_chol_solve_kernel)
    const int nb = D + 1 <= 64 ? 4 : D + 1 <= 80 ? 5 : 8;
    const size_t sm = sizeof(float) * ((size_t)(D + 2) * (16 * nb + 1) + 4);
    if (sm > 64 * 1024) {
      static bool raised = false;
      if (!raised) {
        GL_HIP(hipFuncSetAttribute((const void*)&gl_chol_solve_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
      }
    }
#define GL_CHOL(NB_) hipLaunchKernelGGL((gl_chol_solve_kernel<NB_>), dim3(B), dim3(256), sm, stream, lw.partial, lw.n_chunks, n_sum, D, \
                                        lw.Dp, 1e-6f, coeffs, lw.todo)
    if (nb == 4) GL_CHOL(4); else if (nb == 5) GL_CHOL(5); else GL_CHOL(8);
#undef GL_CHOL
    GL_HIP(hipGetLastError());
    todo = lw.todo;
  }
  if (D <= LS_LDS_MAXN) {  // A and V in LDS: up to 129 KB of the CU's 160 (above 64 KB the kernel has to be told once)
    const size_t sm = sizeof(float) * ((size_t)2 * D * (D | 1) + 8 * D + 8);
    if (sm > 64 * 1024) {
      static bool raised = false;
      if (!raised) {
        GL_HIP(hipFuncSetAttribute((const void*)&gl_eigh_solve_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
      }
    }
    hipLaunchKernelGGL((gl_eigh_solve_kernel<2, false>), dim3(B), dim3(64), sm, stream, lw.partial, lw.n_chunks, n_sum, D, lw.Dp,
                       1e-6f, coeffs, (float*)nullptr, todo);
  } else {  // the two matrices in the workspace (L2), the vectors in LDS; four registers hold the tridiagonal
    const size_t sm = sizeof(float) * ((size_t)8 * D + 8);
    hipLaunchKernelGGL((gl_eigh_solve_kernel<4, true>), dim3(B), dim3(64), sm, stream, lw.partial, lw.n_chunks, n_sum, D, lw.Dp,
                       1e-6f, coeffs, lw.mats, todo);
  }
  GL_HIP(hipGetLastError());
  if (image_or_null && fused) {
    // image = sum_d coeffs_d basis_d = the ordinary render with the solved amplitudes in their parameter columns (no det(T):
    // the stack carries none, tf/simulator.py:226-240)
    hipLaunchKernelGGL(gl_set_amplitudes_kernel, dim3((unsigned)(((long long)B * m->P + 255) / 256)), dim3(256), 0, stream,
                       params, m->P, B, m->d_lin_cols, D, coeffs, w.params);
    GL_HIP(hipGetLastError());
    if ((rc = run_prep(m, w.params, B, w, stream))) return rc;
    MainArgs ia = base_args(m, w, chunk);
    ia.parts = a.parts;
    ia.order = a.order;
    ia.img = image_or_null;
    ia.out_scale = 1.f;
    if ((rc = launch_main<IMG_FWD>(m, ia, B, n_chunks, stream))) return rc;
  } else if (image_or_null) {
    hipLaunchKernelGGL(gl_combine_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, stream, lw.stack, coeffs, D, HW,
                       image_or_null);
    GL_HIP(hipGetLastError());
  }
  return GL_OK;
}

// process-lifetime table for plugin-level table-mode shapelets (n_max = cap), built on first use
static int point_shapelet_table(float** tab_out, int* stride_out) {
  static float* s_tab = nullptr;
  static int s_stride = 0;
  if (!s_tab) {
    std::vector<float> tab;
    glh::build_shapelet_table(GL_SHAPELETS_NMAX_CAP, tab, &s_stride);
    float* p = nullptr;
    GL_HIP(hipMalloc((void**)&p, tab.size() * sizeof(float)));
    GL_HIP(hipMemcpy(p, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    s_tab = p;
  }
  *tab_out = s_tab;
  *stride_out = s_stride;
  return GL_OK;
}

static int series_precompute(bool hessian, int base_kind, int n_galaxies, const int32_t scale_col[3],
                             const float* table_dev, const float* scales, int n_scales, int order, const float* x_dev,
                             const float* y_dev, int64_t n_pts, float* coeffs_dev, void* hip_stream) {
  if (!scale_col || !table_dev || !scales || !x_dev || !y_dev || !coeffs_dev) return fail(GL_EINVAL, "null argument");
  if (base_kind != GL_DPIS && base_kind != GL_DPIE && base_kind != GL_DPIEP)
    return fail(GL_EUNSUPPORTED, "series expansion over profile kind %d is not built (dPIS, dPIE, dPIEP are)", base_kind);
  if (n_galaxies <= 0 || n_pts <= 0 || n_scales < 1 || n_scales > 3) return fail(GL_EINVAL, "bad sizes");
  if (order < 0 || order > SERIES_MAX_ORDER) return fail(GL_EINVAL, "order %d outside [0, %d]", order, SERIES_MAX_ORDER);
  for (int k = 0; k < 3; ++k)
    if (scale_col[k] >= n_scales) return fail(GL_EINVAL, "scale_col[%d]=%d outside the %d scales", k, scale_col[k], n_scales);
  if (scale_col[2] < 0) return fail(GL_EINVAL, "the series variable r_cut must be a scaled parameter");
  ScaledDesc sd{base_kind, n_galaxies, {scale_col[0], scale_col[1], scale_col[2]}};
  float s[3] = {1.f, 1.f, 1.f};
  for (int k = 0; k < n_scales; ++k) s[k] = scales[k];
  dim3 grid((unsigned)((n_pts + 63) / 64)), block(64);
  hipStream_t stream = (hipStream_t)hip_stream;
#define GL_SERIES_LAUNCH(KERNEL, NN) \
  hipLaunchKernelGGL((KERNEL<NN>), grid, block, 0, stream, sd, table_dev, s[0], s[1], s[2], order, x_dev, y_dev, \
                     (long long)n_pts, coeffs_dev)
  if (hessian) {
    if (order <= 3) GL_SERIES_LAUNCH(gl_series_hessian_precompute_kernel, 3);
    else GL_SERIES_LAUNCH(gl_series_hessian_precompute_kernel, 5);
  } else {
    if (order <= 3) GL_SERIES_LAUNCH(gl_series_precompute_kernel, 3);
    else GL_SERIES_LAUNCH(gl_series_precompute_kernel, 5);
  }
#undef GL_SERIES_LAUNCH
  GL_HIP(hipGetLastError());
  return GL_OK;
}

int gl_series_precompute(int base_kind, int n_galaxies, const int32_t scale_col[3], const float* table_dev,
                         const float* scales, int n_scales, int order, const float* x_dev, const float* y_dev,
                         int64_t n_pts, float* coeffs_dev, void* hip_stream) {
  return series_precompute(false, base_kind, n_galaxies, scale_col, table_dev, scales, n_scales, order, x_dev, y_dev,
                           n_pts, coeffs_dev, hip_stream);
}

int gl_series_precompute_hessian(int base_kind, int n_galaxies, const int32_t scale_col[3], const float* table_dev,
                                 const float* scales, int n_scales, int order, const float* x_dev, const float* y_dev,
                                 int64_t n_pts, float* coeffs_dev, void* hip_stream) {
  return series_precompute(true, base_kind, n_galaxies, scale_col, table_dev, scales, n_scales, order, x_dev, y_dev,
                           n_pts, coeffs_dev, hip_stream);
}

int gl_series_hessian_eval(const float* coeffs_dev, int order, int64_t n_pts, int B, const float* theta_E,
                           const float* r_cut, float r0, float* out, void* hip_stream) {
  if (!coeffs_dev || !theta_E || !r_cut || !out) return fail(GL_EINVAL, "null argument");
  if (order < 0 || order > SERIES_MAX_ORDER || n_pts <= 0 || B <= 0) return fail(GL_EINVAL, "bad sizes");
  const long long total = (long long)n_pts * B;
  hipLaunchKernelGGL(gl_series_fields_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)hip_stream, coeffs_dev, 3, order, (long long)n_pts, B, theta_E, r_cut, r0, out);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

int gl_model_set_series_hessian(gl_model* m, int component, const float* coeffs_dev) {
  if (!m || !coeffs_dev) return fail(GL_EINVAL, "null argument");
  if (component < 0 || component >= m->n_lens || m->comps[component].kind != K_SERIES)
    return fail(GL_EINVAL, "component %d is not a GL_SERIES lens", component);
  SeriesDev& sv = m->series[(int)m->comps[component].flags];
  if (!sv.coef) return fail(GL_EINVAL, "gl_model_set_series must be called on component %d first", component);
  const size_t bytes = sizeof(float) * 3 * (size_t)(sv.order + 1) * m->N;
  if (!sv.hcoef) {
    float* p = nullptr;
    GL_HIP(hipMalloc((void**)&p, bytes));
    sv.hcoef = p;
  }
  GL_HIP(hipMemcpy((void*)sv.hcoef, coeffs_dev, bytes, hipMemcpyDeviceToDevice));
  GL_HIP(hipMemcpy(m->d_series, m->series.data(), sizeof(SeriesDev) * m->series.size(), hipMemcpyHostToDevice));
  return GL_OK;
}

int gl_model_set_series(gl_model* m, int component, float r0, const float* coeffs_dev) {
  if (!m || !coeffs_dev) return fail(GL_EINVAL, "null argument");
  if (component < 0 || component >= m->n_lens || m->comps[component].kind != K_SERIES)
    return fail(GL_EINVAL, "component %d is not a GL_SERIES lens", component);
  const int slot = (int)m->comps[component].flags;
  SeriesDev& sv = m->series[slot];
  const size_t bytes = sizeof(float) * 2 * (size_t)(sv.order + 1) * m->N;
  if (!sv.coef) {
    float* p = nullptr;
    GL_HIP(hipMalloc((void**)&p, bytes));
    sv.coef = p;
    ++m->n_series_set;
  }
  GL_HIP(hipMemcpy((void*)sv.coef, coeffs_dev, bytes, hipMemcpyDeviceToDevice));
  sv.r0 = r0;
  if (!m->d_series) GL_HIP(hipMalloc((void**)&m->d_series, sizeof(SeriesDev) * m->series.size()));
  GL_HIP(hipMemcpy(m->d_series, m->series.data(), sizeof(SeriesDev) * m->series.size(), hipMemcpyHostToDevice));
  return GL_OK;
}

int gl_series_eval(const float* coeffs_dev, int order, int64_t n_pts, int B, const float* theta_E, const float* r_cut,
                   float r0, float* out0, float* out1, void* hip_stream) {
  if (!coeffs_dev || !theta_E || !r_cut || !out0 || !out1) return fail(GL_EINVAL, "null argument");
  if (order < 0 || order > SERIES_MAX_ORDER || n_pts <= 0 || B <= 0) return fail(GL_EINVAL, "bad sizes");
  const long long total = (long long)n_pts * B;
  hipLaunchKernelGGL(gl_series_eval_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                     coeffs_dev, order, (long long)n_pts, B, theta_E, r_cut, r0, out0, out1);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

int gl_model_set_catalogue(gl_model* m, int component, int base_kind, int n_galaxies, const int32_t scale_col[3],
                           const float* table) {
  if (!m) return fail(GL_EINVAL, "model is null");
  if (component < 0 || component >= m->n_lens || m->comps[component].kind != K_SCALED)
    return fail(GL_EINVAL, "component %d is not a GL_SCALED lens", component);
  if (base_kind != GL_DPIS && base_kind != GL_DPIE && base_kind != GL_DPIEP)
    return fail(GL_EUNSUPPORTED, "ScalingRelation over profile kind %d is not built (dPIS, dPIE, dPIEP are)", base_kind);
  if (n_galaxies <= 0 || !scale_col || !table) return fail(GL_EINVAL, "empty catalogue");
  CompDesc& cd = m->comps[component];
  int used = 0;
  for (int k = 0; k < 3; ++k) {
    if (scale_col[k] >= cd.n_par) return fail(GL_EINVAL, "scale_col[%d]=%d outside the component's %d scales", k, scale_col[k], cd.n_par);
    if (scale_col[k] >= 0) {
      if (used & (1 << scale_col[k])) return fail(GL_EINVAL, "scale column %d used twice", scale_col[k]);
      used |= 1 << scale_col[k];
    }
  }
  if (used != (1 << cd.n_par) - 1) return fail(GL_EINVAL, "every one of the %d scales must drive one of theta_E, r_core, r_cut", cd.n_par);
  gl_model::Cat cat{};
  cat.dev.base_kind = base_kind;
  cat.dev.n_gal = n_galaxies;
  cat.dev.comp = component;
  for (int k = 0; k < 3; ++k) cat.dev.col[k] = scale_col[k];
  cat.table.assign(table, table + (size_t)7 * n_galaxies);
  if (cd.iparam >= 0) m->cats[cd.iparam] = cat;
  else { cd.iparam = (int)m->cats.size(); m->cats.push_back(cat); }
  // rebuild the model-wide galaxy arrays
  std::vector<CatDev> devs;
  std::vector<float> tab, stat;
  int G = 0;
  for (auto& c : m->cats) {
    c.dev.g_off = G;
    G += c.dev.n_gal;
    devs.push_back(c.dev);
    tab.insert(tab.end(), c.table.begin(), c.table.end());
    for (int g = 0; g < c.dev.n_gal; ++g) {
      float ds[DP_NS];
      scaled_static<float>(c.dev.base_kind, c.table.data() + (size_t)7 * g, ds);
      stat.insert(stat.end(), ds, ds + DP_NS);
    }
  }
  m->G = G;
  if (m->d_cats) { (void)hipFree(m->d_cats); m->d_cats = nullptr; }
  if (m->d_gal_table) { (void)hipFree(m->d_gal_table); m->d_gal_table = nullptr; }
  if (m->d_gal_static) { (void)hipFree(m->d_gal_static); m->d_gal_static = nullptr; }
  GL_HIP(hipMalloc((void**)&m->d_cats, sizeof(CatDev) * devs.size()));
  GL_HIP(hipMalloc((void**)&m->d_gal_table, sizeof(float) * tab.size()));
  GL_HIP(hipMalloc((void**)&m->d_gal_static, sizeof(float) * stat.size()));
  GL_HIP(hipMemcpy(m->d_cats, devs.data(), sizeof(CatDev) * devs.size(), hipMemcpyHostToDevice));
  GL_HIP(hipMemcpy(m->d_gal_table, tab.data(), sizeof(float) * tab.size(), hipMemcpyHostToDevice));
  GL_HIP(hipMemcpy(m->d_gal_static, stat.data(), sizeof(float) * stat.size(), hipMemcpyHostToDevice));
  GL_HIP(hipMemcpy(m->d_comps, m->comps.data(), sizeof(CompDesc) * m->comps.size(), hipMemcpyHostToDevice));
  return GL_OK;
}

int gl_scaled_eval(int base_kind, int n_galaxies, const int32_t scale_col[3], const float* table_dev, const float* x,
                   const float* y, int64_t n_pts, int B, int xy_batched, const float* scales, int n_scales,
                   float* out0, float* out1, void* hip_stream) {
  if (!scale_col || !table_dev || !x || !y || !scales || !out0 || !out1) return fail(GL_EINVAL, "null argument");
  if (base_kind != GL_DPIS && base_kind != GL_DPIE && base_kind != GL_DPIEP)
    return fail(GL_EUNSUPPORTED, "ScalingRelation over profile kind %d is not built (dPIS, dPIE, dPIEP are)", base_kind);
  if (n_galaxies <= 0 || n_pts <= 0 || B <= 0 || n_scales < 1 || n_scales > 3) return fail(GL_EINVAL, "bad sizes");
  for (int k = 0; k < 3; ++k)
    if (scale_col[k] >= n_scales) return fail(GL_EINVAL, "scale_col[%d]=%d outside the %d scales", k, scale_col[k], n_scales);
  ScaledDesc sd{base_kind, n_galaxies, {scale_col[0], scale_col[1], scale_col[2]}};
  long long total = (long long)n_pts * B;
  hipLaunchKernelGGL(gl_scaled_point_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)hip_stream, sd, table_dev, x, y, (long long)n_pts, B, xy_batched, scales, n_scales,
                     out0, out1);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

int gl_scaled_hessian(int base_kind, int n_galaxies, const int32_t scale_col[3], const float* table_dev, const float* x,
                      const float* y, int64_t n_pts, int B, int xy_batched, const float* scales, int n_scales,
                      float* out, void* hip_stream) {
  if (!scale_col || !table_dev || !x || !y || !scales || !out) return fail(GL_EINVAL, "null argument");
  if (base_kind != GL_DPIS && base_kind != GL_DPIE && base_kind != GL_DPIEP)
    return fail(GL_EUNSUPPORTED, "ScalingRelation over profile kind %d is not built (dPIS, dPIE, dPIEP are)", base_kind);
  if (n_galaxies <= 0 || n_pts <= 0 || B <= 0 || n_scales < 1 || n_scales > 3) return fail(GL_EINVAL, "bad sizes");
  for (int k = 0; k < 3; ++k)
    if (scale_col[k] >= n_scales) return fail(GL_EINVAL, "scale_col[%d]=%d outside the %d scales", k, scale_col[k], n_scales);
  ScaledDesc sd{base_kind, n_galaxies, {scale_col[0], scale_col[1], scale_col[2]}};
  long long total = (long long)n_pts * B;
  hipLaunchKernelGGL(gl_scaled_hessian_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, (hipStream_t)hip_stream,
                     sd, table_dev, x, y, (long long)n_pts, B, xy_batched, scales, n_scales, out);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

int gl_model_set_positions(gl_model* m, int n_families, const int* family_sizes, const float* x, const float* y,
                           const float* err_x, const float* err_y) {
  if (!m) return fail(GL_EINVAL, "model is null");
  if (n_families <= 0 || !family_sizes || !x || !y || !err_x || !err_y) return fail(GL_EINVAL, "bad position tables");
  std::vector<int> off(n_families + 1, 0);
  for (int f = 0; f < n_families; ++f) {
    if (family_sizes[f] <= 0) return fail(GL_EINVAL, "image family %d is empty", f);
    off[f + 1] = off[f] + family_sizes[f];
  }
  const int J = off[n_families];
  std::vector<float> tab((size_t)4 * J);
  for (int j = 0; j < J; ++j) { tab[j] = x[j]; tab[J + j] = y[j]; tab[2 * J + j] = err_x[j]; tab[3 * J + j] = err_y[j]; }
  if (m->d_pos) { (void)hipFree(m->d_pos); m->d_pos = nullptr; }
  if (m->d_fam) { (void)hipFree(m->d_fam); m->d_fam = nullptr; }
  GL_HIP(hipMalloc((void**)&m->d_pos, tab.size() * sizeof(float)));
  GL_HIP(hipMalloc((void**)&m->d_fam, off.size() * sizeof(int)));
  GL_HIP(hipMemcpy(m->d_pos, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
  GL_HIP(hipMemcpy(m->d_fam, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice));
  m->pos_J = J;
  m->pos_F = n_families;
  return GL_OK;
}

int gl_positions_fwd_bwd(const gl_model* m, const float* params, int B, float* loglike, float* chi2,
                         float* grad_params_or_null, void* workspace, size_t workspace_bytes, void* hip_stream) {
  int rc = check_call(m, params, B, workspace, workspace_bytes);
  if (rc) return rc;
  if (!m->pos_J) return fail(GL_EINVAL, "gl_model_set_positions has not been called on this model");
  if (!loglike || !chi2) return fail(GL_EINVAL, "loglike / chi2 is null");
  hipStream_t stream = (hipStream_t)hip_stream;
  Workspace w = carve(m, B, workspace);
  if ((rc = run_positions(m, params, B, w, grad_params_or_null != nullptr, stream))) return rc;
  GL_HIP(hipMemcpyAsync(loglike, w.pos_ll, sizeof(float) * B, hipMemcpyDeviceToDevice, stream));
  GL_HIP(hipMemcpyAsync(chi2, w.pos_chi2, sizeof(float) * B, hipMemcpyDeviceToDevice, stream));
  if (grad_params_or_null)
    GL_HIP(hipMemcpyAsync(grad_params_or_null, w.pos_grad, sizeof(float) * (size_t)B * m->P, hipMemcpyDeviceToDevice, stream));
  return GL_OK;
}

int gl_profile_hessian(const gl_component* comp, const float* x, const float* y, int64_t n_pts, int B, int xy_batched,
                       const float* params, float* out, void* hip_stream) {
  if (!comp || !x || !y || !params || !out) return fail(GL_EINVAL, "null argument");
  if (n_pts <= 0 || B <= 0) return fail(GL_EINVAL, "n_pts and B must be positive");
  if (!((comp->kind >= GL_EPL && comp->kind <= GL_DPIEP) || comp->kind == GL_NFW_ELLIPSE || comp->kind == GL_TNFW))
    return fail(GL_EINVAL, "kind %d is not a free-standing mass profile", comp->kind);
  CompDesc cd{};
  cd.kind = comp->kind;
  cd.iparam = comp->iparam;
  cd.flags = comp->flags;
  cd.n_par = kind_num_params(comp->kind, comp->iparam);
  if (cd.kind == GL_EPL && cd.iparam <= 0) cd.iparam = 50;
  const long long total = (long long)n_pts * B;
  hipLaunchKernelGGL(gl_profile_hessian_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, (hipStream_t)hip_stream,
                     cd, x, y, (long long)n_pts, B, xy_batched, params, out);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

int gl_lens_maps(const gl_model* m, const float* params, int B, const float* x, const float* y, int64_t n_pts,
                 int xy_batched, float* out, void* hip_stream) {
  if (!m || !params || !out) return fail(GL_EINVAL, "null argument");
  if (m->has_user)  // the kernel below compiled at run time with the user's bodies (Hessians from the duals)
    if (int rc = compile_user_points(m)) return rc;
  if ((x == nullptr) != (y == nullptr)) return fail(GL_EINVAL, "x and y must both be given or both be null");
  if (B <= 0 || n_pts <= 0) return fail(GL_EINVAL, "B and n_pts must be positive");
  if ((int)m->cats.size() != m->n_scaled) return fail(GL_EINVAL, "GL_SCALED component without a catalogue");
  if (!x) {
    if (n_pts != m->N || xy_batched) return fail(GL_EINVAL, "the model grid has %d points and is not batched", m->N);
    x = m->d_gx;
    y = m->d_gy;
    for (const SeriesDev& sv : m->series)
      if (!sv.coef || !sv.hcoef)
        return fail(GL_EINVAL, "GL_SERIES lens without its deflection / Hessian field (gl_model_set_series, gl_model_set_series_hessian)");
  } else if (m->n_series) {
    return fail(GL_EUNSUPPORTED, "a series-expansion lens lives on the model grid only (series_profile.py:76-89): pass x = y = NULL");
  }
  PosArgs a{};
  a.comps = m->d_comps;
  a.n_lens = m->n_lens;
  a.P = m->P;
  a.B = B;
  a.params = params;
  a.cats = m->d_cats;
  a.gal_table = m->d_gal_table;
  a.gal_static = m->d_gal_static;
  a.series = m->d_series;
  const long long total = (long long)n_pts * B;
  if (m->has_user) {
    long long n_pts_ll = (long long)n_pts;
    void* args[] = {&a, &x, &y, &n_pts_ll, &xy_batched, &out};
    GL_HIP(hipModuleLaunchKernel(m->user_point_fn[4], (unsigned)((total + 63) / 64), 1, 1, 64, 1, 1, 0, (hipStream_t)hip_stream, args, nullptr));
    return GL_OK;
  }
  hipLaunchKernelGGL(gl_lens_maps_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, (hipStream_t)hip_stream, a,
                     x, y, (long long)n_pts, xy_batched, out);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

int gl_model_set_prior(gl_model* m, const gl_zcolumn* cols, int d, const float* const_row) {
  if (!m) return fail(GL_EINVAL, "model is null");
  if (d < 0 || (d > 0 && !cols)) return fail(GL_EINVAL, "bad prior column table");
  std::vector<int> src(std::max(m->P, 1), -1);
  std::vector<ZCol> zc(std::max(d, 1));
  for (int k = 0; k < d; ++k) {
    const gl_zcolumn& c = cols[k];
    if (c.param_col < 0 || c.param_col >= m->P) return fail(GL_EINVAL, "z column %d: param_col %d out of range", k, c.param_col);
    if (src[c.param_col] >= 0) return fail(GL_EINVAL, "packed column %d driven by two z columns", c.param_col);
    if (c.bijector < 0 || c.bijector > 2 || c.prior < 0 || c.prior > 3) return fail(GL_EINVAL, "z column %d: unknown bijector/prior", k);
    src[c.param_col] = k;
    zc[k] = ZCol{c.param_col, c.bijector, c.prior, c.a, c.b, c.lo, c.hi, c.log_norm};
  }
  std::vector<float> cr(std::max(m->P, 1), 0.f);
  for (int p = 0; p < m->P; ++p) {
    if (src[p] < 0) {
      if (!const_row) return fail(GL_EINVAL, "packed column %d has neither a z column nor a constant", p);
      cr[p] = const_row[p];
    }
  }
  if (m->d_zcols) { (void)hipFree(m->d_zcols); m->d_zcols = nullptr; }
  if (m->d_src) { (void)hipFree(m->d_src); m->d_src = nullptr; }
  if (m->d_const) { (void)hipFree(m->d_const); m->d_const = nullptr; }
  GL_HIP(hipMalloc((void**)&m->d_zcols, sizeof(ZCol) * zc.size()));
  GL_HIP(hipMalloc((void**)&m->d_src, sizeof(int) * src.size()));
  GL_HIP(hipMalloc((void**)&m->d_const, sizeof(float) * cr.size()));
  GL_HIP(hipMemcpy(m->d_zcols, zc.data(), sizeof(ZCol) * zc.size(), hipMemcpyHostToDevice));
  GL_HIP(hipMemcpy(m->d_src, src.data(), sizeof(int) * src.size(), hipMemcpyHostToDevice));
  GL_HIP(hipMemcpy(m->d_const, cr.data(), sizeof(float) * cr.size(), hipMemcpyHostToDevice));
  m->d_z = d;
  return GL_OK;
}

int gl_logprob_fwd_bwd(const gl_model* m, const float* z, const float* obs, const float* err_or_null,
                       const float* mask_or_null, float bg_rms, float exp_time, int B, float* logprob, float* loglike,
                       float* chi2, float* grad_z_or_null, float chi2_divisor, unsigned terms, void* workspace,
                       size_t workspace_bytes, void* hip_stream) {
  int rc = check_call(m, z, B, workspace, workspace_bytes);
  if (rc) return rc;
  const bool pix = terms & GL_TERM_PIXELS, pos = terms & GL_TERM_POSITIONS;
  if (!pix && !pos) return fail(GL_EINVAL, "terms selects no likelihood term");
  if (pix && !(chi2_divisor > 0.f)) return fail(GL_EINVAL, "chi2_divisor must be positive");
  if (!m->d_zcols) return fail(GL_EINVAL, "gl_model_set_prior has not been called on this model");
  if (pos && !m->pos_J) return fail(GL_EINVAL, "gl_model_set_positions has not been called on this model");
  if ((pix && !obs) || !logprob || !loglike || !chi2) return fail(GL_EINVAL, "obs / logprob / loglike / chi2 is null");
  hipStream_t stream = (hipStream_t)hip_stream;
  Workspace w = carve(m, B, workspace);
  int chunk, n_chunks;
  chunking(m, B, &chunk, &n_chunks);
  int n_comp = (int)m->comps.size();
  if (wave_front_end(m)) {
    const bool ord = order_in_front_end(m, B);
    const size_t rows = prep_row_bytes(m);
    hipLaunchKernelGGL(gl_prep_wave_kernel, dim3((B + 3) / 4 + (ord ? 1 : 0)), dim3(256), rows, stream, m->d_comps, n_comp,
                       (const float*)nullptr, z, m->d_z, (const ZCol*)m->d_zcols, (const int*)m->d_src, (const float*)m->d_const, m->P,
                       B, w.params, w.derived, m->D, m->epl_comp >= 0 ? w.cost : nullptr, m->epl_comp, ord ? w.order : nullptr,
                       rows ? 1 : 0, prep_tail_from(m, B));
  } else
    hipLaunchKernelGGL(gl_zprep_kernel, dim3((B * n_comp + 127) / 128), dim3(128), 0, stream, m->d_comps, n_comp, z,
                       m->d_z, m->d_zcols, m->d_src, m->d_const, m->P, B, w.params, w.derived, m->D,
                       m->epl_comp >= 0 ? w.cost : nullptr, m->epl_comp);
  GL_HIP(hipGetLastError());
  if ((rc = run_galprep(m, w.params, B, w, stream))) return rc;
  const float* extra = nullptr;
  int use_partial = 0;
  // red_chi2 = (red_pix + red_pos) / n_chi  (tf/model.py:150-162)
  const float n_chi = (pix ? 1.f : 0.f) + (pos ? 1.f : 0.f);
  if (pix && (rc = run_likelihood(m, B, w, chunk, n_chunks, obs, err_or_null, mask_or_null, bg_rms, exp_time,
                                  grad_z_or_null != nullptr, stream, &extra, &use_partial, &n_chunks)))
    return rc;
  if (pos && (rc = run_positions(m, w.params, B, w, grad_z_or_null != nullptr, stream))) return rc;
  return run_finalize(m, w.params, B, n_chunks, w, loglike, chi2, nullptr, stream, z, logprob, grad_z_or_null,
                      pix ? 1.0f / (chi2_divisor * n_chi) : 0.f, extra, use_partial, pos,
                      pos ? 1.0f / (2.0f * (float)m->pos_J * n_chi) : 0.f);
}

int gl_profile_eval(const gl_component* comp, const float* x, const float* y, int64_t n_pts, int B, int xy_batched,
                    const float* params, float* out0, float* out1, void* hip_stream) {
  if (!comp || !x || !y || !params || !out0) return fail(GL_EINVAL, "null argument");
  if (n_pts <= 0 || B <= 0) return fail(GL_EINVAL, "n_pts and B must be positive");
  int npar = kind_num_params(comp->kind, comp->iparam);
  if (npar < 0) return fail(GL_EINVAL, "unknown profile kind %d", comp->kind);
  if (comp->kind == GL_SCALED) return fail(GL_EINVAL, "GL_SCALED needs its catalogue: use gl_scaled_eval");
  if (comp->kind == GL_SERIES) return fail(GL_EINVAL, "GL_SERIES needs its coefficient field: use gl_series_eval");
  const bool mass = comp->kind <= GL_DPIEP || comp->kind == GL_NFW_ELLIPSE || comp->kind == GL_TNFW;
  if (mass && !out1) return fail(GL_EINVAL, "out1 is required for mass profiles");
  CompDesc cd{};
  cd.kind = comp->kind;
  cd.iparam = comp->iparam;
  cd.flags = comp->flags;
  cd.n_par = npar;
  if (cd.kind == GL_EPL && cd.iparam <= 0) cd.iparam = 50;
  hipStream_t stream = (hipStream_t)hip_stream;
  float* s_tab = nullptr;
  int s_stride = 0, rc_tab = 0;
  if (cd.kind == GL_SHAPELETS && (cd.iparam < 0 || cd.iparam > GL_SHAPELETS_NMAX_CAP))
    return fail(GL_EUNSUPPORTED, "shapelets n_max=%d outside [0,%d]", cd.iparam, GL_SHAPELETS_NMAX_CAP);
  if (cd.kind == GL_SHAPELETS && (cd.flags & GL_FLAG_SHAPELETS_INTERPOLATE) && (rc_tab = point_shapelet_table(&s_tab, &s_stride)))
    return rc_tab;
  long long total = (long long)n_pts * B;
  hipLaunchKernelGGL(gl_point_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, cd, x, y,
                     (long long)n_pts, B, xy_batched, params, out0, mass ? out1 : nullptr, s_tab, s_stride);
  GL_HIP(hipGetLastError());
  return GL_OK;
}


#ifdef GL_EIGH_STAMPS
int gl_debug_eigh_stamps(long long* out) {
  GL_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(glk::g_eigh_stamps), sizeof(long long) * 8));
  return GL_OK;
}
#endif

int gl_adam_update(float* x, const float* grad, float* m, float* v, int64_t n, float grad_scale, float lr, float beta1,
                   float beta2, float eps, int64_t t, double* t_dev_or_null, void* hip_stream) {
  if (!x || !grad || !m || !v) return fail(GL_EINVAL, "null argument");
  if (n <= 0) return fail(GL_EINVAL, "n must be positive");
  if (!t_dev_or_null && t < 1) return fail(GL_EINVAL, "the step count t starts at 1");
  // t_dev layout: [0] the counter as a double, [1] 8 bytes of launch ticket (zero-initialised by the caller)
  unsigned* ticket = t_dev_or_null ? reinterpret_cast<unsigned*>(t_dev_or_null + 1) : nullptr;
  const float c1 = (float)(1.0 - std::pow((double)beta1, (double)t)), c2 = (float)(1.0 - std::pow((double)beta2, (double)t));
  hipLaunchKernelGGL(gl_adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, x, grad, m,
                     v, (long long)n, grad_scale, lr, beta1, beta2, eps, (double)t, t_dev_or_null, ticket, c1, c2);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

int gl_svi_sample(const float* mu, const float* l_packed, int d, int full_rank, const float* eps, int n, float diag_shift,
                  float* z, void* hip_stream) {
  if (!mu || !l_packed || !eps || !z) return fail(GL_EINVAL, "null argument");
  if (d <= 0 || n <= 0) return fail(GL_EINVAL, "d and n must be positive");
  const long long total = (long long)n * d;
  hipLaunchKernelGGL(gl_svi_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, mu,
                     l_packed, d, full_rank, eps, n, diag_shift, z);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

int gl_svi_grad(const float* l_packed, int d, int full_rank, const float* eps, const float* logp, const float* grad_z, int n,
                float diag_shift, float* buf, void* hip_stream) {
  if (!l_packed || !eps || !logp || !grad_z || !buf) return fail(GL_EINVAL, "null argument");
  if (d <= 0 || n <= 0) return fail(GL_EINVAL, "d and n must be positive");
  const int n_out = 1 + d + (full_rank ? d * (d + 1) / 2 : d);
  hipLaunchKernelGGL(gl_svi_grad_kernel, dim3(n_out), dim3(256), 0, (hipStream_t)hip_stream, l_packed, d, full_rank, eps,
                     logp, grad_z, n, diag_shift, buf);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

int gl_hmc_kick_drift(const float* p_in, const float* grad, float kick, const float* z_in, const float* sigma, float eps, int n,
                      int d, float* p_out, float* z_out, void* hip_stream) {
  if (!p_in || !grad || !z_in || !sigma || !p_out || !z_out) return fail(GL_EINVAL, "null argument");
  if (n <= 0 || d <= 0 || d > 4096) return fail(GL_EINVAL, "n must be positive and d in [1, 4096]");
  hipLaunchKernelGGL(gl_hmc_kick_drift_kernel, dim3(n), dim3(HMC_WG), sizeof(float) * d, (hipStream_t)hip_stream, p_in, grad,
                     kick, z_in, sigma, eps, n, d, p_out, z_out);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

int gl_hmc_accept(float* z, float* grad, float* logp, const float* z_new, const float* grad_new, const float* logp_new,
                  const float* p0, const float* p_new, float kick, const float* scale_tril, const float* uniforms, int n, int d,
                  float* accept_prob, void* hip_stream) {
  if (!z || !grad || !logp || !z_new || !grad_new || !logp_new || !p0 || !p_new || !scale_tril || !uniforms || !accept_prob)
    return fail(GL_EINVAL, "null argument");
  if (n <= 0 || d <= 0 || d > 4096) return fail(GL_EINVAL, "n must be positive and d in [1, 4096]");
  hipLaunchKernelGGL(gl_hmc_accept_kernel, dim3(n), dim3(HMC_WG), sizeof(float) * 2 * d, (hipStream_t)hip_stream, z, grad, logp,
                     z_new, grad_new, logp_new, p0, p_new, kick, scale_tril, uniforms, n, d, accept_prob);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

int gl_profile_basis(const gl_component* comp, const float* x, const float* y, int64_t n_pts, int B, int xy_batched,
                     const float* params, float* out, void* hip_stream) {
  if (!comp || !x || !y || !params || !out) return fail(GL_EINVAL, "null argument");
  if (n_pts <= 0 || B <= 0) return fail(GL_EINVAL, "n_pts and B must be positive");
  int npar = kind_num_params(comp->kind, comp->iparam);
  if (npar < 0) return fail(GL_EINVAL, "unknown profile kind %d", comp->kind);
  if (kind_num_linear(comp->kind, comp->iparam) <= 0) return fail(GL_EINVAL, "kind %d has no linear amplitudes", comp->kind);
  CompDesc cd{};
  cd.kind = comp->kind;
  cd.iparam = comp->iparam;
  cd.flags = comp->flags;
  cd.n_par = npar;
  float* s_tab = nullptr;
  int s_stride = 0, rc_tab = 0;
  if (cd.kind == GL_SHAPELETS) {
    if (cd.iparam < 0 || cd.iparam > GL_SHAPELETS_NMAX_CAP)
      return fail(GL_EUNSUPPORTED, "shapelets n_max=%d outside [0,%d]", cd.iparam, GL_SHAPELETS_NMAX_CAP);
    if ((cd.flags & GL_FLAG_SHAPELETS_INTERPOLATE) && (rc_tab = point_shapelet_table(&s_tab, &s_stride))) return rc_tab;
  }
  long long total = (long long)n_pts * B;
  hipLaunchKernelGGL(gl_basis_point_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                     cd, x, y, (long long)n_pts, B, xy_batched, params, out, s_tab, s_stride);
  GL_HIP(hipGetLastError());
  return GL_OK;
}

}  // extern "C"
