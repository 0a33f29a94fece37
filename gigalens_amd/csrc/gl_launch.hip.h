// gl_launch.hip.h -- the launcher of the dominant ("main") kernel of a call: picks the kernel instantiation for the model
// (specialised pair / static kernels, the cluster kernel, or the interpreter) and launches it.  Included by the
// gl_launch_mode*.hip translation units, each of which instantiates it for ONE mode so that the kernel families compile
// in parallel.
#pragma once
#include <hip/hip_ext.h>

#include "gl_model.h"
#include "gl_static.hip.h"
#include "gl_pair.hip.h"
#include "gl_shp.hip.h"

namespace glk {

// ---- compile-time-specialised compositions (gl_static.hip.h) ------------------------------------------
// SERSIC and SERSIC_ELLIPSE share one device code path (the spherical profile is the e = 0 member), so
// signatures are matched after folding SERSIC_ELLIPSE -> SERSIC.
template <int MODE>
bool launch_static(const gl_model* m, const MainArgs& a, dim3 grid, dim3 block, size_t shmem, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
  const int T = (MODE == IMG_BWD || MODE == LL_GRAD) ? m->tile_grad : m->tile;
#define GL_PAIR(WW, LK, CK, SK)                                                           \
  do {                                                                                    \
    m->last_main_fn = (const void*)&gl_pair_kernel<MODE, v2f, WW, LK, CK, SK>;            \
    hipExtLaunchKernelGGL((gl_pair_kernel<MODE, v2f, WW, LK, CK, SK>), grid, block, (std::uint32_t)(shmem), stream, ev0, ev1, 0, a); \
  } while (0)
  if (m->pair) {
    // waves/SIMD the register budget is declared for: gradient modes keep the EPL / Sersic state of a pixel
    // pair live between the forward and VJP halves (no spills at 3 resp. 2 waves per SIMD), forward modes fit 4+
    constexpr bool G = (MODE == IMG_BWD || MODE == LL_GRAD);
    constexpr int W1 = G ? 3 : 4, W2 = G ? 2 : 4;
    if (m->light_spherical) {
      switch (m->static_id) {
        case ST_EPLSHEAR_SERSIC: GL_PAIR(W1, L_EplShear, C_None, C_Sersic); return true;  // (a 4-waves-per-SIMD budget spills 22 VGPRs: 98 vs 92 us)
        case ST_EPLSHEAR_SERSIC_SERSIC: GL_PAIR(W2, L_EplShear, C_Sersic, C_Sersic); return true;
        case ST_SIE_SERSIC: GL_PAIR(W1, L_Sie, C_None, C_Sersic); return true;  // gradient mode: 4 VGPRs would spill at 4 waves
        case ST_SIESHEAR_SERSIC_SERSIC: GL_PAIR(W1, L_SieShear, C_Sersic, C_Sersic); return true;
        default: break;
      }
    } else {
      switch (m->static_id) {
        case ST_EPLSHEAR_SERSIC: GL_PAIR(W1, L_EplShear, C_None, C_SersicE); return true;
        case ST_EPLSHEAR_SERSIC_SERSIC: GL_PAIR(W2, L_EplShear, C_SersicE, C_SersicE); return true;
        case ST_SIE_SERSIC: GL_PAIR(W1, L_Sie, C_None, C_SersicE); return true;
        case ST_SIESHEAR_SERSIC_SERSIC: GL_PAIR(W2, L_SieShear, C_SersicE, C_SersicE); return true;  // 168 VGPRs at 3 waves would spill 60
        default: break;
      }
    }
  }
#undef GL_PAIR
  // lenses | [Sersic lens light] | one shapelet source: the order-pair / matrix-pipe kernel (gl_shp.hip.h), every mode
  if (m->shp_kernel && (m->static_id == ST_EPLSHEAR_SHAPELETS || m->static_id == ST_EPLSHEAR_SERSIC_SHAPELETS)) {
    constexpr int NPS = SH_SQ / 2;
    const size_t epi = (size_t)(16 * m->Apad + 4 * 16 * 17) * sizeof(float);
    const size_t sh = (size_t)((m->D + 3) & ~3) * sizeof(float) + std::max(shp_exchange_bytes(NPS), epi);
    const bool interp = (m->comps.back().flags & GL_FLAG_SHAPELETS_INTERPOLATE) != 0;
    // whole 512-pixel tiles, no mask, no pixel list: the instantiation without the ragged-end tile code
    const bool ragged = a.mask || a.pix || (a.N % (2 * WG)) != 0 || (a.chunk % (2 * WG)) != 0;
    // table mode on a whole image: a wave-tile is an 8 x 16 BLOCK of the image instead of 128 consecutive pixels -- the live band
    // of a lensed field is compact in two dimensions (gl_shp.hip.h: fewer live tiles, fewer chain rounds, more tiles culled)
    MainArgs ab = a;
    ab.blk_w = (interp && !ragged && m->shp_blocked && m->width % 16 == 0 && m->height % 8 == 0 &&
                (long long)a.N == (long long)m->width * m->height) ? m->width : 0;
#define GL_SHP2(LLK_, I_, R_)                                                                                 \
  do {                                                                                                        \
    m->last_main_fn = (const void*)&gl_shp_kernel<MODE, 2, L_EplShear, LLK_, NPS, I_, R_>;                     \
    hipExtLaunchKernelGGL((gl_shp_kernel<MODE, 2, L_EplShear, LLK_, NPS, I_, R_>), grid, block, (std::uint32_t)(sh), stream, ev0, ev1, 0, ab);   \
  } while (0)
#define GL_SHP(LLK_, I_) do { if (ragged) GL_SHP2(LLK_, I_, true); else GL_SHP2(LLK_, I_, false); } while (0)
    if (m->static_id == ST_EPLSHEAR_SHAPELETS) { if (interp) GL_SHP(C_None, true); else GL_SHP(C_None, false); }
    else { if (interp) GL_SHP(C_SersicE, true); else GL_SHP(C_SersicE, false); }
#undef GL_SHP
#undef GL_SHP2
    return true;
  }
#define GL_LAUNCH(TT, WW, LK, CK, SK)                                                        \
  do {                                                                                       \
    m->last_main_fn = (const void*)&gl_static_kernel<MODE, TT, WW, LK, CK, SK>;              \
    hipExtLaunchKernelGGL((gl_static_kernel<MODE, TT, WW, LK, CK, SK>), grid, block, (std::uint32_t)(shmem), stream, ev0, ev1, 0, a); \
  } while (0)
  switch (m->static_id) {
    case ST_EPLSHEAR_SERSIC:
      if (T == 4) { if (m->static_variant == 2) GL_LAUNCH(4, 2, L_EplShear, C_None, C_Sersic); else GL_LAUNCH(4, 3, L_EplShear, C_None, C_Sersic); }
      else if (T == 1) GL_LAUNCH(1, 4, L_EplShear, C_None, C_Sersic);
      else { if (m->static_variant == 3) GL_LAUNCH(2, 3, L_EplShear, C_None, C_Sersic); else GL_LAUNCH(2, 4, L_EplShear, C_None, C_Sersic); }
      return true;
    case ST_EPLSHEAR_SERSIC_SERSIC:
      if (T == 4) GL_LAUNCH(4, 2, L_EplShear, C_Sersic, C_Sersic); else GL_LAUNCH(2, 4, L_EplShear, C_Sersic, C_Sersic);
      return true;
    case ST_SIE_SERSIC:
      if (T == 4) GL_LAUNCH(4, 4, L_Sie, C_None, C_Sersic); else GL_LAUNCH(2, 4, L_Sie, C_None, C_Sersic);
      return true;
    case ST_SIESHEAR_SERSIC_SERSIC:
      if (T == 4) GL_LAUNCH(4, 4, L_SieShear, C_Sersic, C_Sersic); else GL_LAUNCH(2, 4, L_SieShear, C_Sersic, C_Sersic);
      return true;
    case ST_EPLSHEAR_SHAPELETS:
      if (T == 2) GL_LAUNCH(2, 2, L_EplShear, C_None, C_Shapelets); else if (T == 1) GL_LAUNCH(1, 2, L_EplShear, C_None, C_Shapelets);
      else return false;
      return true;
    case ST_EPLSHEAR_SERSIC_SHAPELETS:
      if (T == 2) GL_LAUNCH(2, 2, L_EplShear, C_Sersic, C_Shapelets); else if (T == 1) GL_LAUNCH(1, 2, L_EplShear, C_Sersic, C_Shapelets);
      else return false;
      return true;
  }
#undef GL_LAUNCH
  return false;
}

// the interpreter and the cluster kernel live in translation units of their own (gl_launch_generic.hip.h, built without the
// SLP vectoriser): everything that is not a compile-time-specialised composition
template <int MODE>
int launch_generic(const gl_model* m, const MainArgs& a, dim3 grid, dim3 block, size_t shmem, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1);

template <int MODE>
int launch_main(const gl_model* m, const MainArgs& a, int B, int n_chunks, hipStream_t stream) {
  dim3 grid(n_chunks, B), block(WG);
  if (a.tail_rows > 0)  // tail_plan(): the tail samples' workgroups are dealt over whole rows of the grid
    grid.y = a.tail_from + (unsigned)(((long long)(B - a.tail_from) * a.tail_rows + n_chunks - 1) / n_chunks);
  size_t shmem = (size_t)(((m->D + 3) & ~3) + m->ncols * m->Apad) * sizeof(float) + m->nfw_lds;
  // Every timing_stride-th main launch carries an event pair ON ITS OWN DISPATCH PACKET (hipExtLaunchKernel: the runtime writes
  // the kernel's start and end timestamps into them when it completes) -- no extra packets on the stream.  Until round 4 the
  // pair was two hipEventRecord calls around the launch, ~2.5 us of stream time each: bracketing every launch slowed a 0.085 ms
  // step by 5 %, which is why the ring had a stride at all.
  const bool timed = m->timing_slots && (m->timing_calls.fetch_add(1) % m->timing_stride) == 0;
  const int slot = timed ? (int)(m->timing_count.fetch_add(1) % m->timing_slots) : 0;  // the slot is claimed here
  const hipEvent_t ev0 = timed ? m->evs[2 * slot] : nullptr, ev1 = timed ? m->evs[2 * slot + 1] : nullptr;
  // models with user-written profiles, shapelets above n_max = 10, the cluster models in the gradient modes and the basis stack
  // go to the generic launcher first; then the specialised compositions; then the interpreter
  constexpr bool GRADM = (MODE == IMG_BWD || MODE == LL_GRAD);
  const bool generic_first = m->has_user || m->shp_big || (GRADM && m->cluster && a.parts == 7u) || MODE == IMG_BASIS;
  m->last_main_user = -1;
  if (!generic_first && m->static_id && a.parts == 7u && launch_static<MODE>(m, a, grid, block, shmem, stream, ev0, ev1)) {
    // specialised kernel launched
  } else {
    const int rc = launch_generic<MODE>(m, a, grid, block, shmem, stream, ev0, ev1);
    if (rc) {
      // no kernel went out: the claimed ring slot must still hold a complete pair, or a drain would wait on (or read) events
      // of an earlier lap -- both are recorded here, back to back (a ~0 duration)
      if (timed) { (void)hipEventRecord(ev0, stream); (void)hipEventRecord(ev1, stream); }
      return rc;
    }
  }
  GL_HIP(hipGetLastError());
  return GL_OK;
}


}  // namespace glk
