// gl_launch_generic.hip.h -- the launcher of everything that is not a compile-time-specialised composition: the interpreter
// (gl_main_kernel, also for shapelets above n_max = 10 and the basis stack), the cluster kernel, and the run-time compiled
// interpreter of models with user-written profiles.  Included by the gl_generic_noslp_mode*.hip translation units, which
// __graft_entry__.build() compiles with -fno-slp-vectorize: the SLP vectoriser pairs unrelated scalar operations of these two
// kernels (horizontal sums, per-component constants) into packed instructions fed by register moves -- the cluster kernel
// executes 1056 vector instructions per pixel with it and 873 without (C4 1.33 -> see DESIGN.md), the interpreter loses 10-20 %
// of its time -- while the specialised kernels are written in packed form by hand and keep it (their direct-mode shapelet
// variant spills SGPRs to scratch without it).
#pragma once
#include <hip/hip_ext.h>

#include "gl_model.h"
#include "gl_kernels.hip.h"
#include "gl_cluster.hip.h"
#include "gl_clusterw.hip.h"

namespace glk {

template <int MODE>
int launch_generic(const gl_model* m, const MainArgs& a, dim3 grid, dim3 block, size_t shmem, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
#define GL_MAIN(TT, S_, F_)                                                              \
  do {                                                                                   \
    m->last_main_fn = (const void*)&gl_main_kernel<MODE, TT, S_, F_>;                    \
    hipExtLaunchKernelGGL((gl_main_kernel<MODE, TT, S_, F_>), grid, block, (std::uint32_t)(shmem), stream, ev0, ev1, 0, a); \
  } while (0)
#define GL_MAIN_FAM(TT, S_) \
  do { if (m->fam == 2) GL_MAIN(TT, S_, 2); else if (m->fam == 1) GL_MAIN(TT, S_, 1); else GL_MAIN(TT, S_, 0); } while (0)
  bool done = false;
  if (m->has_user) {  // a model with user-written profiles: the interpreter compiled at run time with their bodies (gl_user.hip)
    if (!m->user_fn[MODE]) return fail(GL_EUNSUPPORTED, "this call is not served for models with user-written profiles");
    MainArgs args = a;
    void* kargs[] = {(void*)&args};
    m->last_main_fn = nullptr;
    // the specialised pair kernel of the model's own composition when gl_user.hip could build it (whole renders only: partial
    // renders -- `parts` -- are the interpreter's), else the interpreter with the bodies behind its component switch
    const bool pair = MODE < 4 && m->user_pair_fn[MODE < 4 ? MODE : 0] && a.parts == 7u;
    m->last_main_user = MODE + (pair ? 16 : 0);  // gl_model_last_main_kernel names it (no host function to look up)
    GL_HIP(hipExtModuleLaunchKernel(pair ? m->user_pair_fn[MODE < 4 ? MODE : 0] : m->user_fn[MODE], grid.x * block.x, grid.y, 1, block.x, 1, 1,
                                    (unsigned)shmem, stream, kargs, nullptr, ev0, ev1, 0));
    done = true;
  }
  if (!done && m->shp_big) {  // shapelets above n_max = 10: the runtime-order interpreter variant (basic profile families, T = 2)
    m->last_main_fn = (const void*)&gl_main_kernel<MODE, 2, true, 0, true>;
    hipExtLaunchKernelGGL((gl_main_kernel<MODE, 2, true, 0, true>), grid, block, (std::uint32_t)(shmem), stream, ev0, ev1, 0, a);
    done = true;
  }
  if constexpr (MODE == IMG_BWD || MODE == LL_GRAD) {
#define GL_CLUSTERW(LENS_, SPW_, E_, W_)                                                                 \
  do {                                                                                                 \
    const size_t sh = sizeof(float) * (CW_XCHG_FLOATS + LENS_::kLdsFloats + 16 + m->A);                \
    m->last_main_fn = (const void*)&gl_clusterw_kernel<MODE, LENS_, SPW_, E_, W_>;                       \
    hipExtLaunchKernelGGL((gl_clusterw_kernel<MODE, LENS_, SPW_, E_, W_>), grid, block, (std::uint32_t)(sh), stream, ev0, ev1, 0, a, m->n_lens, m->n_src); \
  } while (0)
    if (!done && m->cluster && m->cluster_w && a.parts == 7u) {  // ... with the components dealt over the four waves (gl_clusterw.hip.h)
      const int size = (m->n_lens <= 4 && m->n_src <= 8) ? 0 : (m->n_src <= 12 ? 1 : 2);
      if (m->cluster == 2) { if (size == 0) GL_CLUSTERW(CwLensNfw<1>, 2, true, 4); else if (size == 1) GL_CLUSTERW(CwLensNfw<2>, 3, true, 3); else GL_CLUSTERW(CwLensNfw<2>, 5, true, 2); }
      else { if (size == 0) GL_CLUSTERW(CwLensNfw<1>, 2, false, 4); else if (size == 1) GL_CLUSTERW(CwLensNfw<2>, 3, false, 4); else GL_CLUSTERW(CwLensNfw<2>, 5, false, 3); }
      done = true;
    }
#undef GL_CLUSTERW
    if (!done && m->cluster && a.parts == 7u) {  // N x same-kind cluster model: forward state of every component kept in registers
      const size_t sh = (size_t)64 * m->Apad * sizeof(float) + sizeof(float) * 2 * NFW_TAB_NODES;  // gradient columns + the h(X) table
#define GL_CLUSTER(NH_, NS_, E_, W_)                                                                     \
  do {                                                                                                 \
    m->last_main_fn = (const void*)&gl_cluster_kernel<MODE, NH_, NS_, E_, W_>;                          \
    hipExtLaunchKernelGGL((gl_cluster_kernel<MODE, NH_, NS_, E_, W_>), grid, block, (std::uint32_t)(sh), stream, ev0, ev1, 0, a, m->n_lens, m->n_src); \
  } while (0)
      const bool small = m->n_lens <= 4 && m->n_src <= 8;
      if (m->cluster == 2) { if (small) GL_CLUSTER(4, 8, true, 3); else GL_CLUSTER(8, 20, true, 2); }
      else { if (small) GL_CLUSTER(4, 8, false, 3); else GL_CLUSTER(8, 20, false, 2); }
#undef GL_CLUSTER
      done = true;
    }
  }
  if (done) {
  } else if constexpr (MODE == IMG_BASIS) {  // basis stack of lstsq_simulate: interpreter kernel, one tile shape
    if (m->has_shapelets) GL_MAIN_FAM(2, true); else GL_MAIN_FAM(2, false);
  } else {
    // four pixels per thread only in the forward modes: the gradient instantiations at T = 4 spill (94 VGPRs on the plain
    // families) and measure no faster than T = 2 (C4 through the interpreter: 2.41 ms either way)
    constexpr bool GRADM = (MODE == IMG_BWD || MODE == LL_GRAD);
    const int Tg = GRADM ? 2 : m->tile;
    if constexpr (GRADM) {
      if (m->has_shapelets) GL_MAIN_FAM(2, true); else GL_MAIN_FAM(2, false);
    } else {
      if (m->has_shapelets) { if (Tg == 4) GL_MAIN_FAM(4, true); else GL_MAIN_FAM(2, true); }
      else { if (Tg == 4) GL_MAIN_FAM(4, false); else GL_MAIN_FAM(2, false); }
    }
  }
#undef GL_MAIN_FAM
#undef GL_MAIN
  GL_HIP(hipGetLastError());
  return GL_OK;
}

}  // namespace glk
