// gl_model.h -- the library-internal model descriptor and error helper shared by the translation units of
// libgigalens_hip.so (gigalens_hip.hip: C ABI + host logic; gl_launch_mode*.hip: one instantiation of the main-kernel
// launcher per mode, compiled in parallel).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include <atomic>

#include "../../include/gigalens_hip.h"
#include "gl_kernels.hip.h"

namespace glk {
// sets the thread-local message gl_last_error() returns; returns `code`
__attribute__((visibility("hidden"))) int fail(int code, const char* fmt, ...);
}  // namespace glk

#define GL_HIP(call)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) return glk::fail(GL_ELAUNCH, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

using glk::CompDesc;
using glk::CatDev;
using glk::SeriesDev;
using glk::ZCol;

namespace glk {
// plans of gl_corr_pair_kernel (gl_post.hip.h)
struct CorrClass {  // one ROW class; its column classes (ncj of them) are computed by the same thread
  int koff;        // offset of this class's [KH][ncj][KWP] kernel block in the kernel buffer
  int KH, pt, pl;  // taps per column, top / left padding of the window (common to the column classes: kernels are shifted)
  int Ho, Wo[4];   // outputs of the class (rows; columns per column class)
  int oo_r, oo_c[4];  // placement of output (0, 0) in the output image
};
struct CorrArgs {
  const float* k;
  int n_class, ncj, B;  // row classes (one per workgroup along z), column classes per thread
  int Hi, Wi;          // input image
  int Hout, Wout, os;  // output image and the placement stride of a class's outputs
  float scale;
  int vec;  // image widths multiples of four and 16-byte aligned buffers: float4 tile fill and output stores
  int dbg;  // -DGL_EXPERIMENTS builds only (GIGALENS_HIP_DBGFLAGS): 16 skip the tile fill, 32 skip the multiply-add loop, 64 skip the output stores
  CorrClass cls[16];
};
}  // namespace glk

struct gl_model {
  std::vector<CompDesc> comps;
  int n_lens = 0, n_ll = 0, n_src = 0;
  int P = 0, D = 0, A = 0, Apad = 0, ncols = 64;
  bool has_shapelets = false, has_table = false;
  bool shp_big = false;  // some shapelet component has n_max > SH_CAP: wide table, runtime-order interpreter variant for all of them
  // user-written profiles inside the model (K_USER_MASS / K_USER_LIGHT): the interpreter kernel compiled at run time with their bodies
  bool has_user = false;
  hipModule_t user_module = nullptr;
  hipFunction_t user_fn[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // per Mode (IMG_BASIS: none)
  // ... and, for the compositions the pair kernel serves (EPL / SIE / Shear / SIS / user lenses | Sersic / user lights), that kernel
  // specialised on the model's component list with the user bodies inside (gl_pair_kernel<MODE, v2f, 2, KindList<...>, ...>)
  hipFunction_t user_pair_fn[4] = {nullptr, nullptr, nullptr, nullptr};
  // ... and the point kernels (gl_positions.hip.h: image-position likelihood P1-P4, lens maps) with the bodies on nested duals,
  // compiled from user_point_src when first asked for (gl_user.hip compile_user_points)
  std::string user_point_src;
  hipModule_t user_point_module = nullptr;
  hipFunction_t user_point_fn[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int height = 0, width = 0, supersample = 1, N = 0;
  float conversion_factor = 1.f;
  // device-resident, immutable
  CompDesc* d_comps = nullptr;
  float* d_gx = nullptr;
  float* d_gy = nullptr;
  int* d_pix = nullptr;
  float* d_shp_tab = nullptr;
  float* d_nfw_tab = nullptr;  // models with NFW lenses: h(X) = g(X) / X^2 on the float format's own grid (gl_host_tables.h)
  int chunk_px_override = 0;  // -DGL_EXPERIMENTS builds only
  int dbg_flags = 0;          // -DGL_EXPERIMENTS builds only
  float grid_rmax = 0.f;      // max |(x, y)| over the pixel grid
  int shp_blocked = 1;        // GIGALENS_HIP_SHP_BLOCKED: wave-tiles of the table-mode shapelet kernel are 8 x 16 blocks of the image
  int shp_cull = 1;           // GIGALENS_HIP_SHP_CULL: wave-tiles provably outside the shapelet table skip the lens (gl_shp.hip.h)
  int corr_max_pairs = 0;     // GIGALENS_HIP_CORR_MAXPAIRS, read once at gl_model_create
  int corr_wide = 1;          // GIGALENS_HIP_CORR_WIDE, read once at gl_model_create
  bool has_nfw = false;
  size_t nfw_lds = 0;          // bytes of that table in a main kernel's LDS
  int shp_stride = 0;
  float* d_psf = nullptr;  // effective kernel flip(psf) (*) box(ss)/ss^2, see gl_post.hip.h
  // the register-blocked pair kernel's plans (gl_post.hip.h gl_corr_pair_kernel): class tables, padded kernels on the device
  struct CorrPlan { glk::CorrArgs args{}; int KWP = 0, ST = 0, max_Ho = 0, max_Wo = 0, max_KH = 0; bool ok = false; };
  CorrPlan corr_fwd, corr_bwd;
  float* d_corr_k = nullptr;
  int psf_h = 0, psf_w = 0;
  int KH = 1, KW = 1, pad_t = 0, pad_l = 0;
  bool has_post = false;
  // unconstrained-space front end (gl_model_set_prior)
  int d_z = 0;
  ZCol* d_zcols = nullptr;
  int* d_src = nullptr;
  float* d_const = nullptr;
  int static_id = 0;   // 0 = generic interpreter kernel, >0 = compile-time-specialised composition
  int static_variant = 0;
  int pair = 1;        // pixel-pair (packed fp32) form of the specialised kernels
  bool light_spherical = false;  // every light profile is the spherical Sersic: the pair kernels take their fast path
  int cluster_w = 0;   // ... in its component-per-wave form (gl_clusterw_kernel)
  int cluster = 0;     // gl_cluster_kernel serves the gradient modes: 1 = halos + spherical Sersic sources, 2 = elliptical sources
  // image-position likelihood (gl_model_set_positions)
  int pos_J = 0, pos_F = 0, lens_params = 0;
  float* d_pos = nullptr;  // [4][J]: x, y, err_x, err_y
  int* d_fam = nullptr;    // [F+1]
  bool has_epl = false;
  int epl_comp = -1;     // the model's only EPL component, or -1 (none / several)
  int fam = 0;  // family level of the interpreter variant (gl_main_kernel FAM): 1 dPIE family / catalogues / series, 2 gl_extra.h
  bool use_order = true;
  int tail_rows = -1, tail_n = -1;  // tapered end of the cost-ordered dispatch (tail_plan); rows 0 = off, -1 = automatic
  bool prep_lds = true;     // gl_prep_wave_kernel keeps each sample's parameter row in LDS
  bool order_fused = true;  // the front end's extra workgroup sorts (gl_prep_wave_kernel); else a launch of gl_order_kernel
  bool wave_prep = true;  // EPL models: one wavefront per sample in the front end (GIGALENS_HIP_WAVE_PREP=0: thread per component)
  int lstsq_wgs = 2048;     // workgroups the normal-matrix kernels of the linear solve aim for (pixel chunks per sample = this / B)
  bool lstsq_chol = true;   // linear solve: Cholesky attempt first (gl_chol_solve_kernel), eigenvalue solve for what it leaves
  bool lstsq_fused = true;  // one-shapelet-source linear solves form the normal matrix from the bases (gl_shp_normal_kernel), no stack
  // measurement hooks (gl_model_set_timing): a ring of event pairs around the main-kernel launches, and the host
  // function of the most recent main launch (gl_model_last_main_kernel)
  int timing_slots = 0, timing_stride = 1;
  // (atomics: two host threads may drive one model on different streams; each timed launch claims its ring slot)
  mutable std::atomic<long long> timing_count{0}, timing_calls{0};
  std::vector<hipEvent_t> evs;  // 2 * timing_slots
  mutable std::atomic<const void*> last_main_fn{nullptr};
  mutable std::atomic<int> last_main_user{-1};  // mode of the run-time compiled kernel the most recent main launch dispatched, or -1
  // galaxy catalogues of the GL_SCALED components (gl_model_set_catalogue)
  struct Cat { CatDev dev; std::vector<float> table; };
  std::vector<Cat> cats;
  int G = 0;           // galaxies over all catalogues
  int n_scaled = 0;    // GL_SCALED components
  CatDev* d_cats = nullptr;
  float* d_gal_table = nullptr;   // [G][7]
  float* d_gal_static = nullptr;  // [G][DP_NS]
  // series-expansion lenses (gl_model_set_series): one coefficient field per GL_SERIES component
  std::vector<SeriesDev> series;      // device pointers owned by the model
  std::vector<int> series_comp;       // component of each slot
  int n_series = 0, n_series_set = 0;
  SeriesDev* d_series = nullptr;
  // linear amplitudes (lstsq_simulate): channel k of the basis stack <-> packed parameter column
  std::vector<int> lin_cols;
  int* d_lin_cols = nullptr;
  int shp_kernel = 0;    // lenses | [Sersic lens lights] | one shapelet source: served by gl_shp.hip.h (GIGALENS_HIP_SHP=0: the round-2 kernels)
  int tile = 2;          // pixels per thread per tile (template T) for forward-only launches
  int tile_grad = 2;     // ... and for launches that also produce gradients
  int target_wgs = 2048;  // work decomposition target of GIGALENS_HIP_TARGET_WGS (see chunking())
  bool target_wgs_set = false;
};

namespace glk {
// launches the dominant kernel of a call for one mode (gl_launch.hip.h); explicit instantiations live in gl_launch_mode*.hip
template <int MODE>
int launch_main(const gl_model* m, const MainArgs& a, int B, int n_chunks, hipStream_t stream);
// gl_user.hip: compile gl_main_kernel with the model's user bodies (n_bodies HIP C++ sources), fill user_module / user_fn
__attribute__((visibility("hidden"))) int compile_user_model(gl_model* m, const char* const* bodies, int n_bodies);
__attribute__((visibility("hidden"))) int compile_user_points(const gl_model* m);
int match_static(const gl_model* m);
extern template int launch_main<IMG_FWD>(const gl_model*, const MainArgs&, int, int, hipStream_t);
extern template int launch_main<IMG_BWD>(const gl_model*, const MainArgs&, int, int, hipStream_t);
extern template int launch_main<LL_FWD>(const gl_model*, const MainArgs&, int, int, hipStream_t);
extern template int launch_main<LL_GRAD>(const gl_model*, const MainArgs&, int, int, hipStream_t);
extern template int launch_main<IMG_BASIS>(const gl_model*, const MainArgs&, int, int, hipStream_t);
}  // namespace glk
