// gl_pair.hip.h -- the specialised kernels in "pixel-pair" form: every per-pixel quantity is a 2-vector
// (pixel j, pixel j+256) so that the arithmetic of the whole fused forward+gradient pass -- not only the EPL
// series loop -- issues as packed fp32 instructions (v_pk_mul/add/fma_f32).  On CDNA4 a wave64 VALU
// instruction occupies its SIMD for 4 cycles whether it is packed or not (measured: SQ_ACTIVE_INST_VALU /
// SQ_INSTS_VALU = 4.17 with 21 % packed instructions), so packed issue is the only way to the 157 TFLOP/s
// vector peak, and this path is VALU-issue bound (profiles/r1_summary.json: VALU busy 0.87).
//
// The per-profile maths is the same as gl_profiles.h (same derived-constant layout, same accumulator
// layout, same finalize), written once over a value type V in {float, v2f}: comparisons yield lane masks and
// `m ? a : b` selects per lane.  Transcendentals (v_rcp/sqrt/log/exp) have no packed form and are applied
// per lane.  Supported components: EPL, SIE, SHEAR, SIS lenses; SERSIC / SERSIC_ELLIPSE lights (K_SERSIC in a kind list
// selects the spherical fast path: only for models whose light profiles are ALL spherical, else K_SERSIC_ELLIPSE serves both).
#pragma once
#include "gl_static.hip.h"
#include "gl_vec.hip.h"

namespace glk {

#ifdef GL_HAVE_USER
// ---- user-written profiles in the pair kernel (run-time compiled: gl_user.hip) --------------------------------------------------
// The model's bodies sit behind glu::mass_fwd / mass_vjp / light_fwd / light_vjp (a switch on the body index, which is a
// compile-time constant here: it folds).  A body is scalar code over a number type, so the two pixels of a pair are evaluated one
// after the other; the VJP comes from the body's forward-mode duals.
template <class V, int CODE> __device__ __forceinline__ void user_mass_fwd_v(const float* d, V x, V y, V& bx, V& by) {
  constexpr unsigned body = (unsigned)user_code_body(CODE);
  if constexpr (sizeof(V) == sizeof(float)) {
    float ax, ay;
    glu::mass_fwd(body, d, x, y, ax, ay);
    bx -= ax; by -= ay;
  } else {
    float ax0, ay0, ax1, ay1;
    glu::mass_fwd(body, d, x.x, y.x, ax0, ay0);
    glu::mass_fwd(body, d, x.y, y.y, ax1, ay1);
    bx -= V{ax0, ax1};
    by -= V{ay0, ay1};
  }
}
template <class V, int CODE> __device__ __forceinline__ void user_mass_vjp_v(const float* d, V x, V y, V gx, V gy, V* acc) {
  constexpr unsigned body = (unsigned)user_code_body(CODE);
  constexpr int NPAR = user_code_npar(CODE);
  float t0[NPAR > 0 ? NPAR : 1], t1[NPAR > 0 ? NPAR : 1];
#pragma unroll
  for (int k = 0; k < NPAR; ++k) { t0[k] = 0.f; t1[k] = 0.f; }
  if constexpr (sizeof(V) == sizeof(float)) {
    glu::mass_vjp(body, d, x, y, gx, gy, t0);
#pragma unroll
    for (int k = 0; k < NPAR; ++k) acc[k] += t0[k];
  } else {
    glu::mass_vjp(body, d, x.x, y.x, gx.x, gy.x, t0);
    glu::mass_vjp(body, d, x.y, y.y, gx.y, gy.y, t1);
#pragma unroll
    for (int k = 0; k < NPAR; ++k) acc[k] += V{t0[k], t1[k]};
  }
}
template <class V, int CODE> __device__ __forceinline__ V user_light_fwd_v(const float* d, V x, V y) {
  constexpr unsigned body = (unsigned)user_code_body(CODE);
  if constexpr (sizeof(V) == sizeof(float)) return glu::light_fwd(body, d, x, y);
  else return V{glu::light_fwd(body, d, x.x, y.x), glu::light_fwd(body, d, x.y, y.y)};
}
template <class V, int CODE, bool SRC>
__device__ __forceinline__ void user_light_vjp_v(const float* d, V x, V y, V g, V* acc, V& gpx, V& gpy) {
  constexpr unsigned body = (unsigned)user_code_body(CODE);
  constexpr int NPAR = user_code_npar(CODE);
  float t0[NPAR > 0 ? NPAR : 1], t1[NPAR > 0 ? NPAR : 1];
#pragma unroll
  for (int k = 0; k < NPAR; ++k) { t0[k] = 0.f; t1[k] = 0.f; }
  if constexpr (sizeof(V) == sizeof(float)) {
    float dgx = 0.f, dgy = 0.f;
    glu::light_vjp(body, d, x, y, g, t0, dgx, dgy);
#pragma unroll
    for (int k = 0; k < NPAR; ++k) acc[k] += t0[k];
    if (SRC) { gpx += dgx; gpy += dgy; }
  } else {
    float dgx0 = 0.f, dgy0 = 0.f, dgx1 = 0.f, dgy1 = 0.f;
    glu::light_vjp(body, d, x.x, y.x, g.x, t0, dgx0, dgy0);
    glu::light_vjp(body, d, x.y, y.y, g.y, t1, dgx1, dgy1);
#pragma unroll
    for (int k = 0; k < NPAR; ++k) acc[k] += V{t0[k], t1[k]};
    if (SRC) { gpx += V{dgx0, dgx1}; gpy += V{dgy0, dgy1}; }
  }
}
#endif

// ---- the pair kernel ----------------------------------------------------------------------------------------
// V = v2f: each thread owns pixels (j, j + 256) of every 512-pixel tile; V = float: one pixel per thread.
template <int MODE, class V, int WAVES, class LK, class LLK, class SK>
__global__ void __launch_bounds__(WG, WAVES) gl_pair_kernel(MainArgs a) {
  constexpr int NL = LK::n, NLL = LLK::n, NS = SK::n, NLIGHT = NLL + NS;
  constexpr bool GRAD = (MODE == IMG_BWD || MODE == LL_GRAD);
  constexpr int W = sizeof(V) / sizeof(float);
  extern __shared__ float smem[];
  float* s_d = smem;
  float* s_acc = smem + ((a.D + 3) & ~3);
  const int tid = threadIdx.x;
  // (sample rank, partial row, pixel range): see MainArgs::tail_rows
  int rank = blockIdx.y, row = blockIdx.x, n_rows = gridDim.x, my_rows = gridDim.x;
  int p0 = row * a.chunk, p1 = min(p0 + a.chunk, a.N);
  if (a.tail_rows > 0) {
    n_rows = a.n_rows;
    if (rank >= a.tail_from) {
      const int j = (rank - a.tail_from) * (int)gridDim.x + row;
      rank = a.tail_from + j / a.tail_rows;
      if (rank >= a.n_samples) return;  // the grid's last row may overhang
      row = j % a.tail_rows;
      my_rows = a.tail_rows;
      const int tiles = (a.N + W * WG - 1) / (W * WG);
      p0 = (int)((long long)row * tiles / a.tail_rows) * (W * WG);
      p1 = min((int)((long long)(row + 1) * tiles / a.tail_rows) * (W * WG), a.N);
    }
  }
  const int b = a.order ? a.order[rank] : rank;
  const CompDesc* __restrict__ comps = a.comps;
  const float* __restrict__ gder = a.derived + (size_t)b * a.D;
  if (!GL_DBG(a.dbg, 4)) {
    if (a.D <= WG) {  // the usual case (EPL at niter = 50: D = 244): one predicated load per thread, no loop scaffolding
      if (tid < a.D) s_d[tid] = gder[tid];
    } else {
      for (int i = tid; i < a.D; i += WG) s_d[i] = gder[i];
    }
  }
  __syncthreads();
  constexpr int NACC_L = [] { int n = 0; for (int i = 0; i < NL; ++i) n += static_nacc(LK::kinds[i]); return n; }();
  constexpr int NACC_C = [] {
    int n = 0;
    for (int i = 0; i < NLL; ++i) n += static_nacc(LLK::kinds[i]);
    for (int i = 0; i < NS; ++i) n += static_nacc(SK::kinds[i]);
    return n;
  }();
  V accL[NACC_L > 0 ? NACC_L : 1];
  V accC[NACC_C > 0 ? NACC_C : 1];
#pragma unroll
  for (int k = 0; k < NACC_L; ++k) accL[k] = V(0.f);
#pragma unroll
  for (int k = 0; k < NACC_C; ++k) accC[k] = V(0.f);
  V st0 = V(0.f), st1 = V(0.f);  // chi^2, and the normalisation as sum of w log2(sigma^2): x ln2 and + log(2 pi) sum w at the end
  V wsum = V(0.f);                // sum of the weights over the ragged / masked tiles (whole tiles count their pixels)
  int n_whole = 0;                // whole unmasked tiles of this workgroup
  const float* dL[NL > 0 ? NL : 1];
  const float* dC[NLIGHT > 0 ? NLIGHT : 1];
#pragma unroll
  for (int i = 0; i < NL; ++i) dL[i] = s_d + comps[i].d_off;
#pragma unroll
  for (int i = 0; i < NLIGHT; ++i) dC[i] = s_d + comps[NL + i].d_off;
  const bool has_err = a.err != nullptr, has_mask = a.mask != nullptr, has_pix = a.pix != nullptr;

  // One tile = W*256 pixels.  CHECK=false is the steady state (whole tile inside the chunk, no pixel mask):
  // no validity selects, no weights.  CHECK=true handles the ragged last tile and img_region weights.
  // err_tag: 0 no error map, 1 error map (both compile the variance and the cotangent of their own case: as a run-time choice the
  // two were computed side by side and selected, 10 instructions per pixel pair), 2 decided at run time (the ragged-end tile)
  // (requesting the NEXT whole tile's planes -- grid, observation, error map -- while a tile is worked on was tried in round 4, after
  // the shapelet kernel's dissection showed bare tile loads costing 1 400 cycles per wave-tile there: 147 VGPRs instead of 141 and
  // 0.0823 against 0.0819 ms per C2 step -- at three waves per SIMD the other waves already cover a tile's opening loads.)
  auto tile = [&](int base, auto check_tag, auto err_tag) {
    constexpr bool CHECK = decltype(check_tag)::value;
    constexpr int ERR = decltype(err_tag)::value;
    const bool herr = ERR == 2 ? has_err : ERR == 1;
    unsigned jj[W], pidx[W];
    bool valid[W];
    V x, y, vmask = V(1.f);
#pragma unroll
    for (int w = 0; w < W; ++w) {
      int j = base + w * WG + tid;
      valid[w] = CHECK ? (j < p1) : true;
      jj[w] = (unsigned)(valid[w] ? j : p1 - 1);
      pidx[w] = (CHECK && has_pix) ? (unsigned)a.pix[jj[w]] : jj[w];  // CHECK=false tiles run only without a pixel list
    }
    // 32-bit BYTE offsets from the (scalar) plane bases: one shift per pixel serves the grid, the observation and the error
    // plane (global_load ... v_off, s[base]) instead of a 64-bit address computation per load
    unsigned jo[W], po[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { jo[w] = jj[w] << 2; po[w] = pidx[w] << 2; }
    auto ldf = [](const float* base, unsigned byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
    if constexpr (W == 2) {
      x = V{ldf(a.gx, jo[0]), ldf(a.gx, jo[1])};
      y = V{ldf(a.gy, jo[0]), ldf(a.gy, jo[1])};
      if (CHECK) vmask = V{valid[0] ? 1.f : 0.f, valid[1] ? 1.f : 0.f};
    } else {
      x = ldf(a.gx, jo[0]);
      y = ldf(a.gy, jo[0]);
      if (CHECK) vmask = valid[0] ? 1.f : 0.f;
    }
    V bx = x, by = y, m = V(0.f);
    EplStateV<V> est[NL > 0 ? NL : 1];
    SerStateV<V> sst[NLIGHT > 0 ? NLIGHT : 1];
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int kind = LK::kinds[i];
      if constexpr (kind == K_EPL) epl_fwd_v<V, GRAD>(dL[i], gder + comps[i].d_off, x, y, bx, by, est[i]);
      else if constexpr (kind == K_SIE) sie_fwd_v<V>(dL[i], x, y, bx, by);
      else if constexpr (kind == K_SHEAR) shear_fwd_v<V>(dL[i], x, y, bx, by);
#ifdef GL_HAVE_USER
      else if constexpr (is_user_code(kind)) user_mass_fwd_v<V, kind>(dL[i], x, y, bx, by);
#endif
      else sis_fwd_v<V>(dL[i], x, y, bx, by);
    }, std::make_integer_sequence<int, NL>{});
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr bool src = i >= NLL;
      constexpr int lkind = src ? SK::kinds[src ? i - NLL : 0] : LLK::kinds[src ? 0 : i];
      constexpr bool ell = lkind != K_SERSIC;  // K_SERSIC in the list = all spherical
#ifdef GL_HAVE_USER
      if constexpr (is_user_code(lkind)) m += user_light_fwd_v<V, lkind>(dC[i], src ? bx : x, src ? by : y);
      else
#endif
      m += sersic_fwd_v<V, ell>(dC[i], src ? bx : x, src ? by : y, sst[i]);
    }, std::make_integer_sequence<int, NLIGHT>{});
    auto nanp = m != m;
    m = (nanp ? V(0.f) : m) * a.out_scale;  // NaN -> 0 (tf/simulator.py:140), then x det(T) (:156)
    if (MODE == IMG_FWD) {
      float* row = a.img + (size_t)b * a.img_stride;
      if constexpr (W == 2) {
        if (valid[0]) row[pidx[0]] = m.x;
        if (valid[1]) row[pidx[1]] = m.y;
      } else {
        if (valid[0]) row[pidx[0]] = m;
      }
      return;
    }
    V gm;
    if (MODE == IMG_BWD) {
      const float* row = a.gimg + (size_t)b * a.img_stride;
      V g;
      if constexpr (W == 2) g = V{row[pidx[0]], row[pidx[1]]}; else g = row[pidx[0]];
      gm = nanp ? V(0.f) : (CHECK ? g * vmask : g) * a.out_scale;
    } else {
      V o, w = vmask, e = V(1.f);
      if constexpr (W == 2) {
        o = V{ldf(a.obs, po[0]), ldf(a.obs, po[1])};
        if (CHECK && has_mask) w = w * V{ldf(a.mask, po[0]), ldf(a.mask, po[1])};
        if (herr) e = V{ldf(a.err, po[0]), ldf(a.err, po[1])};
      } else {
        o = ldf(a.obs, po[0]);
        if (CHECK && has_mask) w = w * ldf(a.mask, po[0]);
        if (herr) e = ldf(a.err, po[0]);
      }
      // tf/model.py:92-99; sigma^2 = bg^2 + m/t (no clip: negative -> NaN like sqrt of a negative)
      V dmo = m - o;
      V s2 = herr ? e * e : m * a.inv_t + a.bg2;
      V is2 = rcp(s2);
      // log(2 pi sigma^2) = ln2 (log2 sigma^2 + log2 2 pi): the pixel loop sums the bare log2 (NaN for sigma^2 < 0, like
      // log(2 pi sqrt(.)^2) in the reference); the two constants join once per workgroup
      V nm = log2_(s2);
      V c2 = __builtin_elementwise_fma(nm, V(0.f), dmo * dmo * is2);  // + 0 * nm: carries that NaN into chi^2
      if (CHECK) {  // invalid lanes carry weight 0; "x * 0" would keep a NaN, so select instead
        auto use = w != V(0.f);
        st0 += use ? c2 * w : V(0.f);
        st1 += use ? nm * w : V(0.f);
        wsum += w;
      } else {
        st0 += c2;
        st1 += nm;
      }
      if (MODE == LL_GRAD) {
        V g = herr ? -(dmo * is2) : (dmo * dmo * is2 - 1.f) * (is2 * (0.5f * a.inv_t)) - dmo * is2;
        gm = nanp ? V(0.f) : (CHECK ? g * w : g) * a.out_scale;
      }
    }
    if constexpr (GRAD) {
      V gbx = V(0.f), gby = V(0.f);
      static_for([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr bool src = i >= NLL;
        constexpr int off = [] {
          int n = 0;
          for (int j = 0; j < i; ++j) n += static_nacc(j < NLL ? LLK::kinds[j < NLL ? j : 0] : SK::kinds[j >= NLL ? j - NLL : 0]);
          return n;
        }();
        constexpr int lkind = src ? SK::kinds[src ? i - NLL : 0] : LLK::kinds[src ? 0 : i];
        constexpr bool ell = lkind != K_SERSIC;
#ifdef GL_HAVE_USER
        if constexpr (is_user_code(lkind)) user_light_vjp_v<V, lkind, src>(dC[i], src ? bx : x, src ? by : y, gm, accC + off, gbx, gby);
        else
#endif
        sersic_vjp_v<V, src, ell>(dC[i], sst[i], gm, accC + off, gbx, gby);
      }, std::make_integer_sequence<int, NLIGHT>{});
      gbx = -gbx;
      gby = -gby;
      static_for([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr int kind = LK::kinds[i];
        constexpr int off = [] { int n = 0; for (int j = 0; j < i; ++j) n += static_nacc(LK::kinds[j]); return n; }();
        if constexpr (kind == K_EPL) epl_vjp_v<V>(dL[i], gbx, gby, est[i], accL + off);
        else if constexpr (kind == K_SIE) sie_vjp_v<V>(dL[i], x, y, gbx, gby, accL + off);
        else if constexpr (kind == K_SHEAR) shear_vjp_v<V>(x, y, gbx, gby, accL + off);
#ifdef GL_HAVE_USER
        else if constexpr (is_user_code(kind)) user_mass_vjp_v<V, kind>(dL[i], x, y, gbx, gby, accL + off);
#endif
        else sis_vjp_v<V>(dL[i], x, y, gbx, gby, accL + off);
      }, std::make_integer_sequence<int, NL>{});
    }
  };
  {
    const bool plain = !has_mask && !has_pix;
    int base = p0;
    if (GL_DBG(a.dbg, 1)) base = p1;
    if (plain) {
      if (has_err)
        for (; base + WG * W <= p1; base += WG * W, ++n_whole) tile(base, std::false_type{}, std::integral_constant<int, 1>{});
      else
        for (; base + WG * W <= p1; base += WG * W, ++n_whole) tile(base, std::false_type{}, std::integral_constant<int, 0>{});
    }
    for (; base < p1; base += WG * W) tile(base, std::true_type{}, std::integral_constant<int, 2>{});
  }
  if (MODE == IMG_FWD) return;
  // this lane's share of sum w log(2 pi sigma^2) from its sum of w log2 sigma^2 and its sum of w
  auto norm_term = [](float l2sum, float w) { return (l2sum + w * 2.6514961294723187f) * (float)kLn2; };  // log2(2 pi)
  float* out = a.partial + ((size_t)b * n_rows + row) * a.A;
  if (row == 0)  // the rows of the sample that no workgroup writes (the other kind of sample has more rows than this one)
    for (int i = my_rows * a.A + tid; i < n_rows * a.A; i += WG) out[i] = 0.f;
  // ---- epilogue: per-sample scale factors deferred out of the pixel loop, lane sum, one reduction ----
  // accumulators live in registers for the whole chunk, so ONE full wave64 reduction per value per workgroup
  // is cheap: lane 63 of each wave stores its sums into the wave's own LDS row (no zero-fill, no read-modify-write)
  if (!GRAD) {  // forward-only: the gradient slots of the partial row are defined (zero), never garbage
    for (int i = tid; i < 4 * a.Apad; i += WG) s_acc[i] = 0.f;
    __syncthreads();
  }
  if constexpr (GRAD) {
    // Gradient modes: the whole accumulator row [chi2, norm, 0, 0 | lenses | lights] is reduced FOUR values per register
    // (the cluster kernel's 4 x 4 transpose inside every quad, then two row shifts): lane 12 + q of each 16-lane row holds
    // the row's sum of value 4 g + q -- 11 instead of 28 cross-lane adds per four values -- and the 16 rows of the workgroup
    // are summed in fixed order below.  Offsets inside the row are the layout gl_model_create assigns (host-checked).
    constexpr int NV = NSTAT + NACC_L + NACC_C, NVP = (NV + 3) & ~3;
    float vals[NVP];
#pragma unroll
    for (int k = 0; k < NVP; ++k) vals[k] = 0.f;
    vals[0] = (MODE == LL_GRAD) ? hsum(st0) : 0.f;
    vals[1] = (MODE == LL_GRAD) ? norm_term(hsum(st1), hsum(wsum) + (float)(W * n_whole)) : 0.f;
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int kind = LK::kinds[i];
      constexpr int off = [] { int n = 0; for (int j = 0; j < i; ++j) n += static_nacc(LK::kinds[j]); return n; }();
      constexpr int G = static_nacc(kind);
      float tmp[G];
#pragma unroll
      for (int k = 0; k < G; ++k) tmp[k] = hsum(accL[off + k]);
      if constexpr (kind == K_EPL) {
        // sum of gP P: the b-gradient is (t - 1) / b times it, the P0-gradient 1 / P0 times it (epl_vjp_v keeps one sum)
        tmp[EPLA_B] = tmp[EPLA_P0] * (dL[i][EPL_TM1] * dL[i][EPL_INVB]);
        tmp[EPLA_P0] *= rcp(dL[i][EPL_P0]);
        // the centre gradients were summed in the lens frame: -(R g) with R the rotation by phi
        const float gxr = tmp[EPLA_CX], gyr = tmp[EPLA_CY], cc = dL[i][EPL_C], ss = dL[i][EPL_S];
        tmp[EPLA_CX] = -(gxr * cc - gyr * ss);
        tmp[EPLA_CY] = -(gxr * ss + gyr * cc);
      }
#pragma unroll
      for (int k = 0; k < G; ++k) vals[NSTAT + off + k] = tmp[k];
    }, std::make_integer_sequence<int, NL>{});
    static_for([&](auto I) {
      constexpr int i = decltype(I)::value;
      constexpr int off = [] {
        int n = 0;
        for (int j = 0; j < i; ++j) n += static_nacc(j < NLL ? LLK::kinds[j < NLL ? j : 0] : SK::kinds[j >= NLL ? j - NLL : 0]);
        return n;
      }();
      constexpr int lkind = i < NLL ? LLK::kinds[i < NLL ? i : 0] : SK::kinds[i >= NLL ? i - NLL : 0];
      if constexpr (is_user_code(lkind)) {  // a user-written light: one sum per parameter, as it stands
#pragma unroll
        for (int k = 0; k < static_nacc(lkind); ++k) vals[NSTAT + NACC_L + off + k] = hsum(accC[off + k]);
      } else {
#pragma unroll
        for (int k = 0; k < SER_NACC; ++k) vals[NSTAT + NACC_L + off + k] = hsum(accC[off + k]) * (k == SERA_INVN ? (float)kLn2 : 1.f);
      }
    }, std::make_integer_sequence<int, NLIGHT>{});
    const bool odd = tid & 1, hi = tid & 2;
    float* s_row16 = s_acc + (tid >> 4) * a.Apad;  // [16 rows of the workgroup][Apad]
#pragma unroll
    for (int g = 0; g < NVP / 4; ++g) {
      float r = quad_transpose_sum(vals[4 * g], vals[4 * g + 1], vals[4 * g + 2], vals[4 * g + 3], odd, hi);
      r = dpp_add(r, 0x114, 0xF);  // row_shr:4
      r = dpp_add(r, 0x118, 0xF);  // row_shr:8 -> lanes 12..15 of the row: the row's sums of values 4g .. 4g + 3
      if ((tid & 15) >= 12 && 4 * g + (tid & 3) < NV) s_row16[4 * g + (tid & 3)] = r;
    }
    __syncthreads();
    for (int k = tid; k < NV; k += WG) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) v += s_acc[j * a.Apad + k];
      out[k] = v;
    }
    return;
  }
  float* s_row = s_acc + (tid >> 6) * a.Apad;
  const bool last_lane = (tid & 63) == 63;
  auto put = [&](float v, int idx) {
    if (!GL_DBG(a.dbg, 2)) v = wave_sum63(v);
    if (last_lane) s_row[idx] = v;
  };
  put((MODE == LL_FWD || MODE == LL_GRAD) ? hsum(st0) : 0.f, 0);
  put((MODE == LL_FWD || MODE == LL_GRAD) ? norm_term(hsum(st1), hsum(wsum) + (float)(W * n_whole)) : 0.f, 1);
  if (last_lane) { s_row[2] = 0.f; s_row[3] = 0.f; }
  __syncthreads();
  for (int k = tid; k < a.A; k += WG) {
    out[k] = (s_acc[k] + s_acc[a.Apad + k]) + (s_acc[2 * a.Apad + k] + s_acc[3 * a.Apad + k]);
  }
}

}  // namespace glk
