/* gigalens_hip.h -- C ABI of the MI355X-native gigalens hot path.
 *
 * Plain pointers and sizes only: no torch / C++ types cross this boundary.
 * Every device buffer is owned by the caller (PyTorch in this repo); the
 * library allocates device memory only inside gl_model_create and the
 * gl_model_set_* set-up calls (model descriptor, grid, PSF, prior table, image
 * positions, galaxy catalogues, series fields) and never per compute call, so
 * every compute entry point is hipGraph-capturable.  All arithmetic is fp32, as in
 * the reference (two documented exceptions run in fp64: the one-off series
 * precompute and the core of the TNFW bracket)
 * (every constant there is tf.float32: tf/simulator.py:27-32,46-51;
 * tf/model.py:63-68,301-306).
 *
 * The reference (furcelay/gigalens) has no FFI -- its plugin boundary is a set
 * of Python ABCs.  Each entry point below names the reference interface it
 * replaces (paths relative to the reference repo root).  INTEGRATION.md shows
 * the ctypes stub a reference maintainer would add.
 */
#ifndef GIGALENS_HIP_H
#define GIGALENS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Error convention: 0 on success, negative gl_status otherwise; the message is
 * available through gl_last_error() (thread local).  Numerical NaN is data,
 * not an error (mirrors the reference: tf/simulator.py:140, tf/inference.py:44). */
typedef enum gl_status {
  GL_OK = 0,
  GL_EINVAL = -1,       /* bad argument (null pointer, negative size, unknown kind) */
  GL_EUNSUPPORTED = -2, /* valid request this build cannot serve */
  GL_ELAUNCH = -3,      /* HIP runtime / launch failure */
  GL_ENOMEM = -4,       /* allocation failure in gl_model_create, or workspace too small */
  GL_ENODEVICE = -5     /* no HIP device */
} gl_status;

/* Profile kinds.  Parameter order inside the packed parameter row is the
 * reference's `_params` list (+ `_amp` for light profiles, profile.py:40-41). */
typedef enum gl_kind {
  /* mass profiles: MassProfile.deriv (profile.py:63-82) */
  GL_EPL = 1,   /* tf/profiles/mass/epl.py:13   [theta_E,gamma,e1,e2,center_x,center_y] */
  GL_SIE = 2,   /* tf/profiles/mass/sie.py:8    [theta_E,e1,e2,center_x,center_y] */
  GL_NFW = 3,   /* tf/profiles/mass/nfw.py:8    [Rs,alpha_Rs,center_x,center_y] */
  GL_SHEAR = 4, /* tf/profiles/mass/shear.py:8  [gamma1,gamma2] */
  GL_SIS = 5,   /* tf/profiles/mass/sis.py:7    [theta_E,center_x,center_y] */
  GL_DPIS = 6,  /* tf/profiles/mass/piemd.py:27 [theta_E,r_core,r_cut,center_x,center_y] */
  GL_DPIE = 7,  /* tf/profiles/mass/piemd.py:99 [theta_E,r_core,r_cut,center_x,center_y,e1,e2] */
  GL_DPIEP = 8, /* tf/profiles/mass/piep.py:23  [theta_E,Ra,Rs,center_x,center_y,e1,e2] */
  GL_SCALED = 9, /* tf/profiles/mass/scaling_relation.py:6-70 (DPIESubhalo, dpie_subhalo.py:6-21): a catalogue of
                    galaxies of one dPIE-family profile whose theta_E / r_core / r_cut follow (L/L*)^power * scale;
                    the packed parameters are the scales, in the order of the reference's `scaling_params`;
                    iparam = their number (1..3); the catalogue is attached with gl_model_set_catalogue */
  GL_SERIES = 10, /* tf/series/series_profile.py:9-95 with dpie_series.py / scaling_series.py / dpie_subhalo_series.py:
                     the (scaled) dPIE deflection expanded in the cut radius around r0, precomputed on the model grid;
                     parameters [theta_E, r_cut]; iparam = order (0..5); field attached with gl_model_set_series */
  GL_NFW_ELLIPSE = 11, /* tf/profiles/mass/nfw.py:100   [Rs,alpha_Rs,e1,e2,center_x,center_y] */
  GL_TNFW = 12,        /* tf/profiles/mass/tnfw.py:12   [Rs,alpha_Rs,r_trunc,center_x,center_y] */
  GL_USER_MASS = 13,   /* profile.py:63-82 as an extension point: a body the user wrote (gl_model_create_user); iparam = parameter
                          count, flags = index of the body */
  /* light profiles: LightProfile.light (profile.py:24-60) */
  GL_SERSIC = 16,         /* tf/profiles/light/sersic.py:23-24 [R_sersic,n_sersic,center_x,center_y,Ie] */
  GL_SERSIC_ELLIPSE = 17, /* sersic.py:68-69 [R_sersic,n_sersic,e1,e2,center_x,center_y,Ie] */
  GL_SHAPELETS = 18,      /* tf/profiles/light/shapelets.py:18,34-36 [beta,center_x,center_y,amp0..amp{L-1}] */
  GL_CORE_SERSIC = 19,    /* sersic.py:85-96 [R_sersic,n_sersic,Rb,alpha,gamma,e1,e2,center_x,center_y,Ie] */
  GL_USER_LIGHT = 20      /* profile.py:24-60 as an extension point: a user-written light body (gl_model_create_user) */
} gl_kind;

#define GL_SHAPELETS_NMAX_CAP 20 /* largest n_max served (231 amplitudes); above 10 the runtime-order path of the interpreter kernel runs */
#define GL_SHAPELETS_TABLE_NODES 6000
#define GL_FLAG_SHAPELETS_INTERPOLATE 1u /* shapelets.py:20 interpolate=True (table mode) */

typedef struct gl_component {
  int32_t kind;   /* gl_kind */
  int32_t iparam; /* EPL: niter cap (epl.py:15, default 50); SHAPELETS: n_max; SCALED: number of scales; else 0 */
  uint32_t flags; /* GL_FLAG_* */
  int32_t reserved; /* GL_USER_LIGHT: 1 = the last parameter is the profile's linear amplitude (lstsq_simulate solves for it); else 0 */
} gl_component;

/* Pixel grid and camera set-up == what LensSimulator.__init__ precomputes
 * (tf/simulator.py:14-70).  All pointers are HOST pointers, copied at create. */
typedef struct gl_grid {
  int32_t height;      /* rows of the supersampled image  (wcs.n_x * supersample) */
  int32_t width;       /* cols of the supersampled image  (wcs.n_y * supersample) */
  int32_t supersample; /* simulator.py:27 */
  int32_t n_region;    /* N = number of evaluated pixels (rows of tf.where(region), tf/simulator.py:43) */
  const float* grid_x; /* [N] img_X, f64 arithmetic cast to f32 (simulator.py:52-55) */
  const float* grid_y; /* [N] img_Y */
  const int32_t* pix_index; /* [N] row*width+col of each evaluated pixel, or NULL == full grid in row-major order */
  float conversion_factor;  /* det(transform_pix2angle) of the un-supersampled transform (tf/simulator.py:27-29) */
  const float* psf;    /* [psf_h*psf_w] PSF already sampled on the supersampled grid and NOT flipped, or NULL
                          (tf/simulator.py:62-70 flips it and cross-correlates, i.e. a true convolution) */
  int32_t psf_h, psf_w;
} gl_grid;

typedef struct gl_model gl_model; /* opaque; immutable once set up (create + the gl_model_set_* calls, which are host-side
                                     set-up and must not race with compute calls); then safe to share between host threads */

/* Build the model descriptor for PhysicalModel(lenses, lens_light, source_light)
 * (model.py:24-44, tf/model.py:290-306) on LensSimulator's grid.  `comps` lists the
 * n_lens mass profiles, then n_lens_light, then n_src light profiles. */
int gl_model_create(const gl_component* comps, int n_lens, int n_lens_light, int n_src,
                    const gl_grid* grid, gl_model** out);
void gl_model_destroy(gl_model* m);

int gl_model_num_params(const gl_model* m);    /* P: packed constrained parameters per sample */
int gl_model_param_offset(const gl_model* m, int component); /* column of the component's first parameter */
int64_t gl_model_num_pixels(const gl_model* m); /* N */

/* Bytes of caller-owned scratch the calls below need for a batch of B samples. */
size_t gl_workspace_bytes(const gl_model* m, int B);

/* LensSimulator.simulate (tf/simulator.py:109-156): params [B,P] -> img [B,H,W]
 * (H = height/supersample, W = width/supersample), NaN->0, PSF, average-pool, x conversion_factor. */
int gl_simulate_fwd(const gl_model* m, const float* params, int B, float* img,
                    void* workspace, size_t workspace_bytes, void* hip_stream);

/* Partial renders (forward only): LensSimulator.simulate(no_deflection=True) (tf/simulator.py:125-126),
 * simulate_source (:242-269), simulate_lens_light (:271-297), simulate_images (:299-328).
 * parts is a bit set: 1 = apply the deflection, 2 = lens light, 4 = source light
 * (simulate == 7, no_deflection == 6, simulate_source == 4, simulate_lens_light == 2, simulate_images == 5). */
#define GL_PART_DEFLECT 1u
#define GL_PART_LENS_LIGHT 2u
#define GL_PART_SOURCE_LIGHT 4u
int gl_simulate_parts_fwd(const gl_model* m, const float* params, int B, unsigned parts, float* img,
                          void* workspace, size_t workspace_bytes, void* hip_stream);

/* Vector-Jacobian product of the above (what tf.GradientTape supplies, tf/inference.py:33-39):
 * grad_img [B,H,W] -> grad_params [B,P]. */
int gl_simulate_bwd(const gl_model* m, const float* params, const float* grad_img, int B,
                    float* grad_params, void* workspace, size_t workspace_bytes, void* hip_stream);

/* ForwardProbModel.stats_pixels (tf/model.py:89-101) fused with simulate() and, when
 * grad_params != NULL, with its gradient d loglike / d params [B,P].
 *   obs [H,W]; err_or_null [H,W] (error_map, wins over bg_rms/exp_time, tf/model.py:92-95);
 *   mask_or_null [H,W] (simulator.img_region weights); loglike [B]; chi2 [B] (= chi^2, NOT reduced:
 *   the caller divides by count_nonzero(img_region), tf/model.py:100). */
int gl_loglike_fwd_bwd(const gl_model* m, const float* params, const float* obs, const float* err_or_null,
                       const float* mask_or_null, float bg_rms, float exp_time, int B, float* loglike,
                       float* chi2, float* grad_params_or_null, void* workspace, size_t workspace_bytes,
                       void* hip_stream);

/* Unconstrained-space front end: ForwardProbModel.log_prob (tf/model.py:126-167) in one launch sequence.
 * Column k of z is the k-th leaf of the prior in tf.nest.flatten order (tf/model.py:76-87); each column
 * carries its default event-space bijector and its scalar prior (TFP semantics restated):
 *   bijector 0 Identity | 1 Exp | 2 Sigmoid(lo,hi);  prior 0 Normal(a,b) | 1 LogNormal(a,b) | 2 Uniform(lo,hi)
 *   | 3 TruncatedNormal(a,b,lo,hi) with log_norm = log(Phi((hi-a)/b) - Phi((lo-a)/b)).
 * const_row [P] supplies the packed columns no z column drives (the *_constants of PhysicalModel). */
typedef struct gl_zcolumn {
  int32_t param_col;
  int32_t bijector;
  int32_t prior;
  float a, b, lo, hi, log_norm;
} gl_zcolumn;
int gl_model_set_prior(gl_model* m, const gl_zcolumn* cols, int d, const float* const_row);

/* z [B,d] -> logprob [B] = loglike + log prior(x) + log|dx/dz|, loglike [B], red_chi2 [B] = chi^2 / chi2_divisor
 * (the caller passes count_nonzero(img_region), tf/model.py:100) and, when grad_z_or_null != NULL,
 * d logprob / d z [B,d] (what tf.GradientTape returns at tf/inference.py:33-39). */
#define GL_TERM_PIXELS 1u    /* include_pixels    (tf/model.py:152-156) */
#define GL_TERM_POSITIONS 2u /* include_positions (tf/model.py:157-161), needs gl_model_set_positions */
int gl_logprob_fwd_bwd(const gl_model* m, const float* z, const float* obs, const float* err_or_null,
                       const float* mask_or_null, float bg_rms, float exp_time, int B, float* logprob,
                       float* loglike, float* red_chi2, float* grad_z_or_null, float chi2_divisor, unsigned terms,
                       void* workspace, size_t workspace_bytes, void* hip_stream);

/* Image-position likelihood, ForwardProbModel.stats_positions (tf/model.py:103-124) with LensSimulator.beta and
 * .magnification (tf/simulator.py:72-91): observed positions of multiply-imaged sources, grouped in families.
 * x, y, err_x, err_y: concatenation over families (HOST pointers, copied).  loglike [B], chi2 [B] (sum over
 * families, not reduced: the caller divides by n_position = 2 * total images, tf/model.py:74,123) and the
 * gradient w.r.t. the packed parameters [B,P] (zero outside the lens columns). */
int gl_model_set_positions(gl_model* m, int n_families, const int* family_sizes, const float* x, const float* y,
                           const float* err_x, const float* err_y);
int gl_positions_fwd_bwd(const gl_model* m, const float* params, int B, float* loglike, float* chi2,
                         float* grad_params_or_null, void* workspace, size_t workspace_bytes, void* hip_stream);

/* ScalingRelation.hessian on arbitrary points (scaling_relation.py:72-83): out [4][n_pts][B] = f_xx, f_xy, f_yx, f_yy
 * summed over the catalogue; other arguments as gl_scaled_eval. */
int gl_scaled_hessian(int base_kind, int n_galaxies, const int32_t scale_col[3], const float* table_dev, const float* x,
                      const float* y, int64_t n_pts, int B, int xy_batched, const float* scales, int n_scales,
                      float* out, void* hip_stream);

/* Series-expansion accelerator (MassSeries.set_grid / set_constants / set_deriv, tf/series/series_profile.py:54-62;
 * ScalingRelationSeries.precompute_deriv, scaling_series.py:19-35; DPIESeries.precompute_deriv, dpie_series.py:19-33).
 * gl_series_precompute: Taylor coefficients C_n, n = 0..order, of the population's deflection per unit amplitude in the
 *   cut-radius scale, at arbitrary points: coeffs [2][order+1][n_pts] (x components then y components).  Catalogue
 *   arguments as gl_model_set_catalogue / gl_scaled_eval (table_dev: DEVICE [n_galaxies][7]); scales: HOST [n_scales],
 *   the entry of theta_E is ignored (amplitude 1), the entry of r_cut is the expansion point r0 and must be scaled.
 *   The reference's f_n (n-th derivatives, summed with 1/n!) are n! C_n.
 * gl_model_set_series: attach a field computed on the model's own pixel list ([2][order+1][N], DEVICE; copied) to a
 *   GL_SERIES lens; at run time  alpha = theta_E * sum_n C_n (r_cut - r0)^n  (series_profile.py:76-95).
 * gl_series_eval: that polynomial on a field, out0/out1 [n_pts][B]  (MassSeries.deriv at plugin level).
 * The Hessian half (MassSeries.set_hessian / .hessian, series_profile.py:64-65,83-89; DPIESeries.precompute_hessian,
 * dpie_series.py:35-49; ScalingRelationSeries.precompute_hessian, scaling_series.py:37-54):
 * gl_series_precompute_hessian: same arguments, coeffs [3][order+1][n_pts] = series of f_xx, f_xy, f_yy.
 * gl_series_hessian_eval: out [3][n_pts][B] = theta_E * sum_n C_n (r_cut - r0)^n per field.
 * gl_model_set_series_hessian: attach the Hessian field of a GL_SERIES lens ([3][order+1][N], DEVICE; copied) so that
 *   gl_lens_maps on the model's own grid (x = y = NULL) can include the lens; gl_model_set_series must come first. */
int gl_series_precompute(int base_kind, int n_galaxies, const int32_t scale_col[3], const float* table_dev,
                         const float* scales, int n_scales, int order, const float* x_dev, const float* y_dev,
                         int64_t n_pts, float* coeffs_dev, void* hip_stream);
int gl_model_set_series(gl_model* m, int component, float r0, const float* coeffs_dev);
int gl_series_eval(const float* coeffs_dev, int order, int64_t n_pts, int B, const float* theta_E, const float* r_cut,
                   float r0, float* out0, float* out1, void* hip_stream);
int gl_series_precompute_hessian(int base_kind, int n_galaxies, const int32_t scale_col[3], const float* table_dev,
                                 const float* scales, int n_scales, int order, const float* x_dev, const float* y_dev,
                                 int64_t n_pts, float* coeffs_dev, void* hip_stream);
int gl_series_hessian_eval(const float* coeffs_dev, int order, int64_t n_pts, int B, const float* theta_E,
                           const float* r_cut, float r0, float* out, void* hip_stream);
int gl_model_set_series_hessian(gl_model* m, int component, const float* coeffs_dev);

/* Linear-amplitude solve, LensSimulator.lstsq_simulate (tf/simulator.py:158-240): every light component is rendered
 * as `depth` basis images of unit amplitude (Sersic: 1; Shapelets: (n_max+1)(n_max+2)/2), NaN -> 0, PSF and
 * pooling per channel, then  coeffs = pinv(X^T X, rcond=1e-6) X^T Y  with X = stack / err_map, Y = obs / err_map.
 *   params   [B,P]; the amplitude columns (gl_model_linear_column) are ignored
 *   parts    GL_PART_DEFLECT | GL_PART_LENS_LIGHT | GL_PART_SOURCE_LIGHT; drop GL_PART_DEFLECT for no_deflection=True
 *   coeffs   [B,D] (return_coeffs) | stacked [B,D,H,W] (return_stacked; the reference's layout is (B,H,W,D)) |
 *   image    [B,H,W] = sum_d coeffs_d stack_d (default return; no det(T) factor, it is absorbed by the coefficients);
 *            any of the three may be NULL.  obs, err: [H,W], required unless only `stacked` is requested.
 * Equivalent forward parameters: amplitude_k = coeffs_k / conversion_factor. */
int gl_model_num_linear(const gl_model* m);            /* D = sum of the light profiles' depth */
int gl_model_linear_column(const gl_model* m, int k);  /* packed-parameter column of linear coefficient k */
size_t gl_lstsq_workspace_bytes(const gl_model* m, int B);
int gl_lstsq_fwd(const gl_model* m, const float* params, const float* obs, const float* err, int B, unsigned parts,
                 float* coeffs_or_null, float* stacked_or_null, float* image_or_null, void* workspace,
                 size_t workspace_bytes, void* hip_stream);
/* Measurement aid: where, inside a workspace of gl_lstsq_workspace_bytes(m, B), the per-sample int32 flags of the most recent
 * solve live: 0 = the Cholesky attempt proved tf.linalg.pinv's rcond cut idle and solved the system (csrc/gl_lstsq.hip.h
 * gl_chol_solve_kernel), 1 = left to the eigenvalue solve.  GL_EUNSUPPORTED for systems the attempt does not serve (> 127). */
int gl_lstsq_solve_flags(const gl_model* m, int B, size_t* offset_bytes);

/* Galaxy catalogue of a GL_SCALED component (ScalingRelation.__init__, scaling_relation.py:27-55).  Must be attached
 * to every GL_SCALED component before gl_workspace_bytes / any compute call (the workspace holds one block of
 * constants per sample and galaxy).
 *   base_kind  GL_DPIS, GL_DPIE or GL_DPIEP (`profile`)
 *   table      HOST [n_galaxies][7] rows (theta_E, r_core, r_cut, center_x, center_y, e1, e2): for a scaling
 *              parameter the entry is (L_g/L*)^power (`_unscaled_params`, :52-55), otherwise the catalogue constant
 *              (`_galaxy_constants`, :48-51; e1, e2 ignored for GL_DPIS)
 *   scale_col  for theta_E, r_core, r_cut: position of its scale inside the component's parameter row, or -1 when
 *              the quantity is a catalogue constant */
int gl_model_set_catalogue(gl_model* m, int component, int base_kind, int n_galaxies, const int32_t scale_col[3],
                           const float* table);

/* ScalingRelation.deriv on arbitrary points (scaling_relation.py:61-70); arguments as gl_profile_eval, with
 * table a DEVICE pointer [n_galaxies][7] and scales [B][n_scales]. */
int gl_scaled_eval(int base_kind, int n_galaxies, const int32_t scale_col[3], const float* table_dev, const float* x,
                   const float* y, int64_t n_pts, int B, int xy_batched, const float* scales, int n_scales,
                   float* out0, float* out1, void* hip_stream);

/* LensSimulator.beta / .magnification / .convergence / .shear on arbitrary points (tf/simulator.py:72-107):
 * out [6][n_pts][B] = beta_x, beta_y, f_xx, f_xy, f_yx, f_yy, the Hessian summed over the lenses as `lens.hessian`
 * resolves it in the reference (tf/profile.py:9-43 autodiff; analytic overrides incl. piemd.py:62-83).
 * x, y as in gl_profile_eval ([n_pts,B] when xy_batched, else [n_pts]); params [B,P] (only lens columns are read).
 * x = y = NULL selects the model's own evaluation grid (n_pts must equal its size, xy_batched 0); that is the only
 * form a model with GL_SERIES lenses accepts -- their fields live on that grid (series_profile.py:76-89) -- and it
 * needs gl_model_set_series_hessian on each of them. */
int gl_lens_maps(const gl_model* m, const float* params, int B, const float* x, const float* y, int64_t n_pts,
                 int xy_batched, float* out, void* hip_stream);

/* Plugin-level point evaluation, the reference's MassProfile.deriv / LightProfile.light called on
 * arbitrary coordinates (tests/test_profiles.py calls exactly these):
 *   x, y [n_pts, B] when xy_batched, else [n_pts] shared by every sample (pixel-major, batch-minor
 *   like the reference, tf/simulator.py:45-51); params [B, n_params(kind)];
 *   out0/out1 [n_pts, B]: (alpha_x, alpha_y) for mass kinds; out0 = light for light kinds (out1 unused). */
int gl_profile_eval(const gl_component* comp, const float* x, const float* y, int64_t n_pts, int B,
                    int xy_batched, const float* params, float* out0, float* out1, void* hip_stream);

/* The optimiser update of the MAP / SVI loops (the reference hands the gradient to a Keras / optax Adam,
 * tf/inference.py:33-39; jax/inference.py:62-68) as ONE launch:
 *   g = grad_scale * grad;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
 *   x -= lr * (m / (1 - b1^t)) / (sqrt(v / (1 - b2^t)) + eps)        (all [n], DEVICE, x / m / v updated in place)
 * t: the 1-based step count from the host; or t_dev (DEVICE, 16 bytes: a double counter holding the number of steps
 * already taken, followed by 8 zero-initialised bytes the library uses) -- the kernel then uses counter + 1 and
 * advances the counter itself, so the call can sit in a captured HIP graph and be replayed. */
int gl_adam_update(float* x, const float* grad, float* m, float* v, int64_t n, float grad_scale, float lr, float beta1,
                   float beta2, float eps, int64_t t, double* t_dev_or_null, void* hip_stream);

/* The Gaussian surrogate of the SVI step (tf/inference.py:64-91: MultivariateNormalTriL over FillScaleTriL(Exp,
 * diag_shift) / MultivariateNormalDiag over Exp; sharded like jax/inference.py:98-128), as the two launches that bracket
 * the forward+gradient call.  l_packed: the row-major lower triangle (d(d+1)/2, full_rank != 0) or the d log-scales.
 * gl_svi_sample: z [n,d] = mu + L eps, eps [n,d] standard normal draws.
 * gl_svi_grad:   buf [1 + d + len(l_packed)] = [ELBO = mean(log q - log p), dELBO/dmu, dELBO/dl_packed] from eps, logp [n]
 *                and grad_z [n,d] = d log p / d z  -- exactly the buffer the one RCCL all-reduce of the step carries. */
int gl_svi_sample(const float* mu, const float* l_packed, int d, int full_rank, const float* eps, int n, float diag_shift,
                  float* z, void* hip_stream);
int gl_svi_grad(const float* l_packed, int d, int full_rank, const float* eps, const float* logp, const float* grad_z, int n,
                float diag_shift, float* buf, void* hip_stream);

/* The leapfrog of the preconditioned HMC loop (tf/inference.py:95-182; momentum precision = the SVI covariance
 * Sigma = L L^T; any d up to 4096 -- cluster models run at d = 132).  All arrays [n,d] row-major DEVICE float32 unless
 * noted.
 * gl_hmc_kick_drift: p_out = p_in + kick * grad;  z_out = z_in + eps * (p_out Sigma)   (may run in place: p_out == p_in,
 *   z_out == z_in).
 * gl_hmc_accept: closes a transition.  p1 = p_new + kick * grad_new; log_acc = (logp_new - |p1 L|^2/2) - (logp - |p0 L|^2/2),
 *   non-finite -> rejected; chain i moves (z, grad, logp overwritten by the proposal) when log(uniforms[i]) < log_acc;
 *   accept_prob [n] = exp(min(log_acc, 0)).  scale_tril: L [d,d] lower triangular. */
int gl_hmc_kick_drift(const float* p_in, const float* grad, float kick, const float* z_in, const float* sigma, float eps, int n,
                      int d, float* p_out, float* z_out, void* hip_stream);
int gl_hmc_accept(float* z, float* grad, float* logp, const float* z_new, const float* grad_new, const float* logp_new,
                  const float* p0, const float* p_new, float kick, const float* scale_tril, const float* uniforms, int n, int d,
                  float* accept_prob, void* hip_stream);

/* LightProfile.light of a `use_lstsq=True` profile (the unit-amplitude basis images: sersic.py:30-34 `Ie = ones`,
 * `ret[tf.newaxis]`; shapelets.py:61-62,71-72): out [depth][n_pts][B], depth = 1 for the Sersic family and
 * (n_max+1)(n_max+2)/2 for Shapelets.  Other arguments as gl_profile_eval; params keeps the kind's full row width,
 * its amplitude columns are not read. */
int gl_profile_basis(const gl_component* comp, const float* x, const float* y, int64_t n_pts, int B, int xy_batched,
                     const float* params, float* out, void* hip_stream);

/* MassProfile.hessian / convergence / shear at plugin level (tf/profile.py:9-43 and the analytic overrides of
 * nfw.py:77-94, shear.py:18-26, sis.py:19-29, piemd.py:62-83,121-138): out [4][n_pts][B] = f_xx, f_xy, f_yx, f_yy.
 * Free-standing mass kinds only (catalogues and series: gl_lens_maps on a model). */
int gl_profile_hessian(const gl_component* comp, const float* x, const float* y, int64_t n_pts, int B, int xy_batched,
                       const float* params, float* out, void* hip_stream);

int gl_kind_num_params(const gl_component* comp); /* length of the reference's params list for this profile */

/* Measurement hooks (bench.py / rocprof cross-check).  gl_model_set_timing(m, slots): slots > 0 allocates a ring of
 * `slots` HIP-event pairs; every subsequent call records one pair on the CALLER'S stream around its dominant ("main")
 * kernel launch -- no host synchronisation, so a sustained loop can be timed launch by launch; slots == 0 turns the
 * hooks off.  gl_model_set_timing_stride(m, k): only every k-th main launch is bracketed (default 1; an event record costs
 * ~2.5 us of stream time, so a sustained loop is sampled rather than slowed).  gl_model_last_main_ms synchronises on the most recent pair and returns its elapsed milliseconds.
 * gl_model_timing_drain synchronises on every pair recorded since the last drain (at most `slots`, oldest first), writes
 * up to `cap` durations to ms[] and their number to *n, and restarts the ring.
 * gl_model_last_main_kernel: the (mangled) symbol of the kernel the most recent main launch dispatched -- the name
 * rocprofv3 lists and tools/isa_flops.py disassembles. */
int gl_model_set_timing(gl_model* m, int slots);
int gl_model_set_timing_stride(gl_model* m, int stride);
int gl_model_last_main_ms(gl_model* m, float* ms);
int gl_model_timing_drain(gl_model* m, float* ms, int cap, int* n);
int gl_model_last_main_kernel(const gl_model* m, char* buf, size_t cap);
/* Measurement aid: the work decomposition of a gradient call on B samples and where, inside a workspace of
 * gl_workspace_bytes(m, B), the per-(sample, chunk) partial rows [B][n_chunks][row_floats] live after the call (n_chunks is the
 * number of rows a sample owns: its pixel chunks, or more where the cheapest samples of a launch run as more, shorter
 * workgroups -- rows a sample does not use are zero).  Slots 2 and 3
 * of a row are pads the finalize kernel never reads; the shapelet kernel (csrc/gl_shp.hip.h) leaves there how many of the
 * workgroup's wave-tiles ran the shapelet chains and how many it saw -- what bench.py's instruction model weights the
 * conditional blocks with. */
int gl_model_launch_shape(const gl_model* m, int B, int* chunk_px, int* n_chunks, int* row_floats, size_t* partial_offset_bytes);

/* ---- Open plugin boundary: user-written profile bodies (csrc/gl_user.hip) -------------------------------------------------------
 * The reference's extension point is a Python subclass with a TensorFlow body: MassProfile.deriv / LightProfile.light are
 * abstract (src/gigalens/profile.py:58-82).  Here the body is ONE function template in HIP C++ over a number type R,
 *     template <class R> __device__ void deriv(R x, R y, const R* p, R& fx, R& fy);     (mass profile: the deflection)
 *     template <class R> __device__ R    light(R x, R y, const R* p);                   (light profile: surface brightness)
 * with p = the profile's n_params parameters in their declared order.  hiprtc compiles it once for gfx950 with a forward-mode
 * dual number (arithmetic, comparisons, sqrt exp log pow sin cos tan atan atan2 sinh cosh tanh atanh abs fmin fmax resolve by
 * argument-dependent lookup; gl::value(r) gives the plain float of a number for branches).  Compile errors come back verbatim
 * through gl_last_error (GL_EINVAL).  gl_user_profile_check compiles only (no device needed).
 *   eval: x, y [n_pts] or [n_pts,B] (xy_batched), params [B,n_params], out0 (, out1 for mass profiles) [n_pts,B];
 *   jac_or_null [n_out][n_params + 2][n_pts,B]: d out / d (x, y, p_0 .. p_{n-1}) of the same pass (n_out = 2 mass, 1 light).
 * Plugin-level calls only: the pixel kernels of the likelihood path take built-in kinds (gl_model_create). */
/* ... and inside a model: components of kind GL_USER_MASS / GL_USER_LIGHT (iparam = parameter count <= 16, flags = index into
 * `bodies`) make gl_model_create_user compile the interpreter kernel of the likelihood path at run time with those bodies in it:
 * simulate / log-likelihood / fused log-prob and their gradients (forward-mode duals of the body) work as for built-in kinds.
 * The linear-amplitude solve serves them too (round 4: a user-written light with `reserved = 1` contributes one basis image).
 * The image-position likelihood and gl_lens_maps serve them as well (round 4): their kernels are compiled -- when first asked for
 * -- with the mass bodies on the nested duals of the point kernels (Hessians and their parameter derivatives by differentiating
 * `deriv`, as the reference does for every profile, tf/profile.py:9-43).  gl_user_points_check: that compile alone, for one mass
 * body, no device needed. */
int gl_model_create_user(const gl_component* comps, int n_lens, int n_lens_light, int n_src, const gl_grid* grid,
                         const char* const* bodies, int n_bodies, gl_model** out);
/* The compiled interpreter is cached per process, keyed on the program text (bodies, parameter counts, shapelet / family
 * switches): models created with the same bodies -- one LensSimulator per stage and batch size of a modelling sequence -- share
 * one hiprtc compile.  The kernel headers the compile includes are embedded in the library (no source checkout needed).
 * gl_user_model_compile_count: hiprtc compiles of model kernels this process has paid for so far (diagnostics, tests). */
long long gl_user_model_compile_count(void);
int gl_user_points_check(const char* body, int n_params);
typedef struct gl_user_profile gl_user_profile;
int gl_user_profile_check(const char* body, int is_light, int n_params);
int gl_user_profile_create(const char* body, int is_light, int n_params, gl_user_profile** out);
int gl_user_profile_eval(const gl_user_profile* u, const float* x, const float* y, int64_t n_pts, int B, int xy_batched,
                         const float* params, float* out0, float* out1, float* jac_or_null, void* hip_stream);
void gl_user_profile_destroy(gl_user_profile* u);

const char* gl_last_error(void);
const char* gl_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GIGALENS_HIP_H */
