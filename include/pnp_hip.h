/* pnp_hip.h -- C ABI of the MI355X (gfx950) PnP-SVRG/SAGA/SARAH hot path.
 *
 * The reference (vmonardo/pnp-svrg @ v1) is pure Python: it has no FFI of its own.  Its
 * boundary for this path is the duck-typed protocol of its algorithms/, problems/ and
 * denoisers/ packages.  This header is the native side a maintainer binds with
 * ctypes (INTEGRATION.md shows the stub); each entry point names the reference code it
 * replaces.
 *
 * Conventions
 *  - every data pointer is a DEVICE pointer (HBM); `stream` is a hipStream_t passed as void*;
 *  - nothing here allocates, frees or synchronises inside a hot call: plans own their
 *    workspaces (created/destroyed explicitly), so every call may be captured in a hipGraph;
 *  - `dtype`: PNP_F32 (production) or PNP_F64 (parity/debug); complex = interleaved (re,im);
 *  - images are row-major [B][H][W]; B independent problems ("batch") per call;
 *  - return 0 on success, nonzero on error; text via pnp_last_error() (thread-local).
 */
#ifndef PNP_HIP_H
#define PNP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PNP_F32 = 0, PNP_F64 = 1 };
enum { PNP_OK = 0, PNP_ERR_ARG = 1, PNP_ERR_HIP = 2, PNP_ERR_UNSUPPORTED = 3 };

int pnp_version(void);
const char* pnp_last_error(void);

/* ------------------------------------------------------------------ CSMRI masked FFT
 * Replaces problems/CSMRI.py:76-81 (grad_full) and :83-89 (grad_stoch), i.e.
 *   g = Re ifft2( sel o fft2(a - b) - sel o Y )
 * computed with real FFTs on the Hermitian-symmetrised k-space residual.            */
typedef struct pnp_csmri_plan pnp_csmri_plan;

/* H == W in {64, 128, 256}.  The plan owns a [batch][W/2][H] complex workspace + twiddles. */
int pnp_csmri_plan_create(pnp_csmri_plan** plan, int H, int W, int batch, int dtype);
int pnp_csmri_plan_destroy(pnp_csmri_plan* plan);

/* Selector (sampling mask, or mask o minibatch) from flat row-major k-space indices, as
 * np.flatnonzero(mask) / problems/CSMRI.py:66-74 produce them.  idx: [batch][n] int32,
 * selT: [batch][W][H] uint8 (TRANSPOSED: the column pass reads along ky).              */
int pnp_csmri_sel_from_indices(pnp_csmri_plan* plan, const int32_t* idx, int n, uint8_t* selT, void* stream);
/* Bit-packed form of a transposed selector: bitsT [batch][W][H/32] uint32, bit (ky & 31) of word [kx][ky >> 5].
 * The sampling mask is kept in this form (8 KiB per 256 x 256 problem instead of 64 KiB).               */
int pnp_csmri_pack_mask(pnp_csmri_plan* plan, const uint8_t* selT, uint32_t* bitsT, void* stream);

/* Device-side minibatch draw (problems/CSMRI.py:66-74 semantics: `mb` of the problem's sampled locations, uniform
 * without replacement; problems of one batch may have different numbers of sampled locations).  Every sampled
 * location i (flat row-major k-space index, as np.flatnonzero(mask) counts) gets a 32-bit key
 *     state = mix64(mix64(mix64(seed) + step) + problem)                       (mix64 = splitmix64 finaliser)
 *     x = lo32(state) ^ i;  x ^= x >> 16;  x *= 0x7feb352d;  x ^= x >> 15;  x *= 0x846ca68b;  x ^= x >> 16
 *     key(i) = x ^ hi32(state)
 * and the mb smallest (key, i) pairs win.  Outputs, for steps step0 .. step0 + nsteps - 1 (a whole outer iteration in
 * one launch):
 *   mbd     [nsteps][batch] descriptors {uint64 state; uint32 T; uint32 P} (16 bytes): i is in the minibatch iff
 *           key(i) < T or (key(i) == T and i <= P);
 *   selbits [nsteps][batch][W][H/32] uint32 (may be NULL): mask o minibatch in the bit-packed layout of bitsT, which
 *           pnp_csmri_grad_sel takes as its selector -- 8 KiB per 256 x 256 problem-step instead of a 64 KiB byte
 *           selector.
 * Deterministic in (seed, step); NOT NumPy's legacy stream (reference-identical draws come from the host).
 * step_dev (may be NULL): device-resident counter added to `step0`, so the call can be replayed from a hipGraph.
 * mb >= the number of sampled locations selects them all.                                                */
int pnp_csmri_draw_thresholds(pnp_csmri_plan* plan, const uint32_t* bitsT, int mb, uint64_t seed, uint32_t step0,
                              int nsteps, const uint32_t* step_dev, void* mbd, uint32_t* selbits, void* stream);
/* mask o minibatch of ONE step as an explicit transposed selector (mbd: [batch] descriptors of that step).   */
int pnp_csmri_sel_from_thresholds(pnp_csmri_plan* plan, const uint32_t* bitsT, const void* mbd, uint8_t* selT,
                                  void* stream);
/* pnp_csmri_draw_thresholds for one step followed by pnp_csmri_sel_from_thresholds (plan-owned descriptors).   */
int pnp_csmri_draw_minibatch(pnp_csmri_plan* plan, const uint32_t* bitsT, int mb, uint64_t seed, uint32_t step,
                             const uint32_t* step_dev, uint8_t* selT, void* stream);
/* Same, from a dense row-major 0/1 indicator [batch][H][W] (uint8).                      */
int pnp_csmri_sel_from_dense(pnp_csmri_plan* plan, const uint8_t* sel, uint8_t* selT, void* stream);

/* Data term for a selector: yh = Hermitian part of (sel o Y) in the packed transposed
 * half-spectrum layout [batch][W/2][H] complex (column 0 carries kx=0 and kx=W/2).
 * YT: [batch][W][H] complex = Y transposed (measurements, CSMRI.py:32-33).               */
int pnp_csmri_pack_y(pnp_csmri_plan* plan, const void* YT, const uint8_t* selT, void* yh, void* stream);

/* out = alpha * Re ifft2( sel o fft2(a - b) - sel o Y ) + beta * c1 + gamma * c2
 *   b, yh, c1, c2 may be NULL (treated as zero).  out may alias a, c1 or c2.
 *   grad_full(z)            : a=z, sel=mask,   yh=pack(mask),    alpha=1/M0
 *   SVRG correction + step  : a=z, b=w, sel=mask o mb, yh=NULL, alpha=-lr/mb, beta=1 (c1=z),
 *                             gamma=-lr (c2=mu), out=z       (pnp_svrg.py:53,57) */
int pnp_csmri_grad(pnp_csmri_plan* plan, const void* a, const void* b, const uint8_t* selT,
                   const void* yh, double alpha, double beta, const void* c1,
                   double gamma, const void* c2, void* out, void* stream);

/* The same gradient with the other selector form and a per-problem scale.  Exactly one of
 *   selT  != NULL     explicit transposed uint8 selector (as pnp_csmri_grad)
 *   bitsT != NULL     bit-packed transposed selector [batch][W][H/32]: the sampling mask itself (grad_full), or one
 *                     step's row of pnp_csmri_draw_thresholds' selbits (mask o device-drawn minibatch)
 * Data term: yh (packed for exactly this selector, pnp_csmri_pack_y) or YT ([batch][W][H] complex = Y transposed:
 * the selector's data term is then formed inside the column pass, which is what a minibatch selector that exists
 * drawn on the device needs -- grad_stoch of pnp_sgd.py:33 / pnp_saga.py:45); at most one of the two.
 * alpha_vec (may be NULL): [batch] values of `dtype`; problem b uses alpha * alpha_vec[b] -- the 1/M0 of
 * problems/CSMRI.py:81 when the masks of a batch have different counts (Bernoulli masks, CSMRI.py:43-45).    */
int pnp_csmri_grad_sel(pnp_csmri_plan* plan, const void* a, const void* b, const uint8_t* selT, const uint32_t* bitsT,
                       const void* yh, const void* YT, double alpha, const void* alpha_vec, double beta,
                       const void* c1, double gamma, const void* c2, void* out, void* stream);

/* One WHOLE inner iteration of pnp_svrg with the TV prox (algorithms/pnp_svrg.py:52-80: minibatch SVRG direction, step,
 * estimate_sigma, TVDenoiser.denoise, Problem.PSNR) in one kernel, one workgroup per problem, the image register-
 * resident from the first load to the last store (f32 plans of 256 x 256):
 *     out = prox_TV( alpha * alpha_vec[b] * Re ifft2( sel o fft2(a - b) ) + beta * c1 + gamma * c2 )
 * bitsT: bit-packed selector as in pnp_csmri_grad_sel (one step's row of pnp_csmri_draw_thresholds' selbits; no data
 * term: the Y terms of the SVRG difference cancel).  sigma_modifier, fallback_sigma, xrec, sse_out, sigma_out as in
 * pnp_prox_tv (the noise estimate is always made in-kernel).  denoise == 0: stop after the noise estimate and store the
 * stepped image (for a prox that is not this one).  out may alias a, c1 or c2.                              */
int pnp_csmri_svrg_step(pnp_csmri_plan* plan, const void* a, const void* b, const uint32_t* bitsT, double alpha,
                        const void* alpha_vec, double beta, const void* c1, double gamma, const void* c2, void* out,
                        int denoise, double sigma_modifier, double fallback_sigma, const void* xrec, double* sse_out,
                        void* sigma_out, void* stream);

/* The outer-loop refresh of the SVRG loop -- algorithms/pnp_svrg.py:32-38: mu = grad_full(z); w = copy(z) -- folded into the first
 * inner iteration of that outer iteration (:52-80 at j = 0).  There w == z, so the minibatch difference
 * grad_stoch(z, mb) - grad_stoch(w, mb) is exactly zero whatever the minibatch and the iteration is z <- prox(z - lr * mu):
 *     mu_out = alpha_vec[b] * Re ifft2( mask o fft2(z) - Y on the mask )         (= problems/CSMRI.py:76-81, yh packed by
 *                                                                                  pnp_csmri_pack_y for mask_bitsT's mask)
 *     w_out  = z
 *     out    = prox_TV( z + (-lr) * mu_out )              [+ noise estimate, PSNR error: as pnp_csmri_svrg_step]
 * in ONE kernel -- bit for bit what pnp_csmri_grad_sel (bits form, batch >= 192) + a copy + pnp_csmri_svrg_step(a = z,
 * b = w, c1 = z, c2 = mu, beta = 1, gamma = -lr) produce, without the second transform pair and the copy.  w_out and mu_out
 * must not alias z, out or each other; out may alias z.                                                      */
int pnp_csmri_svrg_outer_step(pnp_csmri_plan* plan, const void* z, const uint32_t* mask_bitsT, const void* yh,
                              const void* alpha_vec, double lr, void* w_out, void* mu_out, void* out, int denoise,
                              double sigma_modifier, double fallback_sigma, const void* xrec, double* sse_out,
                              void* sigma_out, void* stream);

/* A whole OUTER iteration of the SVRG loop with the TV prox -- algorithms/pnp_svrg.py:32-95 for T2 inner iterations: the refresh
 * mu = grad_full(z), w = z, then T2 times { minibatch SVRG direction, step, estimate_sigma, TVDenoiser.denoise, PSNR } -- in ONE
 * launch: the workgroup that owns a problem runs pnp_csmri_svrg_outer_step and then T2 - 1 times pnp_csmri_svrg_step (a = z,
 * b = w, c1 = z, c2 = mu, alpha = -lr / mini_batch_size, beta = 1, gamma = -lr, out = z) on THAT problem back to back; the
 * results are bit for bit those of the T2 separate calls.  z is updated in place, w and mu are outputs.
 *   selbits  [T2][batch][W][H/32]: slot j = the selector of inner iteration j (pnp_csmri_draw_thresholds' layout; slot 0 is
 *            not read: at j = 0 the minibatch difference is exactly zero);
 *   sse_log  [n_log][batch] double: inner iteration j writes row (log_row0 + j) % n_log (sum (xrec - z)^2 after its prox);
 *   sigma_out [batch]: the noise estimate of the last inner iteration.                                        */
int pnp_csmri_svrg_outer_iteration(pnp_csmri_plan* plan, void* z, void* w, void* mu, const uint32_t* mask_bitsT, const void* yh,
                                   const void* alpha_vec, const uint32_t* selbits, int T2, double lr, int mini_batch_size,
                                   double sigma_modifier, double fallback_sigma, const void* xrec, double* sse_log,
                                   int log_row0, int n_log, void* sigma_out, void* stream);

/* ------------------------------------------------------------------ Deblur / super-resolution
 * Replaces problems/DeblurSR.py:119-147: 1-D circular blur of the raveled image via a length-H*W FFT
 * (spectrum of the kernel computed once at plan creation), optional 4-tap bilinear down-sampler
 * (pylops Bilinear semantics; its adjoint as a deterministic CSR gather).  H*W in {64^2, 128^2, 256^2}.
 * Plan-creation arrays are HOST pointers: Bk [H*W] blur kernel (DeblurSR.py:93, already / N, `dtype`);
 * for scale_percent == 100 pass M = H*W and NULL operators; else g_idx/g_w [M][4] (forward taps) and
 * a_rowptr [H*W+1], a_col/a_val [nnz] (CSR of the adjoint).                                       */
typedef struct pnp_deblur_plan pnp_deblur_plan;
int pnp_deblur_plan_create(pnp_deblur_plan** plan, int H, int W, int batch, int dtype, const void* Bk, int M,
                           const int32_t* g_idx, const void* g_w, const int32_t* a_rowptr,
                           const int32_t* a_col, const void* a_val);
int pnp_deblur_plan_destroy(pnp_deblur_plan* plan);
/* out = scale * B^T S^T ( sel o (S B z - Y) )   (grad_full: sel = NULL, scale = 1/M; grad_stoch:
 * sel = minibatch indicator uint8 [batch][M], scale = 1).  z, out [batch][H*W]; Y [batch][M].   */
int pnp_deblur_grad(pnp_deblur_plan* plan, const void* z, const void* Y, const uint8_t* sel, double scale,
                    void* out, void* stream);
/* grad_stoch with a device-drawn minibatch: mbd = this step's [batch] threshold descriptors from
 * pnp_draw_thresholds(M, ...); the indicator is re-derived where the residual is masked, never stored.   */
int pnp_deblur_grad_mb(pnp_deblur_plan* plan, const void* z, const void* Y, const void* mbd, double scale,
                       void* out, void* stream);
/* forward model S B x (DeblurSR.py:110-112): out [batch][M]                                     */
int pnp_deblur_forward(pnp_deblur_plan* plan, const void* x, void* out, void* stream);

/* ------------------------------------------------------------------ phase retrieval
 * Replaces problems/PR.py:75-87.  A [M][N] row-major, w [N], y [M] (device, `dtype`).
 * out = scale * A_sel^T ( ((|A_sel w| - y_sel)/|A_sel w|) o A_sel w ); rows int32 [nsel] or NULL (all).
 * workspace: pnp_pr_workspace_elems(M, N) elements of `dtype`.                                  */
size_t pnp_pr_workspace_elems(int M, int N);
int pnp_pr_grad(const void* A, const void* w, const void* y, const int32_t* rows, int nsel, int M, int N,
                int dtype, double scale, void* workspace, void* out, void* stream);
/* B independent problems per call: A [batch][M][N], w [batch][N], y [batch][M], rows [batch][nsel] (NULL = all rows),
 * out [batch][N]; workspace: batch * pnp_pr_workspace_elems(M, N) elements.                                  */
int pnp_pr_grad_batch(const void* A, const void* w, const void* y, const int32_t* rows, int nsel, int M, int N, int batch,
                      int dtype, double scale, void* workspace, void* out, void* stream);
/* One power-iteration step of PhaseRetrieval.spec_init (problems/PR.py:50-63): out = scale * A^T (y o (A v)), i.e.
 * D v for D = A^T diag(y) A / M (scale = 1/M) without forming the N x N matrix.  v, out [N]; same workspace.      */
int pnp_pr_spectral_apply(const void* A, const void* v, const void* y, int M, int N, int dtype, double scale,
                          void* workspace, void* out, void* stream);

/* ------------------------------------------------------------------ prox / noise estimate
 * estimate_sigma(z0, multichannel=True, average_sigmas=True) (algorithms/pnp_svrg.py:71):
 * per-column db2 MAD, mean over columns.  sigma_out: [batch] (dtype).                    */
int pnp_sigma_est(const void* z, int H, int W, int batch, int dtype, void* sigma_out, void* stream);

/* TVDenoiser.denoise (denoisers/TV.py:21-26 = per-column Haar BayesShrink) fused with the
 * noise estimate that feeds it and with the squared-error sum of Problem.PSNR
 * (problems/problem.py:33-35).
 *   sigma used = sigma_est*sigma_modifier if sigma_est > 0 else fallback_sigma
 *   sigma_est  = sigma_in[b] if sigma_in != NULL else estimated in-kernel
 *   xrec, sse_out may be NULL; sse_out: [batch] double = sum (xrec - out)^2
 *   sigma_out (may be NULL): [batch] (dtype) the sigma_est that was used.               */
int pnp_prox_tv(const void* z_in, void* z_out, int H, int W, int batch, int dtype,
                const void* sigma_in, double sigma_modifier, double fallback_sigma,
                const void* xrec, double* sse_out, void* sigma_out, void* stream);

/* NLMDenoiser.denoise (denoisers/NLM.py:22-27 -> skimage 0.18 _nl_means_denoising_2d, slow mode,
 * Schraudolph fast_exp).  patch_size as the caller passes it (even sizes are bumped to the
 * next odd one like skimage: 4 -> 5; supported sides 3/5/7), patch_distance in [1, 8].
 *   sigma_in != NULL : h = sigma = sigma_in[b]*sigma_modifier, var = 2 sigma^2   (NLM.py:25)
 *   sigma_in == NULL : h = fixed_h, var = 0                                     (NLM.py:27)
 *   w0 [side*side] (device, double) = exp(-(x^2+y^2)/(2A^2)), A = (side-1)/4, and w0_sum = its sum as
 *   NumPy computes it (the caller builds both once: bit-for-bit the reference's normalisation).
 *   z_out must not alias z_in.  sse_out [batch] double (optional; needs xrec and sse_workspace of
 *   batch*ceil(H/16)*ceil(W/16) doubles).                                                     */
int pnp_nlm2d(const void* z_in, void* z_out, int H, int W, int batch, int dtype, int patch_size,
              int patch_distance, const void* sigma_in, double sigma_modifier, double fixed_h,
              const double* w0, double w0_sum, const void* xrec, double* sse_out, double* sse_workspace,
              void* stream);

/* sum (xrec - z)^2 per problem (Problem.PSNR, problems/problem.py:33-35). sse_out: [batch] double */
int pnp_sse(const void* z, const void* xrec, int n_per_problem, int batch, int dtype, double* sse_out, void* stream);

/* per-problem min and max (RealSN_DnCNN.py:20-22). out: [batch][2] (dtype)               */
int pnp_minmax(const void* z, int n_per_problem, int batch, int dtype, void* out, void* stream);

/* ------------------------------------------------------------------ DnCNN prox (MFMA)
 * Replaces RealSN_DnCNNDenoiser.denoise (denoisers/RealSN_DnCNN.py:16-42) around the 17-layer
 * network of denoisers/DeepDenoisers/model/models.py:5-22 (realSN_models.py:4-21 at inference:
 * the spectral-norm hook only detaches the stored weight).
 * Weights are HOST pointers (plan creation is setup): BatchNorm already folded by the caller.
 *   w_first [64][3][3]            conv(1->64), no bias
 *   w_mid   [n_mid][64][64][3][3] conv(64->64) x BN scale;  b_mid [n_mid][64] folded BN bias
 *   w_last  [64][3][3]            conv(64->1), no bias
 * H % 8 == 0, W % 32 == 0.  The plan owns two [batch][64][H][W] fp32 activation buffers.   */
typedef struct pnp_dncnn_plan pnp_dncnn_plan;
int pnp_dncnn_plan_create(pnp_dncnn_plan** plan, int n_mid, const float* w_first, const float* w_mid,
                          const float* b_mid, const float* w_last, int H, int W, int batch);
int pnp_dncnn_plan_destroy(pnp_dncnn_plan* plan);
/* Biases of the first / last layer and the activation, for networks of the same 3x3-conv shape that are not
 * bias-free ReLU nets: the MMO `simple_CNN` (denoisers/MMODenoise.py:73-101: every conv has a bias, LeakyReLU(0.01),
 * b_mid goes in through plan_create).  b_first: HOST [64] or NULL (= zeros); negative_slope 0 = ReLU.         */
int pnp_dncnn_set_affine(pnp_dncnn_plan* plan, const float* b_first, float b_last, float negative_slope);
/* Conv kernel choice for the 64->64 layers (0, 1, 5: fp32 on the f32 matrix cores):
 *   5 = Winograd F(4x4,3x3) (default where H % 8 == 0 and W % 64 == 0; executes 1/4 of the direct form's multiply-adds;
 *       accuracy envelope: <= 2e-5 absolute against the reference network on its own weights (measured 8e-7), <= 1e-5
 *       relative against a float64 evaluation on white-noise weights -- about 5x the direct form's rounding error),
 *   1 = Winograd F(2,3) along x (2/3; the default for the other sizes),
 *   0 = direct implicit GEMM (bit-for-bit an fmaf chain; the one-flag way back for parity runs),
 *   6 = opt-in: F(4x4,3x3) on the BF16 matrix cores, every fp32 factor split exactly into three bf16 terms and the six
 *       products above 2^-24 summed in fp32 -- the same error envelope as 5 (measured: equal to 5's against float64),
 *       not the reference's arithmetic operation for operation, and at present SLOWER than 5; H % 8 == 0,
 *       W % 64 == 0.
 * The default comes from the environment variable PNP_DNCNN_WINOGRAD at plan creation (6 as above; unset or any other value = 5,
 * falling back to 1 where the image size does not allow it).                                                          */
int pnp_dncnn_set_winograd(pnp_dncnn_plan* plan, int enable);
/* raw network: r = net(x), x and r [batch][H][W] fp32 (the predicted noise residual)          */
int pnp_dncnn_forward(pnp_dncnn_plan* plan, const float* x, float* r, void* stream);
/* the whole denoise(): min-max normalise, scale by 1 + sigma_net/255/2, x - net(x), undo both;
 * z_in/z_out/xrec in `dtype` (may alias); sse_out [batch] double = sum (xrec - z_out)^2 or NULL. */
int pnp_dncnn_denoise(pnp_dncnn_plan* plan, const void* z_in, void* z_out, int dtype, double sigma_net,
                      const void* xrec, double* sse_out, void* stream);

/* MMODenoiser.denoise (denoisers/MMODenoise.py:122-128 around apply_model :18-40 and simple_CNN.forward :88-101):
 * z_out = clip(xc + net(xc), 0, 1) with xc = clip(z_in, 0, 1) in fp32.  The reference feeds the TRANSPOSED image
 * (np.moveaxis on a 2-D array); the caller gets the same result by creating the plan with every 3x3 kernel
 * transposed (conv(x^T, w)^T == conv(x, w^T)).  z_in/z_out/xrec in `dtype`; sse_out as in pnp_dncnn_denoise.  */
int pnp_mmo_denoise(pnp_dncnn_plan* plan, const void* z_in, void* z_out, int dtype, const void* xrec,
                    double* sse_out, void* stream);

/* In-band timing of the MFMA conv launches (measurement aid for bench.py): between begin and end
 * every forward/denoise call brackets its n_mid conv launches with hipEvents on the caller's
 * stream (no synchronisation until _end).  _end returns the mean duration of one conv launch.  */
int pnp_dncnn_profile_begin(pnp_dncnn_plan* plan, int max_calls);
int pnp_dncnn_profile_end(pnp_dncnn_plan* plan, double* avg_ms_per_launch, long* launches);
/* Diagnostic (allocates + synchronises; never on the hot path): median in-kernel shader cycles and 100 MHz
 * reference ticks of the conv tile loop after `reps` back-to-back launches -> the clock held under load. */
int pnp_dncnn_debug_clock(pnp_dncnn_plan* plan, int reps, double* cycles, double* ref_ticks, void* stream);
/* Test hooks (never on the hot path): ONE 64->64 layer of the plan's network, kernel as selected by pnp_dncnn_set_winograd, on
 * CALLER-provided activation buffers in/out [batch][64][H][W] fp32 -- so that a test can put guard bands around them
 * (tests/test_gpu_dncnn.py::test_wino44_guard_bands).  w44_override (may be NULL; mode 5 only): packed F(4x4,3x3) weights of
 * the layer in the caller's memory, pnp_dncnn_debug_w44_floats() floats as pnp_dncnn_debug_w44_weights copies them out.
 * w44_rows: 0 = the production choice of region form, 1 / 2 = 4 x 64 / 8 x 64 regions for the whole layer (mode 5 only).        */
size_t pnp_dncnn_debug_w44_floats(void);
int pnp_dncnn_debug_w44_weights(pnp_dncnn_plan* plan, int layer, float* dst, void* stream);
int pnp_dncnn_debug_mid_layer(pnp_dncnn_plan* plan, int layer, const float* in, float* out, const float* w44_override,
                              int w44_rows, void* stream);

/* Device-resident step counter and log ring (hipGraph replay of a whole outer iteration: nothing in the graph
 * depends on a host-side step index).  pnp_log_append: log[(*step_dev % n_log)][0..n) = src[0..n).        */
int pnp_counter_add(uint32_t* counter, uint32_t inc, void* stream);
int pnp_log_append(const double* src, int n, double* log, int n_log, const uint32_t* step_dev, void* stream);
/* the same with a log-owned counter that the call also advances: log[(*counter % n_log)] = src; ++*counter          */
int pnp_log_append_inc(const double* src, int n, double* log, int n_log, uint32_t* counter, void* stream);

/* ------------------------------------------------------------------ minibatches over M measurements, SAGA table
 * Problem.select_mb (problems/problem.py:110-117: `np.random.choice(M, size, replace=False)` -> 0/1 indicator) on the
 * device, with the same key / threshold construction as pnp_csmri_draw_thresholds (candidates 0 .. M-1):
 * mbd [nsteps][batch] descriptors; the Deblur gradient consumes a step's descriptors directly (pnp_deblur_grad_mb),
 * or they are expanded to an indicator sel [batch][M] (uint8) / an ascending row list rows [batch][mb] (int32, the
 * form pnp_pr_grad takes).                                                                                */
int pnp_draw_thresholds(int M, int batch, int mb, uint64_t seed, uint32_t step0, int nsteps, const uint32_t* step_dev,
                        void* mbd, void* stream);
int pnp_indicator_from_thresholds(int M, int batch, const void* mbd, uint8_t* sel, void* stream);
int pnp_rows_from_thresholds(int M, int batch, int mb, const void* mbd, int32_t* rows, void* stream);
/* indicator from host-drawn index lists (np.random.choice output): idx [batch][n] int32 -> sel [batch][M] uint8   */
int pnp_indicator_from_indices(const int32_t* idx, int n, int M, int batch, uint8_t* sel, void* stream);
/* One SAGA step (algorithms/pnp_saga.py:43-57) in one pass over the vectors (n = batch * N elements):
 *   old = slot; sum += g - old; z -= lr * ((g - prev) + sum * inv_hist); slot = g
 * g = the new minibatch gradient (already / mini_batch_size), slot = the table row being replaced, prev = the row
 * written by the previous step (may be the same row), sum = running sum of the table (== sum(table), :47).   */
int pnp_saga_table_update(void* z, const void* g, void* slot, const void* prev, void* sum, double lr, double inv_hist,
                          size_t n, int dtype, void* stream);

/* ------------------------------------------------------------------ elementwise
 * out = a*x + b*y + c*w   (y, w may be NULL); n = total element count.
 * Covers z -= lr*v (pnp_gd.py:35), SAGA/SARAH combines (pnp_saga.py:47, pnp_sarah.py:72). */
int pnp_axpbypcz(double a, const void* x, double b, const void* y, double c, const void* w,
                 void* out, size_t n, int dtype, void* stream);

/* ------------------------------------------------------------------ host: the legacy RNG draw of select_mb
 * np.random.choice(pool, size, replace=False) (problems/CSMRI.py:72, problems/problem.py:114) on the legacy MT19937 stream,
 * restated in C (no device work): out[k] = pool[perm[k]] (pool NULL: perm[k]) for the first `size` entries of
 * permutation(pop).  mt_key [624] / *mt_pos are the stream's state as np.random.get_state() returns it and are advanced
 * exactly as NumPy advances them; work: [pop] scratch.                                                               */
int pnp_legacy_choice(uint32_t* mt_key, int* mt_pos, const int64_t* pool, int pop, int size, int32_t* work, int64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* PNP_HIP_H */
