/* dpsx -- C ABI of the MI355X (gfx950) DPS test-time-compute hot path.
 *
 * Drop-in boundary for vishnutez/dps-ttc.  The reference is pure Python and has
 * no FFI of its own; each entry point below names the reference function
 * (file:line under /root/reference) whose device work it replaces, and
 * INTEGRATION.md shows the ctypes binding a maintainer adds on the reference
 * side.  Conventions:
 *   - every pointer is a DEVICE pointer to contiguous fp32 (NCHW) unless the
 *     name ends in _host; the caller owns all buffers;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*),
 *     does no allocation and no host sync (graph-capturable);
 *   - a dpsx_op serves ONE stream at a time: its workspace (caller-provided, sized by
 *     dpsx_op_workspace_bytes) carries the per-tile partial sums from dpsx_step_fwd_f32 to
 *     dpsx_step_bwd_f32, and its arrival counters belong to the launch in flight.  Calls on
 *     one op must be stream-ordered; concurrency is across ops (one op + workspace per
 *     stream -- kernels.ParticleGroups does exactly that).  Calls that take no op are
 *     re-entrant;
 *   - return value: DPSX_OK or a negative DPSX_E* code; nothing throws/exits;
 *   - `n` = particles, `c` = channels, `h`,`w` = image size, chw = c*h*w.
 */
#ifndef DPSX_H
#define DPSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPSX_ABI_VERSION 3

enum {
    DPSX_OK = 0,
    DPSX_EINVAL = -1,     /* null pointer / bad size / bad enum */
    DPSX_EUNSUPPORTED = -2,
    DPSX_ELAUNCH = -3,    /* HIP reported a launch/runtime error */
    DPSX_ENOMEM = -4,
    DPSX_EWORKSPACE = -5  /* caller workspace too small */
};

int dpsx_abi_version(void);
const char *dpsx_strerror(int code);
/* last HIP error string seen by this thread (for DPSX_ELAUNCH / DPSX_ENOMEM) */
const char *dpsx_last_hip_error(void);

/* ---- per-step scalars -----------------------------------------------------
 * The six fp32 table entries one DDPM step reads, replacing eight
 * extract_and_expand() H2D copies per step (gaussian_diffusion.py:769-773,
 * posterior_mean_variance.py:116-117, 121-122, 235-236). */
typedef struct dpsx_coefs {
    float a;        /* sqrt_recip_alphas_cumprod[t]    posterior_mean_variance.py:121 */
    float b;        /* sqrt_recipm1_alphas_cumprod[t]  :122 */
    float c1;       /* posterior_mean_coef1[t]         :116 */
    float c2;       /* posterior_mean_coef2[t]         :117 */
    float min_log;  /* posterior_log_variance_clipped[t]  :235 */
    float max_log;  /* log(betas[t])                   :236 */
    int32_t add_noise; /* bit 0: t != 0 (add the noise term)  gaussian_diffusion.py:473, :503
                        * bit 1: DDIM step instead of DDPM (DDIM.p_sample, gaussian_diffusion.py:481-509); the
                        *        record then carries c1 = sqrt(alphas_cumprod_prev[t]),
                        *        c2 = sqrt(1 - alphas_cumprod_prev[t] - sigma^2), min_log = sigma (eta-scaled, :487-491),
                        *        max_log unused; a, b as above (also predict_eps_from_x_start, :506-509).
                        *        The variance channels of model_out are then not read. */
} dpsx_coefs;

/* ---- S1: p_mean_variance + DDPM.p_sample ----------------------------------
 * gaussian_diffusion.py:308-330, 466-476; posterior_mean_variance.py:96-129
 * (epsilon), :40-45 (clip), :211-242 (learned_range).
 * model_out is [n, 2c, h, w] (eps | v).  x0_hat, sample: [n, c, h, w].
 * inside (optional, may be NULL): uint8 [n*chw], 1 where the pre-clamp value
 * lies in [-1, 1] (the set on which torch's clamp passes gradient). */
int dpsx_posterior_fwd_f32(const float *x_t, const float *model_out, const float *noise,
                           float *x0_hat, float *sample, uint8_t *inside,
                           int64_t n, int64_t chw, const dpsx_coefs *coefs_host, void *stream);

/* VJP of the above, as torch.autograd.grad walks it (condition_methods.py:48,185).
 * g_x0 / g_sample may be NULL (zero cotangent).  g_model_out is [n, 2c, h, w]. */
int dpsx_posterior_bwd_f32(const float *g_x0, const float *g_sample, const float *x_t,
                           const float *model_out, const float *noise,
                           float *g_x, float *g_model_out,
                           int64_t n, int64_t chw, const dpsx_coefs *coefs_host, void *stream);

/* ---- measurement operators A (measurements.py) ---------------------------- */
typedef struct dpsx_op dpsx_op; /* opaque; owns only its constant tables */

enum { DPSX_BLUR_AUTO = 0, DPSX_BLUR_FORCE_TAPS = 1 };

/* GaussialBlurOperator / MotionBlurOperator: ReflectionPad2d(ks/2) + depthwise
 * cross-correlation, same ks x ks kernel on every channel
 * (measurements.py:93-149, util/img_utils.py:268-308).  kernel_host: ks*ks fp32,
 * row-major, HOST memory.  Rank-1 kernels take the separable LDS path unless
 * mode == DPSX_BLUR_FORCE_TAPS. */
int dpsx_op_create_blur(const float *kernel_host, int ks, int mode, dpsx_op **out);

/* SuperResolutionOperator -> Resizer (measurements.py:76-91, util/resizer.py:55-74).
 * w_*_host [taps_*, out_*] fp32 and i_*_host [taps_*, out_*] int64 are
 * Resizer.weights / Resizer.field_of_view for the H and W axes. */
int dpsx_op_create_resize(int64_t in_h, int64_t in_w,
                          const float *w_h_host, const int64_t *i_h_host, int64_t taps_h, int64_t out_h,
                          const float *w_w_host, const int64_t *i_w_host, int64_t taps_w, int64_t out_w,
                          dpsx_op **out);

/* InpaintingOperator (measurements.py:151-168): mask_dev is a DEVICE pointer to
 * h*w fp32 ([1,1,h,w], broadcast over n and c); borrowed, must outlive the op. */
int dpsx_op_create_mask(const float *mask_dev, int64_t h, int64_t w, dpsx_op **out);

/* DenoiseOperator (measurements.py:57-73): identity. */
int dpsx_op_create_identity(dpsx_op **out);

/* PhaseRetrievalOperator (measurements.py:179-189 -> util/img_utils.py:26-30 ->
 * util/fastmri_utils.py:67-89): zero-pad by `pad`, centred orthonormal 2-D FFT,
 * modulus.  Square images of side h; plans are built for up to max_planes = n*c. */
int dpsx_op_create_phase(int64_t h, int64_t pad, int64_t max_planes, dpsx_op **out);

void dpsx_op_destroy(dpsx_op *op);
/* measurement shape for an [*, *, h, w] input */
int dpsx_op_out_shape(const dpsx_op *op, int64_t h, int64_t w, int64_t *out_h, int64_t *out_w);
/* 0 generic taps, 1 separable, 2 resize, 3 mask, 4 identity, 5 phase */
int dpsx_op_kind(const dpsx_op *op);
/* bytes of scratch the op-level and step-level calls below need for n particles */
int64_t dpsx_op_workspace_bytes(const dpsx_op *op, int64_t n, int64_t c, int64_t h, int64_t w);

/* operator.forward(data)  (measurements.py:84,108,142,158,186) */
int dpsx_op_forward_f32(dpsx_op *op, const float *x, float *y,
                        int64_t n, int64_t c, int64_t h, int64_t w,
                        void *workspace, int64_t workspace_bytes, void *stream);
/* exact VJP of operator.forward (what autograd computes for condition_methods.py:48,185).
 * Linear ops ignore x; phase retrieval differentiates at x (re-runs the forward FFT). */
int dpsx_op_adjoint_f32(dpsx_op *op, const float *u, const float *x, float *g,
                        int64_t n, int64_t c, int64_t h, int64_t w,
                        void *workspace, int64_t workspace_bytes, void *stream);

/* ---- residual norm (condition_methods.py:37-39, 179-181; gaussian_diffusion.py:627-630)
 * r = y - ax (y has y_n in {1, n} particles); norm[p] = ||r_p||_2.  Deterministic
 * two-pass reduction (no float atomics).  r may be NULL. */
int dpsx_residual_norm_f32(const float *y, int64_t y_n, const float *ax, float *r, float *norm,
                           int64_t n, int64_t m, void *workspace, int64_t workspace_bytes, void *stream);
/* cotangent on ax of sum_p g_norm[p] * norm_p^power, power in {1, 2}; 0 where norm == 0 */
int dpsx_norm_bwd_f32(const float *r, const float *norm, const float *g_norm, int power,
                      float *g_ax, int64_t n, int64_t m, void *stream);

/* ---- fused DPS step (gaussian_diffusion.py:207-257 + condition_methods.py:33-60,
 *      94-106, 145-187, 198-212), three launches per step:
 *  fwd : S1 + A(x0_hat) + residual + norm   -> x0_hat, sample, inside, resid, norm
 *  bwd : cotangent of scale*norm^power back through A and the clamp to the UNet
 *        output: g_model_out[:, :c] = -b * g_pre,  g_model_out[:, c:] untouched
 *        (the caller zeroes that half once), where g_pre = dLoss/d(pre-clamp x0)
 *  upd : x_{t-1} = sample - (a * g_pre + g_unet)   with a*g_pre = (-a/b) * g_eps
 * `resid` is an op-defined scratch of dpsx_step_resid_bytes() bytes that carries
 * the residual (or, for phase retrieval, the complex cotangent) from fwd to bwd. */
int64_t dpsx_step_resid_bytes(const dpsx_op *op, int64_t n, int64_t c, int64_t h, int64_t w);

/* norm != NULL: the launch finishes the per-particle norms itself (each particle's last block re-sums the partials
 * in a fixed order: deterministic, no extra launch).  norm == NULL: they are finalised by dpsx_step_bwd_f32's
 * prologue from the partial sums left in `workspace` (see there).
 * x0_hat == NULL (blur and resize operators, phase retrieval at the hand-written 384-point geometry): pred_xstart is consumed inside the launch -- A(x0_hat), the clamp gate --
 * and not written out; the `ps` step reads it nowhere afterwards (posterior_mean_variance.py:96-129 returns it to
 * condition_methods.py:33-60, which uses it for the norm only).  The other operators read it back and require it. */
int dpsx_step_fwd_f32(dpsx_op *op, const float *x_t, const float *model_out, const float *noise,
                      const float *y, int64_t y_n,
                      float *x0_hat, float *sample, uint8_t *inside, void *resid, float *norm,
                      int64_t n, int64_t c, int64_t h, int64_t w, const dpsx_coefs *coefs_host,
                      void *workspace, int64_t workspace_bytes, void *stream);

/* norm: the norms dpsx_step_fwd_f32 produced, or NULL when that call was given norm == NULL: the partial sums
 * it left in `workspace` (same buffer, untouched in between) are then finalised in this launch's prologue --
 * saving a launch per step -- and written to norm_out (required in that case, optional otherwise). */
int dpsx_step_bwd_f32(dpsx_op *op, const void *resid, const float *norm, float *norm_out, const uint8_t *inside,
                      const float *x0_hat, const float *y, int64_t y_n,
                      float scale, int power, float *g_model_out,
                      int64_t n, int64_t c, int64_t h, int64_t w, const dpsx_coefs *coefs_host,
                      void *workspace, int64_t workspace_bytes, void *stream);

/* The same launch with one more cotangent on x0_hat: g_x0_extra [n, c, h, w] (NULL = none) is added to
 * coef * A^T r before the clamp gate and the -b scaling.  It carries the gradient of any further loss term that
 * depends on x0_hat only -- the semantic-guidance term of PosteriorSamplingSemanticGuid.measurement_semantic_guidance
 * (condition_methods.py:155-187: sem_guid_scale_t * ||emb(x0_hat) - emb_ref||^p, its VJP taken by the caller
 * through the pluggable embedder) -- so that configuration also runs on the three fused launches. */
int dpsx_step_bwd_extra_f32(dpsx_op *op, const void *resid, const float *norm, float *norm_out,
                            const uint8_t *inside, const float *x0_hat, const float *y, int64_t y_n,
                            float scale, int power, const float *g_x0_extra, float *g_model_out,
                            int64_t n, int64_t c, int64_t h, int64_t w, const dpsx_coefs *coefs_host,
                            void *workspace, int64_t workspace_bytes, void *stream);

int dpsx_step_update_f32(const float *sample, const float *g_model_out, const float *g_unet,
                         float *x_next, int64_t n, int64_t chw, const dpsx_coefs *coefs_host,
                         void *stream);

/* plain x_{t-1} = sample - (g_a + g_b)   (g_b may be NULL)  gaussian_diffusion.py:255 */
int dpsx_update_f32(const float *sample, const float *g_a, const float *g_b, float *out,
                    int64_t count, void *stream);

/* ---- best-of-N scoring / select (gaussian_diffusion.py:626-633, 687-698;
 *      best_of_n_simple.py:32-40) */
/* costs[p] = ||y - A(x_p)||_2 without materialising the residual */
int dpsx_score_f32(dpsx_op *op, const float *x, const float *y, int64_t y_n, float *costs,
                   int64_t n, int64_t c, int64_t h, int64_t w,
                   void *workspace, int64_t workspace_bytes, void *stream);
/* The same scoring with the select fused in (gaussian_diffusion.py:626-631): costs[p] as above, *best_idx_dev =
 * argmin_p costs[p] (torch.argmin: first minimum, NaN counts as the minimum) and, if best_val_dev != NULL, its cost
 * -- the per-particle reduction and the select share one small follow-up launch. */
int dpsx_score_argmin_f32(dpsx_op *op, const float *x, const float *y, int64_t y_n, float *costs,
                          int64_t *best_idx_dev, float *best_val_dev,
                          int64_t n, int64_t c, int64_t h, int64_t w,
                          void *workspace, int64_t workspace_bytes, void *stream);

/* One step of SearchDDPM.p_sample_loop (gaussian_diffusion.py:618-633): S1 (p_sample: `sample` out; x0_hat is not
 * needed by this loop), costs[p] = ||y - A(sample_p)||_2, best = argmin (torch.argmin order) and, if x_next != NULL,
 * x_next[p] = sample[best] for every p (img[best_path.repeat(n_paths)]).  Four launches: S1, the scoring launch, one
 * small launch that finishes the costs and selects, the replication.  x_next == NULL: up to the select only (multi-GPU callers
 * exchange the champions first). */
int dpsx_search_step_f32(dpsx_op *op, const float *x_t, const float *model_out, const float *noise,
                         const float *y, int64_t y_n, float *sample, float *costs,
                         int64_t *best_idx_dev, float *best_val_dev, float *x_next,
                         int64_t n, int64_t c, int64_t h, int64_t w, const dpsx_coefs *coefs_host,
                         void *workspace, int64_t workspace_bytes, void *stream);

/* The same step with the loop's state held as ONE particle (gaussian_diffusion.py:618-633): after a select every particle
 * is a copy of the winner (img[best_path.repeat(n_paths)], :633), so x_t is [1, c, h, w] and model_out [1, 2c, h, w] -- one
 * model evaluation per step for the caller -- while noise, sample and costs stay per particle ([n, ...]).  x_next (nullable)
 * receives the winner ONCE, [1, c, h, w].  Results are those of dpsx_search_step_f32 on n copies of the state, bit for bit. */
int dpsx_search_step_one_f32(dpsx_op *op, const float *x_t, const float *model_out, const float *noise,
                             const float *y, int64_t y_n, float *sample, float *costs,
                             int64_t *best_idx_dev, float *best_val_dev, float *x_next,
                             int64_t n, int64_t c, int64_t h, int64_t w, const dpsx_coefs *coefs_host,
                             void *workspace, int64_t workspace_bytes, void *stream);

/* SearchDDPM.resample_update's cost update (gaussian_diffusion.py:556-585):
 *   curr[p] = ||y - A(x_p)||_1^2 / (c*h*w)                                         (:557-563)
 *   net[p]  = curr + prev (MEAN) | min(curr, prev) (MIN, NaN propagates as torch.min) | curr - prev (DIFF) | curr (CURR)
 * prev_costs == NULL (first call, :566-579) gives net = curr for every potential.  curr_costs may be NULL. */
enum { DPSX_POT_MEAN = 1, DPSX_POT_MIN = 2, DPSX_POT_DIFF = 3, DPSX_POT_CURR = 4 };
int dpsx_resample_cost_f32(dpsx_op *op, const float *x, const float *y, int64_t y_n, const float *prev_costs,
                           int potential, float *curr_costs, float *net_costs,
                           int64_t n, int64_t c, int64_t h, int64_t w,
                           void *workspace, int64_t workspace_bytes, void *stream);

/* torch.argmin semantics: first minimum wins, NaN counts as the minimum (gaussian_diffusion.py:631).
 * val_out_dev (optional, may be NULL) receives v[argmin] -- `costs[best_path]` of :632 without a host index. */
int dpsx_argmin_f32(const float *v, int64_t n, int64_t *idx_out_dev, float *val_out_dev, void *stream);
/* dst[p] = src[ids[p]]   (ids: device int64 [n_out]; an id outside [0, n_src) fills dst[p] with NaN) */
int dpsx_gather_f32(const float *src, const int64_t *ids_dev, float *dst,
                    int64_t n_out, int64_t n_src, int64_t chw, void *stream);
/* dst[p] = src[*idx_dev] for all p  (img[best.repeat(n)], gaussian_diffusion.py:633) */
int dpsx_replicate_f32(const float *src, const int64_t *idx_dev, float *dst,
                       int64_t n_out, int64_t n_src, int64_t chw, void *stream);

/* ---- the device half of the multi-GPU champion exchange (best-of-N across ranks: gaussian_diffusion.py:626-633 and
 * best_of_n_simple.py:32-40 over a sharded particle set).  The collective itself stays with the caller's communicator
 * (RCCL through torch.distributed); these two launches replace the seven small device ops around it -- argmin, copy,
 * concatenation, two strided copies, argmin, replication -- whose launch gaps and host calls cost more than the collective.
 *   pack:   out[0 .. chw) = particles[best], out[chw .. chw+4) = (cost, (float)best, 0, 0): the record one all-gather
 *           carries (the header sits BEHIND the image so that the image stays 16-byte aligned).  best_idx_dev == NULL:
 *           the torch.argmin-order select over costs runs in this launch; else costs may be NULL if best_val_dev is given.
 *   select: table is the gathered [world][chw + 4]; the winner is the torch.argmin-order minimum of table[r][chw]
 *           (lowest rank wins ties, NaN counts as the minimum); dst[p] = its image for p < n_out; win_rank_dev /
 *           win_local_dev (nullable) receive the winning rank and its local particle index. */
int dpsx_pack_champion_f32(const float *particles, const float *costs, const int64_t *best_idx_dev,
                           const float *best_val_dev, float *out, int64_t n, int64_t chw, void *stream);
int dpsx_select_champion_f32(const float *table, int64_t world, int64_t chw, float *dst, int64_t n_out,
                             int64_t *win_rank_dev, int64_t *win_local_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DPSX_H */
