/* ORACLE -- test infrastructure only, never linked or called by the product.
 *
 * Plain-C CPU restatement of the per-step DPS hot path of vishnutez/dps-ttc
 * (reference paths relative to /root/reference).  fp32 data; reductions and
 * tap sums accumulate in double and round once, so this is at least as
 * accurate as the reference's fp32 ATen kernels.  Element-wise formulas keep
 * the reference's operation order (compiled with -ffp-contract=off so a*x-b*e
 * is two roundings, exactly as two ATen ops).
 *
 * Pinned against the reference by tests/golden/make_golden.py (fixtures under
 * tests/golden/, checked in tests/test_oracle_golden.py).
 *
 * Build: oracle/Makefile -> oracle/_build/libdps_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------
 * S1  x0_hat / posterior mean / learned-range log-variance / DDPM sample
 *   posterior_mean_variance.py:120-123 (eps -> x0), :40-45 (clamp),
 *   :110-118 (mean), :230-242 (log-variance), gaussian_diffusion.py:466-476.
 *   model_out is [N, 2C, H, W]: channels [0,C) = eps, [C,2C) = v
 *   (gaussian_diffusion.py:314-315).  `chw` = C*H*W.
 *   add_noise: bit 0 = add the noise term (t != 0), bit 1 = DDIM step (:479-509) instead of DDPM.
 * ---------------------------------------------------------------------- */
API void orc_posterior_fwd(const float *x, const float *model_out, const float *noise,
                           float *x0_hat, float *mean, float *logvar, float *sample,
                           uint8_t *inside, int64_t n, int64_t chw,
                           float a, float b, float c1, float c2,
                           float min_log, float max_log, int add_noise)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p) {
        const float *xp = x + p * chw, *ep = model_out + p * 2 * chw, *vp = ep + chw;
        const float *zp = noise ? noise + p * chw : NULL;
        for (int64_t i = 0; i < chw; ++i) {
            float t1 = a * xp[i];
            float t2 = b * ep[i];
            float pre = t1 - t2;
            float x0 = pre < -1.0f ? -1.0f : (pre > 1.0f ? 1.0f : pre);
            float m1 = c1 * x0;
            float m2 = c2 * xp[i];
            float mu = m1 + m2;
            float frac = (vp[i] + 1.0f) / 2.0f;
            float l1 = frac * max_log;
            float l2 = (1.0f - frac) * min_log;
            float lv = l1 + l2;
            float s = mu;
            if (add_noise & 2) {
                /* DDIM (gaussian_diffusion.py:481-509): c1 = sqrt(abar_prev), c2 = sqrt(1 - abar_prev - sigma^2),
                 * min_log carries sigma; eps re-derived from x0_hat (:506-509) */
                float eps = (t1 - x0) / b;
                float d1 = x0 * c1;
                float d2 = c2 * eps;
                mu = d1 + d2;
                s = mu;
                if (add_noise & 1) s = mu + min_log * zp[i];
                lv = 0.0f;
            } else if (add_noise) {
                float sd = expf(0.5f * lv);
                s = mu + sd * zp[i];
            }
            int64_t o = p * chw + i;
            if (x0_hat) x0_hat[o] = x0;
            if (mean) mean[o] = mu;
            if (logvar) logvar[o] = lv;
            if (sample) sample[o] = s;
            if (inside) inside[o] = (uint8_t)(pre >= -1.0f && pre <= 1.0f);
        }
    }
}

/* VJP of the block above w.r.t. x and model_out, as torch.autograd derives it
 * (clamp passes the cotangent on the closed interval [-1, 1]).  g_x0 and
 * g_sample may each be NULL (= zero cotangent). */
API void orc_posterior_bwd(const float *g_x0, const float *g_sample,
                           const float *x, const float *model_out, const float *noise,
                           float *g_x, float *g_model_out, int64_t n, int64_t chw,
                           float a, float b, float c1, float c2,
                           float min_log, float max_log, int add_noise)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p) {
        const float *xp = x + p * chw, *ep = model_out + p * 2 * chw, *vp = ep + chw;
        float *ge = g_model_out + p * 2 * chw, *gv = ge + chw;
        for (int64_t i = 0; i < chw; ++i) {
            int64_t o = p * chw + i;
            float pre = a * xp[i] - b * ep[i];
            int in = (pre >= -1.0f && pre <= 1.0f);
            double gs = g_sample ? g_sample[o] : 0.0;
            /* DDIM: sample = c1 x0 + c2 (a x - x0) / b (+ sigma z) */
            double ds_dx0 = (add_noise & 2) ? (double)c1 - (double)c2 / (double)b : (double)c1;
            double ds_dx = (add_noise & 2) ? (double)c2 * (double)a / (double)b : (double)c2;
            double g0 = (g_x0 ? g_x0[o] : 0.0) + ds_dx0 * gs;
            double gp = in ? g0 : 0.0;
            g_x[o] = (float)((double)a * gp + ds_dx * gs);
            ge[i] = (float)(-(double)b * gp);
            double gvv = 0.0;
            if (add_noise == 1 && g_sample) {
                float frac = (vp[i] + 1.0f) / 2.0f;
                float lv = frac * max_log + (1.0f - frac) * min_log;
                gvv = gs * noise[o] * exp(0.5 * (double)lv) * 0.5 * 0.5 *
                      ((double)max_log - (double)min_log);
            }
            gv[i] = (float)gvv;
        }
    }
}

/* ------------------------------------------------------------------------
 * Blur: ReflectionPad2d(K/2) + depthwise cross-correlation with one K x K
 * kernel shared by all channels (util/img_utils.py:268-283, 301-305;
 * measurements.py:108-110, 142-143).  planes = N*C.
 * ---------------------------------------------------------------------- */
static inline int64_t refl(int64_t i, int64_t n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

API void orc_blur_fwd(const float *x, const float *k, float *y,
                      int64_t planes, int64_t h, int64_t w, int64_t ks, int skip_zero_taps)
{
    const int64_t r = ks / 2;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int64_t p = 0; p < planes; ++p)
        for (int64_t i = 0; i < h; ++i) {
            const float *xp = x + p * h * w;
            for (int64_t j = 0; j < w; ++j) {
                double acc = 0.0;
                for (int64_t u = 0; u < ks; ++u) {
                    const float *row = xp + refl(i + u - r, h) * w;
                    for (int64_t v = 0; v < ks; ++v) {
                        float kv = k[u * ks + v];
                        if (skip_zero_taps && kv == 0.0f) continue;
                        acc += (double)kv * (double)row[refl(j + v - r, w)];
                    }
                }
                y[(p * h + i) * w + j] = (float)acc;
            }
        }
}

/* Exact adjoint, literally: scatter each cotangent through the taps into the
 * padded plane (convolution_backward w.r.t. input), then fold the reflected
 * border back (reflection_pad2d_backward). */
API void orc_blur_adj(const float *u_in, const float *k, float *g,
                      int64_t planes, int64_t h, int64_t w, int64_t ks)
{
    const int64_t r = ks / 2, ph = h + 2 * r, pw = w + 2 * r;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t p = 0; p < planes; ++p) {
        double *pad = (double *)calloc((size_t)(ph * pw), sizeof(double));
        double *acc = (double *)calloc((size_t)(h * w), sizeof(double));
        const float *up = u_in + p * h * w;
        for (int64_t uu = 0; uu < ks; ++uu)
            for (int64_t vv = 0; vv < ks; ++vv) {
                double kv = k[uu * ks + vv];
                if (kv == 0.0) continue;
                for (int64_t i = 0; i < h; ++i) {
                    double *dst = pad + (i + uu) * pw + vv;
                    const float *src = up + i * w;
                    for (int64_t j = 0; j < w; ++j) dst[j] += kv * (double)src[j];
                }
            }
        for (int64_t a = 0; a < ph; ++a)
            for (int64_t b = 0; b < pw; ++b)
                acc[refl(a - r, h) * w + refl(b - r, w)] += pad[a * pw + b];
        for (int64_t i = 0; i < h * w; ++i) g[p * h * w + i] = (float)acc[i];
        free(pad);
        free(acc);
    }
}

/* ------------------------------------------------------------------------
 * Resizer, one axis (util/resizer.py:59-72): out[o] = sum_k wt[k,o]*x[idx[k,o]]
 * along `axis` (0 = H, 1 = W) of each [h, w] plane.  wt/idx are [K, n_out].
 * ---------------------------------------------------------------------- */
API void orc_resize_axis_fwd(const float *x, float *y, int64_t planes, int64_t h, int64_t w,
                             int axis, int64_t taps, int64_t n_out,
                             const float *wt, const int64_t *idx)
{
    const int64_t oh = axis == 0 ? n_out : h, ow = axis == 1 ? n_out : w;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < planes; ++p) {
        const float *xp = x + p * h * w;
        float *yp = y + p * oh * ow;
        for (int64_t i = 0; i < oh; ++i)
            for (int64_t j = 0; j < ow; ++j) {
                int64_t o = axis == 0 ? i : j;
                double acc = 0.0;
                for (int64_t t = 0; t < taps; ++t) {
                    int64_t s = idx[t * n_out + o];
                    float xv = axis == 0 ? xp[s * w + j] : xp[i * w + s];
                    acc += (double)wt[t * n_out + o] * (double)xv;
                }
                yp[i * ow + j] = (float)acc;
            }
    }
}

/* adjoint of the above = index_put_(accumulate=True) of wt*u (autograd of :69) */
API void orc_resize_axis_adj(const float *u_in, float *g, int64_t planes, int64_t h, int64_t w,
                             int axis, int64_t taps, int64_t n_out,
                             const float *wt, const int64_t *idx)
{
    const int64_t oh = axis == 0 ? n_out : h, ow = axis == 1 ? n_out : w;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < planes; ++p) {
        double *acc = (double *)calloc((size_t)(h * w), sizeof(double));
        const float *up = u_in + p * oh * ow;
        for (int64_t i = 0; i < oh; ++i)
            for (int64_t j = 0; j < ow; ++j) {
                int64_t o = axis == 0 ? i : j;
                double uv = up[i * ow + j];
                for (int64_t t = 0; t < taps; ++t) {
                    int64_t s = idx[t * n_out + o];
                    double c = (double)wt[t * n_out + o] * uv;
                    if (axis == 0) acc[s * w + j] += c; else acc[i * w + s] += c;
                }
            }
        for (int64_t i = 0; i < h * w; ++i) g[p * h * w + i] = (float)acc[i];
        free(acc);
    }
}

/* ------------------------------------------------------------------------
 * Inpainting: x * mask, mask [1,1,H,W] broadcast (measurements.py:158-162).
 * Exact in fp32 (one multiply); its adjoint is the same multiply.
 * ---------------------------------------------------------------------- */
API void orc_mask_mul(const float *x, const float *mask, float *y, int64_t planes, int64_t hw)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < planes; ++p)
        for (int64_t i = 0; i < hw; ++i) y[p * hw + i] = x[p * hw + i] * mask[i];
}

/* ------------------------------------------------------------------------
 * Residual and per-particle L2 norm (condition_methods.py:37-39, 179-181;
 * gaussian_diffusion.py:627-630).  y has y_n particles (1 broadcasts).
 * ---------------------------------------------------------------------- */
API void orc_residual_norm(const float *y, int64_t y_n, const float *ax, float *r, float *norm,
                           int64_t n, int64_t m)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p) {
        const float *yp = y + (y_n == 1 ? 0 : p) * m;
        double ss = 0.0;
        for (int64_t i = 0; i < m; ++i) {
            float d = yp[i] - ax[p * m + i];
            if (r) r[p * m + i] = d;
            ss += (double)d * (double)d;
        }
        norm[p] = (float)sqrt(ss);
    }
}

/* cotangent of A(x0_hat) for loss_p = g_norm[p] * norm_p^power, power 1 or 2
 * (condition_methods.py:41-48, 184-185; torch gives 0 where norm == 0) */
API void orc_norm_bwd(const float *r, const float *norm, const float *g_norm, int power,
                      float *g_ax, int64_t n, int64_t m)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p) {
        double nv = norm[p], gn = g_norm[p];
        double coef = power == 2 ? -2.0 * gn : (nv == 0.0 ? 0.0 : -gn / nv);
        for (int64_t i = 0; i < m; ++i) g_ax[p * m + i] = (float)(coef * (double)r[p * m + i]);
    }
}

/* x_{t-1} = sample - grad  (gaussian_diffusion.py:255) */
API void orc_update(const float *sample, const float *grad, float *out, int64_t count)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < count; ++i) out[i] = sample[i] - grad[i];
}

/* torch.argmin over a 1-D fp32 vector: first minimum wins; a NaN compares as
 * the minimum (gaussian_diffusion.py:631, best_of_n_simple.py:34). */
API int64_t orc_argmin(const float *v, int64_t n)
{
    int64_t best = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (isnan(v[i])) return i;
        if (v[i] < v[best]) best = i;
    }
    return best;
}

/* img[ids]  (gaussian_diffusion.py:633, 697) */
API void orc_gather(const float *src, const int64_t *ids, float *dst, int64_t n_out, int64_t chw)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n_out; ++p)
        memcpy(dst + p * chw, src + ids[p] * chw, (size_t)chw * sizeof(float));
}

/* ------------------------------------------------------------------------
 * Phase retrieval (measurements.py:186-189 -> img_utils.py:26-30 ->
 * fastmri_utils.py:67-89): zero-pad by `pad`, ifftshift, orthonormal 2-D DFT,
 * fftshift, modulus.  Direct O(n^3) DFT in double; s = h + 2*pad (even).
 * Also returns the complex spectrum (re, im) for the adjoint.
 * ---------------------------------------------------------------------- */
static void dft_tables(int64_t s, double **co, double **si)
{
    *co = (double *)malloc((size_t)s * sizeof(double));
    *si = (double *)malloc((size_t)s * sizeof(double));
    for (int64_t i = 0; i < s; ++i) {
        (*co)[i] = cos(2.0 * M_PI * (double)i / (double)s);
        (*si)[i] = sin(2.0 * M_PI * (double)i / (double)s);
    }
}

API void orc_phase_fwd(const float *x, float *amp, float *spec_re, float *spec_im,
                       int64_t planes, int64_t h, int64_t pad)
{
    const int64_t s = h + 2 * pad, half = s / 2;
    double *co, *si;
    dft_tables(s, &co, &si);
    const double scale = 1.0 / (double)s; /* ortho: 1/sqrt(s*s) */
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t p = 0; p < planes; ++p) {
        /* rows: T[k1][q] = sum_{q'} in[k1][q'] e^{-2 pi i q q'/s}, only image rows/cols non-zero */
        double *tr = (double *)calloc((size_t)(h * s), sizeof(double));
        double *ti = (double *)calloc((size_t)(h * s), sizeof(double));
        const float *xp = x + p * h * h;
        for (int64_t a = 0; a < h; ++a)          /* image row a sits at padded row a+pad */
            for (int64_t q = 0; q < s; ++q) {
                double sr = 0.0, sim = 0.0;
                for (int64_t bcol = 0; bcol < h; ++bcol) {
                    /* ifftshift (even s): element at padded col c moves to (c - half) mod s */
                    int64_t c = ((bcol + pad) - half + s) % s;
                    int64_t ph = (q * c) % s;
                    double v = xp[a * h + bcol];
                    sr += v * co[ph];
                    sim -= v * si[ph];
                }
                tr[a * s + q] = sr;
                ti[a * s + q] = sim;
            }
        for (int64_t k = 0; k < s; ++k)
            for (int64_t q = 0; q < s; ++q) {
                double sr = 0.0, sim = 0.0;
                for (int64_t a = 0; a < h; ++a) {
                    int64_t rr = ((a + pad) - half + s) % s;
                    int64_t ph = (k * rr) % s;
                    double cr = co[ph], ci = -si[ph];
                    sr += tr[a * s + q] * cr - ti[a * s + q] * ci;
                    sim += tr[a * s + q] * ci + ti[a * s + q] * cr;
                }
                /* fftshift: spectrum index k lands at (k + half) mod s */
                int64_t ok = (k + half) % s, oq = (q + half) % s;
                int64_t o = (p * s + ok) * s + oq;
                double re = sr * scale, im = sim * scale;
                if (spec_re) spec_re[o] = (float)re;
                if (spec_im) spec_im[o] = (float)im;
                amp[o] = (float)sqrt(re * re + im * im);
            }
        free(tr);
        free(ti);
    }
    free(co);
    free(si);
}

/* VJP of orc_phase_fwd: cotangent u on the amplitude -> cotangent on x.
 * d|z| = Re(conj(z)/|z| dz) (0 where |z| = 0, as torch's abs backward), then
 * the adjoint of the shifted orthonormal DFT (= the shifted inverse DFT) and
 * the crop that undoes the zero padding; real part only since x is real. */
API void orc_phase_adj(const float *u_in, const float *spec_re, const float *spec_im,
                       float *g, int64_t planes, int64_t h, int64_t pad)
{
    const int64_t s = h + 2 * pad, half = s / 2;
    double *co, *si;
    dft_tables(s, &co, &si);
    const double scale = 1.0 / (double)s;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t p = 0; p < planes; ++p) {
        double *wr = (double *)malloc((size_t)(s * s) * sizeof(double));
        double *wi = (double *)malloc((size_t)(s * s) * sizeof(double));
        for (int64_t ok = 0; ok < s; ++ok)
            for (int64_t oq = 0; oq < s; ++oq) {
                int64_t o = (p * s + ok) * s + oq;
                double re = spec_re[o], im = spec_im[o];
                double mag = sqrt(re * re + im * im);
                double c = mag == 0.0 ? 0.0 : (double)u_in[o] / mag;
                /* undo fftshift: position ok holds spectrum index (ok - half) mod s */
                int64_t k = (ok - half + s) % s, q = (oq - half + s) % s;
                wr[k * s + q] = c * re;
                wi[k * s + q] = c * im;
            }
        /* g[a][b] = Re sum_{k,q} w[k][q] e^{+2 pi i (k r + q c)/s} / s */
        double *tr = (double *)malloc((size_t)(h * s) * sizeof(double));
        double *ti = (double *)malloc((size_t)(h * s) * sizeof(double));
        for (int64_t a = 0; a < h; ++a) {
            int64_t rr = ((a + pad) - half + s) % s;
            for (int64_t q = 0; q < s; ++q) {
                double sr = 0.0, sim = 0.0;
                for (int64_t k = 0; k < s; ++k) {
                    int64_t ph = (k * rr) % s;
                    sr += wr[k * s + q] * co[ph] - wi[k * s + q] * si[ph];
                    sim += wr[k * s + q] * si[ph] + wi[k * s + q] * co[ph];
                }
                tr[a * s + q] = sr;
                ti[a * s + q] = sim;
            }
        }
        for (int64_t a = 0; a < h; ++a)
            for (int64_t bcol = 0; bcol < h; ++bcol) {
                int64_t c = ((bcol + pad) - half + s) % s;
                double sr = 0.0;
                for (int64_t q = 0; q < s; ++q) {
                    int64_t ph = (q * c) % s;
                    sr += tr[a * s + q] * co[ph] - ti[a * s + q] * si[ph];
                }
                g[(p * h + a) * h + bcol] = (float)(sr * scale);
            }
        free(wr); free(wi); free(tr); free(ti);
    }
    free(co);
    free(si);
}
