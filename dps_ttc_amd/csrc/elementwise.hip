// Element-wise / reduction kernels of the DPS step (HBM-bound; 16 B per lane).
//
// Layout: particles are the outer dimension; one particle = chw contiguous
// fp32; model_out holds 2*chw per particle (eps | v).  Fast paths need
// chw % 4 == 0 and 16-byte aligned bases (torch allocations are), otherwise a
// scalar kernel with identical arithmetic runs.
#include "common.h"

namespace dpsx {

constexpr int kThreads = 256;

static inline dim3 grid_for(int64_t work_per_particle, int64_t n)
{
    int64_t bx = (work_per_particle + kThreads - 1) / kThreads;
    if (bx < 1) bx = 1;
    return dim3((unsigned)bx, (unsigned)n, 1);
}

// ===================================================================== S1 forward
template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_posterior_fwd(const float *__restrict__ x,
                                                            const float *__restrict__ mo,
                                                            const float *__restrict__ z,
                                                            float *__restrict__ x0o, float *__restrict__ so,
                                                            uint8_t *__restrict__ ins, int64_t chw, Coefs k,
                                                            int64_t xs)
{
    // xs: particle stride of x (and half that of model_out): chw, or 0 when ONE state feeds all particles (search_ddpm)
    const int64_t p = blockIdx.y;
    const int64_t i = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * (VEC ? 4 : 1);
    if (i >= chw) return;
    const float *xp = x + p * xs + i, *ep = mo + p * 2 * xs + i, *vp = ep + chw;
    const int64_t o = p * chw + i;
    if constexpr (VEC) {
        const float4 xv = *reinterpret_cast<const float4 *>(xp);
        const float4 ev = *reinterpret_cast<const float4 *>(ep);
        float4 vv = make_float4(0, 0, 0, 0), zv = vv;
        if (k.add_noise & 1) {
            vv = *reinterpret_cast<const float4 *>(vp);
            zv = *reinterpret_cast<const float4 *>(z + o);
        }
        bool b0, b1, b2, b3;
        float4 x0, sm;
        x0.x = post_x0(xv.x, ev.x, k, b0);
        x0.y = post_x0(xv.y, ev.y, k, b1);
        x0.z = post_x0(xv.z, ev.z, k, b2);
        x0.w = post_x0(xv.w, ev.w, k, b3);
        sm.x = post_sample(xv.x, x0.x, vv.x, zv.x, k);
        sm.y = post_sample(xv.y, x0.y, vv.y, zv.y, k);
        sm.z = post_sample(xv.z, x0.z, vv.z, zv.z, k);
        sm.w = post_sample(xv.w, x0.w, vv.w, zv.w, k);
        if (x0o) *reinterpret_cast<float4 *>(x0o + o) = x0;
        if (so) *reinterpret_cast<float4 *>(so + o) = sm;
        if (ins) *reinterpret_cast<uchar4 *>(ins + o) = make_uchar4(b0, b1, b2, b3);
    } else {
        bool b;
        float x0 = post_x0(*xp, *ep, k, b);
        float sm = post_sample(*xp, x0, (k.add_noise & 1) ? *vp : 0.f, (k.add_noise & 1) ? z[o] : 0.f, k);
        if (x0o) x0o[o] = x0;
        if (so) so[o] = sm;
        if (ins) ins[o] = b;
    }
}

int posterior_fwd(const float *x, const float *mo, const float *z, float *x0, float *sample, uint8_t *inside,
                  int64_t n, int64_t chw, const Coefs &k, hipStream_t s, bool one_state)
{
    if (n == 0 || chw == 0) return DPSX_OK;
    const int64_t xs = one_state ? 0 : chw;
    const bool vec = chw % 4 == 0 && aligned16(x) && aligned16(mo) && aligned16(z) && aligned16(x0) &&
                     aligned16(sample) && (reinterpret_cast<uintptr_t>(inside) & 3u) == 0;
    if (vec)
        k_posterior_fwd<true><<<grid_for(chw / 4, n), kThreads, 0, s>>>(x, mo, z, x0, sample, inside, chw, k, xs);
    else
        k_posterior_fwd<false><<<grid_for(chw, n), kThreads, 0, s>>>(x, mo, z, x0, sample, inside, chw, k, xs);
    return check_launch();
}

// ===================================================================== S1 backward
// g_pre = [pre in [-1,1]] * (g_x0 + c1 g_s);  g_x = a g_pre + c2 g_s;
// g_eps = -b g_pre;  g_v = g_s * z * sd * (max_log - min_log) / 4
template <int U>      // U = 4: one float4 unit per lane (chw % 4 == 0, 16-byte aligned planes); U = 1: scalar
__global__ __launch_bounds__(kThreads) void k_posterior_bwd(const float *__restrict__ g_x0,
                                                            const float *__restrict__ g_s,
                                                            const float *__restrict__ x,
                                                            const float *__restrict__ mo,
                                                            const float *__restrict__ z,
                                                            float *__restrict__ g_x, float *__restrict__ g_mo,
                                                            int64_t chw, Coefs k)
{
    const int64_t p = blockIdx.y;
    const int64_t i = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * U;
    if (i >= chw) return;
    const int64_t o = p * chw + i, e = p * 2 * chw + i;
    float xv[U], ev[U], vv[U], zv[U], g0v[U], gsv[U], ox[U], oe[U], ov[U];
    const bool noisy = k.add_noise == 1 && g_s;
    if constexpr (U == 4) {
        auto ld4 = [](const float *q, float (&d)[U]) {
            const float4 t = *reinterpret_cast<const float4 *>(q);
            d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
        };
        ld4(x + o, xv);
        ld4(mo + e, ev);
        if (g_x0) ld4(g_x0 + o, g0v);
        if (g_s) ld4(g_s + o, gsv);
        if (noisy) { ld4(mo + e + chw, vv); ld4(z + o, zv); }
    } else {
        xv[0] = x[o]; ev[0] = mo[e];
        if (g_x0) g0v[0] = g_x0[o];
        if (g_s) gsv[0] = g_s[o];
        if (noisy) { vv[0] = mo[e + chw]; zv[0] = z[o]; }
    }
    // d sample / d x0_hat and d sample / d x (direct): DDPM c1, c2;  DDIM c1 - c2 / b, c2 a / b
    const bool ddim = k.add_noise & 2;
    const float ds_dx0 = ddim ? k.c1 - k.c2 / k.b : k.c1, ds_dx = ddim ? k.c2 * k.a / k.b : k.c2;
#pragma unroll
    for (int q = 0; q < U; ++q) {
        bool in;
        (void)post_x0(xv[q], ev[q], k, in);
        const float gs = g_s ? gsv[q] : 0.0f;
        const float g0 = (g_x0 ? g0v[q] : 0.0f) + ds_dx0 * gs;
        const float gp = in ? g0 : 0.0f;
        ox[q] = k.a * gp + ds_dx * gs;
        oe[q] = -k.b * gp;
        ov[q] = 0.0f;
        if (noisy) {
            const float sd = expf(0.5f * post_logvar(vv[q], k));
            ov[q] = gs * zv[q] * sd * (0.25f * (k.max_log - k.min_log));
        }
    }
    if constexpr (U == 4) {
        *reinterpret_cast<float4 *>(g_x + o) = make_float4(ox[0], ox[1], ox[2], ox[3]);
        *reinterpret_cast<float4 *>(g_mo + e) = make_float4(oe[0], oe[1], oe[2], oe[3]);
        *reinterpret_cast<float4 *>(g_mo + e + chw) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    } else {
        g_x[o] = ox[0];
        g_mo[e] = oe[0];
        g_mo[e + chw] = ov[0];
    }
}

int posterior_bwd(const float *g_x0, const float *g_s, const float *x, const float *mo, const float *z,
                  float *g_x, float *g_mo, int64_t n, int64_t chw, const Coefs &k, hipStream_t s)
{
    if (n == 0 || chw == 0) return DPSX_OK;
    const bool vec = chw % 4 == 0 && aligned16(g_x0) && aligned16(g_s) && aligned16(x) && aligned16(mo) && aligned16(z) &&
                     aligned16(g_x) && aligned16(g_mo);
    if (vec) k_posterior_bwd<4><<<grid_for(chw / 4, n), kThreads, 0, s>>>(g_x0, g_s, x, mo, z, g_x, g_mo, chw, k);
    else k_posterior_bwd<1><<<grid_for(chw, n), kThreads, 0, s>>>(g_x0, g_s, x, mo, z, g_x, g_mo, chw, k);
    return check_launch();
}

// ===================================================================== inpainting mask
__global__ __launch_bounds__(kThreads) void k_mask_mul(const float *__restrict__ x, const float *__restrict__ m,
                                                       float *__restrict__ y, int64_t hw)
{
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= hw) return;
    const int64_t o = (int64_t)blockIdx.y * hw + i;
    y[o] = __fmul_rn(x[o], m[i]);
}

int mask_mul(const float *x, const float *mask, float *y, int64_t planes, int64_t hw, hipStream_t s)
{
    if (planes == 0 || hw == 0) return DPSX_OK;
    k_mask_mul<<<grid_for(hw, planes), kThreads, 0, s>>>(x, mask, y, hw);
    return check_launch();
}

// ===================================================================== residual + norm
// grid (parts, n): block (q, p) reduces elements [q*chunk, (q+1)*chunk) of particle p.
// mask (nullable, hw elements, broadcast over particles and channels): the inpainting operator applied on the fly,
// d = y - mask * ax with the product rounded once, exactly as mask_mul + this kernel did in two launches (the search
// step's scoring of a masked proposal: 44 -> one launch without the 2P round trip through scratch).
__global__ __launch_bounds__(kThreads) void k_residual_partials(const float *__restrict__ y, int64_t y_n,
                                                                const float *__restrict__ ax,
                                                                float *__restrict__ r,
                                                                float *__restrict__ partials, int64_t m,
                                                                int64_t chunk, int l1, Tail tail,
                                                                const float *__restrict__ mask, int hw)
{
    __shared__ float scratch[kThreads / kWave];
    const int64_t p = blockIdx.y, q = blockIdx.x;
    const float *yp = y + (y_n == 1 ? 0 : p) * m, *ap = ax + p * m;
    const int64_t lo = q * chunk, hi = min(m, lo + chunk);
    float acc = 0.0f;
    if (mask) {
        int mi = (int)((lo + threadIdx.x) % hw);                 // walks the mask plane with the element index
        const int step = kThreads % hw;
        for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) {
            const float d = __fsub_rn(yp[i], __fmul_rn(ap[i], mask[mi]));
            if (r) r[p * m + i] = d;
            acc = l1 ? acc + fabsf(d) : fmaf(d, d, acc);
            mi += step;
            if (mi >= hw) mi -= hw;
        }
    } else {
        for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) {
            const float d = __fsub_rn(yp[i], ap[i]);
            if (r) r[p * m + i] = d;
            acc = l1 ? acc + fabsf(d) : fmaf(d, d, acc);
        }
    }
    const float t = block_sum(acc, scratch);
    if (threadIdx.x == 0) tail_publish(&partials[p * gridDim.x + q], t, tail.counters != nullptr);
    tail_arrive(tail, (int)p);
}

int residual_partials(const float *y, int64_t y_n, const float *ax, float *r, float *partials, int64_t n,
                      int64_t m, int parts, hipStream_t s, int l1, const Tail &tail, const float *mask, int64_t hw)
{
    if (n == 0) return DPSX_OK;
    if (mask && (hw < 1 || hw > (1 << 30))) return DPSX_EINVAL;
    const int64_t chunk = (m + parts - 1) / parts;
    Tail t = tail;
    t.blocks_per_particle = parts;
    k_residual_partials<<<dim3(parts, (unsigned)n), kThreads, 0, s>>>(y, y_n, ax, r, partials, m, chunk, l1, t, mask,
                                                                     (int)hw);
    return check_launch();
}

// one wave per particle; partial sums added in index order within a lane, then a fixed tree
__global__ __launch_bounds__(kWave) void k_finalize_norm(const float *__restrict__ partials, int parts,
                                                         float *__restrict__ norm)
{
    const int64_t p = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < parts; i += kWave) acc += (double)partials[p * parts + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, kWave);
    if (threadIdx.x == 0) norm[p] = (float)sqrt(acc);
}

int finalize_norm(const float *partials, int parts, float *norm, int64_t n, hipStream_t s)
{
    if (n == 0) return DPSX_OK;
    k_finalize_norm<<<(unsigned)n, kWave, 0, s>>>(partials, parts, norm);
    return check_launch();
}

// Finalisation + select in ONE small launch (replaces k_finalize_norm + k_argmin after a scoring launch, and does the
// cost combine of SearchDDPM.resample_update): wave w finishes particles w, w + nw, ... in the order of k_finalize_norm
// (bit-identical values), then the block runs the torch.argmin-order select over them.
// Measured alternative (r02): finishing inside the scoring launch ("last block done", common.h: Tail) costs every short
// scoring block two dependent memory round trips while it holds its LDS -- 42 us instead of ~25 us at N = 64.
constexpr int kSelThreads = 1024;
// the body of the finalisation + select; `writer`: this block stores the costs / the select's outputs (with several blocks
// -- k_finalize_select_copy -- every block computes the same values in the same order and ONE of them stores).
// -> the winner's index (block-uniform, valid in every thread), -1 when no select was asked for
__device__ __forceinline__ int64_t finalize_select_body(const Tail &t, const bool writer)
{
    __shared__ float s_v[kSelThreads / kWave];
    __shared__ int64_t s_i[kSelThreads / kWave];
    __shared__ int64_t s_best;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nw = kSelThreads / kWave;
    ArgMin best{0.0f, -1};
    constexpr int B = 4;          // particles per wave in flight: their partial loads are issued together (one latency)
    for (int p0 = wave; p0 < t.n; p0 += nw * B) {
        double acc[B];
#pragma unroll
        for (int b = 0; b < B; ++b) acc[b] = 0.0;
        for (int i = lane; i < t.parts; i += kWave) {
            float v[B];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const int p = min(p0 + b * nw, t.n - 1);             // surplus slots re-read the last particle
                v[b] = t.partials[(int64_t)p * t.parts + i];
            }
#pragma unroll
            for (int b = 0; b < B; ++b) acc[b] += (double)v[b];
        }
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int p = p0 + b * nw;
            if (p >= t.n) break;                                      // wave-uniform
            double a = acc[b];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, kWave);
            float v = t.mode == TAIL_L1SQ ? (float)(a * a * (double)t.l1_scale) : (float)sqrt(a);
            if (lane == 0) {
                if (t.raw_out && writer) t.raw_out[p] = v;
                if (t.prev) {
                    const float q = t.prev[p];
                    if (t.potential == POT_MEAN) v = v + q;
                    else if (t.potential == POT_MIN) v = (v != v || q != q) ? __builtin_nanf("") : fminf(v, q);
                    else if (t.potential == POT_DIFF) v = v - q;
                }
                if (writer) t.out[p] = v;
                const ArgMin c{v, p};
                if (argmin_better(c, best)) best = c;
            }
        }
    }
    if (!t.best_idx) return -1;
    if (lane == 0) { s_v[wave] = best.v; s_i[wave] = best.i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) {
            const ArgMin c{s_v[w], s_i[w]};
            if (argmin_better(c, best)) best = c;
        }
        s_best = best.i < 0 ? 0 : best.i;
        if (writer) {
            *t.best_idx = s_best;
            if (t.best_val) *t.best_val = best.v;
        }
    }
    __syncthreads();
    return s_best;
}

__global__ __launch_bounds__(kSelThreads) void k_finalize_select(Tail t)
{
    (void)finalize_select_body(t, true);
}

// the same + ONE copy of the winner (the single-state search step: dpsx_search_step_one_f32): every block of the copy
// finishes the costs and the select for itself (n * parts floats from the L2 -- 12 KB at N = 64) and copies its slice;
// one launch and one launch boundary less than finalisation + dpsx_replicate_f32(n_out = 1)
__global__ __launch_bounds__(kSelThreads) void k_finalize_select_copy(Tail t, const float *__restrict__ src,
                                                                      float *__restrict__ dst, int64_t chw4)
{
    const int64_t b = finalize_select_body(t, blockIdx.x == 0);
    const int64_t i = (int64_t)blockIdx.x * kSelThreads + threadIdx.x;
    if (i < chw4) reinterpret_cast<float4 *>(dst)[i] = (reinterpret_cast<const float4 *>(src) + b * chw4)[i];
}

int finalize_select(const Tail &t, hipStream_t s)
{
    if (t.n == 0) return DPSX_OK;
    k_finalize_select<<<1, kSelThreads, 0, s>>>(t);
    return check_launch();
}

int finalize_select_copy(const Tail &t, const float *src, float *dst, int64_t chw, hipStream_t s)
{
    if (t.n == 0) return DPSX_OK;
    const int64_t chw4 = chw / 4;
    k_finalize_select_copy<<<(unsigned)((chw4 + kSelThreads - 1) / kSelThreads), kSelThreads, 0, s>>>(t, src, dst, chw4);
    return check_launch();
}

__device__ __forceinline__ float norm_coef(float nv, float gn, int power)
{
    // d(gn * norm^power)/d(ax) = -coef * r ;  torch yields 0 where norm == 0
    return power == 2 ? -2.0f * gn : (nv == 0.0f ? 0.0f : -gn / nv);
}

__global__ __launch_bounds__(kThreads) void k_norm_bwd(const float *__restrict__ r, const float *__restrict__ norm,
                                                       const float *__restrict__ g_norm, int power,
                                                       float *__restrict__ g_ax, int64_t m)
{
    const int64_t p = blockIdx.y, i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= m) return;
    g_ax[p * m + i] = norm_coef(norm[p], g_norm[p], power) * r[p * m + i];
}

int norm_bwd(const float *r, const float *norm, const float *g_norm, int power, float *g_ax, int64_t n,
             int64_t m, hipStream_t s)
{
    if (n == 0 || m == 0) return DPSX_OK;
    k_norm_bwd<<<grid_for(m, n), kThreads, 0, s>>>(r, norm, g_norm, power, g_ax, m);
    return check_launch();
}

// ===================================================================== unfused tail of step_bwd
__global__ __launch_bounds__(kThreads) void k_clamp_scale(const float *__restrict__ g_x0,
                                                          const float *__restrict__ norm,
                                                          const uint8_t *__restrict__ ins, float scale, int power,
                                                          float *__restrict__ g_mo, int64_t chw, Coefs k,
                                                          const float *__restrict__ g_extra)
{
    const int64_t p = blockIdx.y, i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= chw) return;
    const float coef = norm_coef(norm[p], scale, power);  // g_x0 holds A^T r; cotangent is coef * A^T r
    float g0 = coef * g_x0[p * chw + i];
    if (g_extra) g0 += g_extra[p * chw + i];
    const float gp = ins[p * chw + i] ? g0 : 0.0f;
    g_mo[p * 2 * chw + i] = -k.b * gp;
}

int clamp_scale_to_eps(const float *g_x0, const float *norm, const uint8_t *inside, float scale, int power,
                       float *g_model_out, int64_t n, int64_t chw, const Coefs &k, hipStream_t s,
                       const float *g_extra)
{
    if (n == 0 || chw == 0) return DPSX_OK;
    k_clamp_scale<<<grid_for(chw, n), kThreads, 0, s>>>(g_x0, norm, inside, scale, power, g_model_out, chw, k,
                                                        g_extra);
    return check_launch();
}

// ===================================================================== S4 update
// x_{t-1} = sample - (a*g_pre + g_unet),  a*g_pre = (-a/b) * g_eps
template <bool VEC>
__global__ __launch_bounds__(kThreads) void k_step_update(const float *__restrict__ sm,
                                                          const float *__restrict__ g_mo,
                                                          const float *__restrict__ gu, float *__restrict__ out,
                                                          int64_t chw, float ratio)
{
    const int64_t p = blockIdx.y;
    const int64_t i = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * (VEC ? 4 : 1);
    if (i >= chw) return;
    const int64_t o = p * chw + i, e = p * 2 * chw + i;
    if constexpr (VEC) {
        const float4 s4 = *reinterpret_cast<const float4 *>(sm + o);
        const float4 g4 = *reinterpret_cast<const float4 *>(g_mo + e);
        float4 u4 = make_float4(0, 0, 0, 0);
        if (gu) u4 = *reinterpret_cast<const float4 *>(gu + o);
        float4 r4;
        r4.x = s4.x - (ratio * g4.x + u4.x);
        r4.y = s4.y - (ratio * g4.y + u4.y);
        r4.z = s4.z - (ratio * g4.z + u4.z);
        r4.w = s4.w - (ratio * g4.w + u4.w);
        *reinterpret_cast<float4 *>(out + o) = r4;
    } else {
        out[o] = sm[o] - (ratio * g_mo[e] + (gu ? gu[o] : 0.0f));
    }
}

int step_update(const float *sample, const float *g_mo, const float *g_unet, float *x_next, int64_t n,
                int64_t chw, const Coefs &k, hipStream_t s)
{
    if (n == 0 || chw == 0) return DPSX_OK;
    const float ratio = -k.a / k.b;
    const bool vec = chw % 4 == 0 && aligned16(sample) && aligned16(g_mo) && aligned16(g_unet) && aligned16(x_next);
    if (vec)
        k_step_update<true><<<grid_for(chw / 4, n), kThreads, 0, s>>>(sample, g_mo, g_unet, x_next, chw, ratio);
    else
        k_step_update<false><<<grid_for(chw, n), kThreads, 0, s>>>(sample, g_mo, g_unet, x_next, chw, ratio);
    return check_launch();
}

__global__ __launch_bounds__(kThreads) void k_plain_update(const float *__restrict__ sm,
                                                           const float *__restrict__ ga,
                                                           const float *__restrict__ gb, float *__restrict__ out,
                                                           int64_t count)
{
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= count) return;
    const float g = gb ? __fadd_rn(ga[i], gb[i]) : ga[i];
    out[i] = __fsub_rn(sm[i], g);
}

int plain_update(const float *sample, const float *ga, const float *gb, float *out, int64_t count, hipStream_t s)
{
    if (count == 0) return DPSX_OK;
    k_plain_update<<<(unsigned)((count + kThreads - 1) / kThreads), kThreads, 0, s>>>(sample, ga, gb, out, count);
    return check_launch();
}

// index of element i of a [C, H, W] particle in the [H, W] mask.  A 64-bit remainder by a run-time divisor is a software
// routine of a hundred-odd instructions -- more than the rest of these kernels; particles below 2^32 elements (all of them)
// take the 32-bit form.
__device__ __forceinline__ int64_t mask_index(int64_t i, int64_t chw, int64_t hw)
{
    if (chw <= 0xffffffffll) return (int64_t)((unsigned)i % (unsigned)hw);      // launch-uniform
    return i % hw;
}

// ===================================================================== inpainting fused step
// fwd: S1 + r = y - mask*x0 (not stored) + norm partials.   bwd: recompute r from x0_hat.
__global__ __launch_bounds__(kThreads) void k_mask_step_fwd(StepFwdArgs a, const float *__restrict__ mask,
                                                            int64_t chw, int64_t hw)
{
    __shared__ float scratch[kThreads / kWave];
    const int64_t p = blockIdx.y;
    float acc = 0.0f;
    // each block covers 4*kThreads consecutive elements of one particle
    const int64_t i = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 4;
    if (i < chw) {
        const int64_t o = p * chw + i, e = p * 2 * chw + i;
        const float4 xv = *reinterpret_cast<const float4 *>(a.x_t + o);
        const float4 ev = *reinterpret_cast<const float4 *>(a.model_out + e);
        float4 vv = make_float4(0, 0, 0, 0), zv = vv;
        if (a.k.add_noise & 1) {
            vv = *reinterpret_cast<const float4 *>(a.model_out + e + chw);
            zv = *reinterpret_cast<const float4 *>(a.noise + o);
        }
        const float4 mv = *reinterpret_cast<const float4 *>(mask + mask_index(i, chw, hw));
        const float4 yv = *reinterpret_cast<const float4 *>(a.y + (a.y_n == 1 ? 0 : p) * chw + i);
        bool b0, b1, b2, b3;
        float4 x0, sm;
        x0.x = post_x0(xv.x, ev.x, a.k, b0);
        x0.y = post_x0(xv.y, ev.y, a.k, b1);
        x0.z = post_x0(xv.z, ev.z, a.k, b2);
        x0.w = post_x0(xv.w, ev.w, a.k, b3);
        sm.x = post_sample(xv.x, x0.x, vv.x, zv.x, a.k);
        sm.y = post_sample(xv.y, x0.y, vv.y, zv.y, a.k);
        sm.z = post_sample(xv.z, x0.z, vv.z, zv.z, a.k);
        sm.w = post_sample(xv.w, x0.w, vv.w, zv.w, a.k);
        *reinterpret_cast<float4 *>(a.x0_hat + o) = x0;
        *reinterpret_cast<float4 *>(a.sample + o) = sm;
        *reinterpret_cast<uchar4 *>(a.inside + o) = make_uchar4(b0, b1, b2, b3);
        const float d0 = __fsub_rn(yv.x, __fmul_rn(x0.x, mv.x)), d1 = __fsub_rn(yv.y, __fmul_rn(x0.y, mv.y));
        const float d2 = __fsub_rn(yv.z, __fmul_rn(x0.z, mv.z)), d3 = __fsub_rn(yv.w, __fmul_rn(x0.w, mv.w));
        acc = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    const float t = block_sum(acc, scratch);
    if (threadIdx.x == 0) tail_publish(&a.partials[p * gridDim.x + blockIdx.x], t, a.tail.counters != nullptr);
    tail_arrive(a.tail, (int)p);
}

int mask_step_fwd(const dpsx_op *op, const StepFwdArgs &f, int parts, hipStream_t s)
{
    const int64_t chw = f.c * f.h * f.w;
    StepFwdArgs a = f;
    a.tail.blocks_per_particle = parts;
    k_mask_step_fwd<<<dim3(parts, (unsigned)a.n), kThreads, 0, s>>>(a, op->mask, chw, a.h * a.w);
    return check_launch();
}

__global__ __launch_bounds__(kThreads) void k_mask_step_bwd(StepBwdArgs a, const float *__restrict__ mask,
                                                            int64_t chw, int64_t hw)
{
    __shared__ float s_nrm[1];
    const int64_t p = blockIdx.y;
    const int64_t i = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * 4;
    if (!a.norm) {                                   // block-uniform: derive the norm from the forward half's partials
        particle_norm_to_lds(a.partials, a.parts, p, s_nrm);
        __syncthreads();
        if (a.norm_out && blockIdx.x == 0 && threadIdx.x == 0) a.norm_out[p] = s_nrm[0];
    }
    if (i >= chw) return;
    const int64_t o = p * chw + i;
    const float coef = norm_coef(a.norm ? a.norm[p] : s_nrm[0], a.scale, a.power);  // cotangent on A x0 is coef * r
    const float4 x0 = *reinterpret_cast<const float4 *>(a.x0_hat + o);
    const float4 mv = *reinterpret_cast<const float4 *>(mask + mask_index(i, chw, hw));
    const float4 yv = *reinterpret_cast<const float4 *>(a.y + (a.y_n == 1 ? 0 : p) * chw + i);
    const uchar4 in = *reinterpret_cast<const uchar4 *>(a.inside + o);
    float4 g;
    // A^T = multiply by mask again; then clamp gate; then d/d eps = -b
    float4 ex = make_float4(0, 0, 0, 0);
    if (a.g_extra) ex = *reinterpret_cast<const float4 *>(a.g_extra + o);     // block-uniform
    g.x = in.x ? -a.k.b * (coef * __fsub_rn(yv.x, __fmul_rn(x0.x, mv.x)) * mv.x + ex.x) : 0.0f;
    g.y = in.y ? -a.k.b * (coef * __fsub_rn(yv.y, __fmul_rn(x0.y, mv.y)) * mv.y + ex.y) : 0.0f;
    g.z = in.z ? -a.k.b * (coef * __fsub_rn(yv.z, __fmul_rn(x0.z, mv.z)) * mv.z + ex.z) : 0.0f;
    g.w = in.w ? -a.k.b * (coef * __fsub_rn(yv.w, __fmul_rn(x0.w, mv.w)) * mv.w + ex.w) : 0.0f;
    *reinterpret_cast<float4 *>(a.g_model_out + p * 2 * chw + i) = g;
}

int mask_step_bwd(const dpsx_op *op, const StepBwdArgs &a, hipStream_t s)
{
    const int64_t chw = a.c * a.h * a.w;
    k_mask_step_bwd<<<grid_for(chw / 4, a.n), kThreads, 0, s>>>(a, op->mask, chw, a.h * a.w);
    return check_launch();
}

// ===================================================================== select
// torch.argmin: first minimum; NaN is the minimum.  One block; n is small (<= a few thousand).
// (value, index) pairs are reduced with a total order -- NaN before everything, then the smaller value, then the
// smaller index -- by wave shuffles and one LDS hop, so the result does not depend on the reduction shape.
// (ArgMin / argmin_better: common.h, shared with the in-launch tail)
__global__ __launch_bounds__(kThreads) void k_argmin(const float *__restrict__ v, int64_t n, int64_t *__restrict__ out,
                                                     float *__restrict__ val_out)
{
    __shared__ float s_val[kThreads / kWave];
    __shared__ int64_t s_idx[kThreads / kWave];
    ArgMin best{0.0f, -1};
    for (int64_t i = threadIdx.x; i < n; i += kThreads) {
        const ArgMin c{v[i], i};
        if (argmin_better(c, best)) best = c;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        ArgMin c;
        c.v = __shfl_down(best.v, o, kWave);
        c.i = __shfl_down(best.i, o, kWave);
        if (argmin_better(c, best)) best = c;
    }
    const int wave = threadIdx.x / kWave;
    if (threadIdx.x % kWave == 0) { s_val[wave] = best.v; s_idx[wave] = best.i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kThreads / kWave; ++w) {
            const ArgMin c{s_val[w], s_idx[w]};
            if (argmin_better(c, best)) best = c;
        }
        *out = best.i < 0 ? 0 : best.i;
        if (val_out) *val_out = best.v;
    }
}

int argmin_f32(const float *v, int64_t n, int64_t *idx, float *val, hipStream_t s)
{
    k_argmin<<<1, kThreads, 0, s>>>(v, n, idx, val);
    return check_launch();
}

__global__ __launch_bounds__(kThreads) void k_gather(const float *__restrict__ src, const int64_t *__restrict__ ids,
                                                     float *__restrict__ dst, int64_t n_src, int64_t chw4,
                                                     int replicate)
{
    const int64_t p = blockIdx.y;
    int64_t sidx = replicate ? ids[0] : ids[p];
    float4 *d4 = reinterpret_cast<float4 *>(dst) + p * chw4;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= chw4) return;
    // never read out of bounds: an id outside [0, n_src) poisons its destination particle with NaN (visible in every
    // later score) instead of faulting -- the ids come from the device (argmin / multinomial), no host check
    if (sidx < 0 || sidx >= n_src) {
        const float q = __builtin_nanf("");
        d4[i] = make_float4(q, q, q, q);
        return;
    }
    d4[i] = (reinterpret_cast<const float4 *>(src) + sidx * chw4)[i];
}

__global__ __launch_bounds__(kThreads) void k_gather_scalar(const float *__restrict__ src,
                                                            const int64_t *__restrict__ ids,
                                                            float *__restrict__ dst, int64_t n_src, int64_t chw,
                                                            int replicate)
{
    const int64_t p = blockIdx.y;
    int64_t sidx = replicate ? ids[0] : ids[p];
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= chw) return;
    dst[p * chw + i] = (sidx < 0 || sidx >= n_src) ? __builtin_nanf("") : src[sidx * chw + i];
}

// dst[p] = src[*idx] for all p: a lane reads its float4 of the winner once and stores it to kRepl destinations
// (one block per (slice, group of kRepl particles): 8x fewer, fatter blocks than one per destination particle)
constexpr int kRepl = 8;
__global__ __launch_bounds__(kThreads) void k_replicate(const float *__restrict__ src, const int64_t *__restrict__ idx,
                                                        float *__restrict__ dst, int64_t n_out, int64_t n_src,
                                                        int64_t chw4)
{
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= chw4) return;
    const int64_t sidx = idx[0];
    const float q = __builtin_nanf("");
    const float4 v = (sidx < 0 || sidx >= n_src) ? make_float4(q, q, q, q)
                                                 : (reinterpret_cast<const float4 *>(src) + sidx * chw4)[i];
    const int64_t p0 = (int64_t)blockIdx.y * kRepl;
    float4 *d4 = reinterpret_cast<float4 *>(dst) + p0 * chw4 + i;
#pragma unroll
    for (int k = 0; k < kRepl; ++k)
        if (p0 + k < n_out) d4[(int64_t)k * chw4] = v;
}

// ---- the device half of the per-rank champion exchange (distributed.py: _exchange_champions)
// pack: out[0 .. chw) = particles[best], out[chw] = value (costs[best] or *val), out[chw + 1] = (float)best, two zeros --
// what ONE all-gather then carries to every rank.  best == nullptr: the torch.argmin-order select over costs runs here
// (every block recomputes it: n <= a few thousand floats from L2, cheaper than a launch of its own).
__global__ __launch_bounds__(kThreads) void k_pack_champion(const float *__restrict__ particles, const float *__restrict__ costs,
                                                            const int64_t *__restrict__ best_in,
                                                            const float *__restrict__ val_in, float *__restrict__ out,
                                                            int64_t n, int64_t chw4)
{
    __shared__ float s_val[kThreads / kWave];
    __shared__ int64_t s_idx[kThreads / kWave];
    __shared__ float s_bv;
    __shared__ int64_t s_bi;
    if (best_in) {
        if (threadIdx.x == 0) {         // (an index outside [0, n) yields a NaN record, never an out-of-bounds read)
            const int64_t bi = best_in[0];
            s_bi = bi;
            s_bv = val_in ? val_in[0] : ((bi >= 0 && bi < n) ? costs[bi] : __builtin_nanf(""));
        }
    } else {
        ArgMin best{0.0f, -1};
        for (int64_t i = threadIdx.x; i < n; i += kThreads) {
            const ArgMin c{costs[i], i};
            if (argmin_better(c, best)) best = c;
        }
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            ArgMin c;
            c.v = __shfl_down(best.v, o, kWave);
            c.i = __shfl_down(best.i, o, kWave);
            if (argmin_better(c, best)) best = c;
        }
        const int wave = threadIdx.x / kWave;
        if (threadIdx.x % kWave == 0) { s_val[wave] = best.v; s_idx[wave] = best.i; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < kThreads / kWave; ++w) {
                const ArgMin c{s_val[w], s_idx[w]};
                if (argmin_better(c, best)) best = c;
            }
            s_bi = best.i < 0 ? 0 : best.i;
            s_bv = best.v;
        }
    }
    __syncthreads();
    const int64_t b = s_bi;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    float4 *o4 = reinterpret_cast<float4 *>(out);
    if (i < chw4) {
        const float q = __builtin_nanf("");
        o4[i] = (b < 0 || b >= n) ? make_float4(q, q, q, q) : (reinterpret_cast<const float4 *>(particles) + b * chw4)[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) o4[chw4] = make_float4(s_bv, (float)b, 0.0f, 0.0f);
}

// select: table [world][chw + 4] as gathered; winner rank = torch.argmin order over table[r][chw] (first minimum, NaN
// counts as the minimum: lowest rank wins ties); dst[p] = table[winner][0 .. chw) for p < n_out; optional outputs: the
// winner's rank and its local index (table[winner][chw + 1]).
__global__ __launch_bounds__(kThreads) void k_select_champion(const float *__restrict__ table, int world, int64_t chw4,
                                                              float *__restrict__ dst, int64_t n_out,
                                                              int64_t *__restrict__ win_rank, int64_t *__restrict__ win_local)
{
    __shared__ int s_w;
    if (threadIdx.x < kWave) {
        ArgMin best{0.0f, -1};
        for (int r = threadIdx.x; r < world; r += kWave) {
            const ArgMin c{table[(int64_t)r * (chw4 + 1) * 4 + chw4 * 4], (int64_t)r};
            if (argmin_better(c, best)) best = c;
        }
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            ArgMin c;
            c.v = __shfl_down(best.v, o, kWave);
            c.i = __shfl_down(best.i, o, kWave);
            if (argmin_better(c, best)) best = c;
        }
        if (threadIdx.x == 0) s_w = best.i < 0 ? 0 : (int)best.i;
    }
    __syncthreads();
    const int wr = s_w;
    const float4 *src = reinterpret_cast<const float4 *>(table) + (int64_t)wr * (chw4 + 1);
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        if (win_rank) *win_rank = wr;
        if (win_local) *win_local = (int64_t)src[chw4].y;
    }
    if (i >= chw4) return;
    const float4 v = src[i];
    const int64_t p0 = (int64_t)blockIdx.y * kRepl;
    float4 *d4 = reinterpret_cast<float4 *>(dst) + p0 * chw4 + i;
#pragma unroll
    for (int k = 0; k < kRepl; ++k)
        if (p0 + k < n_out) d4[(int64_t)k * chw4] = v;
}

int pack_champion(const float *particles, const float *costs, const int64_t *best, const float *val, float *out, int64_t n,
                  int64_t chw, hipStream_t s)
{
    const int64_t chw4 = chw / 4;
    k_pack_champion<<<(unsigned)((chw4 + kThreads - 1) / kThreads), kThreads, 0, s>>>(particles, costs, best, val, out, n, chw4);
    return check_launch();
}

int select_champion(const float *table, int world, int64_t chw, float *dst, int64_t n_out, int64_t *win_rank,
                    int64_t *win_local, hipStream_t s)
{
    const int64_t chw4 = chw / 4;
    const dim3 grid((unsigned)((chw4 + kThreads - 1) / kThreads), (unsigned)((n_out + kRepl - 1) / kRepl));
    k_select_champion<<<grid, kThreads, 0, s>>>(table, world, chw4, dst, n_out, win_rank, win_local);
    return check_launch();
}

int gather_f32(const float *src, const int64_t *ids, float *dst, int64_t n_out, int64_t n_src, int64_t chw,
               bool replicate, hipStream_t s)
{
    if (n_out == 0 || chw == 0) return DPSX_OK;
    if (replicate && chw % 4 == 0 && aligned16(src) && aligned16(dst)) {
        const dim3 grid((unsigned)((chw / 4 + kThreads - 1) / kThreads), (unsigned)((n_out + kRepl - 1) / kRepl));
        k_replicate<<<grid, kThreads, 0, s>>>(src, ids, dst, n_out, n_src, chw / 4);
        return check_launch();
    }
    if (chw % 4 == 0 && aligned16(src) && aligned16(dst))
        k_gather<<<grid_for(chw / 4, n_out), kThreads, 0, s>>>(src, ids, dst, n_src, chw / 4, replicate);
    else
        k_gather_scalar<<<grid_for(chw, n_out), kThreads, 0, s>>>(src, ids, dst, n_src, chw, replicate);
    return check_launch();
}

}  // namespace dpsx
