// Phase-retrieval operator  A(x) = | F_c( zero-pad(x) ) |  with F_c the centred
// orthonormal 2-D FFT (reference: measurements.py:179-189, util/img_utils.py:26-30,
// util/fastmri_utils.py:67-89: ifftshift -> fftn(norm="ortho") -> fftshift).
//
// The FFT itself is the library transform (hipFFT/rocFFT, batched C2C, in place);
// everything around it is hand-written and fused into two element-wise passes:
//   pre  : zero-pad + ifftshift folded into the load index, real -> complex;
//   post : fftshift folded into the index, 1/s scaling, modulus, and -- in the
//          fused step -- residual y-|z|, per-block sums of squares, and the
//          cotangent w = (y-|z|) z/|z| written over the spectrum in place;
//   crop : after the (unnormalised) inverse FFT, undo shift + padding, real part.
// s = h + 2*pad is even, so both shifts are a roll by s/2.
#include <hipfft/hipfft.h>

#include <map>

#include "common.h"
#include "phase_fft.h"

namespace dpsx {

constexpr int PT = 256;
constexpr int kChunk = 4096;  // spectrum elements per block in the reducing pass

struct PhaseHost {
    std::map<int64_t, hipfftHandle> plans;  // batch * 4 + kind -> plan   (kind: 0 C2C, 1 R2C, 2 C2R)
    float2 *d_tw = nullptr;                 // exp(-2 pi i t / 384) for the hand-written spectral step
};

static PhaseHost *host_of(const dpsx_op *op) { return static_cast<PhaseHost *>(op->fft_plan); }

// the hand-written three-pass spectral step (phase_fft.h) covers the BASELINE geometry: 256 x 256, oversample 2.0
static bool spectral(const dpsx_op *op)
{
    static const bool off = getenv("DPSX_PHASE_LIBRARY_FFT") != nullptr;      // A/B switch for tools/kbench.py
    return !off && op->pr_h == prfft::IMG && op->pr_pad == prfft::PADW && host_of(op) && host_of(op)->d_tw;
}


// A/B switch (tools/kbench.py): the round-2 passes B and C of the hand-written spectral step
static bool phase_v1()
{
    static const bool on = getenv("DPSX_PHASE_V1") != nullptr;
    return on;
}
bool phase_norm_in_bwd(const dpsx_op *op) { return spectral(op) && !phase_v1(); }

static int get_plan(dpsx_op *op, int64_t planes, hipfftHandle *out, int kind = 0)
{
    PhaseHost *h = host_of(op);
    auto it = h->plans.find(planes * 4 + kind);
    if (it != h->plans.end()) { *out = it->second; return DPSX_OK; }
    const int s = (int)(op->pr_h + 2 * op->pr_pad);
    int dims[2] = {s, s};
    hipfftHandle plan;
    const hipfftType type = kind == 0 ? HIPFFT_C2C : (kind == 1 ? HIPFFT_R2C : HIPFFT_C2R);
    // default (packed) layouts: real [s][s], half spectrum [s][s/2 + 1]
    if (hipfftPlanMany(&plan, 2, dims, nullptr, 1, s * s, nullptr, 1, s * s, type, (int)planes) != HIPFFT_SUCCESS)
        return DPSX_ENOMEM;
    h->plans[planes * 4 + kind] = plan;
    *out = plan;
    return DPSX_OK;
}

int phase_create(dpsx_op *op)
{
    op->fft_plan = new PhaseHost();
    op->has_plan = true;
    hipfftHandle p;
    int rc = get_plan(op, op->pr_planes, &p);  // build the plan for the expected batch up front
    if (rc == DPSX_OK && op->pr_h == prfft::IMG && op->pr_pad == prfft::PADW) {
        float2 tw[prfft::N];
        for (int t = 0; t < prfft::N; ++t) {
            const double a = -2.0 * 3.14159265358979323846 * (double)t / (double)prfft::N;
            tw[t] = make_float2((float)cos(a), (float)sin(a));
        }
        PhaseHost *h = host_of(op);
        if (hipMalloc((void **)&h->d_tw, sizeof(tw)) != hipSuccess ||
            hipMemcpy(h->d_tw, tw, sizeof(tw), hipMemcpyHostToDevice) != hipSuccess)
            rc = DPSX_ENOMEM;
    }
    if (rc != DPSX_OK) phase_destroy(op);
    return rc;
}

void phase_destroy(dpsx_op *op)
{
    PhaseHost *h = host_of(op);
    if (!h) return;
    for (auto &kv : h->plans) (void)hipfftDestroy(kv.second);
    if (h->d_tw) (void)hipFree(h->d_tw);
    delete h;
    op->fft_plan = nullptr;
}

int64_t phase_workspace_bytes(const dpsx_op *op, int64_t planes)
{
    const int64_t s = op->pr_h + 2 * op->pr_pad;
    return (planes * s * s * 8 + 255) / 256 * 256;
}

// The fused step works on the Hermitian half: x0_hat is real, so Z[-k] = conj(Z[k]) and the s x (s/2 + 1) half
// spectrum of an R2C transform carries everything; the real part of the inverse transform of the cotangent W is the
// C2R transform of its Hermitian part  Wh[k] = (W[k] + conj(W[-k])) / 2 = (r(k) + r(-k)) / 2 * z[k] / |z[k]|.
// Half the FFT work and half the spectrum traffic of the complex-to-complex form (which the plain operator calls keep).
int64_t phase_parts_per_particle(const dpsx_op *op, int64_t c)
{
    const int64_t s = op->pr_h + 2 * op->pr_pad;
    if (spectral(op)) return c * ((prfft::HS + prfft::CT - 1) / prfft::CT);    // one partial per column tile
    return c * ((s * (s / 2 + 1) + kChunk - 1) / kChunk);
}

// fused-step scratch: [planes][s][s] real | [planes][s][s/2 + 1] complex
int64_t phase_step_resid_bytes(const dpsx_op *op, int64_t planes)
{
    const int64_t s = op->pr_h + 2 * op->pr_pad;
    return (planes * s * s * 4 + 255) / 256 * 256 + (planes * s * (s / 2 + 1) * 8 + 255) / 256 * 256;
}

// ---------------------------------------------------------------- kernels
__global__ __launch_bounds__(PT) void k_phase_pre(const float *__restrict__ x, float2 *__restrict__ q, int h,
                                                  int pad, int s)
{
    const int64_t plane = blockIdx.y;
    const int idx = blockIdx.x * PT + threadIdx.x;
    if (idx >= s * s) return;
    const int u = idx / s, v = idx - u * s, half = s / 2;
    int i = u + half, j = v + half;  // ifftshift: Q[u][v] = P[(u - s/2) mod s][(v - s/2) mod s]
    i = i >= s ? i - s : i;
    j = j >= s ? j - s : j;
    float val = 0.0f;
    if (i >= pad && i < pad + h && j >= pad && j < pad + h) val = x[plane * h * h + (int64_t)(i - pad) * h + (j - pad)];
    q[plane * s * s + idx] = make_float2(val, 0.0f);
}

// MODE 0: amp (+ optional centred spectrum) out.
// MODE 1: fused step: r = y - |z|, sums of squares, z <- r z/|z| in place.
// MODE 2: adjoint at a point: z <- u z/|z| in place (u given in centred layout).
template <int MODE>
__global__ __launch_bounds__(PT) void k_phase_post(float2 *__restrict__ z, float *__restrict__ amp,
                                                   float2 *__restrict__ spec, const float *__restrict__ yu,
                                                   int y_n, int c, float *__restrict__ partials, int s)
{
    __shared__ float scratch[PT / kWave];
    const int64_t plane = blockIdx.y;
    const int64_t ss = (int64_t)s * s;
    const int half = s / 2;
    const float inv = 1.0f / (float)s;
    float acc = 0.0f;
    const int base = blockIdx.x * kChunk;
    for (int t = threadIdx.x; t < kChunk; t += PT) {
        const int idx = base + t;
        if (idx >= ss) break;
        const int k = idx / s, q = idx - k * s;
        int ok = k + half, oq = q + half;  // fftshift: centred[(k + s/2) mod s] = Z[k]
        ok = ok >= s ? ok - s : ok;
        oq = oq >= s ? oq - s : oq;
        const int64_t cidx = (int64_t)ok * s + oq;
        float2 zz = z[plane * ss + idx];
        zz.x *= inv;
        zz.y *= inv;
        const float mag = sqrtf(zz.x * zz.x + zz.y * zz.y);
        if (MODE == 0) {
            amp[plane * ss + cidx] = mag;
            if (spec) spec[plane * ss + cidx] = zz;
        } else {
            const int n = (int)(plane / c), ch = (int)(plane % c);
            const int64_t yo = ((int64_t)(y_n == 1 ? 0 : n) * c + ch) * ss + cidx;
            float g;
            if (MODE == 1) {
                g = yu[yo] - mag;
                acc = fmaf(g, g, acc);
            } else {
                g = yu[plane * ss + cidx];
            }
            const float f = mag == 0.0f ? 0.0f : g / mag;  // torch: d|z| = 0 at z = 0
            z[plane * ss + idx] = make_float2(f * zz.x, f * zz.y);
        }
    }
    if (MODE == 1) {
        const float t = block_sum(acc, scratch);
        const int chunks = gridDim.x;
        const int n = (int)(plane / c), ch = (int)(plane % c);
        if (threadIdx.x == 0) partials[((int64_t)n * c + ch) * chunks + blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(PT) void k_phase_crop(const float2 *__restrict__ v, float *__restrict__ g, int h,
                                                   int pad, int s)
{
    const int64_t plane = blockIdx.y;
    const int idx = blockIdx.x * PT + threadIdx.x;
    if (idx >= h * h) return;
    const int a = idx / h, b = idx - a * h, half = s / 2;
    int i = a + pad + half, j = b + pad + half;  // undo ifftshift: P[i][j] = Q[(i + s/2) mod s][...]
    i = i >= s ? i - s : i;
    j = j >= s ? j - s : j;
    g[plane * h * h + idx] = v[plane * s * s + (int64_t)i * s + j].x * (1.0f / (float)s);
}

// ---- Hermitian-half kernels of the fused step
__global__ __launch_bounds__(PT) void k_phase_pre_real(const float *__restrict__ x, float *__restrict__ q, int h,
                                                       int pad, int s)
{
    const int64_t plane = blockIdx.y;
    const int idx = blockIdx.x * PT + threadIdx.x;
    if (idx >= s * s) return;
    const int u = idx / s, v = idx - u * s, half = s / 2;
    int i = u + half, j = v + half;  // ifftshift
    i = i >= s ? i - s : i;
    j = j >= s ? j - s : j;
    float val = 0.0f;
    if (i >= pad && i < pad + h && j >= pad && j < pad + h) val = x[plane * h * h + (int64_t)(i - pad) * h + (j - pad)];
    q[plane * s * s + idx] = val;
}

// z: [planes][s][s/2 + 1] from the R2C transform.  For k = (ky, kx <= s/2) and its mirror m = (-ky, -kx):
// r(k) = y[shift k] - |z[k]|/s, r(m) = y[shift m] - |z[k]|/s; sums of squares over the FULL spectrum (the mirror
// counts here unless it lies in the half itself, kx in {0, s/2}); z[k] <- (r(k) + r(m)) / 2 * z[k] / |z[k]|.
__global__ __launch_bounds__(PT) void k_phase_post_half(float2 *__restrict__ z, const float *__restrict__ y, int y_n,
                                                        int c, float *__restrict__ partials, int s)
{
    __shared__ float scratch[PT / kWave];
    const int64_t plane = blockIdx.y;
    const int hs = s / 2 + 1, half = s / 2;
    const int64_t hh = (int64_t)s * hs, ss = (int64_t)s * s;
    const float inv = 1.0f / (float)s;
    const int n = (int)(plane / c), ch = (int)(plane % c);
    const float *yp = y + ((int64_t)(y_n == 1 ? 0 : n) * c + ch) * ss;
    float acc = 0.0f;
    const int base = blockIdx.x * kChunk;
    for (int t = threadIdx.x; t < kChunk; t += PT) {
        const int idx = base + t;
        if (idx >= hh) break;
        const int ky = idx / hs, kx = idx - ky * hs;
        const int my = ky == 0 ? 0 : s - ky, mx = kx == 0 ? 0 : s - kx;          // the mirror frequency
        int oy = ky + half, ox = kx + half, py = my + half, px = mx + half;      // fftshift of both
        oy = oy >= s ? oy - s : oy; ox = ox >= s ? ox - s : ox;
        py = py >= s ? py - s : py; px = px >= s ? px - s : px;
        float2 zz = z[plane * hh + idx];
        zz.x *= inv;
        zz.y *= inv;
        const float mag = sqrtf(zz.x * zz.x + zz.y * zz.y);
        const float r1 = yp[(int64_t)oy * s + ox] - mag, r2 = yp[(int64_t)py * s + px] - mag;
        acc = fmaf(r1, r1, acc);
        if (kx != 0 && kx != half) acc = fmaf(r2, r2, acc);
        const float f = mag == 0.0f ? 0.0f : 0.5f * (r1 + r2) / mag;  // torch: d|z| = 0 at z = 0
        z[plane * hh + idx] = make_float2(f * zz.x, f * zz.y);
    }
    const float t = block_sum(acc, scratch);
    if (threadIdx.x == 0) partials[((int64_t)n * c + ch) * gridDim.x + blockIdx.x] = t;
}

__global__ __launch_bounds__(PT) void k_phase_crop_real(const float *__restrict__ v, float *__restrict__ g, int h,
                                                        int pad, int s)
{
    const int64_t plane = blockIdx.y;
    const int idx = blockIdx.x * PT + threadIdx.x;
    if (idx >= h * h) return;
    const int a = idx / h, b = idx - a * h, half = s / 2;
    int i = a + pad + half, j = b + pad + half;  // undo ifftshift
    i = i >= s ? i - s : i;
    j = j >= s ? j - s : j;
    g[plane * h * h + idx] = v[plane * s * s + (int64_t)i * s + j] * (1.0f / (float)s);
}

// S1 fused with the zero-pad + ifftshift staging of the fused step: one float4 unit of the padded (shifted) image per
// lane; interior units do the posterior arithmetic (x0_hat, sample, gate out) and write x0_hat into the transform's
// input, the others write zeros.  Needs h, pad, s/2 multiples of 4 (so a unit never straddles the wrap or the border).
__global__ __launch_bounds__(PT) void k_phase_s1_pre(const float *__restrict__ x, const float *__restrict__ mo,
                                                     const float *__restrict__ z, float *__restrict__ x0o,
                                                     float *__restrict__ so, uint8_t *__restrict__ ins,
                                                     float *__restrict__ q, int h, int pad, int s, int c, Coefs k)
{
    const int64_t plane = blockIdx.y;
    const int unit = blockIdx.x * PT + threadIdx.x, su = s / 4;
    if (unit >= s * su) return;
    const int u = unit / su, v = (unit - u * su) * 4, half = s / 2;
    int i = u + half, j = v + half;
    i = i >= s ? i - s : i;
    j = j >= s ? j - s : j;
    float4 val = make_float4(0, 0, 0, 0);
    if (i >= pad && i < pad + h && j >= pad && j < pad + h) {
        const int64_t hw = (int64_t)h * h, o = plane * hw + (int64_t)(i - pad) * h + (j - pad);
        const int64_t n = plane / c, ch = plane % c;
        const int64_t e = (n * 2 * c + ch) * hw + (int64_t)(i - pad) * h + (j - pad);
        const float4 xv = *reinterpret_cast<const float4 *>(x + o);
        const float4 ev = *reinterpret_cast<const float4 *>(mo + e);
        float4 vv = make_float4(0, 0, 0, 0), zv = vv;
        if (k.add_noise & 1) {
            vv = *reinterpret_cast<const float4 *>(mo + e + (int64_t)c * hw);
            zv = *reinterpret_cast<const float4 *>(z + o);
        }
        bool b0, b1, b2, b3;
        float4 sm;
        val.x = post_x0(xv.x, ev.x, k, b0);
        val.y = post_x0(xv.y, ev.y, k, b1);
        val.z = post_x0(xv.z, ev.z, k, b2);
        val.w = post_x0(xv.w, ev.w, k, b3);
        sm.x = post_sample(xv.x, val.x, vv.x, zv.x, k);
        sm.y = post_sample(xv.y, val.y, vv.y, zv.y, k);
        sm.z = post_sample(xv.z, val.z, vv.z, zv.z, k);
        sm.w = post_sample(xv.w, val.w, vv.w, zv.w, k);
        *reinterpret_cast<float4 *>(x0o + o) = val;
        *reinterpret_cast<float4 *>(so + o) = sm;
        *reinterpret_cast<uchar4 *>(ins + o) = make_uchar4(b0, b1, b2, b3);
    }
    *reinterpret_cast<float4 *>(q + plane * s * s + (int64_t)u * s + v) = val;
}

// crop + clamp gate + -b * coef (+ the optional extra cotangent) of the backward half in one pass over the image
__global__ __launch_bounds__(PT) void k_phase_crop_clamp(const float *__restrict__ v, const float *__restrict__ norm,
                                                         const uint8_t *__restrict__ ins,
                                                         const float *__restrict__ g_extra, float scale, int power,
                                                         float neg_b, float *__restrict__ g_mo, int h, int pad, int s,
                                                         int c)
{
    const int64_t plane = blockIdx.y;
    const int unit = blockIdx.x * PT + threadIdx.x, hu = h / 4;
    if (unit >= h * hu) return;
    const int a = unit / hu, b = (unit - a * hu) * 4, half = s / 2;
    int i = a + pad + half, j = b + pad + half;
    i = i >= s ? i - s : i;
    j = j >= s ? j - s : j;
    const float4 t = *reinterpret_cast<const float4 *>(v + plane * s * s + (int64_t)i * s + j);
    const int64_t n = plane / c, ch = plane % c, hw = (int64_t)h * h, o = plane * hw + (int64_t)a * h + b;
    const float nv = norm[n];
    const float coef = (power == 2 ? -2.0f * scale : (nv == 0.0f ? 0.0f : -scale / nv)) * (1.0f / (float)s);
    const uchar4 in = *reinterpret_cast<const uchar4 *>(ins + o);
    float4 ex = make_float4(0, 0, 0, 0);
    if (g_extra) ex = *reinterpret_cast<const float4 *>(g_extra + o);
    float4 g;
    g.x = in.x ? neg_b * (coef * t.x + ex.x) : 0.0f;
    g.y = in.y ? neg_b * (coef * t.y + ex.y) : 0.0f;
    g.z = in.z ? neg_b * (coef * t.z + ex.z) : 0.0f;
    g.w = in.w ? neg_b * (coef * t.w + ex.w) : 0.0f;
    *reinterpret_cast<float4 *>(g_mo + (n * 2 * c + ch) * hw + (int64_t)a * h + b) = g;
}

// ---------------------------------------------------------------- host
static int run_fft(dpsx_op *op, float2 *buf, int64_t planes, int dir, hipStream_t s)
{
    hipfftHandle plan;
    int rc = get_plan(op, planes, &plan);
    if (rc != DPSX_OK) return rc;
    if (hipfftSetStream(plan, s) != HIPFFT_SUCCESS) return DPSX_ELAUNCH;
    if (hipfftExecC2C(plan, reinterpret_cast<hipfftComplex *>(buf), reinterpret_cast<hipfftComplex *>(buf), dir) !=
        HIPFFT_SUCCESS)
        return DPSX_ELAUNCH;
    return DPSX_OK;
}

static int spectrum_of(dpsx_op *op, const float *x, float2 *buf, int64_t planes, hipStream_t s)
{
    const int h = (int)op->pr_h, pad = (int)op->pr_pad, sz = h + 2 * pad;
    k_phase_pre<<<dim3((sz * sz + PT - 1) / PT, (unsigned)planes), PT, 0, s>>>(x, buf, h, pad, sz);
    int rc = check_launch();
    if (rc != DPSX_OK) return rc;
    return run_fft(op, buf, planes, HIPFFT_FORWARD, s);
}

int phase_forward(dpsx_op *op, const float *x, float *amp, float *spec, int64_t planes, void *ws,
                  int64_t ws_bytes, hipStream_t s)
{
    if (planes == 0) return DPSX_OK;
    if (!ws || ws_bytes < phase_workspace_bytes(op, planes)) return DPSX_EWORKSPACE;
    const int sz = (int)(op->pr_h + 2 * op->pr_pad);
    float2 *buf = static_cast<float2 *>(ws);
    int rc = spectrum_of(op, x, buf, planes, s);
    if (rc != DPSX_OK) return rc;
    const unsigned chunks = (unsigned)((sz * sz + kChunk - 1) / kChunk);
    k_phase_post<0><<<dim3(chunks, (unsigned)planes), PT, 0, s>>>(buf, amp, reinterpret_cast<float2 *>(spec),
                                                                   nullptr, 1, 1, nullptr, sz);
    return check_launch();
}

int phase_adjoint(dpsx_op *op, const float *u, const float *x, float *g, int64_t planes, void *ws,
                  int64_t ws_bytes, hipStream_t s)
{
    if (planes == 0) return DPSX_OK;
    if (!ws || ws_bytes < phase_workspace_bytes(op, planes)) return DPSX_EWORKSPACE;
    const int h = (int)op->pr_h, pad = (int)op->pr_pad, sz = h + 2 * pad;
    float2 *buf = static_cast<float2 *>(ws);
    int rc = spectrum_of(op, x, buf, planes, s);
    if (rc != DPSX_OK) return rc;
    const unsigned chunks = (unsigned)((sz * sz + kChunk - 1) / kChunk);
    k_phase_post<2><<<dim3(chunks, (unsigned)planes), PT, 0, s>>>(buf, nullptr, nullptr, u, 0, 1, nullptr, sz);
    if ((rc = check_launch()) != DPSX_OK) return rc;
    if ((rc = run_fft(op, buf, planes, HIPFFT_BACKWARD, s)) != DPSX_OK) return rc;
    k_phase_crop<<<dim3((h * h + PT - 1) / PT, (unsigned)planes), PT, 0, s>>>(buf, g, h, pad, sz);
    return check_launch();
}

bool phase_is_spectral(const dpsx_op *op) { return spectral(op); }

bool phase_vec4_ok(const dpsx_op *op)
{
    const int64_t h = op->pr_h, pad = op->pr_pad, sz = h + 2 * pad;
    return h % 4 == 0 && pad % 4 == 0 && (sz / 2) % 4 == 0;
}

int phase_step_fwd(dpsx_op *op, const StepFwdArgs &f, float *resid_c, hipStream_t s)
{
    const int64_t n = f.n, c = f.c, planes = n * c;
    if (planes == 0) return DPSX_OK;
    const int h = (int)op->pr_h, pad = (int)op->pr_pad, sz = h + 2 * pad;
    float *real = resid_c;
    float2 *half = reinterpret_cast<float2 *>(reinterpret_cast<char *>(resid_c) +
                                              (planes * sz * sz * 4 + 255) / 256 * 256);
    int rc;
    const bool vec = phase_vec4_ok(op) && aligned16(f.x_t) && aligned16(f.model_out) && aligned16(f.noise) &&
                     aligned16(f.x0_hat) && aligned16(f.sample) && (reinterpret_cast<uintptr_t>(f.inside) & 3u) == 0;
    if (spectral(op) && !vec) return DPSX_EUNSUPPORTED;   // the partial-sum layout is fixed per operator
    if (vec && spectral(op)) {       // passes A and B of the hand-written spectral step (phase_fft.h)
        float2 *hbuf = reinterpret_cast<float2 *>(resid_c);
        const float2 *tw = host_of(op)->d_tw;
        prfft::k_pr_rows_fwd<<<dim3(prfft::IMG / prfft::RPB, (unsigned)planes), 256, 0, s>>>(
            f.x_t, f.model_out, f.noise, f.x0_hat, f.sample, f.inside, hbuf, tw, (int)c, f.k);
        if ((rc = check_launch()) != DPSX_OK) return rc;
        const unsigned tiles = (prfft::HS + prfft::CT - 1) / prfft::CT;
        if (phase_v1()) {
            const size_t lds = (size_t)(prfft::N + prfft::N * prfft::CT) * sizeof(float2);
            prfft::k_pr_cols<<<dim3(tiles, (unsigned)planes), prfft::BT, lds, s>>>(hbuf, f.y, (int)f.y_n, (int)c, f.partials, tw);
            return check_launch();
        }
        const size_t lds = (size_t)prfft::B2LDS * sizeof(float2);
        const unsigned nyq_blocks = (unsigned)((planes + prfft::CT - 1) / prfft::CT);
        const unsigned grid = nyq_blocks + (unsigned)planes * (prfft::HS / prfft::CT);
        prfft::k_pr_cols2<<<grid, prfft::B2T, lds, s>>>(hbuf, f.y, (int)f.y_n, (int)c, f.partials, tw, (int)planes,
                                                        (int)nyq_blocks);
        return check_launch();
    }
    if (vec) {      // S1 and the transform's input staging in one pass
        k_phase_s1_pre<<<dim3((sz * (sz / 4) + PT - 1) / PT, (unsigned)planes), PT, 0, s>>>(
            f.x_t, f.model_out, f.noise, f.x0_hat, f.sample, f.inside, real, h, pad, sz, (int)c, f.k);
        rc = check_launch();
    } else {
        rc = posterior_fwd(f.x_t, f.model_out, f.noise, f.x0_hat, f.sample, f.inside, n, c * h * h, f.k, s);
        if (rc != DPSX_OK) return rc;
        k_phase_pre_real<<<dim3((sz * sz + PT - 1) / PT, (unsigned)planes), PT, 0, s>>>(f.x0_hat, real, h, pad, sz);
        rc = check_launch();
    }
    if (rc != DPSX_OK) return rc;
    const float *y = f.y;
    const int64_t y_n = f.y_n;
    float *partials = f.partials;
    hipfftHandle plan;
    if ((rc = get_plan(op, planes, &plan, 1)) != DPSX_OK) return rc;
    if (hipfftSetStream(plan, s) != HIPFFT_SUCCESS) return DPSX_ELAUNCH;
    if (hipfftExecR2C(plan, real, reinterpret_cast<hipfftComplex *>(half)) != HIPFFT_SUCCESS) return DPSX_ELAUNCH;
    const unsigned chunks = (unsigned)((sz * (sz / 2 + 1) + kChunk - 1) / kChunk);
    k_phase_post_half<<<dim3(chunks, (unsigned)planes), PT, 0, s>>>(half, y, (int)y_n, (int)c, partials, sz);
    return check_launch();
}

// the whole backward half when the 16-byte form applies: C2R, then crop + gate + scaling in one pass
int phase_step_bwd_fused(dpsx_op *op, float *resid_c, const StepBwdArgs &b, hipStream_t s)
{
    const int64_t planes = b.n * b.c;
    if (planes == 0) return DPSX_OK;
    if (spectral(op)) {              // pass C
        if (phase_v1()) {
            if (!b.norm) return DPSX_EINVAL;
            prfft::k_pr_rows_inv<<<dim3(prfft::IMG / prfft::RPB, (unsigned)planes), 256, 0, s>>>(
                reinterpret_cast<const float2 *>(resid_c), b.norm, b.inside, b.g_extra, b.scale, b.power, -b.k.b,
                b.g_model_out, host_of(op)->d_tw, (int)b.c);
            return check_launch();
        }
        // b.norm == nullptr: the launch finalises the per-particle norm from the forward half's partial sums itself
        prfft::k_pr_rows_inv2<<<dim3(prfft::IMG / prfft::C2ROWS, (unsigned)planes), prfft::C2T, 0, s>>>(
            reinterpret_cast<const float2 *>(resid_c), b.norm, b.partials, b.parts, b.norm_out, b.inside, b.g_extra,
            b.scale, b.power, -b.k.b, b.g_model_out, host_of(op)->d_tw, (int)b.c);
        return check_launch();
    }
    const int h = (int)op->pr_h, pad = (int)op->pr_pad, sz = h + 2 * pad;
    float *real = resid_c;
    float2 *half = reinterpret_cast<float2 *>(reinterpret_cast<char *>(resid_c) +
                                              (planes * sz * sz * 4 + 255) / 256 * 256);
    hipfftHandle plan;
    int rc = get_plan(op, planes, &plan, 2);
    if (rc != DPSX_OK) return rc;
    if (hipfftSetStream(plan, s) != HIPFFT_SUCCESS) return DPSX_ELAUNCH;
    if (hipfftExecC2R(plan, reinterpret_cast<hipfftComplex *>(half), real) != HIPFFT_SUCCESS) return DPSX_ELAUNCH;
    k_phase_crop_clamp<<<dim3((h * (h / 4) + PT - 1) / PT, (unsigned)planes), PT, 0, s>>>(
        real, b.norm, b.inside, b.g_extra, b.scale, b.power, -b.k.b, b.g_model_out, h, pad, sz, (int)b.c);
    return check_launch();
}

int phase_step_bwd(dpsx_op *op, float *resid_c, float *g_x0, int64_t planes, hipStream_t s)
{
    if (planes == 0) return DPSX_OK;
    const int h = (int)op->pr_h, pad = (int)op->pr_pad, sz = h + 2 * pad;
    float *real = resid_c;
    float2 *half = reinterpret_cast<float2 *>(reinterpret_cast<char *>(resid_c) +
                                              (planes * sz * sz * 4 + 255) / 256 * 256);
    hipfftHandle plan;
    int rc = get_plan(op, planes, &plan, 2);
    if (rc != DPSX_OK) return rc;
    if (hipfftSetStream(plan, s) != HIPFFT_SUCCESS) return DPSX_ELAUNCH;
    if (hipfftExecC2R(plan, reinterpret_cast<hipfftComplex *>(half), real) != HIPFFT_SUCCESS) return DPSX_ELAUNCH;
    k_phase_crop_real<<<dim3((h * h + PT - 1) / PT, (unsigned)planes), PT, 0, s>>>(real, g_x0, h, pad, sz);
    return check_launch();
}

}  // namespace dpsx
