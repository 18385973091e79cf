// C ABI of libdpsx (see include/dpsx.h): argument checks, operator objects and
// dispatch to the kernels.  No torch types, no allocation on the launch path.
#include <cmath>
#include <cstring>
#include <new>
#include <algorithm>
#include <vector>

#include "common.h"

using namespace dpsx;

namespace dpsx {
static thread_local char g_hip_err[256] = "";
void set_last_hip_error(hipError_t e)
{
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s", hipGetErrorName(e), hipGetErrorString(e));
}
}  // namespace dpsx

#pragma GCC visibility push(default)
extern "C" {

int dpsx_abi_version(void) { return DPSX_ABI_VERSION; }

const char *dpsx_strerror(int code)
{
    switch (code) {
    case DPSX_OK: return "ok";
    case DPSX_EINVAL: return "invalid argument";
    case DPSX_EUNSUPPORTED: return "unsupported configuration";
    case DPSX_ELAUNCH: return "HIP launch/runtime error";
    case DPSX_ENOMEM: return "out of device memory";
    case DPSX_EWORKSPACE: return "workspace too small";
    }
    return "unknown error";
}

const char *dpsx_last_hip_error(void) { return g_hip_err; }

// ------------------------------------------------------------------ S1
int dpsx_posterior_fwd_f32(const float *x_t, const float *model_out, const float *noise, float *x0_hat,
                           float *sample, uint8_t *inside, int64_t n, int64_t chw,
                           const dpsx_coefs *coefs_host, void *stream)
{
    if (!coefs_host || n < 0 || chw < 0) return DPSX_EINVAL;
    if (n == 0 || chw == 0) return DPSX_OK;   // empty particle set: nothing to do (pointers may be null)
    if (!x_t || !model_out) return DPSX_EINVAL;
    if ((coefs_host->add_noise & 1) && !noise) return DPSX_EINVAL;
    return posterior_fwd(x_t, model_out, noise, x0_hat, sample, inside, n, chw, to_coefs(coefs_host),
                         (hipStream_t)stream);
}

int dpsx_posterior_bwd_f32(const float *g_x0, const float *g_sample, const float *x_t, const float *model_out,
                           const float *noise, float *g_x, float *g_model_out, int64_t n, int64_t chw,
                           const dpsx_coefs *coefs_host, void *stream)
{
    if (!coefs_host || n < 0 || chw < 0) return DPSX_EINVAL;
    if (n == 0 || chw == 0) return DPSX_OK;
    if (!x_t || !model_out || !g_x || !g_model_out) return DPSX_EINVAL;
    if ((coefs_host->add_noise & 1) && g_sample && !noise) return DPSX_EINVAL;
    return posterior_bwd(g_x0, g_sample, x_t, model_out, noise, g_x, g_model_out, n, chw, to_coefs(coefs_host),
                         (hipStream_t)stream);
}

// ------------------------------------------------------------------ operator objects
// arrival counters of the in-launch reductions (common.h: Tail); zero between launches
static int alloc_counters(dpsx_op *op)
{
    const size_t bytes = (size_t)(1 + kTailMaxParticles) * sizeof(unsigned);
    DPSX_HIP_TRY(hipMalloc((void **)&op->d_counters, bytes));
    DPSX_HIP_TRY(hipMemset(op->d_counters, 0, bytes));
    return DPSX_OK;
}

static int finish_create(dpsx_op *op, dpsx_op **out)
{
    int rc = alloc_counters(op);
    if (rc != DPSX_OK) { dpsx_op_destroy(op); return rc; }
    *out = op;
    return DPSX_OK;
}

static int upload_taps(dpsx_op *op, const std::vector<int> &dy, const std::vector<int> &dx,
                       const std::vector<float> &w)
{
    const size_t n = std::max<size_t>(w.size(), 1);
    DPSX_HIP_TRY(hipMalloc((void **)&op->d_tap_dy, n * sizeof(int)));
    DPSX_HIP_TRY(hipMalloc((void **)&op->d_tap_dx, n * sizeof(int)));
    DPSX_HIP_TRY(hipMalloc((void **)&op->d_tap_w, n * sizeof(float)));
    if (!w.empty()) {
        DPSX_HIP_TRY(hipMemcpy(op->d_tap_dy, dy.data(), w.size() * sizeof(int), hipMemcpyHostToDevice));
        DPSX_HIP_TRY(hipMemcpy(op->d_tap_dx, dx.data(), w.size() * sizeof(int), hipMemcpyHostToDevice));
        DPSX_HIP_TRY(hipMemcpy(op->d_tap_w, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return DPSX_OK;
}

int dpsx_op_create_blur(const float *kernel_host, int ks, int mode, dpsx_op **out)
{
    if (!kernel_host || !out || ks < 1 || ks % 2 == 0 || ks > 2 * kMaxRadius + 1) return DPSX_EINVAL;
    dpsx_op *op = new (std::nothrow) dpsx_op();
    if (!op) return DPSX_ENOMEM;
    const int R = ks / 2;
    op->ks = ks;
    // effective support: largest |offset| with a non-zero tap (the Gaussian sigma=3 kernel
    // occupies 25x25 of its 61x61 grid) -- smaller halos, same result (0 * x contributes nothing)
    int reach = 0;
    double peak = 0.0;
    int pi = R, pj = R;
    for (int i = 0; i < ks; ++i)
        for (int j = 0; j < ks; ++j) {
            const float v = kernel_host[i * ks + j];
            if (v != 0.0f) reach = std::max(reach, std::max(std::abs(i - R), std::abs(j - R)));
            if (std::fabs((double)v) > peak) { peak = std::fabs((double)v); pi = i; pj = j; }
        }
    op->radius = R;  // the reflection pad is always ks/2 (fold geometry), taps may reach less
    const int reach4 = std::max(4, (reach + 3) / 4 * 4);
    op->radius4 = reach4;
    op->reach = reach;
    // rank-1 test in double: K ~= col (x) row with col = K[:, pj], row = K[pi, :] / K[pi, pj]
    bool separable = mode != DPSX_BLUR_FORCE_TAPS && peak > 0.0;
    std::vector<double> col((size_t)ks), row((size_t)ks);
    if (separable) {
        for (int i = 0; i < ks; ++i) col[(size_t)i] = kernel_host[i * ks + pj];
        for (int j = 0; j < ks; ++j) row[(size_t)j] = (double)kernel_host[pi * ks + j] / (double)kernel_host[pi * ks + pj];
        for (int i = 0; i < ks && separable; ++i)
            for (int j = 0; j < ks; ++j)
                if (std::fabs(col[(size_t)i] * row[(size_t)j] - (double)kernel_host[i * ks + j]) > 1e-6 * peak) {
                    separable = false;
                    break;
                }
    }
    op->kind = separable ? OP_SEP : OP_TAPS;
    if (separable)
        for (int d = -reach; d <= reach; ++d) {
            op->sep.v[reach4 + d] = (float)col[(size_t)(R + d)];
            op->sep.h[reach4 + d] = (float)row[(size_t)(R + d)];
        }
    // the list of non-zero taps always exists: it is the whole implementation for non-separable kernels
    // and the fallback of separable ones for shapes the 16-byte loader cannot take (w % 4 != 0)
    {
        std::vector<int> dy, dx;
        std::vector<float> w;
        for (int i = 0; i < ks; ++i)
            for (int j = 0; j < ks; ++j)
                if (kernel_host[i * ks + j] != 0.0f) {
                    dy.push_back(i - R);
                    dx.push_back(j - R);
                    w.push_back(kernel_host[i * ks + j]);
                }
        op->nnz = (int)w.size();
        int rc = upload_taps(op, dy, dx, w);
        if (rc != DPSX_OK) { dpsx_op_destroy(op); return rc; }
        // per-side halo of the forward tap list (the adjoint swaps top <-> bottom, left <-> right)
        {
            int t = 0, b = 0, l = 0, r = 0;
            for (size_t i = 0; i < w.size(); ++i) {
                t = std::max(t, -dy[i]); b = std::max(b, dy[i]);
                l = std::max(l, -dx[i]); r = std::max(r, dx[i]);
            }
            if (t + b < 3) b = 3 - t;                                  // a run window spans four rows
            op->halo_t = t; op->halo_b = b;
            op->halo_l = (l + 3) / 4 * 4; op->halo_r = (r + 3) / 4 * 4;
        }
        // vertical runs of 4 or 2 taps per kernel column, window kept inside [-halo_t, halo_b]; four classes
        // (dx parity x run length) stored one after the other -- see TapRun
        std::vector<TapRun> cls_f[4], cls_a[4];
        for (int j = 0; j < ks; ++j) {
            const int dx = j - R, odd = dx & 1;
            int i = 0;
            while (i < ks) {
                if (kernel_host[i * ks + j] == 0.0f) { ++i; continue; }
                int k = 0;                                              // consecutive non-zero taps from row i
                while (i + k < ks && k < 4 && kernel_host[(i + k) * ks + j] != 0.0f) ++k;
                const int L = k >= 3 ? 4 : 2;
                const int first = i - R;                               // dy of the run's first tap
                const int dy0 = std::min(first, op->halo_b - (L - 1)); // shift up so dy0 + L - 1 <= halo_b (>= -halo_t)
                TapRun r{};
                r.dy0 = dy0; r.dx = dx;
                for (int q = 0; q < L; ++q) {
                    const int ii = dy0 + q + R;
                    r.w[q] = (ii >= 0 && ii < ks && ii >= i) ? kernel_host[ii * ks + j] : 0.0f;
                }
                cls_f[2 * odd + (L == 2)].push_back(r);
                // adjoint (correlation-transpose): V[p][q] = sum w_t u[p - dy_t][q - dx_t] -> negated offsets
                // (same dx parity), reversed weights, same loop
                TapRun t{};
                t.dy0 = -(dy0 + L - 1); t.dx = -dx;
                for (int q = 0; q < L; ++q) t.w[q] = r.w[L - 1 - q];
                cls_a[2 * odd + (L == 2)].push_back(t);
                i = dy0 + L + R;                                       // first row not covered by this run
            }
        }
        // an even number of runs per class (the pipelined loop works on pairs): a dummy run repeats the last offsets
        // with zero weights
        for (int c = 0; c < 4; ++c)
            if (cls_f[c].size() % 2) {
                TapRun f = cls_f[c].back(), t = cls_a[c].back();
                for (int q = 0; q < 4; ++q) f.w[q] = t.w[q] = 0.0f;
                cls_f[c].push_back(f);
                cls_a[c].push_back(t);
            }
        std::vector<TapRun> fwd, adj;
        for (int c = 0; c < 4; ++c) {
            std::stable_sort(cls_a[c].begin(), cls_a[c].end(), [](const TapRun &x, const TapRun &y) { return x.dx > y.dx; });
            for (const TapRun &t : cls_a[c]) {
                op->nrun_adj_pos[c] += t.dx >= 1;
                op->nrun_adj_neg[c] += t.dx <= -1;
            }
            op->nrun[c] = (int)cls_f[c].size();
            fwd.insert(fwd.end(), cls_f[c].begin(), cls_f[c].end());
            adj.insert(adj.end(), cls_a[c].begin(), cls_a[c].end());
        }
        op->nruns = (int)fwd.size();
        const size_t bytes = std::max<size_t>(fwd.size(), 1) * sizeof(TapRun);
        if (hipMalloc(&op->d_runs_fwd, bytes) != hipSuccess || hipMalloc(&op->d_runs_adj, bytes) != hipSuccess ||
            (!fwd.empty() && (hipMemcpy(op->d_runs_fwd, fwd.data(), fwd.size() * sizeof(TapRun), hipMemcpyHostToDevice) != hipSuccess ||
                              hipMemcpy(op->d_runs_adj, adj.data(), adj.size() * sizeof(TapRun), hipMemcpyHostToDevice) != hipSuccess))) {
            dpsx_op_destroy(op);
            return DPSX_ENOMEM;
        }
    }
    return finish_create(op, out);
}

int dpsx_op_create_resize(int64_t in_h, int64_t in_w, const float *w_h_host, const int64_t *i_h_host,
                          int64_t taps_h, int64_t out_h, const float *w_w_host, const int64_t *i_w_host,
                          int64_t taps_w, int64_t out_w, dpsx_op **out)
{
    if (!w_h_host || !i_h_host || !w_w_host || !i_w_host || !out) return DPSX_EINVAL;
    if (in_h < 1 || in_w < 1 || out_h < 1 || out_w < 1 || taps_h < 1 || taps_w < 1) return DPSX_EINVAL;
    if (in_h > 16384 || in_w > 16384) return DPSX_EUNSUPPORTED;
    dpsx_op *op = new (std::nothrow) dpsx_op();
    if (!op) return DPSX_ENOMEM;
    op->kind = OP_RESIZE;
    op->in_h = in_h; op->in_w = in_w; op->out_h = out_h; op->out_w = out_w; op->taps_h = taps_h; op->taps_w = taps_w;
    int rc = resize_create(op, w_h_host, i_h_host, w_w_host, i_w_host);
    if (rc != DPSX_OK) { delete op; return rc; }
    return finish_create(op, out);
}

int dpsx_op_create_mask(const float *mask_dev, int64_t h, int64_t w, dpsx_op **out)
{
    if (!mask_dev || !out || h < 1 || w < 1) return DPSX_EINVAL;
    dpsx_op *op = new (std::nothrow) dpsx_op();
    if (!op) return DPSX_ENOMEM;
    op->kind = OP_MASK;
    op->mask = mask_dev;
    op->in_h = h; op->in_w = w;
    return finish_create(op, out);
}

int dpsx_op_create_identity(dpsx_op **out)
{
    if (!out) return DPSX_EINVAL;
    dpsx_op *op = new (std::nothrow) dpsx_op();
    if (!op) return DPSX_ENOMEM;
    op->kind = OP_IDENT;
    return finish_create(op, out);
}

int dpsx_op_create_phase(int64_t h, int64_t pad, int64_t max_planes, dpsx_op **out)
{
    if (!out || h < 1 || pad < 0 || max_planes < 1 || (h + 2 * pad) % 2 != 0) return DPSX_EINVAL;
    dpsx_op *op = new (std::nothrow) dpsx_op();
    if (!op) return DPSX_ENOMEM;
    op->kind = OP_PHASE;
    op->pr_h = h; op->pr_pad = pad; op->pr_planes = max_planes;
    int rc = phase_create(op);
    if (rc != DPSX_OK) { delete op; return rc; }
    return finish_create(op, out);
}

void dpsx_op_destroy(dpsx_op *op)
{
    if (!op) return;
    if (op->d_tap_dy) (void)hipFree(op->d_tap_dy);
    if (op->d_tap_dx) (void)hipFree(op->d_tap_dx);
    if (op->d_tap_w) (void)hipFree(op->d_tap_w);
    if (op->d_runs_fwd) (void)hipFree(op->d_runs_fwd);
    if (op->d_runs_adj) (void)hipFree(op->d_runs_adj);
    if (op->d_counters) (void)hipFree(op->d_counters);
    if (op->kind == OP_RESIZE) resize_destroy(op);
    if (op->kind == OP_PHASE) phase_destroy(op);
    delete op;
}

int dpsx_op_kind(const dpsx_op *op) { return op ? op->kind : DPSX_EINVAL; }

int dpsx_op_out_shape(const dpsx_op *op, int64_t h, int64_t w, int64_t *out_h, int64_t *out_w)
{
    if (!op || !out_h || !out_w) return DPSX_EINVAL;
    switch (op->kind) {
    case OP_RESIZE:
        if (h != op->in_h || w != op->in_w) return DPSX_EINVAL;
        *out_h = op->out_h; *out_w = op->out_w;
        return DPSX_OK;
    case OP_PHASE:
        if (h != op->pr_h || w != op->pr_h) return DPSX_EINVAL;
        *out_h = *out_w = op->pr_h + 2 * op->pr_pad;
        return DPSX_OK;
    case OP_MASK:
        if (h != op->in_h || w != op->in_w) return DPSX_EINVAL;
        [[fallthrough]];
    default:
        *out_h = h; *out_w = w;
        return DPSX_OK;
    }
}

// partial sums-of-squares slots per particle for the fused paths
static int64_t parts_per_particle(const dpsx_op *op, int64_t c, int64_t h, int64_t w)
{
    switch (op->kind) {
    case OP_SEP:
    case OP_TAPS: return blur_parts_per_particle(op, c, h, w);
    case OP_RESIZE: return resize_parts_per_particle(op, c);
    case OP_MASK:
    case OP_IDENT: return (c * h * w + 1023) / 1024;
    case OP_PHASE: return phase_parts_per_particle(op, c);
    }
    return 1;
}

static inline int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

static int64_t meas_elems(const dpsx_op *op, int64_t c, int64_t h, int64_t w)
{
    int64_t oh = h, ow = w;
    if (dpsx_op_out_shape(op, h, w, &oh, &ow) != DPSX_OK) return -1;
    return c * oh * ow;
}

// workspace layout: [partials | scratch A (measurement-sized) | scratch B (image-sized) | op private]
int64_t dpsx_op_workspace_bytes(const dpsx_op *op, int64_t n, int64_t c, int64_t h, int64_t w)
{
    if (!op || n < 0 || c < 1 || h < 1 || w < 1) return DPSX_EINVAL;
    const int64_t m = meas_elems(op, c, h, w);
    if (m < 0) return DPSX_EINVAL;
    int64_t parts = std::max<int64_t>(parts_per_particle(op, c, h, w), 64);
    int64_t bytes = align256(n * parts * 4);
    if (op->kind == OP_SEP || op->kind == OP_TAPS) bytes += blur_adjoint_scratch_bytes(op, n * c, h, w);
    if (op->kind == OP_PHASE || op->kind == OP_IDENT || op->kind == OP_MASK) {
        bytes += align256(n * m * 4) + align256(n * c * h * w * 4);
        if (op->kind == OP_PHASE) bytes += phase_workspace_bytes(op, n * c);
    }
    return bytes;
}

struct Ws {
    float *partials;
    float *meas;   // measurement-sized scratch (only for unfused ops)
    float *img;    // image-sized scratch (only for unfused ops)
    void *priv;
    int64_t priv_bytes;
};

static int carve(const dpsx_op *op, void *ws, int64_t ws_bytes, int64_t n, int64_t c, int64_t h, int64_t w, Ws &o)
{
    const int64_t need = dpsx_op_workspace_bytes(op, n, c, h, w);
    if (need < 0) return DPSX_EINVAL;
    if (!ws || ws_bytes < need || !aligned16(ws)) return DPSX_EWORKSPACE;
    char *p = static_cast<char *>(ws);
    const int64_t parts = std::max<int64_t>(parts_per_particle(op, c, h, w), 64);
    o.partials = reinterpret_cast<float *>(p);
    p += align256(n * parts * 4);
    o.meas = o.img = nullptr;
    o.priv = nullptr;
    o.priv_bytes = 0;
    if (op->kind == OP_SEP || op->kind == OP_TAPS) {
        o.priv = p;
        o.priv_bytes = blur_adjoint_scratch_bytes(op, n * c, h, w);
    }
    if (op->kind == OP_PHASE || op->kind == OP_IDENT || op->kind == OP_MASK) {
        const int64_t m = meas_elems(op, c, h, w);
        o.meas = reinterpret_cast<float *>(p);
        p += align256(n * m * 4);
        o.img = reinterpret_cast<float *>(p);
        p += align256(n * c * h * w * 4);
        o.priv = p;
        o.priv_bytes = ws_bytes - (p - static_cast<char *>(ws));
    }
    return DPSX_OK;
}

static int check_geom(const dpsx_op *op, int64_t n, int64_t c, int64_t h, int64_t w)
{
    if (!op || n < 0 || c < 1 || h < 1 || w < 1) return DPSX_EINVAL;
    if (n * c > (1 << 24)) return DPSX_EUNSUPPORTED;
    int64_t oh, ow;
    return dpsx_op_out_shape(op, h, w, &oh, &ow);
}

int dpsx_op_forward_f32(dpsx_op *op, const float *x, float *y, int64_t n, int64_t c, int64_t h, int64_t w,
                        void *workspace, int64_t workspace_bytes, void *stream)
{
    int rc = check_geom(op, n, c, h, w);
    if (rc != DPSX_OK) return rc;
    if (!x || !y) return DPSX_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    switch (op->kind) {
    case OP_SEP:
    case OP_TAPS: return blur_forward(op, x, y, n * c, h, w, s);
    case OP_RESIZE: return resize_forward(op, x, y, n * c, s);
    case OP_MASK: return mask_mul(x, op->mask, y, n * c, h * w, s);
    case OP_IDENT:
        DPSX_HIP_TRY(hipMemcpyAsync(y, x, (size_t)(n * c * h * w) * 4, hipMemcpyDeviceToDevice, s));
        return DPSX_OK;
    case OP_PHASE: {
        Ws ws;
        if ((rc = carve(op, workspace, workspace_bytes, n, c, h, w, ws)) != DPSX_OK) return rc;
        return phase_forward(op, x, y, nullptr, n * c, ws.priv, ws.priv_bytes, s);
    }
    }
    return DPSX_EUNSUPPORTED;
}

int dpsx_op_adjoint_f32(dpsx_op *op, const float *u, const float *x, float *g, int64_t n, int64_t c, int64_t h,
                        int64_t w, void *workspace, int64_t workspace_bytes, void *stream)
{
    int rc = check_geom(op, n, c, h, w);
    if (rc != DPSX_OK) return rc;
    if (!u || !g) return DPSX_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    switch (op->kind) {
    case OP_SEP:
    case OP_TAPS: {
        Ws ws;
        if ((rc = carve(op, workspace, workspace_bytes, n, c, h, w, ws)) != DPSX_OK) return rc;
        return blur_adjoint(op, u, g, n * c, h, w, static_cast<float *>(ws.priv), ws.priv_bytes, s);
    }
    case OP_RESIZE: return resize_adjoint(op, u, g, n * c, s);
    case OP_MASK: return mask_mul(u, op->mask, g, n * c, h * w, s);
    case OP_IDENT:
        DPSX_HIP_TRY(hipMemcpyAsync(g, u, (size_t)(n * c * h * w) * 4, hipMemcpyDeviceToDevice, s));
        return DPSX_OK;
    case OP_PHASE: {
        if (!x) return DPSX_EINVAL;
        Ws ws;
        if ((rc = carve(op, workspace, workspace_bytes, n, c, h, w, ws)) != DPSX_OK) return rc;
        return phase_adjoint(op, u, x, g, n * c, ws.priv, ws.priv_bytes, s);
    }
    }
    return DPSX_EUNSUPPORTED;
}

// ------------------------------------------------------------------ residual norm
int dpsx_residual_norm_f32(const float *y, int64_t y_n, const float *ax, float *r, float *norm, int64_t n,
                           int64_t m, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (!y || !ax || !norm || n < 0 || m < 0 || (y_n != 1 && y_n != n)) return DPSX_EINVAL;
    if (n == 0) return DPSX_OK;
    const int parts = (int)std::min<int64_t>(256, std::max<int64_t>(1, (m + 4095) / 4096));
    if (!workspace || workspace_bytes < n * parts * 4) return DPSX_EWORKSPACE;
    float *partials = static_cast<float *>(workspace);
    hipStream_t s = (hipStream_t)stream;
    int rc = residual_partials(y, y_n, ax, r, partials, n, m, parts, s);
    if (rc != DPSX_OK) return rc;
    return finalize_norm(partials, parts, norm, n, s);
}

int dpsx_norm_bwd_f32(const float *r, const float *norm, const float *g_norm, int power, float *g_ax, int64_t n,
                      int64_t m, void *stream)
{
    if (!r || !norm || !g_norm || !g_ax || n < 0 || m < 0 || (power != 1 && power != 2)) return DPSX_EINVAL;
    return norm_bwd(r, norm, g_norm, power, g_ax, n, m, (hipStream_t)stream);
}

// ------------------------------------------------------------------ fused DPS step
int64_t dpsx_step_resid_bytes(const dpsx_op *op, int64_t n, int64_t c, int64_t h, int64_t w)
{
    if (!op) return DPSX_EINVAL;
    const int64_t m = meas_elems(op, c, h, w);
    if (m < 0) return DPSX_EINVAL;
    switch (op->kind) {
    case OP_MASK: return 256;                         // residual is recomputed from x0_hat
    case OP_PHASE: return phase_step_resid_bytes(op, n * c);   // padded real image + Hermitian half spectrum
    default: return align256(n * m * 4);
    }
}

static bool mask_fused_ok(const dpsx_op *op, int64_t c, int64_t h, int64_t w,
                          std::initializer_list<const void *> ptrs)
{
    if ((h * w) % 4 != 0 || (c * h * w) % 4 != 0) return false;
    if (!aligned16(op->mask)) return false;
    for (const void *p : ptrs)
        if (p && !aligned16(p)) return false;
    return true;
}

int dpsx_step_fwd_f32(dpsx_op *op, const float *x_t, const float *model_out, const float *noise, const float *y,
                      int64_t y_n, float *x0_hat, float *sample, uint8_t *inside, void *resid, float *norm,
                      int64_t n, int64_t c, int64_t h, int64_t w, const dpsx_coefs *coefs_host, void *workspace,
                      int64_t workspace_bytes, void *stream)
{
    int rc = check_geom(op, n, c, h, w);
    if (rc != DPSX_OK) return rc;
    if (!x_t || !model_out || !y || !sample || !inside || !resid || !coefs_host) return DPSX_EINVAL;
    // x0_hat is an optional OUTPUT where the launch consumes it on the fly (blur, resize): the `ps` loop reads it nowhere
    // after this call (the backward half works from the clamp gate), so a caller that does not need the image saves its
    // store (blur, resize, the hand-written spectral phase step); inpainting's backward half, the identity and the
    // library-FFT phase paths read it back and need it
    if (!x0_hat && op && op->kind != OP_SEP && op->kind != OP_TAPS && op->kind != OP_RESIZE &&
        !(op->kind == OP_PHASE && phase_is_spectral(op)))
        return DPSX_EINVAL;
    if ((coefs_host->add_noise & 1) && !noise) return DPSX_EINVAL;
    if (y_n != 1 && y_n != n) return DPSX_EINVAL;
    if (n == 0) return DPSX_OK;
    Ws ws;
    if ((rc = carve(op, workspace, workspace_bytes, n, c, h, w, ws)) != DPSX_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const Coefs k = to_coefs(coefs_host);
    const int64_t chw = c * h * w;
    StepFwdArgs a{x_t, model_out, noise, y, y_n, x0_hat, sample, inside, static_cast<float *>(resid),
                  ws.partials, n, c, h, w, k};
    int parts = (int)parts_per_particle(op, c, h, w);
    // norm != NULL: the launch itself finishes the per-particle reduction (each particle's last block re-sums its
    // partials in the order of k_finalize_norm -- bit-identical, no extra launch)
    bool tail_done = false;
    if (norm && op->d_counters && n <= kTailMaxParticles && op->kind != OP_PHASE) {
        a.tail.counters = op->d_counters;
        a.tail.partials = ws.partials;
        a.tail.parts = op->kind == OP_IDENT ? 64 : parts;
        a.tail.mode = TAIL_L2;
        a.tail.out = norm;
        a.tail.n = (int)n;
        tail_done = true;
    }
    switch (op->kind) {
    case OP_SEP:
    case OP_TAPS: rc = blur_step_fwd(op, a, s); break;
    case OP_RESIZE: rc = resize_step_fwd(op, a, s); break;
    case OP_MASK:
        // shapes that are not a multiple of 4: the host composes the step from the op-level calls
        if (!mask_fused_ok(op, c, h, w, {x_t, model_out, noise, y, x0_hat, sample}) ||
            (reinterpret_cast<uintptr_t>(inside) & 3u) != 0)
            return DPSX_EUNSUPPORTED;
        rc = mask_step_fwd(op, a, parts, s);
        break;
    case OP_IDENT:
        rc = posterior_fwd(x_t, model_out, noise, x0_hat, sample, inside, n, chw, k, s);
        if (rc != DPSX_OK) return rc;
        parts = 64;
        rc = residual_partials(y, y_n, x0_hat, static_cast<float *>(resid), ws.partials, n, chw, parts, s, 0, a.tail);
        break;
    case OP_PHASE:
        rc = phase_step_fwd(op, a, static_cast<float *>(resid), s);      // S1 + staging + R2C + residual / cotangent
        break;
    default: return DPSX_EUNSUPPORTED;
    }
    if (rc != DPSX_OK) {
        // a launch that failed part-way may leave arrival counters non-zero: clear them behind whatever did run, so that
        // the handle's next in-launch reduction does not start from a stale count
        if (tail_done)
            (void)hipMemsetAsync(op->d_counters, 0, (size_t)(1 + kTailMaxParticles) * sizeof(unsigned), s);
        return rc;
    }
    // norm == NULL: the partial sums stay in `workspace` and dpsx_step_bwd_f32 finalises them in its prologue
    return (norm && !tail_done) ? finalize_norm(ws.partials, parts, norm, n, s) : DPSX_OK;
}

int dpsx_step_bwd_f32(dpsx_op *op, const void *resid, const float *norm, float *norm_out, const uint8_t *inside,
                      const float *x0_hat, const float *y, int64_t y_n, float scale, int power,
                      float *g_model_out, int64_t n, int64_t c, int64_t h, int64_t w,
                      const dpsx_coefs *coefs_host, void *workspace, int64_t workspace_bytes, void *stream)
{
    return dpsx_step_bwd_extra_f32(op, resid, norm, norm_out, inside, x0_hat, y, y_n, scale, power, nullptr,
                                   g_model_out, n, c, h, w, coefs_host, workspace, workspace_bytes, stream);
}

int dpsx_step_bwd_extra_f32(dpsx_op *op, const void *resid, const float *norm, float *norm_out,
                            const uint8_t *inside, const float *x0_hat, const float *y, int64_t y_n, float scale,
                            int power, const float *g_x0_extra, float *g_model_out, int64_t n, int64_t c, int64_t h,
                            int64_t w, const dpsx_coefs *coefs_host, void *workspace, int64_t workspace_bytes,
                            void *stream)
{
    int rc = check_geom(op, n, c, h, w);
    if (rc != DPSX_OK) return rc;
    if (!resid || !inside || !g_model_out || !coefs_host || (power != 1 && power != 2)) return DPSX_EINVAL;
    if (!norm && !norm_out) return DPSX_EINVAL;
    if (n == 0) return DPSX_OK;
    Ws ws;
    if ((rc = carve(op, workspace, workspace_bytes, n, c, h, w, ws)) != DPSX_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const Coefs k = to_coefs(coefs_host);
    const int64_t chw = c * h * w;
    int parts = (int)parts_per_particle(op, c, h, w);
    if (op->kind == OP_IDENT || op->kind == OP_PHASE) {
        if (op->kind == OP_IDENT) parts = 64;
        // (the hand-written spectral phase step finalises the norm in its backward launch's prologue)
        if (!norm && !(op->kind == OP_PHASE && phase_norm_in_bwd(op))) {     // no fused prologue: finalise with the small kernel
            if ((rc = finalize_norm(ws.partials, parts, norm_out, n, s)) != DPSX_OK) return rc;
            norm = norm_out;
        }
    }
    StepBwdArgs b{static_cast<const float *>(resid), norm, ws.partials, parts, norm_out, inside, x0_hat, y, y_n,
                  scale, power, g_model_out, n, c, h, w, k, g_x0_extra};
    switch (op->kind) {
    case OP_SEP:
    case OP_TAPS: return blur_step_bwd(op, b, static_cast<float *>(ws.priv), ws.priv_bytes, s);
    case OP_RESIZE: return resize_step_bwd(op, b, s);
    case OP_MASK:
        if (!x0_hat || !y || (y_n != 1 && y_n != n)) return DPSX_EINVAL;
        if (!mask_fused_ok(op, c, h, w, {x0_hat, y, g_model_out}) || (reinterpret_cast<uintptr_t>(inside) & 3u))
            return DPSX_EUNSUPPORTED;
        return mask_step_bwd(op, b, s);
    case OP_IDENT:
        return clamp_scale_to_eps(static_cast<const float *>(resid), norm, inside, scale, power, g_model_out, n,
                                  chw, k, s, g_x0_extra);
    case OP_PHASE:
        if (phase_vec4_ok(op) && aligned16(g_model_out) && aligned16(g_x0_extra) &&
            (reinterpret_cast<uintptr_t>(inside) & 3u) == 0)
            return phase_step_bwd_fused(op, const_cast<float *>(static_cast<const float *>(resid)), b, s);
        if (phase_is_spectral(op)) return DPSX_EUNSUPPORTED;     // its buffers have no unaligned form
        rc = phase_step_bwd(op, const_cast<float *>(static_cast<const float *>(resid)), ws.img, n * c, s);
        if (rc != DPSX_OK) return rc;
        return clamp_scale_to_eps(ws.img, norm, inside, scale, power, g_model_out, n, chw, k, s, g_x0_extra);
    }
    return DPSX_EUNSUPPORTED;
}

int dpsx_step_update_f32(const float *sample, const float *g_model_out, const float *g_unet, float *x_next,
                         int64_t n, int64_t chw, const dpsx_coefs *coefs_host, void *stream)
{
    if (!sample || !g_model_out || !x_next || !coefs_host || n < 0 || chw < 0) return DPSX_EINVAL;
    if (coefs_host->b == 0.0f) return DPSX_EINVAL;
    return step_update(sample, g_model_out, g_unet, x_next, n, chw, to_coefs(coefs_host), (hipStream_t)stream);
}

int dpsx_update_f32(const float *sample, const float *g_a, const float *g_b, float *out, int64_t count,
                    void *stream)
{
    if (!sample || !g_a || !out || count < 0) return DPSX_EINVAL;
    return plain_update(sample, g_a, g_b, out, count, (hipStream_t)stream);
}

// ------------------------------------------------------------------ best-of-N
// Residual partials per block (one launch; two for the operators that materialise A x first), then one small launch
// that finishes costs[p] and, when asked, the combine with the previous costs and the argmin over all particles.
// the scoring launch of one contiguous range of particles (partial sums only; finalize_select finishes them)
static int score_launch(dpsx_op *op, const Ws &ws, const float *x, const float *y, int64_t y_n, int l1, int parts,
                        int64_t n, int64_t c, int64_t h, int64_t w, hipStream_t s)
{
    const int64_t chw = c * h * w;
    const Tail tail{};                     // no in-launch tail
    int rc;
    switch (op->kind) {
    case OP_SEP:
    case OP_TAPS: return blur_score(op, x, y, y_n, ws.partials, n, c, h, w, l1, tail, s);
    case OP_RESIZE: return resize_score(op, x, y, y_n, ws.partials, n, c, l1, tail, s);
    case OP_IDENT: return residual_partials(y, y_n, x, nullptr, ws.partials, n, chw, parts, s, l1, tail);
    case OP_MASK:      // y - mask * x in the reduction itself (one launch, no scratch)
        return residual_partials(y, y_n, x, nullptr, ws.partials, n, chw, parts, s, l1, tail, op->mask, h * w);
    case OP_PHASE: {
        // A x into scratch, then the generic residual reduction
        const int64_t m = meas_elems(op, c, h, w);
        float *ax = ws.meas;
        rc = phase_forward(op, x, ax, nullptr, n * c, ws.priv, ws.priv_bytes, s);
        if (rc != DPSX_OK) return rc;
        return residual_partials(y, y_n, ax, nullptr, ws.partials, n, m, parts, s, l1, tail);
    }
    default: return DPSX_EUNSUPPORTED;
    }
}

static int score_parts(const dpsx_op *op, int64_t c, int64_t h, int64_t w)
{
    if (op->kind == OP_IDENT || op->kind == OP_MASK || op->kind == OP_PHASE) return 64;
    return (int)parts_per_particle(op, c, h, w);
}

static Tail score_tail(const Ws &ws, int parts, int l1, const float *prev, int potential, float *raw_out, float *costs,
                       int64_t *best_idx, float *best_val, int64_t n, int64_t chw)
{
    // the reduction is finished by one small follow-up launch (finalize_select), not inside the scoring launch: see
    // the measurement at k_finalize_select
    Tail fin{};
    fin.partials = ws.partials;
    fin.parts = parts;
    fin.mode = l1 ? TAIL_L1SQ : TAIL_L2;
    fin.l1_scale = (float)(1.0 / (double)chw);
    fin.prev = prev;
    fin.potential = potential;
    fin.raw_out = raw_out;
    fin.out = costs;
    fin.best_idx = best_idx;
    fin.best_val = best_val;
    fin.n = (int)n;
    return fin;
}

static int score_impl(dpsx_op *op, const float *x, const float *y, int64_t y_n, int l1, const float *prev,
                      int potential, float *raw_out, float *costs, int64_t *best_idx, float *best_val, int64_t n,
                      int64_t c, int64_t h, int64_t w, void *workspace, int64_t workspace_bytes, void *stream)
{
    int rc = check_geom(op, n, c, h, w);
    if (rc != DPSX_OK) return rc;
    if (!x || !y || !costs || (y_n != 1 && y_n != n)) return DPSX_EINVAL;
    if (n == 0) return best_idx ? DPSX_EINVAL : DPSX_OK;
    Ws ws;
    if ((rc = carve(op, workspace, workspace_bytes, n, c, h, w, ws)) != DPSX_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int parts = score_parts(op, c, h, w);
    if ((rc = score_launch(op, ws, x, y, y_n, l1, parts, n, c, h, w, s)) != DPSX_OK) return rc;
    return finalize_select(score_tail(ws, parts, l1, prev, potential, raw_out, costs, best_idx, best_val, n, c * h * w), s);
}

int dpsx_search_step_f32(dpsx_op *op, const float *x_t, const float *model_out, const float *noise, const float *y,
                         int64_t y_n, float *sample, float *costs, int64_t *best_idx_dev, float *best_val_dev,
                         float *x_next, int64_t n, int64_t c, int64_t h, int64_t w, const dpsx_coefs *coefs_host,
                         void *workspace, int64_t workspace_bytes, void *stream)
{
    int rc = check_geom(op, n, c, h, w);
    if (rc != DPSX_OK) return rc;
    if (!x_t || !model_out || !y || !sample || !costs || !best_idx_dev || !coefs_host) return DPSX_EINVAL;
    if ((coefs_host->add_noise & 1) && !noise) return DPSX_EINVAL;
    if ((y_n != 1 && y_n != n) || n == 0 || x_next == sample) return DPSX_EINVAL;
    // S1 (no x0_hat store) -> scoring launch -> one launch for costs + select -> the winner's replication.
    // Measured and dropped (N = 64, Gaussian, 89.7 us for this sequence): S1 fused into the separable scoring kernel (the
    // proposal's halo needs all four input streams: 72 us for the fused launch against 37 + 30 for the two); costs +
    // select + replication in one launch (92.5 us per step); the particles in 2 / 3 / 4 chunks with the (latency-bound)
    // scoring of chunk k on a second stream beside the (bandwidth-bound) S1 of chunk k + 1, forked and joined by events:
    // 94.8 / 100.8 / 113.6 us -- the cross-stream dependencies cost more than the overlap returns.
    Ws ws;
    if ((rc = carve(op, workspace, workspace_bytes, n, c, h, w, ws)) != DPSX_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int64_t chw = c * h * w;
    const int parts = score_parts(op, c, h, w);
    rc = posterior_fwd(x_t, model_out, noise, nullptr, sample, nullptr, n, chw, to_coefs(coefs_host), s);
    if (rc != DPSX_OK) return rc;
    if ((rc = score_launch(op, ws, sample, y, y_n, 0, parts, n, c, h, w, s)) != DPSX_OK) return rc;
    rc = finalize_select(score_tail(ws, parts, 0, nullptr, POT_NONE, nullptr, costs, best_idx_dev, best_val_dev, n, chw), s);
    if (rc != DPSX_OK || !x_next) return rc;
    return gather_f32(sample, best_idx_dev, x_next, n, n, chw, true, s);
}

int dpsx_search_step_one_f32(dpsx_op *op, const float *x_t, const float *model_out, const float *noise, const float *y,
                             int64_t y_n, float *sample, float *costs, int64_t *best_idx_dev, float *best_val_dev,
                             float *x_next, int64_t n, int64_t c, int64_t h, int64_t w, const dpsx_coefs *coefs_host,
                             void *workspace, int64_t workspace_bytes, void *stream)
{
    int rc = check_geom(op, n, c, h, w);
    if (rc != DPSX_OK) return rc;
    if (!x_t || !model_out || !y || !sample || !costs || !best_idx_dev || !coefs_host) return DPSX_EINVAL;
    if ((coefs_host->add_noise & 1) && !noise) return DPSX_EINVAL;
    if ((y_n != 1 && y_n != n) || n == 0 || x_next == sample || x_next == x_t) return DPSX_EINVAL;
    // After a select every particle of SearchDDPM is a copy of the winner (img[best_path.repeat(n_paths)], :633), so the
    // loop's state is ONE particle: S1 reads that state and its model output once for all n proposals (the per-particle
    // noise makes them differ), the scoring launch and the select are the n-particle ones, and the winner is copied out
    // once instead of n times: 3P of traffic per particle-step (noise in, proposal out, proposal scored) instead of the
    // 8P of the replicated form -- and one model evaluation per step instead of n for the caller.
    Ws ws;
    if ((rc = carve(op, workspace, workspace_bytes, n, c, h, w, ws)) != DPSX_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int64_t chw = c * h * w;
    const int parts = score_parts(op, c, h, w);
    rc = posterior_fwd(x_t, model_out, noise, nullptr, sample, nullptr, n, chw, to_coefs(coefs_host), s, true);
    if (rc != DPSX_OK) return rc;
    if ((rc = score_launch(op, ws, sample, y, y_n, 0, parts, n, c, h, w, s)) != DPSX_OK) return rc;
    const Tail tail = score_tail(ws, parts, 0, nullptr, POT_NONE, nullptr, costs, best_idx_dev, best_val_dev, n, chw);
    static const bool unfused = getenv("DPSX_SEARCH_ONE_UNFUSED") != nullptr;       // A/B switch for tools/kbench_search.py
    if (!unfused && x_next && chw % 4 == 0 && aligned16(sample) && aligned16(x_next))   // costs + select + the one copy: one launch
        return finalize_select_copy(tail, sample, x_next, chw, s);
    rc = finalize_select(tail, s);
    if (rc != DPSX_OK || !x_next) return rc;
    return gather_f32(sample, best_idx_dev, x_next, 1, n, chw, true, s);
}

int dpsx_score_f32(dpsx_op *op, const float *x, const float *y, int64_t y_n, float *costs, int64_t n, int64_t c,
                   int64_t h, int64_t w, void *workspace, int64_t workspace_bytes, void *stream)
{
    return score_impl(op, x, y, y_n, 0, nullptr, POT_NONE, nullptr, costs, nullptr, nullptr, n, c, h, w, workspace,
                      workspace_bytes, stream);
}

int dpsx_score_argmin_f32(dpsx_op *op, const float *x, const float *y, int64_t y_n, float *costs,
                          int64_t *best_idx_dev, float *best_val_dev, int64_t n, int64_t c, int64_t h, int64_t w,
                          void *workspace, int64_t workspace_bytes, void *stream)
{
    if (!best_idx_dev) return DPSX_EINVAL;
    return score_impl(op, x, y, y_n, 0, nullptr, POT_NONE, nullptr, costs, best_idx_dev, best_val_dev, n, c, h, w,
                      workspace, workspace_bytes, stream);
}

int dpsx_resample_cost_f32(dpsx_op *op, const float *x, const float *y, int64_t y_n, const float *prev_costs,
                           int potential, float *curr_costs, float *net_costs, int64_t n, int64_t c, int64_t h,
                           int64_t w, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (potential < DPSX_POT_MEAN || potential > DPSX_POT_CURR) return DPSX_EINVAL;
    return score_impl(op, x, y, y_n, 1, prev_costs, potential, curr_costs, net_costs, nullptr, nullptr, n, c, h, w,
                      workspace, workspace_bytes, stream);
}

int dpsx_argmin_f32(const float *v, int64_t n, int64_t *idx_out_dev, float *val_out_dev, void *stream)
{
    if (!v || !idx_out_dev || n < 1) return DPSX_EINVAL;
    return argmin_f32(v, n, idx_out_dev, val_out_dev, (hipStream_t)stream);
}

int dpsx_gather_f32(const float *src, const int64_t *ids_dev, float *dst, int64_t n_out, int64_t n_src,
                    int64_t chw, void *stream)
{
    if (n_out < 0 || chw < 0) return DPSX_EINVAL;
    if (n_out == 0 || chw == 0) return DPSX_OK;
    if (!src || !ids_dev || !dst || n_src < 1 || src == dst) return DPSX_EINVAL;
    return gather_f32(src, ids_dev, dst, n_out, n_src, chw, false, (hipStream_t)stream);
}

int dpsx_replicate_f32(const float *src, const int64_t *idx_dev, float *dst, int64_t n_out, int64_t n_src,
                       int64_t chw, void *stream)
{
    if (n_out < 0 || chw < 0) return DPSX_EINVAL;
    if (n_out == 0 || chw == 0) return DPSX_OK;
    if (!src || !idx_dev || !dst || n_src < 1 || src == dst) return DPSX_EINVAL;
    return gather_f32(src, idx_dev, dst, n_out, n_src, chw, true, (hipStream_t)stream);
}

int dpsx_pack_champion_f32(const float *particles, const float *costs, const int64_t *best_idx_dev,
                           const float *best_val_dev, float *out, int64_t n, int64_t chw, void *stream)
{
    if (!particles || !out || n < 1 || chw < 4 || chw % 4 != 0) return DPSX_EINVAL;
    if (!best_idx_dev && !costs) return DPSX_EINVAL;
    if (best_idx_dev && !best_val_dev && !costs) return DPSX_EINVAL;
    if (!aligned16(particles) || !aligned16(out)) return DPSX_EUNSUPPORTED;
    return pack_champion(particles, costs, best_idx_dev, best_val_dev, out, n, chw, (hipStream_t)stream);
}

int dpsx_select_champion_f32(const float *table, int64_t world, int64_t chw, float *dst, int64_t n_out,
                             int64_t *win_rank_dev, int64_t *win_local_dev, void *stream)
{
    if (!table || !dst || world < 1 || world > (1 << 20) || chw < 4 || chw % 4 != 0 || n_out < 1) return DPSX_EINVAL;
    if (!aligned16(table) || !aligned16(dst)) return DPSX_EUNSUPPORTED;
    return select_champion(table, (int)world, chw, dst, n_out, win_rank_dev, win_local_dev, (hipStream_t)stream);
}

}  // extern "C"
#pragma GCC visibility pop
