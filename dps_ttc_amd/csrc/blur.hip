// Blur operator A = ReflectionPad2d(R) + depthwise cross-correlation, and its
// exact adjoint A^T = reflection-fold o full-correlation-transpose
// (reference: util/img_utils.py:268-283, measurements.py:93-149).
//
// One workgroup (256 threads = 4 waves) owns a 64x64 output tile of one
// (particle, channel) plane:
//   stage A  HBM -> LDS: the tile plus a halo of R4 (= R rounded up to 4) as
//            16-byte loads; in the fused DPS step the loader computes x0_hat
//            from x_t/eps on the fly (S1), and writes x0_hat / sample / clamp
//            gate for the tile interior, so x0_hat never makes a round trip;
//   separable kernels (Gaussian): horizontal pass LDS->LDS, 8 outputs per
//            thread from a register sliding window (taps live in SGPRs),
//            then vertical pass LDS->registers, 4x4 outputs per thread;
//   generic kernels (motion): list of non-zero taps, 16 outputs per thread;
//   epilogue residual y - A x0 (+ per-tile sum of squares, deterministic), or for the
//            adjoint the clamp gate and the -b * coef scaling into g_model_out.
// Blocks are numbered so that the tiles of one plane share blockIdx % 8, i.e.
// one XCD and its L2 (halo re-reads then hit L2, not HBM).
#include <cstdlib>

#include "common.h"

namespace dpsx {

constexpr int TH = 64, TW = 64, NT = 256;
constexpr size_t kScratchBytes = 1024;     // reduction scratch behind every kernel's LDS image

struct BlurArgs {
    // plain input / output
    const float *x;       // [planes, h, w]            (fwd !POST, adjoint input u)
    float *out;           // fwd: A x or residual r (may be null when RESID); adj !EPI: g
    // fused S1 prologue (POST)
    const float *x_t, *model_out, *noise;
    float *x0_hat, *sample;
    uint8_t *inside_w;
    // residual epilogue (RESID)
    const float *y;
    int y_n;
    float *partials;      // [planes * tiles]
    // adjoint epilogue (EPI)
    const float *norm_in; // [n] finalized norms, or nullptr: derive from norm_partials (fused bwd half)
    const float *norm_partials;
    int norm_parts;
    float *norm_out;
    const uint8_t *inside_r;
    const float *g_extra;  // optional extra cotangent on x0_hat, added before the clamp gate (nullable)
    float *g_model_out;
    float scale;
    int power;
    // geometry
    int c, h, w, tiles_x, tiles_y, planes;
    Coefs k;
    // generic taps
    // vertical runs of taps (forward table, or the adjoint's negated / reversed one), grouped by class:
    // [even dx, 4 taps | even dx, 2 taps | odd dx, 4 taps | odd dx, 2 taps]; nrun[k] = runs in class k
    const TapRun *runs;
    int nrun[4];
    int nhlo[4], nhhi[4];  // one-launch adjoint: runs [nhlo, nhhi) of class k can reach this tile's mirrored-column strip
    int dbg;   // phase-ablation mask: 0 unless built with -DDPSX_ABLATION=1 (see blur_sep.h)
    // zero-extended loads (adjoint): the source plane is src_h x src_w and sits at (src_off, src_off)
    // inside the h x w domain the kernel tiles (src_off = 0 and src = domain for everything else)
    int src_h, src_w, src_off;
    // scoring variants of the residual epilogue (plain input only): sums of |r| instead of r^2; in-launch finalisation
    int l1;
    Tail tail;
};

__device__ __forceinline__ bool block_to_tile(const BlurArgs &a, int &plane, int &ty, int &tx)
{
    const int tiles = a.tiles_x * a.tiles_y;
    const int b = blockIdx.x;
    const int xcd = b & 7, k = b >> 3;
    plane = (k / tiles) * 8 + xcd;
    const int t = k % tiles;
    ty = t / a.tiles_x;
    tx = t % a.tiles_x;
    return plane < a.planes;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---------------------------------------------------------------- stage A
// Fill s_in[RH][SW] with the region whose top-left image coordinate is (gy0, gx0).
// REFLECT: forward operator (reflection padding); else zero extension (adjoint).
template <bool POST, bool REFLECT, bool VEC>
__device__ __forceinline__ void load_region(float *s_in, const int SW, const int RH, const int RW,
                                            const int gy0, const int gx0, const int h0, const int w0,
                                            const BlurArgs &a, const int plane, const int tid = threadIdx.x)
{
    const int h = REFLECT ? a.h : a.src_h, w = REFLECT ? a.w : a.src_w, off = REFLECT ? 0 : a.src_off;
    const int64_t hw = (int64_t)h * w;
    const int n = plane / a.c, ch = plane % a.c;
    const float *src = nullptr, *eps = nullptr, *vv = nullptr, *zz = nullptr;
    if constexpr (POST) {
        src = a.x_t + (int64_t)plane * hw;
        eps = a.model_out + ((int64_t)n * 2 * a.c + ch) * hw;
        vv = eps + (int64_t)a.c * hw;
        zz = a.noise + (int64_t)plane * hw;
    } else {
        src = a.x + (int64_t)plane * hw;
    }
    constexpr int U = VEC ? 4 : 1;
    const int RWu = RW / U;
    for (int u = tid; u < RH * RWu; u += NT) {
        const int rr = u / RWu, cu = u - rr * RWu;
        const int gy = gy0 + rr - off, gx = gx0 + cu * U - off;     // coordinates in the source plane
        float val[U];
        float ev[U];
        bool rowok = true;
        int sy = gy;
        if constexpr (REFLECT) sy = clampi(reflect_idx(gy, h), 0, h - 1);
        else rowok = gy >= 0 && gy < h;
        const bool fast = VEC && gx >= 0 && gx + U - 1 < w;
        if (!rowok) {
#pragma unroll
            for (int e = 0; e < U; ++e) val[e] = 0.0f;
        } else if (fast) {
            if constexpr (VEC) {
                const float4 t = *reinterpret_cast<const float4 *>(src + (int64_t)sy * w + gx);
                val[0] = t.x; val[1] = t.y; val[2] = t.z; val[3] = t.w;
                if constexpr (POST) {
                    const float4 q = *reinterpret_cast<const float4 *>(eps + (int64_t)sy * w + gx);
                    ev[0] = q.x; ev[1] = q.y; ev[2] = q.z; ev[3] = q.w;
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < U; ++e) {
                int sx = gx + e;
                bool ok = true;
                if constexpr (REFLECT) sx = clampi(reflect_idx(sx, w), 0, w - 1);
                else ok = sx >= 0 && sx < w;
                val[e] = ok ? src[(int64_t)sy * w + sx] : 0.0f;
                if constexpr (POST) ev[e] = eps[(int64_t)sy * w + sx];
            }
        }
        if constexpr (POST) {
            bool ins[U];
            float xin[U];
#pragma unroll
            for (int e = 0; e < U; ++e) {
                xin[e] = val[e];
                val[e] = post_x0(xin[e], ev[e], a.k, ins[e]);
            }
            // tile interior: emit x0_hat, sample and the clamp gate
            const bool interior = gy >= h0 && gy < h0 + TH && gy < h && gx >= w0 && gx < w0 + TW && gx < w;
            if (interior) {
                const int64_t o = (int64_t)gy * w + gx;
                float sm[U];
                if (fast || !VEC) {
                    float vq[U], zq[U];
                    if constexpr (VEC) {
                        float4 t4 = make_float4(0, 0, 0, 0), z4 = t4;
                        if (a.k.add_noise & 1) {
                            t4 = *reinterpret_cast<const float4 *>(vv + o);
                            z4 = *reinterpret_cast<const float4 *>(zz + o);
                        }
                        vq[0] = t4.x; vq[1] = t4.y; vq[2] = t4.z; vq[3] = t4.w;
                        zq[0] = z4.x; zq[1] = z4.y; zq[2] = z4.z; zq[3] = z4.w;
                    } else {
                        vq[0] = (a.k.add_noise & 1) ? vv[o] : 0.0f;
                        zq[0] = (a.k.add_noise & 1) ? zz[o] : 0.0f;
                    }
#pragma unroll
                    for (int e = 0; e < U; ++e) sm[e] = post_sample(xin[e], val[e], vq[e], zq[e], a.k);
                    float *x0p = a.x0_hat + (int64_t)plane * hw + o;
                    float *smp = a.sample + (int64_t)plane * hw + o;
                    uint8_t *inp = a.inside_w + (int64_t)plane * hw + o;
                    if constexpr (VEC) {
                        if (a.x0_hat) *reinterpret_cast<float4 *>(x0p) = make_float4(val[0], val[1], val[2], val[3]);
                        *reinterpret_cast<float4 *>(smp) = make_float4(sm[0], sm[1], sm[2], sm[3]);
                        *reinterpret_cast<uchar4 *>(inp) = make_uchar4(ins[0], ins[1], ins[2], ins[3]);
                    } else {
                        if (a.x0_hat) *x0p = val[0];
                        *smp = sm[0];
                        *inp = ins[0];
                    }
                }
            }
        }
        float *dst = s_in + rr * SW + cu * U;
        if constexpr (VEC) *reinterpret_cast<float4 *>(dst) = make_float4(val[0], val[1], val[2], val[3]);
        else dst[0] = val[0];
    }
}

// ---------------------------------------------------------------- epilogues
// residual of up to 4 consecutive columns of one output row; returns their sum of squares
template <bool VEC4>
__device__ __forceinline__ float resid_epilogue(const BlurArgs &a, int plane, int oy, int ox, const float *acc4,
                                                int count = 4)
{
    const int64_t hw = (int64_t)a.h * a.w;
    float ss = 0.0f;
    if (oy >= a.h) return 0.0f;
    const int n = plane / a.c, ch = plane % a.c;
    const float *yp = a.y + ((int64_t)(a.y_n == 1 ? 0 : n) * a.c + ch) * hw + (int64_t)oy * a.w;
    float *rp = a.out ? a.out + (int64_t)plane * hw + (int64_t)oy * a.w : nullptr;
    if (VEC4 && count == 4 && ox + 3 < a.w) {
        const float4 yv = *reinterpret_cast<const float4 *>(yp + ox);
        float4 r;
        r.x = yv.x - acc4[0]; r.y = yv.y - acc4[1]; r.z = yv.z - acc4[2]; r.w = yv.w - acc4[3];
        if (rp) *reinterpret_cast<float4 *>(rp + ox) = r;
        ss = a.l1 ? fabsf(r.x) + fabsf(r.y) + fabsf(r.z) + fabsf(r.w) : r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w;
    } else {
        for (int e = 0; e < count; ++e)
            if (ox + e < a.w) {
                const float r = yp[ox + e] - acc4[e];
                if (rp) rp[ox + e] = r;
                ss += a.l1 ? fabsf(r) : r * r;
            }
    }
    return ss;
}

__device__ __forceinline__ float norm_coef_dev(float nv, float gn, int power)
{
    return power == 2 ? -2.0f * gn : (nv == 0.0f ? 0.0f : -gn / nv);
}

// plain output (epi == false) or clamp gate + -b*coef scaling into g_model_out[:, :c]
template <bool VEC4>
__device__ __forceinline__ void out_epilogue(const BlurArgs &a, int plane, int oy, int ox, const float *acc4,
                                             float coef, bool epi, int count = 4)
{
    if (oy >= a.h) return;
    const int64_t hw = (int64_t)a.h * a.w;
    const int64_t o = (int64_t)oy * a.w + ox;
    if (!epi) {
        float *gp = a.out + (int64_t)plane * hw + o;
        if (VEC4 && count == 4 && ox + 3 < a.w)
            *reinterpret_cast<float4 *>(gp) = make_float4(acc4[0], acc4[1], acc4[2], acc4[3]);
        else
            for (int e = 0; e < count; ++e)
                if (ox + e < a.w) gp[e] = acc4[e];
        return;
    }
    const int n = plane / a.c, ch = plane % a.c;
    const uint8_t *ip = a.inside_r + (int64_t)plane * hw + o;
    float *gp = a.g_model_out + ((int64_t)n * 2 * a.c + ch) * hw + o;
    const float mb = -a.k.b;
    const float *ep = a.g_extra ? a.g_extra + (int64_t)plane * hw + o : nullptr;
    if (VEC4 && count == 4 && ox + 3 < a.w) {
        const uchar4 in = *reinterpret_cast<const uchar4 *>(ip);
        float4 ex = make_float4(0, 0, 0, 0);
        if (ep) ex = *reinterpret_cast<const float4 *>(ep);
        float4 g;
        g.x = in.x ? mb * (coef * acc4[0] + ex.x) : 0.0f;
        g.y = in.y ? mb * (coef * acc4[1] + ex.y) : 0.0f;
        g.z = in.z ? mb * (coef * acc4[2] + ex.z) : 0.0f;
        g.w = in.w ? mb * (coef * acc4[3] + ex.w) : 0.0f;
        *reinterpret_cast<float4 *>(gp) = g;
    } else {
        for (int e = 0; e < count; ++e)
            if (ox + e < a.w) gp[e] = ip[e] ? mb * (coef * acc4[e] + (ep ? ep[e] : 0.0f)) : 0.0f;
    }
}

#include "blur_sep.h"

// =====================================================================
// Generic path: list of non-zero taps (dy, dx in [-R, R], weight) grouped into vertical runs, read through
// wave-uniform scalar loads.  thread -> columns 2 (tid % 32), +1; rows 8 (tid / 32) .. +7:
// the 32 lanes of a half-wave read 256 consecutive LDS bytes (conflict-free ds_read_b64).
// LDS: s_in[RH][SW] | 256 floats of scratch
// =====================================================================
// halo of the staged region on each side of the 64 x 64 tile (columns in multiples of 4)
struct TapGeom {
    int t, b, l, r;
    // halo-unit enumeration without integer division (the halo is a run-time quantity here): magic = ceil(2^32 / d)
    unsigned rwu_magic, side_magic;     // d = region units per row;  d = units per row of the left + right bands
};

__device__ __forceinline__ unsigned fastdiv(unsigned x, unsigned magic, unsigned d)
{
    return d == 1 ? x : __umulhi(x, magic);       // exact for x < 2^16, d <= 128 (checked exhaustively on the host side)
}

// halo unit hu in [0, H) of a staged region (TH + t + b rows, RWu = (TW + l + r) / 4 units per row) -> (region row,
// unit column).  [0, (t + b) RWu): the t rows above and the b rows below the tile; the rest: TH rows x (l/4 + r/4) units.
__device__ __forceinline__ void taps_halo_unit(const TapGeom &g, const int RWu, int hu, int &rr, int &cu)
{
    const int band = (g.t + g.b) * RWu;
    if (hu < band) {
        const int row = (int)fastdiv((unsigned)hu, g.rwu_magic, (unsigned)RWu);
        cu = hu - row * RWu;
        rr = row < g.t ? row : row + TH;
    } else {
        const int q = (g.l + g.r) >> 2, v = hu - band;
        const int row = (int)fastdiv((unsigned)v, g.side_magic, (unsigned)q);
        const int c = v - row * q;
        rr = row + g.t;
        cu = c < (g.l >> 2) ? c : c + TW / 4;
    }
}

// Loads-first loader of the tap-list kernels for REGULAR geometry (the recipe of the separable kernel's loader): every
// lane issues ALL its loads -- 4 interior float4 units x up to 4 streams, then up to NHB halo units x up to 2 streams
// -- before anything is consumed: one memory round trip per tile instead of one per 1024 units (round 1: 2-4 serialised
// round trips per tile, with four co-resident blocks in lockstep).  Halo units beyond NHB x 256 (kernels whose path
// reaches far on several sides) follow in rounds of four.
template <bool POST, bool REFLECT, int NHB>
__device__ __forceinline__ void load_region_taps_first(float *s_in, const int SW, const int RWu, const TapGeom g,
                                                       const int h0, const int w0, const BlurArgs &a, const int plane)
{
    const int h = REFLECT ? a.h : a.src_h, w = REFLECT ? a.w : a.src_w, off = REFLECT ? 0 : a.src_off;
    const unsigned hw = (unsigned)(h * w);
    const int n = plane / a.c, ch = plane % a.c;
    const float *src, *eps = nullptr, *vv = nullptr, *zz = nullptr;
    if constexpr (POST) {
        src = a.x_t + (int64_t)plane * hw;
        eps = a.model_out + ((int64_t)n * 2 * a.c + ch) * hw;
        vv = eps + (int64_t)a.c * hw;
        zz = a.noise + (int64_t)plane * hw;
    } else {
        src = a.x + (int64_t)plane * hw;
    }
    constexpr int NI = TH * TW / 4 / NT;            // interior units per thread (4)
    const int H = (g.t + g.b) * RWu + TH * ((g.l + g.r) >> 2);
    const bool noisy = POST && (a.k.add_noise & 1);
    float4 xi[NI], ei[NI], vi[NI], zi[NI], xh[NHB], eh[NHB];
    // streams with reuse (x, eps: a neighbour's halo is this tile's interior) first, grouped per stream; once-read
    // streams (v, noise) last and non-temporal
#pragma unroll
    for (int pass = 0; pass < (POST ? 2 : 1); ++pass) {
        const float *p = pass ? eps : src;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int u = threadIdx.x + k * NT, row = u >> 4, cu = u & 15;
            float4 v;
            // forward operator on regular geometry: an interior unit lies inside the image -- one plain 16-byte load, none of
            // the reflection selects of the halo path (they were a tenth of this loader's vector instructions)
            if constexpr (REFLECT) v = *reinterpret_cast<const float4 *>(p + (unsigned)((h0 + row) * w + w0 + 4 * cu));
            else v = load_unit_reg<false>(p, h0 + row - off, w0 + 4 * cu - off, h, w);
            if (pass) ei[k] = v; else xi[k] = v;
        }
#pragma unroll
        for (int k = 0; k < NHB; ++k) {
            const int hu = min((int)threadIdx.x + k * NT, H - 1);        // surplus lanes repeat the last unit
            int rr, cu;
            taps_halo_unit(g, RWu, max(hu, 0), rr, cu);
            const float4 v = load_unit_reg<REFLECT || POST>(p, h0 - g.t + rr - off, w0 - g.l + 4 * cu - off, h, w);
            if (pass) eh[k] = v; else xh[k] = v;
        }
    }
    if constexpr (POST) {
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int u = threadIdx.x + k * NT, row = u >> 4, cu = u & 15;
            const unsigned o = (unsigned)((h0 + row) * w + w0 + 4 * cu);
            vi[k] = ld_stream(noisy ? vv + o : g_zero_unit, true);
            zi[k] = ld_stream(noisy ? zz + o : g_zero_unit, true);
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // every load above issues before the first use below
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int u = threadIdx.x + k * NT, row = u >> 4, cu = u & 15;
        float4 val = xi[k];
        if constexpr (POST) {
            float4 x0, sm;
            uchar4 gate;
            post_unit(xi[k], ei[k], vi[k], zi[k], a.k, x0, sm, gate);       // packed S1 (common.h)
            const int64_t o = (int64_t)plane * hw + (unsigned)((h0 + row) * w + w0 + 4 * cu);
            if (a.x0_hat) *reinterpret_cast<float4 *>(a.x0_hat + o) = x0;
            *reinterpret_cast<float4 *>(a.sample + o) = sm;
            *reinterpret_cast<uchar4 *>(a.inside_w + o) = gate;
            val = x0;
        }
        *reinterpret_cast<float4 *>(s_in + (g.t + row) * SW + g.l + 4 * cu) = val;
    }
#pragma unroll
    for (int k = 0; k < NHB; ++k) {
        const int hu = threadIdx.x + k * NT;
        if (hu >= H) continue;
        int rr, cu;
        taps_halo_unit(g, RWu, hu, rr, cu);
        float4 val = xh[k];
        if constexpr (POST) val = post_x0_unit(xh[k], eh[k], a.k);
        *reinterpret_cast<float4 *>(s_in + rr * SW + 4 * cu) = val;
    }
    // far-reaching kernels: the remaining halo units, four per lane in flight
    for (int base = NHB * NT + threadIdx.x; base < H; base += 4 * NT) {
        float4 xv[4], ev[4];
        int rr[4], cu[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            taps_halo_unit(g, RWu, min(base + k * NT, H - 1), rr[k], cu[k]);
            xv[k] = load_unit_reg<REFLECT || POST>(src, h0 - g.t + rr[k] - off, w0 - g.l + 4 * cu[k] - off, h, w);
            if constexpr (POST) ev[k] = load_unit_reg<true>(eps, h0 - g.t + rr[k], w0 - g.l + 4 * cu[k], h, w);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (base + k * NT >= H) break;
            float4 val = xv[k];
            if constexpr (POST) val = post_x0_unit(xv[k], ev[k], a.k);
            *reinterpret_cast<float4 *>(s_in + rr[k] * SW + 4 * cu[k]) = val;
        }
    }
}

// ---------------------------------------------------------------- tap loop
// A lane owns TWO adjacent columns x PRW consecutive rows of the tile.  One run = L vertically consecutive taps of one
// kernel column: the lane reads its (PRW + L - 1) x 2 window once and issues PRW * L packed FMAs on it (pair of columns
// x broadcast weight: v_pk_fma_f32 keeps a SIMD's FMA pipe full from ONE wave, which is what lets half of a
// workgroup's waves do nothing but load -- below).  Window reads: 8 bytes per lane as `ds_read_b64` when the column
// offset is even (256 B per LDS clock, twice the rate of the dword reads the round-1 loop sat on: 19 `ds_read_b32` per
// 64 FMAs were 38 LDS cycles per wave-run against 32 VALU cycles per CU), as `ds_read2_b32` when it is odd (an 8-byte
// LDS read must be 8-byte aligned).  SWC > 0: compile-time row stride, so every read of a run is one base register plus
// an immediate offset (one address computation per run instead of one per row -- 11 v_add per 32 v_pk_fma otherwise).
// Run lengths 4 and 2: a mostly horizontal motion path has two vertically adjacent non-zero taps per kernel column
// (bilinear splat); padding those to four wasted half of the FMAs.
constexpr int PRW = 8;    // rows per lane (x 2 columns = 16 outputs)

struct RunClip {          // zero-extended sources (adjoint): the source plane's extent in tile coordinates, for run skipping
    int row_lo, row_hi, col_lo, col_hi;      // source rows / columns [lo, hi) relative to the wave's first row / tile column 0
};

// Run records are read-only for the whole launch: fetched through the constant address space so that the wave-uniform
// index turns into scalar loads (s_load_dwordx8 into SGPRs; weights then feed v_pk_fma_f32 as scalar operands).  Through
// the generic pointer the compiler must assume the kernel's own stores may alias the table and falls back to vector
// loads plus per-lane selects -- 2.5x the VALU instructions of the FMAs themselves (measured: SQ_INSTS_VALU).
// Measured and dropped in round 2 (plain forward, N = 64, 50.5 us with this loop): two runs per iteration behind one
// wait (52.4), the table in LDS with a software-pipelined loop (next window's reads issued before this run's FMAs: 57.2),
// the table in VGPRs broadcast with v_readlane_b32 (61-83: the selects end up in VGPRs), and a pipeline around ONE explicit
// s_waitcnt lgkmcnt(0) per run placed BEFORE the next window's reads are issued, so that the FMAs run on data that has
// already arrived (55.5; the ISA had exactly the intended wait / read / FMA order).  An ablation build shows the record
// fetches themselves cost 1.4 us.  By the counters the launch is bound by no single pipe: VALU ~25 us, LDS ~16 us and the
// 20 us load / store phases of a tile add up with little overlap (occupancy probe: T = 30 + 78 / n us for n workgroups
// per CU, n = 5 here).
typedef const __attribute__((address_space(4))) TapRun *ConstRuns;
__device__ __forceinline__ TapRun load_run(const TapRun *runs, int i)
{
    const ConstRuns c = (ConstRuns)(runs);
    TapRun r;
    r.dy0 = c[i].dy0; r.dx = c[i].dx; r.pad0 = 0; r.pad1 = 0;
    r.w[0] = c[i].w[0]; r.w[1] = c[i].w[1]; r.w[2] = c[i].w[2]; r.w[3] = c[i].w[3];
    return r;
}

template <bool CLIP, int L>
__device__ __forceinline__ bool run_live(const TapRun &r, const RunClip &clip)
{
    if constexpr (!CLIP) return true;
    // wave-uniform: the window rows [dy0, dy0 + 15 + L - 1] / columns [dx, dx + 63] of this wave miss the source
    // plane entirely (a ring tile of the padded domain): every product would be with a staged zero
    return r.dy0 + 2 * PRW + L - 2 >= clip.row_lo && r.dy0 < clip.row_hi && r.dx + TW - 1 >= clip.col_lo &&
           r.dx < clip.col_hi;
}

// One run on the lane's window: (PRW + L - 1) 8-byte reads, PRW * L packed FMAs -- for BOTH parities of dx.  An 8-byte
// LDS read must be 8-byte aligned, so a lane cannot fetch the inputs (2cp + dx, 2cp + dx + 1) of its column pair when dx
// is odd.  Round 2's first form read the two aligned words that straddle the pair and used scalar `v_fma_f32` on the inner
// halves: twice the reads and -- the SQ counters showed it (SQ_INSTS_VALU = 2.0x the packed-FMA count of the tap list,
// VALU busy 60 % of the launch) -- twice the FMA-pipe time, a wave-wide v_fma_f32 costs the same four cycles as a
// v_pk_fma_f32.  Now odd-dx runs accumulate the SHIFTED output pair (2cp - 1, 2cp), whose inputs (2cp - 1 + dx, 2cp + dx)
// are aligned, into a second accumulator set `aco`; the two sets are merged once per tile (merge_shifted: the lane's own
// .y plus its right neighbour's .x, one lane shift).  The shifted pairs cover columns -1 .. 62, so column 63's odd-dx part
// is accumulated separately: lane l of a wave keeps `c63` for the wave's row (l & 15), L dword reads + L FMAs per run.
// The reads are volatile so that the backend does not fuse two of them into one `ds_read2_b64`, which moves 128 B per LDS
// clock where two `ds_read_b64` move 256.
typedef const volatile __attribute__((address_space(3))) v2f *LdsPair;     // explicit LDS pointer: volatile must not demote the read to a flat load

// p: the lane's window base for this run (lowest row; the even column the 8-byte reads start at).
// REV: the lane's rows are visited in descending order (mirrored rows of the one-launch adjoint): acc[PRW-1-i] pairs with
// window row i + q.  SWAP: the pair is mirrored horizontally (low input column feeds the high output column).
template <int L, int SWC, bool REV, bool SWAP, int PR = PRW>
__device__ __forceinline__ void run_apply(v2f (&acc)[PR], const float *p, const int SW, const TapRun &r)
{
    v2f wa[PR + L - 1];
#pragma unroll
    for (int m = 0; m < PR + L - 1; ++m) wa[m] = *(LdsPair)(p + m * SW);
#pragma unroll
    for (int q = 0; q < L; ++q) {
        const v2f w2 = v2f{r.w[q], r.w[q]};
#pragma unroll
        for (int i = 0; i < PR; ++i) {
            const v2f in = SWAP ? v2f{wa[i + q].y, wa[i + q].x} : wa[i + q];      // op_sel of v_pk_fma_f32, no move
            v2f &dst = acc[REV ? PR - 1 - i : i];
            dst = __builtin_elementwise_fma(w2, in, dst);                           // v_pk_fma_f32
        }
    }
}

// the column the shifted pairs leave out (63): one row per lane, 16 rows per wave.  All 16 lanes read the SAME column, so
// with a row stride that is a multiple of 32 words every read would be a 16-way bank conflict (measured: the plain
// forward went from 58 to 72 us); the compile-time strides are 100 / 132 words and the read is the aligned 8-byte pair
// (63 + dx is even): 16 rows x 100 words land on 16 distinct bank pairs of the 64-bank b64 map.
template <int L>
__device__ __forceinline__ void run_apply_col(float &c, const float *p, const int SW, const TapRun &r)
{
    float in[L];
#pragma unroll
    for (int q = 0; q < L; ++q) in[q] = (*(LdsPair)(p + q * SW)).x;
#pragma unroll
    for (int q = 0; q < L; ++q) c = fmaf(r.w[q], in[q], c);
}

// per-lane accumulators of one pass over the tap table
template <int PR>
struct TapAccT {
    v2f ac[PR];       // columns (2cp, 2cp + 1), rows r0 .. r0 + PR - 1
    v2f aco[PR];      // columns (2cp - 1, 2cp): the runs whose inputs for (2cp, 2cp + 1) would be unaligned
    float c63;        // the left-out column of `aco`, row (lane & 15) of the wave
};
typedef TapAccT<PRW> TapAcc;

template <int PR>
__device__ __forceinline__ void tapacc_zero(TapAccT<PR> &t)
{
#pragma unroll
    for (int i = 0; i < PR; ++i) t.ac[i] = t.aco[i] = v2f{0.0f, 0.0f};
    t.c63 = 0.0f;
}

// ac += the shifted set: out[2cp] += aco.y, out[2cp + 1] += aco.x of the lane to the right (column 63: c63 of the row)
__device__ __forceinline__ void merge_shifted(TapAcc &t)
{
    const int lane = threadIdx.x & 63, cp = lane & 31, half = lane >> 5;
#pragma unroll
    for (int i = 0; i < PRW; ++i) {
        const float nb = __shfl_down(t.aco[i].x, 1, kWave);
        const float cc = __shfl(t.c63, half * PRW + i, kWave);
        t.ac[i].x += t.aco[i].y;
        t.ac[i].y += cp == 31 ? cc : nb;
    }
}

// The runs of one class.  Latency is hidden by the four or more waves that share a SIMD (one workgroup per tile, four
// per CU), not by pipelining inside a wave: a hand-pipelined form (next run's window loaded before this run's FMAs, two
// register sets) cost 44 register moves per pair of runs and was slower.  The next record's scalar loads are issued a
// run ahead.
// The runs of one class.  Latency is hidden by the four or more waves that share a SIMD (one workgroup per tile, four
// per CU), not by pipelining inside a wave (see load_run).  The next record's scalar loads are issued a run ahead.
template <int L, bool ODD, int SWC, bool CLIP>
__device__ __forceinline__ void tap_runs_pk(TapAcc &t, const float *base, const float *base63, const int sw_rt,
                                            const TapRun *runs, const int nruns, const RunClip clip)
{
    const int SW = SWC > 0 ? SWC : sw_rt;
    if (nruns <= 0) return;
    TapRun r = load_run(runs, 0);
    for (int k = 0; k < nruns; ++k) {
        const TapRun nxt = load_run(runs, min(k + 1, nruns - 1));
        if (run_live<CLIP, L>(r, clip)) {
            const int o = r.dy0 * SW + r.dx;
            if constexpr (ODD) {
                run_apply<L, SWC, false, false>(t.aco, base + o - 1, SW, r);
                run_apply_col<L>(t.c63, base63 + o, SW, r);
            } else {
                run_apply<L, SWC, false, false>(t.ac, base + o, SW, r);
            }
        }
        r = nxt;
    }
}

// all runs of a table on the staged image s_in (row stride SW, even): acc[i] = columns (2cp, 2cp+1) of row r0 + i.
// org: LDS word of (row r0, column 2cp) of the tile; org63: of (the wave's row lane & 15, column 63)
template <int SWC, bool CLIP>
__device__ __forceinline__ void tap_all_runs(v2f (&acc)[PRW], const float *s_in, const int SW, const int org,
                                             const int org63, const BlurArgs &a, const RunClip clip)
{
    TapAcc t;
    tapacc_zero(t);
    const TapRun *r = a.runs;
    tap_runs_pk<4, false, SWC, CLIP>(t, s_in + org, s_in + org63, SW, r, a.nrun[0], clip);
    r += a.nrun[0];
    tap_runs_pk<2, false, SWC, CLIP>(t, s_in + org, s_in + org63, SW, r, a.nrun[1], clip);
    r += a.nrun[1];
    tap_runs_pk<4, true, SWC, CLIP>(t, s_in + org, s_in + org63, SW, r, a.nrun[2], clip);
    r += a.nrun[2];
    tap_runs_pk<2, true, SWC, CLIP>(t, s_in + org, s_in + org63, SW, r, a.nrun[3], clip);
    if (a.nrun[2] + a.nrun[3] > 0) merge_shifted(t);          // launch-uniform
#pragma unroll
    for (int i = 0; i < PRW; ++i) acc[i] = t.ac[i];
}

// ---------------------------------------------------------------- kernels
// One workgroup (4 waves) per 64 x 64 tile, four or more resident per CU: loads-first stage, tap loop, epilogue.
// MODE 0: plain A x -> out;  1: residual y - A x (+ out if non-null) and the per-tile sum (score / fused forward);
// MODE 2: correlation-transpose of the zero-extended cotangent on the padded domain -> out (adjoint, first launch).
// Measured and dropped (r02): a persistent wave-specialised form (4 loader + 4 compute waves per workgroup, two LDS
// images, one barrier per tile) -- 144 us for the fused forward against 122 for this structure before the loads-first
// stage: with half the waves loading, two compute waves per SIMD do not cover the tap loop's LDS and scalar-load
// latencies (LDS reads and scalar loads share one counter, so a run record's wait drains the window reads too).
// NHB: halo units per lane the loads-first stage keeps in flight (2: a halo of at most 512 units -- the compact kernels, and
// 16 fewer live registers, which is what keeps the fused forward's 24 loads free of a spill in their midst; else 4)
template <bool POST, int MODE, bool VEC, int SWC, int NHB = 4>
__global__ __launch_bounds__(NT, 4) void k_blur_taps(BlurArgs a, TapGeom g)
{
    constexpr bool RESID = MODE == 1, REFLECT = MODE != 2;
    const int RH = TH + g.t + g.b, RW = TW + g.l + g.r, SW = SWC > 0 ? SWC : RW;
    extern __shared__ __align__(16) float lds[];
    float *s_in = lds, *s_red = lds + RH * SW;
    int plane, ty, tx;
    if (!block_to_tile(a, plane, ty, tx)) return;
    const int h0 = ty * TH, w0 = tx * TW;
    bool regular;
    if constexpr (REFLECT) regular = VEC && a.h % TH == 0 && a.w % TW == 0 && max(g.t, g.b) < a.h && max(g.l, g.r) < a.w;
    else regular = VEC && a.src_w % 4 == 0 && a.src_off % 4 == 0;   // the SOURCE decides whether a unit is wholly in or out
    if constexpr (VEC) {
        if (regular) load_region_taps_first<POST, REFLECT, NHB>(s_in, SW, RW / 4, g, h0, w0, a, plane);
        else load_region<POST, REFLECT, VEC>(s_in, SW, RH, RW, h0 - g.t, w0 - g.l, h0, w0, a, plane);
    } else {
        load_region<POST, REFLECT, VEC>(s_in, SW, RH, RW, h0 - g.t, w0 - g.l, h0, w0, a, plane);
    }
    __syncthreads();
    const int tid = threadIdx.x;
    const int cp = tid & 31, r0 = (tid >> 5) * PRW;       // columns 2cp, 2cp+1; rows r0 .. r0 + 7
    RunClip clip{0, 0, 0, 0};
    if constexpr (!REFLECT) {
        const int wr0 = h0 + ((tid >> 6) * 2 * PRW);      // first row of this WAVE, in domain coordinates
        clip = RunClip{a.src_off - wr0, a.src_off + a.src_h - wr0, a.src_off - w0, a.src_off + a.src_w - w0};
    }
    v2f acc[PRW];
    const int row63 = (tid >> 6) * 2 * PRW + (tid & 15);       // the shifted pairs' left-out column: one row per lane
    if (ABL(16)) {      // ablation: no tap loop at all (what the load / store phases of the launch cost by themselves)
#pragma unroll
        for (int i = 0; i < PRW; ++i) acc[i] = *(LdsPair)(s_in + (r0 + g.t + i) * SW + 2 * cp + g.l);
    } else
    tap_all_runs<SWC, !REFLECT>(acc, s_in, SW, (r0 + g.t) * SW + 2 * cp + g.l, (row63 + g.t) * SW + 63 + g.l, a, clip);
    const int ox = w0 + 2 * cp;
    const bool full = VEC && a.h % TH == 0 && a.w % TW == 0;
    float ss = 0.0f;
    if (RESID && full) {
        // full tile: the lane's 8 x 2 measurement values are fetched together (one wait), then r = y - A(x0_hat)
        const unsigned hw = (unsigned)(a.h * a.w), o = (unsigned)((h0 + r0) * a.w + ox);
        const int n = plane / a.c, ch = plane % a.c;
        const float *yp = a.y + ((int64_t)(a.y_n == 1 ? 0 : n) * a.c + ch) * hw + o;
        float *rp = a.out ? a.out + (int64_t)plane * hw + o : nullptr;
        v2f yv[PRW];
#pragma unroll
        for (int i = 0; i < PRW; ++i) yv[i] = *reinterpret_cast<const v2f *>(yp + (unsigned)(i * a.w));
#pragma unroll
        for (int i = 0; i < PRW; ++i) {
            const v2f r = yv[i] - acc[i];
            if (rp) *reinterpret_cast<v2f *>(rp + (unsigned)(i * a.w)) = r;
            if constexpr (POST) ss += r.x * r.x + r.y * r.y;
            else ss += a.l1 ? fabsf(r.x) + fabsf(r.y) : r.x * r.x + r.y * r.y;
        }
    } else if (!RESID && (a.w & 1) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 7u) == 0 && ox + 1 < a.w &&
               h0 + r0 + PRW <= a.h) {
        float *gp = a.out + (int64_t)plane * a.h * a.w + (unsigned)((h0 + r0) * a.w + ox);
#pragma unroll
        for (int i = 0; i < PRW; ++i) *reinterpret_cast<v2f *>(gp + (unsigned)(i * a.w)) = acc[i];
    } else {
#pragma unroll
        for (int i = 0; i < PRW; ++i) {
            const int oy = h0 + r0 + i;
            const float two[2] = {acc[i].x, acc[i].y};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (ox + e >= a.w) continue;
                if constexpr (RESID) ss += resid_epilogue<false>(a, plane, oy, ox + e, &two[e], 1);
                else out_epilogue<false>(a, plane, oy, ox + e, &two[e], 0.0f, false, 1);
            }
        }
    }
    if constexpr (RESID) {
        const float t = block_sum(ss, s_red);
        if (threadIdx.x == 0) tail_publish(&a.partials[(int64_t)plane * (a.tiles_x * a.tiles_y) + ty * a.tiles_x + tx], t, a.tail.counters != nullptr);
        tail_arrive(a.tail, plane / a.c);
    }
}

// =====================================================================
// One-launch adjoint for regular geometry (h, w multiples of the tile): A^T u on the IMAGE domain, no padded-domain
// scratch, no fold launch.  With V = C^T u_z the correlation-transpose of the zero-extended cotangent,
//     (A^T u)[jy][jx] = sum over my in M(jy), mx in M(jx) of V[my][mx],   M(j) = {j} + {-j : j >= 1} + {2(n-1)-j : j <= n-2}
// (the reflection fold; V vanishes beyond the taps' reach, so the far mirror terms drop out by themselves).  A mirrored
// term is the SAME tap loop on a mirrored window of the staged tile: V[-jy][.] = sum_d w'[d] u_z[-jy + dy][.] reads the
// rows of u in DESCENDING order as jy ascends, so a lane applies each run to the window starting at row
// (c - r0 - 7 + dy0) with its eight accumulators visited backwards (c = 0 for the top border, 2 * 63 for the bottom one,
// in tile coordinates); mirrored columns likewise swap the two halves of every 8-byte pair (op_sel of v_pk_fma_f32) and
// exchange the roles of the plain and the shifted accumulator sets (the parity of c - column + dx flips).  Windows that
// slide off the staged region are clamped onto its zero rows / columns: the staged halo is at least 11 rows / 4 columns,
// so a clamped window lies wholly in the zero extension whenever the true one does.  Per tile up to nine passes
// {plain, top, bottom} x {plain, left, right}; every pass skips, wave-uniformly, the runs that cannot reach the image from
// the wave's rows, and whole passes when the wave's rows are out of the taps' reach of the border.  Deterministic, no
// atomics; round 2's two-launch form (padded-domain V in scratch + fold kernel: 111 + 18 us at N = 64) stays as the
// fallback for ragged sizes.  A previous one-launch attempt computed V on the tile PLUS its reflection ring and folded
// through LDS (140-173 us): the ring patches landed on one or two of the four waves and the LDS image grew by the ring.
// =====================================================================
struct AdjReach { int t, b, l, r; };     // the adjoint taps reach dy in [-t, b], dx in [-l, r]

struct AdjLane {
    int rb, cb, cbo, rb63;               // window bases before the run's offsets: rows, plain / shifted column pair, c63 row
    int row_lo, row_hi, col_lo, col_hi;  // clamps of the window base (tile coordinates)
    int dyl, dyh, dxl, dxh;              // wave-uniform liveness of a run: dy0 + L - 1 >= dyl, dy0 <= dyh, dxl <= dx <= dxh
};

template <int L, bool ODD, int SWC, bool RY, bool RX, int PR>
__device__ __forceinline__ void adj_runs(TapAccT<PR> &t, const float *s0, const int sw_rt, const AdjLane &ln,
                                         const TapRun *runs, const int nruns)
{
    const int SW = SWC > 0 ? SWC : sw_rt;
    if (nruns <= 0) return;
    constexpr bool SHIFTED = RX ? !ODD : ODD;        // which accumulator set has aligned inputs for this dx parity
    TapRun r = load_run(runs, 0);
    for (int k = 0; k < nruns; ++k) {
        const TapRun nxt = load_run(runs, min(k + 1, nruns - 1));
        if (r.dy0 + L - 1 >= ln.dyl && r.dy0 <= ln.dyh && r.dx >= ln.dxl && r.dx <= ln.dxh) {
            int row = ln.rb + r.dy0, col = (SHIFTED ? ln.cbo : ln.cb) + r.dx;
            if constexpr (RY) row = min(max(row, ln.row_lo), ln.row_hi - (PR + L - 1));
            if constexpr (RX) col = min(max(col, ln.col_lo), ln.col_hi - 2);
            run_apply<L, SWC, RY, RX, PR>(SHIFTED ? t.aco : t.ac, s0 + row * SW + col, SW, r);
            if constexpr (SHIFTED && !RX) {          // (mirrored-column passes never owe column 63 anything: see the caller)
                int r63 = ln.rb63 + r.dy0;
                if constexpr (RY) r63 = min(max(r63, ln.row_lo), ln.row_hi - L);
                run_apply_col<L>(t.c63, s0 + r63 * SW + 63 + r.dx, SW, r);
            }
        }
        r = nxt;
    }
}

template <int SWC, bool RY, bool RX, int PR>
__device__ __forceinline__ void adj_all_runs(TapAccT<PR> &t, const float *s0, const int SW, const AdjLane &ln, const BlurArgs &a)
{
    tapacc_zero(t);
    const TapRun *r = a.runs;
    adj_runs<4, false, SWC, RY, RX, PR>(t, s0, SW, ln, r, a.nrun[0]);
    r += a.nrun[0];
    adj_runs<2, false, SWC, RY, RX, PR>(t, s0, SW, ln, r, a.nrun[1]);
    r += a.nrun[1];
    adj_runs<4, true, SWC, RY, RX, PR>(t, s0, SW, ln, r, a.nrun[2]);
    r += a.nrun[2];
    adj_runs<2, true, SWC, RY, RX, PR>(t, s0, SW, ln, r, a.nrun[3]);
}

template <int SWC, bool RY, bool RX>
__device__ __forceinline__ void adj_pass(TapAcc &t, const float *s0, const int SW, const AdjLane &ln, const BlurArgs &a)
{
    adj_all_runs<SWC, RY, RX, PRW>(t, s0, SW, ln, a);
    merge_shifted(t);
}

// Mirrored-COLUMN pass on a narrow strip.  Only the columns within the taps' reach of the left / right border receive
// anything from it (12 of 64 for the bench kernel), but every wave owns all 64 columns of its 16 rows: in the plain lane
// layout 80 % of the lanes multiplied staged zeros (ablation at N = 64: +26 us for the column mirrors against 62 us for
// the whole plain pass).  Here the wave's 64 lanes are re-dealt over the strip alone -- W = 8 (16) column pairs x 8 (4)
// row groups of PR = 2 (4) rows -- so all lanes carry live outputs; the results return to the owner lanes of the plain
// layout with one `ds_bpermute` per value (same wave, no barrier, no LDS storage).  Row groups of a half-wave are 4 (8)
// rows apart: with the 100 / 132-word strides their 8-byte reads fall on disjoint bank ranges.
// The strip is wide enough that its outermost column pair's neighbour contributes nothing (reach <= 2 W - 2).
template <int PR>
struct StripLane { int cps, row_s, cp_base, cp; };

template <int PR>
__device__ __forceinline__ StripLane<PR> strip_lane(const int px, const int wr0)
{
    constexpr int W = PR == 2 ? 8 : 16, NH = 32 / W;          // column pairs per strip row; row groups per half-wave
    const int lane = threadIdx.x & 63;
    StripLane<PR> sl;
    const int k = lane / W, half = k / NH, j = k % NH;
    sl.cps = lane & (W - 1);
    sl.row_s = wr0 + j * (2 * PRW / NH) + half * PR;            // first of the lane's PR rows (tile coordinates)
    sl.cp_base = px == 2 ? 32 - W : 0;
    sl.cp = sl.cp_base + sl.cps;
    return sl;
}

// merge the strip's shifted set, drop the self-mirrored border pixel, hand the values to the owner lanes of the plain layout
template <int PR>
__device__ __forceinline__ void strip_finish(v2f (&acc)[PRW], TapAccT<PR> &t, const StripLane<PR> &sl, const int py,
                                             const int px, const int wr0, const int r0)
{
    constexpr int W = PR == 2 ? 8 : 16;
#pragma unroll
    for (int i = 0; i < PR; ++i) {
        const float nb = __shfl_down(t.aco[i].x, 1, kWave);
        t.ac[i].x += t.aco[i].y;
        t.ac[i].y += sl.cps == W - 1 ? 0.0f : nb;
        // the border pixel itself is its own mirror image
        const int trow = sl.row_s + i;
        const bool xr = (py == 1 && trow == 0) || (py == 2 && trow == TH - 1);
        if (xr || (px == 1 && sl.cp == 0)) t.ac[i].x = 0.0f;
        if (xr || (px == 2 && sl.cp == 31)) t.ac[i].y = 0.0f;
    }
    // owner lane (column pair cpo, rows r0 .. r0 + 7) pulls row rho = r0 - wr0 + i of the strip
    const int rel = (int)(threadIdx.x & 31) - sl.cp_base;
    const bool mine = rel >= 0 && rel < W;
#pragma unroll
    for (int i = 0; i < PRW; ++i) {
        const int rho = r0 - wr0 + i;
        const int grp = PR == 2 ? 4 * ((rho >> 1) & 1) + (rho >> 2) : 2 * ((rho >> 2) & 1) + (rho >> 3);
        const int src = (rel & (W - 1)) + W * grp;
        const float vx = __shfl(t.ac[i % PR].x, src, kWave), vy = __shfl(t.ac[i % PR].y, src, kWave);
        acc[i].x += mine ? vx : 0.0f;
        acc[i].y += mine ? vy : 0.0f;
    }
}

// ---------------------------------------------------------------- one scan of the table for the whole tile
// The passes above each walk the run table again, and a walk costs its fixed per-run latency whether the pass has 32 or 8
// FMAs per run to show for it: the occupancy probe (tools/occ_taps.sh) gives T = 55 + 174 / n us for n resident workgroups
// per CU against 30 + 78 / n for the plain operator -- the adjoint's per-tile serial path was 2.2x as long.  A tile with at
// most one mirrored side per axis (every tile of an image with two or more tiles per axis) therefore applies, per run and
// in ONE iteration: the plain window, the mirrored-row window (into the same accumulators: the self-mirrored border row is
// restored from a saved copy), the mirrored-column strip window and the corner strip window.
struct FusedLane {
    // plain window
    int rb, cb, cbo, rb63;
    // mirrored rows, plain layout (wave-uniform switch v_on)
    bool v_on, v_x0, v_x7, v_xc;
    int v_rb, v_rb63;
    // mirrored columns / corner: strip layout
    bool h_on, vh_on;
    int h_rb, vh_rb, h_cb, h_cbo;
    int h_lo[4], h_hi[4];      // per class: the runs a strip on this side can use (adjoint table sorted by dx)
    int row_lo, row_hi, col_lo, col_hi;
};

// window reads and FMAs as separate steps, so that the windows of one run are read behind ONE wait
template <int L, int PR>
struct Win { v2f w[PR + L - 1]; };

template <int L, int PR>
__device__ __forceinline__ void win_read(Win<L, PR> &x, const float *p, const int SW)
{
#pragma unroll
    for (int m = 0; m < PR + L - 1; ++m) x.w[m] = *(LdsPair)(p + m * SW);
}

template <int L, bool REV, bool SWAP, int PR>
__device__ __forceinline__ void win_fma(v2f (&acc)[PR], const Win<L, PR> &x, const TapRun &r)
{
#pragma unroll
    for (int q = 0; q < L; ++q) {
        const v2f w2 = v2f{r.w[q], r.w[q]};
#pragma unroll
        for (int i = 0; i < PR; ++i) {
            const v2f in = SWAP ? v2f{x.w[i + q].y, x.w[i + q].x} : x.w[i + q];
            v2f &dst = acc[REV ? PR - 1 - i : i];
            dst = __builtin_elementwise_fma(w2, in, dst);
        }
    }
}

// HASV / HASH: the wave has a mirrored-row window / the tile a mirrored-column strip (and then, with HASV, the corner
// strip) -- wave-uniform, hoisted out of the loop.  No liveness tests inside: a run that cannot reach the image from
// these rows reads a window of staged zeros (that is what the clamps guarantee) and adds exact zeros.
template <int L, bool ODD, int SWC, int PRS, bool HASV, bool HASH>
__device__ __forceinline__ void adj_runs_fused(TapAcc &t, TapAccT<PRS> &h, TapAccT<PRS> &vh, const float *s0,
                                               const int sw_rt, const FusedLane &g, const TapRun *runs, const int nruns)
{
    const int SW = SWC > 0 ? SWC : sw_rt;
    if (nruns <= 0) return;
    TapRun r = load_run(runs, 0);
    for (int k = 0; k < nruns; ++k) {
        const TapRun nxt = load_run(runs, min(k + 1, nruns - 1));
        v2f (&dst)[PRW] = ODD ? t.aco : t.ac;
        int hcol = 0;
        {   // plain window (+ column 63 of the shifted set) and the mirrored-column strip: one wait
            Win<L, PRW> wm;
            Win<L, PRS> wh;
            float c[L];
            win_read<L, PRW>(wm, s0 + (g.rb + r.dy0) * SW + (ODD ? g.cbo : g.cb) + r.dx, SW);
            if constexpr (ODD) {
                const float *p63 = s0 + (g.rb63 + r.dy0) * SW + 63 + r.dx;
#pragma unroll
                for (int q = 0; q < L; ++q) c[q] = (*(LdsPair)(p63 + q * SW)).x;
            }
            if constexpr (HASH) {
                hcol = min(max((ODD ? g.h_cb : g.h_cbo) + r.dx, g.col_lo), g.col_hi - 2);
                win_read<L, PRS>(wh, s0 + (g.h_rb + r.dy0) * SW + hcol, SW);
            }
            win_fma<L, false, false, PRW>(dst, wm, r);
            if constexpr (ODD) {
#pragma unroll
                for (int q = 0; q < L; ++q) t.c63 = fmaf(r.w[q], c[q], t.c63);
            }
            if constexpr (HASH) win_fma<L, false, true, PRS>(ODD ? h.ac : h.aco, wh, r);
        }
        if constexpr (HASV) {   // mirrored rows (same accumulators; the self-mirrored border row is put back) + corner strip
            Win<L, PRW> wv;
            Win<L, PRS> wc;
            float c[L];
            const int row = min(max(g.v_rb + r.dy0, g.row_lo), g.row_hi - (PRW + L - 1));
            win_read<L, PRW>(wv, s0 + row * SW + (ODD ? g.cbo : g.cb) + r.dx, SW);
            if constexpr (ODD) {
                const int r63 = min(max(g.v_rb63 + r.dy0, g.row_lo), g.row_hi - L);
                const float *p63 = s0 + r63 * SW + 63 + r.dx;
#pragma unroll
                for (int q = 0; q < L; ++q) c[q] = (*(LdsPair)(p63 + q * SW)).x;
            }
            if constexpr (HASH) {
                const int rowc = min(max(g.vh_rb + r.dy0, g.row_lo), g.row_hi - (PRS + L - 1));
                win_read<L, PRS>(wc, s0 + rowc * SW + hcol, SW);
            }
            const v2f s0v = dst[0], s7v = dst[PRW - 1];
            win_fma<L, true, false, PRW>(dst, wv, r);
            if (g.v_x0) dst[0] = s0v;                 // the border row is its own mirror image
            if (g.v_x7) dst[PRW - 1] = s7v;
            if constexpr (ODD) {
                const float sc = t.c63;
#pragma unroll
                for (int q = 0; q < L; ++q) t.c63 = fmaf(r.w[q], c[q], t.c63);
                if (g.v_xc) t.c63 = sc;
            }
            if constexpr (HASH) win_fma<L, true, true, PRS>(ODD ? vh.ac : vh.aco, wc, r);
        }
        r = nxt;
    }
}

// one class of the table: the adjoint table is sorted by dx inside a class, so the runs a left (right) strip can use are
// its first (last) ones -- up to three segments, the middle one with the strip windows
template <int L, bool ODD, int SWC, int PRS>
__device__ __forceinline__ void adj_class_fused(TapAcc &t, TapAccT<PRS> &h, TapAccT<PRS> &vh, const float *s0, const int SW,
                                                const FusedLane &g, const TapRun *runs, const int n, const int hlo,
                                                const int hhi)
{
#pragma unroll 1
    for (int seg = 0; seg < 3; ++seg) {
        const int lo = seg == 0 ? 0 : (seg == 1 ? hlo : hhi), hi = seg == 0 ? hlo : (seg == 1 ? hhi : n);
        if (hi <= lo) continue;
        const bool hs = seg == 1;
        if (g.v_on) {                                  // wave-uniform
            if (hs) adj_runs_fused<L, ODD, SWC, PRS, true, true>(t, h, vh, s0, SW, g, runs + lo, hi - lo);
            else adj_runs_fused<L, ODD, SWC, PRS, true, false>(t, h, vh, s0, SW, g, runs + lo, hi - lo);
        } else {
            if (hs) adj_runs_fused<L, ODD, SWC, PRS, false, true>(t, h, vh, s0, SW, g, runs + lo, hi - lo);
            else adj_runs_fused<L, ODD, SWC, PRS, false, false>(t, h, vh, s0, SW, g, runs + lo, hi - lo);
        }
    }
}

template <int SWC, int PRS>
__device__ __forceinline__ void adj_fused(v2f (&acc)[PRW], const float *s0, const int SW, FusedLane &g, const BlurArgs &a,
                                          const int py, const int px, const int cy, const int cx, const int wr0,
                                          const int r0)
{
    const StripLane<PRS> sl = strip_lane<PRS>(px, wr0);
    g.h_rb = sl.row_s;
    g.vh_rb = cy - sl.row_s - (PRS - 1);
    g.h_cb = cx - 2 * sl.cp - 1;
    g.h_cbo = cx - 2 * sl.cp;
    TapAcc t;
    TapAccT<PRS> h, vh;
    tapacc_zero(t); tapacc_zero(h); tapacc_zero(vh);
    const TapRun *r = a.runs;
    adj_class_fused<4, false, SWC, PRS>(t, h, vh, s0, SW, g, r, a.nrun[0], g.h_lo[0], g.h_hi[0]);
    r += a.nrun[0];
    adj_class_fused<2, false, SWC, PRS>(t, h, vh, s0, SW, g, r, a.nrun[1], g.h_lo[1], g.h_hi[1]);
    r += a.nrun[1];
    adj_class_fused<4, true, SWC, PRS>(t, h, vh, s0, SW, g, r, a.nrun[2], g.h_lo[2], g.h_hi[2]);
    r += a.nrun[2];
    adj_class_fused<2, true, SWC, PRS>(t, h, vh, s0, SW, g, r, a.nrun[3], g.h_lo[3], g.h_hi[3]);
    merge_shifted(t);
#pragma unroll
    for (int i = 0; i < PRW; ++i) acc[i] = t.ac[i];
    if (g.h_on) strip_finish<PRS>(acc, h, sl, 0, px, wr0, r0);          // block-uniform
    if (g.vh_on) strip_finish<PRS>(acc, vh, sl, py, px, wr0, r0);        // wave-uniform
}

// PRS: rows per lane of the mirrored-column strips -- 2 (8 column pairs x 8 row groups: a 16-column strip, taps reaching
// <= 14 columns past the border) or 4 (16 x 4: 32 columns, <= 30); chosen by the host from the kernel's reach
template <bool EPI, int SWC, int PRS>
__global__ __launch_bounds__(NT, 4) void k_blur_taps_adj(BlurArgs a, TapGeom g, AdjReach reach)
{
    const int RH = TH + g.t + g.b, RW = TW + g.l + g.r, SW = SWC > 0 ? SWC : RW;
    extern __shared__ __align__(16) float lds[];
    float *s_in = lds, *s_red = lds + RH * SW;
    int plane, ty, tx;
    if (!block_to_tile(a, plane, ty, tx)) return;
    const int h0 = ty * TH, w0 = tx * TW;
    const int n = plane / a.c, ch = plane % a.c;
    NormPartials np{};
    const bool own_norm = EPI && !a.norm_in;
    if constexpr (EPI) {
        if (own_norm && threadIdx.x < kWave) np = particle_norm_issue(a.norm_partials, a.norm_parts, n);
    }
    load_region_taps_first<false, false, 4>(s_in, SW, RW / 4, g, h0, w0, a, plane);
    if constexpr (EPI) {
        if (own_norm) particle_norm_reduce(np, a.norm_parts, s_red);
    }
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63;
    // which 16 rows a wave owns rotates from tile to tile: the mirrored-row passes always fall on the first / last row
    // block of a border tile, and would otherwise pile up on one SIMD of the CU
    const int wv = ((tid >> 6) + tx + ty + plane) & 3;
    const int cp = tid & 31, wr0 = wv * 2 * PRW, r0 = wr0 + ((tid >> 5) & 1) * PRW, l15 = lane & 15;
    const float *s0 = s_in + g.t * SW + g.l;          // tile coordinates (0, 0)
    const bool top = ty == 0, bot = ty == a.tiles_y - 1, lef = tx == 0, rig = tx == a.tiles_x - 1;
    constexpr int BIG = 1 << 20, CM = 2 * (TH - 1);   // mirror constant of the far border in tile coordinates
    v2f acc[PRW];
    TapAcc t;
    AdjLane ln;
    ln.row_lo = -g.t; ln.row_hi = TH + g.b; ln.col_lo = -g.l; ln.col_hi = TW + g.r;
    // at most one mirrored side per axis, strip wide enough: one scan of the table (adj_fused)
    const int hrch = lef ? reach.r : reach.l;
    const bool fused = !(top && bot) && !(lef && rig) && !((lef || rig) && hrch > (PRS == 2 ? 14 : 30)) && !ABL(4);      // block-uniform
    if (fused) {
        const int py = top ? 1 : (bot ? 2 : 0), px = (lef && reach.r >= 1) ? 1 : ((rig && reach.l >= 1) ? 2 : 0);
        const int cy = py == 2 ? CM : 0, cx = px == 2 ? CM : 0;
        FusedLane f;
        f.row_lo = ln.row_lo; f.row_hi = ln.row_hi; f.col_lo = ln.col_lo; f.col_hi = ln.col_hi;
        f.rb = r0; f.cb = 2 * cp; f.cbo = 2 * cp - 1; f.rb63 = wr0 + l15;
        f.v_on = (py == 1 && max(wr0, 1) <= reach.b) || (py == 2 && min(wr0 + 2 * PRW - 1, TH - 2) - (TH - 1) >= -reach.t);
        f.v_on = f.v_on && !ABL(1);
        f.v_rb = cy - r0 - (PRW - 1); f.v_rb63 = cy - (wr0 + l15);
        f.v_x0 = py == 1 && r0 == 0; f.v_x7 = py == 2 && r0 == TH - PRW;
        f.v_xc = (py == 1 && wr0 + l15 == 0) || (py == 2 && wr0 + l15 == TH - 1);
        f.h_on = px != 0 && !ABL(2);
        f.vh_on = f.h_on && f.v_on;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f.h_lo[c] = px == 1 ? 0 : a.nrun[c] - a.nhhi[c];        // left: the first nhlo runs (dx >= 1); right: the last nhhi
            f.h_hi[c] = px == 1 ? a.nhlo[c] : a.nrun[c];
            if (!f.h_on) f.h_hi[c] = f.h_lo[c] = 0;
        }
        adj_fused<SWC, PRS>(acc, s0, SW, f, a, py, px, cy, cx, wr0, r0);
    } else
#pragma unroll 1
    for (int py = 0; py < 3; ++py) {
        // rows: plain | mirrored about the top border (image row 0) | about the bottom border (image row h - 1)
        if (py && ABL(1)) continue;
        if (py == 1 && !(top && max(wr0, 1) <= reach.b)) continue;               // wave-uniform
        if (py == 2 && !(bot && min(wr0 + 2 * PRW - 1, TH - 2) - (TH - 1) >= -reach.t)) continue;
        const int cy = py == 2 ? CM : 0;
        ln.rb = py ? cy - r0 - (PRW - 1) : r0;
        ln.rb63 = py ? cy - (wr0 + l15) : wr0 + l15;
        ln.dyl = py == 0 ? -h0 - wr0 - (2 * PRW - 1) : (py == 1 ? max(wr0, 1) : -BIG);
        ln.dyh = py == 0 ? a.h - 1 - h0 - wr0 : (py == 1 ? BIG : min(wr0 + 2 * PRW - 1, TH - 2) - (TH - 1));
#pragma unroll 1
        for (int px = 0; px < 3; ++px) {
            if (px && ABL(2)) continue;
            if (px == 1 && !(lef && reach.r >= 1)) continue;                       // launch / block uniform
            if (px == 2 && !(rig && reach.l >= 1)) continue;
            const int cx = px == 2 ? CM : 0;
            ln.cb = px ? cx - 2 * cp - 1 : 2 * cp;
            ln.cbo = px ? cx - 2 * cp : 2 * cp - 1;
            ln.dxl = px == 0 ? -w0 - (TW - 1) : (px == 1 ? 1 : -BIG);
            ln.dxh = px == 0 ? a.w - 1 - w0 : (px == 1 ? BIG : -1);
            if (py == 0 && px == 0) adj_pass<SWC, false, false>(t, s0, SW, ln, a);
            else if (px == 0) adj_pass<SWC, true, false>(t, s0, SW, ln, a);
            else if (py == 0) adj_pass<SWC, false, true>(t, s0, SW, ln, a);
            else adj_pass<SWC, true, true>(t, s0, SW, ln, a);
            if (py == 0 && px == 0) {
#pragma unroll
                for (int i = 0; i < PRW; ++i) acc[i] = t.ac[i];
            } else {
                // the border pixel itself is its own mirror image: M(0) has no second copy of row / column 0
                const bool xr0 = py == 1 && r0 == 0, xr7 = py == 2 && r0 == TH - PRW;
                const bool xcx = px == 1 && cp == 0, xcy = px == 2 && cp == 31;
#pragma unroll
                for (int i = 0; i < PRW; ++i) {
                    const bool xr = (i == 0 && xr0) || (i == PRW - 1 && xr7);
                    acc[i].x += (xr || xcx) ? 0.0f : t.ac[i].x;
                    acc[i].y += (xr || xcy) ? 0.0f : t.ac[i].y;
                }
            }
        }
    }
    // epilogue: plain gradient, or clamp gate + -b * (coef * g + extra) into g_model_out[:, :c]
    const unsigned hw = (unsigned)(a.h * a.w), o = (unsigned)((h0 + r0) * a.w + w0 + 2 * cp);
    if constexpr (EPI) {
        const float nv = a.norm_in ? a.norm_in[n] : s_red[0];
        const float coef = norm_coef_dev(nv, a.scale, a.power), mb = -a.k.b;
        if (own_norm && a.norm_out && ch == 0 && ty == 0 && tx == 0 && threadIdx.x == 0) a.norm_out[n] = nv;
        const uint8_t *ip = a.inside_r + (int64_t)plane * hw + o;
        const float *ep = a.g_extra ? a.g_extra + (int64_t)plane * hw + o : nullptr;
        float *gp = a.g_model_out + ((int64_t)n * 2 * a.c + ch) * hw + o;
        uchar2 in[PRW];
        v2f ex[PRW];
#pragma unroll
        for (int i = 0; i < PRW; ++i) {
            in[i] = *reinterpret_cast<const uchar2 *>(ip + (unsigned)(i * a.w));
            ex[i] = ep ? *reinterpret_cast<const v2f *>(ep + (unsigned)(i * a.w)) : v2f{0.0f, 0.0f};
        }
#pragma unroll
        for (int i = 0; i < PRW; ++i) {
            v2f gq;
            gq.x = in[i].x ? mb * (coef * acc[i].x + ex[i].x) : 0.0f;
            gq.y = in[i].y ? mb * (coef * acc[i].y + ex[i].y) : 0.0f;
            *reinterpret_cast<v2f *>(gp + (unsigned)(i * a.w)) = gq;
        }
    } else {
        float *gp = a.out + (int64_t)plane * hw + o;
#pragma unroll
        for (int i = 0; i < PRW; ++i) *reinterpret_cast<v2f *>(gp + (unsigned)(i * a.w)) = acc[i];
    }
}

struct FoldArgs {
    const float *v;       // [planes, ph, pw] padded-domain correlation-transpose
    float *g;             // !EPI: [planes, h, w]
    const float *norm_in, *norm_partials;
    int norm_parts;
    float *norm_out;
    const uint8_t *inside;
    const float *g_extra;
    float *g_model_out;
    float scale, neg_b;
    int power, c, h, w, ph, pw, off, reach;
};

template <bool EPI>
__global__ __launch_bounds__(NT) void k_blur_fold(FoldArgs f)
{
    __shared__ float s_nrm[1];
    const int plane = blockIdx.y, n = plane / f.c, ch = plane % f.c;
    float coef = 0.0f;
    if constexpr (EPI) {
        if (!f.norm_in) {
            particle_norm_to_lds(f.norm_partials, f.norm_parts, n, s_nrm);
            __syncthreads();
        }
        const float nv = f.norm_in ? f.norm_in[n] : s_nrm[0];
        coef = norm_coef_dev(nv, f.scale, f.power);
        if (!f.norm_in && f.norm_out && blockIdx.x == 0 && threadIdx.x == 0 && ch == 0) f.norm_out[n] = nv;
    }
    const int idx = blockIdx.x * NT + threadIdx.x;
    if (idx >= f.h * f.w) return;
    const int i = idx / f.w, j = idx - i * f.w;
    int sy[2], sx[2];
    const int ny = fold_sources(i, f.h, f.reach, sy), nx = fold_sources(j, f.w, f.reach, sx);
    const int ps[3] = {i, sy[0], sy[1]}, qs[3] = {j, sx[0], sx[1]};
    const float *vp = f.v + (int64_t)plane * f.ph * f.pw;
    float acc = 0.0f;
    for (int a = 0; a <= ny; ++a)
        for (int b = 0; b <= nx; ++b) acc += vp[(int64_t)(ps[a] + f.off) * f.pw + (qs[b] + f.off)];
    const int64_t o = (int64_t)i * f.w + j, hw = (int64_t)f.h * f.w;
    if constexpr (EPI)
        f.g_model_out[((int64_t)n * 2 * f.c + ch) * hw + o] =
            f.inside[(int64_t)plane * hw + o]
                ? f.neg_b * (coef * acc + (f.g_extra ? f.g_extra[(int64_t)plane * hw + o] : 0.0f)) : 0.0f;
    else
        f.g[(int64_t)plane * hw + o] = acc;
}

// four consecutive pixels of a row per lane (w, pw, off multiples of 4; 16-byte aligned planes): the main term and the
// row folds are float4 loads, only the columns within `reach` of the left / right edge add scalar loads.  Per pixel the
// terms are added in the order of k_blur_fold (rows outer, columns inner): bit-identical results.
template <bool EPI>
__global__ __launch_bounds__(NT) void k_blur_fold4(FoldArgs f)
{
    __shared__ float s_nrm[1];
    const int plane = blockIdx.y, n = plane / f.c, ch = plane % f.c;
    float coef = 0.0f;
    if constexpr (EPI) {
        if (!f.norm_in) {
            particle_norm_to_lds(f.norm_partials, f.norm_parts, n, s_nrm);
            __syncthreads();
        }
        const float nv = f.norm_in ? f.norm_in[n] : s_nrm[0];
        coef = norm_coef_dev(nv, f.scale, f.power);
        if (!f.norm_in && f.norm_out && blockIdx.x == 0 && threadIdx.x == 0 && ch == 0) f.norm_out[n] = nv;
    }
    const int unit = blockIdx.x * NT + threadIdx.x, wu = f.w / 4;
    if (unit >= f.h * wu) return;
    const int i = unit / wu, j0 = (unit - i * wu) * 4;
    int sy[2];
    const int ny = fold_sources(i, f.h, f.reach, sy);
    const int ps[3] = {i, sy[0], sy[1]};
    const float *vp = f.v + (int64_t)plane * f.ph * f.pw;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const bool edge = j0 <= f.reach || j0 + 3 >= f.w - 1 - f.reach;     // some column of the unit has column folds
    for (int a = 0; a <= ny; ++a) {
        const float *row = vp + (int64_t)(ps[a] + f.off) * f.pw + f.off;
        const float4 m = *reinterpret_cast<const float4 *>(row + j0);
        const float mv[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[e] += mv[e];
            if (edge) {
                int sx[2];
                const int nx = fold_sources(j0 + e, f.w, f.reach, sx);
                for (int b = 0; b < nx; ++b) acc[e] += row[sx[b]];
            }
        }
    }
    const int64_t hw = (int64_t)f.h * f.w, o = (int64_t)i * f.w + j0;
    if constexpr (EPI) {
        const uchar4 in = *reinterpret_cast<const uchar4 *>(f.inside + (int64_t)plane * hw + o);
        float4 ex = make_float4(0, 0, 0, 0);
        if (f.g_extra) ex = *reinterpret_cast<const float4 *>(f.g_extra + (int64_t)plane * hw + o);
        float4 g;
        g.x = in.x ? f.neg_b * (coef * acc[0] + ex.x) : 0.0f;
        g.y = in.y ? f.neg_b * (coef * acc[1] + ex.y) : 0.0f;
        g.z = in.z ? f.neg_b * (coef * acc[2] + ex.z) : 0.0f;
        g.w = in.w ? f.neg_b * (coef * acc[3] + ex.w) : 0.0f;
        *reinterpret_cast<float4 *>(f.g_model_out + ((int64_t)n * 2 * f.c + ch) * hw + o) = g;
    } else {
        *reinterpret_cast<float4 *>(f.g + (int64_t)plane * hw + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// =====================================================================
// host dispatch
// =====================================================================
// small: only the reduction slots (regular multi-tile geometry: no slow folds, which keep 160 tap words in the scratch)
static inline size_t sep_lds_bytes(int rr, bool small_scratch = false)
{
    size_t pad = 0;
#if defined(DPSX_ABLATION) && DPSX_ABLATION
    static const size_t env_pad = getenv("DPSX_LDS_PAD") ? (size_t)atoi(getenv("DPSX_LDS_PAD")) : 0;   // occupancy probe
    pad = env_pad;
#endif
    return (size_t)((TH + 2 * rr) * (TW + 2 * rr + 4)) * 4 + (small_scratch ? 256 : kScratchBytes) + pad;
}

static void fill_geometry(BlurArgs &a, int64_t planes, int64_t c, int64_t h, int64_t w)
{
    a.c = (int)c; a.h = (int)h; a.w = (int)w; a.planes = (int)planes;
    a.tiles_x = (int)((w + TW - 1) / TW);
    a.tiles_y = (int)((h + TH - 1) / TH);
#if defined(DPSX_ABLATION) && DPSX_ABLATION
    static const int dbg = getenv("DPSX_DBG") ? atoi(getenv("DPSX_DBG")) : 0;   // ablation build only (blur_sep.h)
    a.dbg = dbg;
#else
    a.dbg = 0;
#endif
    a.src_h = (int)h; a.src_w = (int)w; a.src_off = 0;
}

static inline unsigned grid_blocks(const BlurArgs &a)
{
    const int groups = (a.planes + 7) / 8;
    return (unsigned)(groups * 8 * a.tiles_x * a.tiles_y);
}

int64_t blur_parts_per_particle(const dpsx_op *, int64_t c, int64_t h, int64_t w)
{
    return c * ((h + TH - 1) / TH) * ((w + TW - 1) / TW);
}

// dynamic LDS above 64 KiB needs the attribute once per kernel symbol
template <typename K>
static int allow_lds(K kernel, size_t bytes, bool &done)
{
    if (!done) {
        DPSX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)(144 * 1024)));
        done = true;
    }
    (void)bytes;
    return DPSX_OK;
}

#define DPSX_LAUNCH(KERNEL, GRID, LDS, STREAM, ...)                      \
    do {                                                                  \
        static bool s_attr_done = false;                                  \
        int rc_ = allow_lds(&KERNEL, LDS, s_attr_done);                   \
        if (rc_ != DPSX_OK) return rc_;                                   \
        hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(NT), LDS, STREAM, __VA_ARGS__); \
        return check_launch();                                            \
    } while (0)

template <int R4, bool POST, bool RESID>
static int launch_sep_fwd(const BlurArgs &a, const SepTaps &t, hipStream_t s)
{
    const size_t lds = sep_lds_bytes(4 * R4);
    DPSX_LAUNCH((k_blur_sep_fwd<R4, POST, RESID>), grid_blocks(a), lds, s, a, t);
}

template <bool POST, bool RESID>
static int dispatch_sep_fwd(const dpsx_op *op, const BlurArgs &a, hipStream_t s)
{
    switch (op->radius4 / 4) {
    case 1: return launch_sep_fwd<1, POST, RESID>(a, op->sep, s);
    case 2: return launch_sep_fwd<2, POST, RESID>(a, op->sep, s);
    case 3: return launch_sep_fwd<3, POST, RESID>(a, op->sep, s);
    case 4: return launch_sep_fwd<4, POST, RESID>(a, op->sep, s);
    case 5: return launch_sep_fwd<5, POST, RESID>(a, op->sep, s);
    case 6: return launch_sep_fwd<6, POST, RESID>(a, op->sep, s);
    case 7: return launch_sep_fwd<7, POST, RESID>(a, op->sep, s);
    case 8: return launch_sep_fwd<8, POST, RESID>(a, op->sep, s);
    }
    return DPSX_EUNSUPPORTED;
}

template <int R4, bool EPI>
static int launch_sep_adj(const BlurArgs &a, const SepTaps &t, int reach, hipStream_t s)
{
    // two or more full tiles per axis: every fold is a fast (register-window) fold and the scratch holds one norm slot
    const size_t lds = sep_lds_bytes(4 * R4, a.h % TH == 0 && a.w % TW == 0 && a.tiles_x >= 2 && a.tiles_y >= 2);
    DPSX_LAUNCH((k_blur_sep_adj<R4, EPI>), grid_blocks(a), lds, s, a, t, reach);
}

template <int R4, bool EPI>
static int launch_sep_adj_sym(const BlurArgs &a, const SepTaps &t, hipStream_t s)
{
    const size_t lds = sep_lds_bytes(4 * R4, true);
    DPSX_LAUNCH((k_blur_sep_adj_sym<R4, EPI>), grid_blocks(a), lds, s, a, t);
}

// symmetric taps on both axes (bitwise), whole tiles, two or more of them per axis, reach inside the image: the adjoint
// without fold terms (k_blur_sep_adj_sym)
static bool sep_adj_sym_ok(const dpsx_op *op, const BlurArgs &a)
{
    static const char *force = getenv("DPSX_SEP_ADJ");          // A/B switch for tools/kbench.py: "fold" = the general kernel
    if (force && force[0] == 'f') return false;
    const int rr = op->radius4;
    if (a.h % TH != 0 || a.w % TW != 0 || a.tiles_x < 2 || a.tiles_y < 2 || rr + 1 >= a.h || rr + 1 >= a.w) return false;
    for (int i = 0; i < rr; ++i)
        if (op->sep.h[i] != op->sep.h[2 * rr - i] || op->sep.v[i] != op->sep.v[2 * rr - i]) return false;
    return true;
}

template <bool EPI>
static int dispatch_sep_adj(const dpsx_op *op, const BlurArgs &a, hipStream_t s)
{
    SepTaps f{};  // adjoint of a correlation = correlation with the reversed taps
    const int rr = op->radius4;
    if (sep_adj_sym_ok(op, a)) {
        switch (rr / 4) {
        case 1: return launch_sep_adj_sym<1, EPI>(a, op->sep, s);
        case 2: return launch_sep_adj_sym<2, EPI>(a, op->sep, s);
        case 3: return launch_sep_adj_sym<3, EPI>(a, op->sep, s);
        case 4: return launch_sep_adj_sym<4, EPI>(a, op->sep, s);
        case 5: return launch_sep_adj_sym<5, EPI>(a, op->sep, s);
        case 6: return launch_sep_adj_sym<6, EPI>(a, op->sep, s);
        case 7: return launch_sep_adj_sym<7, EPI>(a, op->sep, s);
        case 8: return launch_sep_adj_sym<8, EPI>(a, op->sep, s);
        }
        return DPSX_EUNSUPPORTED;
    }
    for (int i = 0; i <= 2 * rr; ++i) {
        f.h[i] = op->sep.h[2 * rr - i];
        f.v[i] = op->sep.v[2 * rr - i];
    }
    switch (rr / 4) {
    case 1: return launch_sep_adj<1, EPI>(a, f, op->reach, s);
    case 2: return launch_sep_adj<2, EPI>(a, f, op->reach, s);
    case 3: return launch_sep_adj<3, EPI>(a, f, op->reach, s);
    case 4: return launch_sep_adj<4, EPI>(a, f, op->reach, s);
    case 5: return launch_sep_adj<5, EPI>(a, f, op->reach, s);
    case 6: return launch_sep_adj<6, EPI>(a, f, op->reach, s);
    case 7: return launch_sep_adj<7, EPI>(a, f, op->reach, s);
    case 8: return launch_sep_adj<8, EPI>(a, f, op->reach, s);
    }
    return DPSX_EUNSUPPORTED;
}

static void set_taps(const dpsx_op *op, BlurArgs &a)
{
    a.runs = static_cast<const TapRun *>(op->d_runs_fwd);
    for (int k = 0; k < 4; ++k) a.nrun[k] = op->nrun[k];
}

// compile-time LDS row-stride classes of the 16-byte-load variants (0 = runtime stride, the scalar-load variants):
// multiples of 4 words (16-byte staging stores) that are NOT multiples of 32 (run_apply_col reads one column of 16 rows)
constexpr int kSwNarrow = 100, kSwWide = 132;
static inline int taps_swc(const TapGeom &g, bool vec)
{
    const int rw = TW + g.l + g.r;
    return !vec ? 0 : (rw <= 96 ? kSwNarrow : kSwWide);
}
static inline size_t taps_lds(const TapGeom &g, int swc)
{
    const int rh = TH + g.t + g.b, sw = swc > 0 ? swc : TW + g.l + g.r;
    size_t pad = 0;
#if defined(DPSX_ABLATION) && DPSX_ABLATION
    static const size_t env_pad = getenv("DPSX_LDS_PAD") ? (size_t)atoi(getenv("DPSX_LDS_PAD")) : 0;   // occupancy probe
    pad = env_pad;
#endif
    return ((size_t)rh * sw) * 4 + kScratchBytes + pad;
}
static inline unsigned magic_of(unsigned d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); }

static TapGeom make_geom(int t, int b, int l, int r)
{
    TapGeom g{t, b, l, r, 0u, 0u};
    g.rwu_magic = magic_of((unsigned)((TW + l + r) / 4));
    g.side_magic = magic_of((unsigned)((l + r) / 4));
    return g;
}

template <bool POST, int MODE, bool VEC, int SWC, int NHB>
static int launch_taps_kn(const BlurArgs &a, const TapGeom &g, hipStream_t s)
{
    const size_t lds = taps_lds(g, SWC);
    DPSX_LAUNCH((k_blur_taps<POST, MODE, VEC, SWC, NHB>), grid_blocks(a), lds, s, a, g);
}

template <bool POST, int MODE, bool VEC, int SWC>
static int launch_taps_k(const BlurArgs &a, const TapGeom &g, hipStream_t s)
{
    if constexpr (VEC && POST) {
        const int halo_units = (g.t + g.b) * ((TW + g.l + g.r) / 4) + TH * ((g.l + g.r) / 4);
        if (halo_units <= 2 * NT) return launch_taps_kn<POST, MODE, VEC, SWC, 2>(a, g, s);
    }
    return launch_taps_kn<POST, MODE, VEC, SWC, 4>(a, g, s);
}

template <bool POST, int MODE>
static int launch_taps(const BlurArgs &a, const TapGeom &g, bool vec, hipStream_t s)
{
    if (!vec) return launch_taps_k<POST, MODE, false, 0>(a, g, s);
    if (taps_swc(g, true) == kSwNarrow) return launch_taps_k<POST, MODE, true, kSwNarrow>(a, g, s);
    return launch_taps_k<POST, MODE, true, kSwWide>(a, g, s);
}

template <bool POST, bool RESID>
static int launch_taps_fwd(const dpsx_op *op, BlurArgs a, bool vec, hipStream_t s)
{
    set_taps(op, a);
    const TapGeom g = make_geom(op->halo_t, op->halo_b, op->halo_l, op->halo_r);
    return launch_taps<POST, RESID ? 1 : 0>(a, g, vec, s);
}

int64_t blur_adjoint_scratch_bytes(const dpsx_op *op, int64_t planes, int64_t h, int64_t w)
{
    // the padded-domain buffer of the two-launch adjoint; the separable 16-byte path does not need it
    if (op->kind == OP_SEP && w % 4 == 0 && (h * w) % 4 == 0) return 0;
    const int64_t r4 = op->radius4;
    return ((planes * (h + 2 * r4) * (w + 2 * r4) * 4 + 255) / 256) * 256;
}

template <bool EPI, int SWC, int PRS>
static int launch_taps_adj1_kp(const BlurArgs &a, const TapGeom &g, const AdjReach &reach, hipStream_t s)
{
    const size_t lds = taps_lds(g, SWC);
    DPSX_LAUNCH((k_blur_taps_adj<EPI, SWC, PRS>), grid_blocks(a), lds, s, a, g, reach);
}

template <bool EPI, int SWC>
static int launch_taps_adj1_k(const BlurArgs &a, const TapGeom &g, const AdjReach &reach, hipStream_t s)
{
    // strip width by the taps' horizontal reach (both sides use the same kernel instance)
    if (std::max(reach.l, reach.r) <= 14) return launch_taps_adj1_kp<EPI, SWC, 2>(a, g, reach, s);
    return launch_taps_adj1_kp<EPI, SWC, 4>(a, g, reach, s);
}

// regular geometry: one launch on the image domain (k_blur_taps_adj)
template <bool EPI>
static int launch_taps_adj1(const dpsx_op *op, BlurArgs a, hipStream_t s)
{
    a.runs = static_cast<const TapRun *>(op->d_runs_adj);
    for (int k = 0; k < 4; ++k) {
        a.nrun[k] = op->nrun[k];
        a.nhlo[k] = op->nrun_adj_pos[k];      // (as counts: runs with dx >= 1 lead the class, runs with dx <= -1 end it)
        a.nhhi[k] = op->nrun_adj_neg[k];
    }
    // negated offsets: the forward halos swap sides; at least 11 zero rows / 4 zero columns for the clamped mirrored windows
    const AdjReach reach{op->halo_b, op->halo_t, op->halo_r, op->halo_l};
    const TapGeom g = make_geom(std::max(reach.t, 11), std::max(reach.b, 11), std::max(reach.l, 4), std::max(reach.r, 4));
    if (taps_swc(g, true) == kSwNarrow) return launch_taps_adj1_k<EPI, kSwNarrow>(a, g, reach, s);
    return launch_taps_adj1_k<EPI, kSwWide>(a, g, reach, s);
}

template <bool EPI>
static int launch_taps_adj(const dpsx_op *op, BlurArgs a, bool vec, float *scratch, int64_t scratch_bytes,
                           hipStream_t s)
{
    if (vec && a.h % TH == 0 && a.w % TW == 0 && (!EPI || (aligned16(a.g_extra) && (reinterpret_cast<uintptr_t>(a.inside_r) & 3u) == 0)))
        return launch_taps_adj1<EPI>(op, a, s);
    const int r4 = op->radius4;
    const int ph = a.h + 2 * r4, pw = a.w + 2 * r4;
    if (!scratch || scratch_bytes < (int64_t)a.planes * ph * pw * 4) return DPSX_EWORKSPACE;
    // 1. V = C^T u on the padded domain
    BlurArgs c{};
    c.x = a.x; c.out = scratch;
    fill_geometry(c, a.planes, 1, ph, pw);
    c.src_h = a.h; c.src_w = a.w; c.src_off = r4;
    c.runs = static_cast<const TapRun *>(op->d_runs_adj);
    for (int k = 0; k < 4; ++k) c.nrun[k] = op->nrun[k];
    const TapGeom g = make_geom(op->halo_b, op->halo_t, op->halo_r, op->halo_l);      // negated offsets: the sides swap
    const bool v2 = vec && aligned16(scratch);
    {
        int rc = launch_taps<false, 2>(c, g, v2, s);
        if (rc != DPSX_OK) return rc;
    }
    // 2. fold + epilogue
    FoldArgs f{};
    f.v = scratch; f.g = a.out; f.norm_in = a.norm_in; f.norm_partials = a.norm_partials; f.norm_parts = a.norm_parts;
    f.norm_out = a.norm_out; f.inside = a.inside_r; f.g_extra = a.g_extra; f.g_model_out = a.g_model_out; f.scale = a.scale;
    f.neg_b = -a.k.b; f.power = a.power; f.c = a.c; f.h = a.h; f.w = a.w; f.ph = ph; f.pw = pw; f.off = r4;
    f.reach = op->reach;
    const bool vec4 = v2 && a.w % 4 == 0 && pw % 4 == 0 && r4 % 4 == 0 && aligned16(EPI ? (const void *)a.g_model_out : (const void *)a.out) &&
                      aligned16(a.g_extra) && (reinterpret_cast<uintptr_t>(a.inside_r) & 3u) == 0;
    if (vec4) {
        const dim3 grid4((unsigned)((a.h * (a.w / 4) + NT - 1) / NT), (unsigned)a.planes);
        hipLaunchKernelGGL(k_blur_fold4<EPI>, grid4, dim3(NT), 0, s, f);
        return check_launch();
    }
    const dim3 grid((unsigned)((a.h * a.w + NT - 1) / NT), (unsigned)a.planes);
    hipLaunchKernelGGL(k_blur_fold<EPI>, grid, dim3(NT), 0, s, f);
    return check_launch();
}

static bool geometry_ok(const dpsx_op *op, int64_t h, int64_t w)
{
    // ReflectionPad2d requires pad < dim (torch raises otherwise)
    return h > op->radius && w > op->radius && h < (1 << 15) && w < (1 << 15);
}

static bool vec_ok(int64_t h, int64_t w, std::initializer_list<const void *> ptrs)
{
    if (w % 4 != 0 || (h * w) % 4 != 0) return false;
    for (const void *p : ptrs)
        if (p && !aligned16(p)) return false;
    return true;
}

int blur_forward(const dpsx_op *op, const float *x, float *y, int64_t planes, int64_t h, int64_t w, hipStream_t s)
{
    if (!geometry_ok(op, h, w)) return DPSX_EINVAL;
    if (planes == 0) return DPSX_OK;
    BlurArgs a{};
    a.x = x; a.out = y;
    fill_geometry(a, planes, 1, h, w);
    const bool vec = vec_ok(h, w, {x, y});
    return op->kind == OP_SEP && vec ? dispatch_sep_fwd<false, false>(op, a, s)
                                     : launch_taps_fwd<false, false>(op, a, vec, s);
}

int blur_adjoint(const dpsx_op *op, const float *u, float *g, int64_t planes, int64_t h, int64_t w, float *scratch,
                 int64_t scratch_bytes, hipStream_t s)
{
    if (!geometry_ok(op, h, w)) return DPSX_EINVAL;
    if (planes == 0) return DPSX_OK;
    BlurArgs a{};
    a.x = u; a.out = g;
    fill_geometry(a, planes, 1, h, w);
    const bool vec = vec_ok(h, w, {u, g});
    return op->kind == OP_SEP && vec ? dispatch_sep_adj<false>(op, a, s)
                                     : launch_taps_adj<false>(op, a, vec, scratch, scratch_bytes, s);
}

int blur_step_fwd(const dpsx_op *op, const StepFwdArgs &f, hipStream_t s)
{
    if (!geometry_ok(op, f.h, f.w)) return DPSX_EINVAL;
    if (f.n == 0) return DPSX_OK;
    BlurArgs a{};
    a.x_t = f.x_t; a.model_out = f.model_out; a.noise = f.noise;
    a.x0_hat = f.x0_hat; a.sample = f.sample; a.inside_w = f.inside;
    a.y = f.y; a.y_n = (int)f.y_n; a.out = f.resid; a.partials = f.partials;
    a.k = f.k;
    fill_geometry(a, f.n * f.c, f.c, f.h, f.w);
    a.tail = f.tail;
    a.tail.blocks_per_particle = a.c * a.tiles_x * a.tiles_y;
    const bool vec = vec_ok(f.h, f.w, {f.x_t, f.model_out, f.noise, f.x0_hat, f.sample, f.y, f.resid}) &&
                     (reinterpret_cast<uintptr_t>(f.inside) & 3u) == 0;
    return op->kind == OP_SEP && vec ? dispatch_sep_fwd<true, true>(op, a, s) : launch_taps_fwd<true, true>(op, a, vec, s);
}

int blur_step_bwd(const dpsx_op *op, const StepBwdArgs &b, float *scratch, int64_t scratch_bytes, hipStream_t s)
{
    if (!geometry_ok(op, b.h, b.w)) return DPSX_EINVAL;
    if (b.n == 0) return DPSX_OK;
    BlurArgs a{};
    a.x = b.resid;
    a.norm_in = b.norm; a.norm_partials = b.partials; a.norm_parts = b.parts; a.norm_out = b.norm_out;
    a.inside_r = b.inside; a.g_extra = b.g_extra; a.g_model_out = b.g_model_out; a.scale = b.scale; a.power = b.power;
    a.k = b.k;
    fill_geometry(a, b.n * b.c, b.c, b.h, b.w);
    const bool vec = vec_ok(b.h, b.w, {b.resid, b.g_model_out, b.g_extra}) && (reinterpret_cast<uintptr_t>(b.inside) & 3u) == 0;
    return op->kind == OP_SEP && vec ? dispatch_sep_adj<true>(op, a, s)
                                     : launch_taps_adj<true>(op, a, vec, scratch, scratch_bytes, s);
}

int blur_score(const dpsx_op *op, const float *x, const float *y, int64_t y_n, float *partials, int64_t n,
               int64_t c, int64_t h, int64_t w, int l1, const Tail &tail, hipStream_t s)
{
    if (!geometry_ok(op, h, w)) return DPSX_EINVAL;
    if (n == 0) return DPSX_OK;
    BlurArgs a{};
    a.x = x; a.y = y; a.y_n = (int)y_n; a.out = nullptr; a.partials = partials;
    fill_geometry(a, n * c, c, h, w);
    a.l1 = l1;
    a.tail = tail;
    a.tail.blocks_per_particle = a.c * a.tiles_x * a.tiles_y;
    const bool vec = vec_ok(h, w, {x, y});
    return op->kind == OP_SEP && vec ? dispatch_sep_fwd<false, true>(op, a, s) : launch_taps_fwd<false, true>(op, a, vec, s);
}

}  // namespace dpsx
