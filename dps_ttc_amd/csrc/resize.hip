// Super-resolution operator: Shocher's Resizer as used by SuperResolutionOperator
// (reference: util/resizer.py:55-74, measurements.py:76-91).  Per axis
//     out[o] = sum_k wt[k, o] * x[idx[k, o]]
// W axis first, then H (Resizer.sorted_dims == [3, 2] for equal scales).  The
// tables are arbitrary (reflection is baked into idx), so the kernels are
// table-driven gathers staged through LDS:
//   forward : one workgroup per (plane, block of output rows): the input rows
//             that block needs (plus the rows it "owns" in the fused step) are
//             streamed HBM -> LDS once with 16-byte loads, W pass LDS -> LDS,
//             H pass LDS -> registers, residual / sum-of-squares epilogue;
//   adjoint : gather form over a CSR inverse of the tables (input index ->
//             list of (output index, weight)), so no atomics and a
//             deterministic summation order; one workgroup per (plane, block
//             of input rows); u (the small measurement-space tensor) is read
//             once into LDS, g is written with coalesced rows.
#include <algorithm>
#include <vector>

#include "common.h"

namespace dpsx {

constexpr int RT = 256;  // threads

struct ResizeDev {
    const float *w_h, *w_w;   // [taps, out]
    const int *i_h, *i_w;     // [taps, out]
    const int *inv_h_ptr, *inv_h_idx, *inv_w_ptr, *inv_w_idx;
    const float *inv_h_w, *inv_w_w;
    int in_h, in_w, out_h, out_w, taps_h, taps_w;
    // forward blocking
    int tp;                   // output rows per block
    int fwd_rows;             // max input rows staged per block
    const int *blk_lo;        // [nblk] first staged input row
    const int *blk_cnt;       // [nblk] number of staged rows
    const int *own_lo;        // [nblk+1] input-row ownership partition (fused step)
    // adjoint blocking
    int ti;                   // input rows per block
    int adj_rows;             // max u rows staged per block
    const int *ablk_lo, *ablk_cnt;
    int nnz_w;                // entries of the W inverse
    int ell_w;                // ELL width: max entries of one input column
    const int *ell_w_idx;     // [ell_w][in_w] output column (0 where padded)
    const float *ell_w_w;     // [ell_w][in_w] weight (0 where padded)
    int max_he;               // max H-inverse entries of one adjoint block
    int no_ellh;              // A/B switch (DPSX_RESIZE_ADJ_CSR): keep the CSR form of the H pass
    int ell_h;                // ELL width of the H inverse: max entries of one input row (same entry order as the CSR form)
    const int *ell_h_idx;     // [ell_h][in_h] output row (the row's first entry where padded)
    const float *ell_h_w;     // [ell_h][in_h] weight (0 where padded)
    // a second, finer adjoint blocking for small particle counts (planes x blocks of the first one would not fill the
    // chip: 192 workgroups at N = 16, 256 x 256): chosen at launch, copied over ti / adj_rows / ablk_* / max_he
    int ti2, adj_rows2, max_he2;
    const int *ablk_lo2, *ablk_cnt2;
    // a second, finer FORWARD blocking for the same case: tp2 = RT / out_w output rows per block (ONE trip of the H-pass
    // loop) when tp = gparts * tp2, gparts in {2, 4}.  Both blockings leave the SAME partial sums behind -- one per tp2
    // output rows, formed by the same threads in the same tree: a block of the coarse blocking publishes gparts of them
    // -- so a particle's norm does not depend on how many particles were launched with it.  gparts = 1: no fine blocking.
    int gparts, tp2, fwd_rows2;
    const int *blk_lo2, *blk_cnt2, *own_lo2;
};

struct ResizeArgs {
    const float *x;           // plain input [planes, in_h, in_w] / adjoint input u [planes, out_h, out_w]
    float *out;               // fwd: A x or residual (nullable when RESID); adj: g
    const float *x_t, *model_out, *noise;
    float *x0_hat, *sample;
    uint8_t *inside_w;
    const float *y;
    int y_n;
    float *partials;
    const float *norm_in, *norm_partials;
    int norm_parts;
    float *norm_out;
    const uint8_t *inside_r;
    const float *g_extra;     // optional extra cotangent on x0_hat (nullable)
    float *g_model_out;
    float scale;
    int power;
    int c, planes;
    Coefs k;
    int l1;                   // scoring (plain input): partials are sums of |r|
    Tail tail;                // in-launch finalisation of the per-particle reduction (common.h)
};

__device__ __forceinline__ float norm_coef_r(float nv, float gn, int power)
{
    return power == 2 ? -2.0f * gn : (nv == 0.0f ? 0.0f : -gn / nv);
}

// block_sum of up to four values at once (same tree per value as common.h's block_sum: wave_sum, waves added in index
// order); results valid in thread 0.  scratch: 16 floats (RT = 4 waves).
__device__ __forceinline__ void block_sum4(float (&v)[4], float *scratch)
{
#pragma unroll
    for (int g = 0; g < 4; ++g) v[g] = wave_sum(v[g]);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) scratch[g * 4 + wv] = v[g];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float t = 0.0f;
            for (int i = 0; i < RT / 64; ++i) t += scratch[g * 4 + i];
            v[g] = t;
        }
    }
}

// The H pass of one block and its partial sums.  gparts == 1: one partial per block (every trip of the loop accumulates
// into one value per thread).  gparts > 1 (the block covers exactly gparts * RT outputs): trip j keeps its own value --
// partial j of the block is what a block of the fine blocking, whose only trip it is, publishes as its one partial.
template <bool RESID, typename F>
__device__ __forceinline__ void resize_h_pass(const ResizeArgs &a, const ResizeDev &d, F &h_out, int total, int plane,
                                              int nblk, int blk, float *s_red)
{
    if (d.gparts > 1) {                                          // launch-uniform
        float sg[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < d.gparts) sg[j] = h_out((int)threadIdx.x + j * RT, 0.0f);
        if constexpr (RESID) {
            block_sum4(sg, s_red);
            if (threadIdx.x == 0) {
                float *dst = &a.partials[((int64_t)plane * nblk + blk) * d.gparts];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < d.gparts) tail_publish(dst + j, sg[j], a.tail.counters != nullptr);
            }
            tail_arrive(a.tail, plane / a.c);
        }
        return;
    }
    float ss = 0.0f;
    for (int it = threadIdx.x; it < total; it += RT) ss = h_out(it, ss);
    if constexpr (RESID) {
        const float t = block_sum(ss, s_red);
        if (threadIdx.x == 0) tail_publish(&a.partials[(int64_t)plane * nblk + blk], t, a.tail.counters != nullptr);
        tail_arrive(a.tail, plane / a.c);
    }
}

// LDS: s_in[fwd_rows][iws] | s_tmp[fwd_rows][out_w] | s_red[16]
// A staged row is stored RESIDUE-MAJOR: column j sits at word (j % 4) * q4 + j / 4 (q4 = ceil(in_w / 4), iws = 4 q4).
// The W pass reads column ~(in_w / out_w) * o + const in lane o: in plain order that is a stride of 4 words for x4
// (8-way bank conflict, 20 us of the 84 us launch at N = 64); residue-major, consecutive lanes read consecutive words.
template <bool POST, bool RESID, bool VEC>
__global__ __launch_bounds__(RT) void k_resize_fwd(ResizeArgs a, ResizeDev d)
{
    extern __shared__ __align__(16) float lds[];
    const int q4 = (d.in_w + 3) / 4, iws = 4 * q4;
    float *s_in = lds, *s_tmp = lds + d.fwd_rows * iws, *s_red = s_tmp + d.fwd_rows * d.out_w;
    const int nblk = (d.out_h + d.tp - 1) / d.tp;
    const int plane = blockIdx.x / nblk, blk = blockIdx.x % nblk;
    const int lo = d.blk_lo[blk], cnt = d.blk_cnt[blk];
    const int64_t ihw = (int64_t)d.in_h * d.in_w, ohw = (int64_t)d.out_h * d.out_w;
    const int n = plane / a.c, ch = plane % a.c;
    // ---- tables -> LDS (tiny; every lane then reads them at LDS speed instead of through L1)
    float *s_ww = s_red + 16, *s_wh = s_ww + d.taps_w * d.out_w;
    int *s_iw = reinterpret_cast<int *>(s_wh + d.taps_h * d.tp), *s_ih = s_iw + d.taps_w * d.out_w;
    for (int i = threadIdx.x; i < d.taps_w * d.out_w; i += RT) {
        s_ww[i] = d.w_w[i];
        const int j = d.i_w[i];
        s_iw[i] = (j & 3) * q4 + (j >> 2);
    }
    const int p0 = blk * d.tp, p1 = min(d.out_h, p0 + d.tp);
    for (int i = threadIdx.x; i < d.taps_h * d.tp; i += RT) {
        const int k = i / d.tp, p = p0 + i % d.tp;
        s_wh[i] = p < d.out_h ? d.w_h[k * d.out_h + p] : 0.0f;
        s_ih[i] = p < d.out_h ? d.i_h[k * d.out_h + p] - lo : 0;
    }
    // ---- stage A: rows [lo, lo+cnt) -> LDS, four units per lane in flight before any is consumed
    {
        const float *src, *eps = nullptr, *vv = nullptr, *zz = nullptr;
        int olo = 0, ohi = 0;
        if constexpr (POST) {
            src = a.x_t + (int64_t)plane * ihw;
            eps = a.model_out + ((int64_t)n * 2 * a.c + ch) * ihw;
            vv = eps + (int64_t)a.c * ihw;
            zz = a.noise + (int64_t)plane * ihw;
            olo = d.own_lo[blk];
            ohi = d.own_lo[blk + 1];
        } else {
            src = a.x + (int64_t)plane * ihw;
        }
        constexpr int U = VEC ? 4 : 1, B = 4;
        const int wu = d.in_w / U, total = cnt * wu;
        for (int base = threadIdx.x; base < total; base += RT * B) {
            float xv[B][U], ev[B][U], vq[B][U], zq[B][U];
#pragma unroll
            for (int bb = 0; bb < B; ++bb) {
                const int u = base + bb * RT;
                if (u >= total) continue;
                const int rr = u / wu, cu = u - rr * wu;
                const int gy = lo + rr;
                const int64_t o = (int64_t)gy * d.in_w + cu * U;
                const bool own = POST && gy >= olo && gy < ohi && (a.k.add_noise & 1);
                if constexpr (VEC) {
                    const float4 t = *reinterpret_cast<const float4 *>(src + o);
                    xv[bb][0] = t.x; xv[bb][1] = t.y; xv[bb][2] = t.z; xv[bb][3] = t.w;
                    if constexpr (POST) {
                        const float4 q = *reinterpret_cast<const float4 *>(eps + o);
                        ev[bb][0] = q.x; ev[bb][1] = q.y; ev[bb][2] = q.z; ev[bb][3] = q.w;
                        float4 t4 = make_float4(0, 0, 0, 0), z4 = t4;
                        if (own) {
                            t4 = *reinterpret_cast<const float4 *>(vv + o);
                            z4 = *reinterpret_cast<const float4 *>(zz + o);
                        }
                        vq[bb][0] = t4.x; vq[bb][1] = t4.y; vq[bb][2] = t4.z; vq[bb][3] = t4.w;
                        zq[bb][0] = z4.x; zq[bb][1] = z4.y; zq[bb][2] = z4.z; zq[bb][3] = z4.w;
                    }
                } else {
                    xv[bb][0] = src[o];
                    if constexpr (POST) {
                        ev[bb][0] = eps[o];
                        vq[bb][0] = own ? vv[o] : 0.0f;
                        zq[bb][0] = own ? zz[o] : 0.0f;
                    }
                }
            }
#pragma unroll
            for (int bb = 0; bb < B; ++bb) {
                const int u = base + bb * RT;
                if (u >= total) continue;
                const int rr = u / wu, cu = u - rr * wu;
                const int gy = lo + rr, gx = cu * U;
                float val[U];
#pragma unroll
                for (int e = 0; e < U; ++e) val[e] = xv[bb][e];
                if constexpr (POST) {
                    bool ins[U];
#pragma unroll
                    for (int e = 0; e < U; ++e) val[e] = post_x0(xv[bb][e], ev[bb][e], a.k, ins[e]);
                    if (gy >= olo && gy < ohi) {
                        float sm[U];
#pragma unroll
                        for (int e = 0; e < U; ++e) sm[e] = post_sample(xv[bb][e], val[e], vq[bb][e], zq[bb][e], a.k);
                        const int64_t po = (int64_t)plane * ihw + (int64_t)gy * d.in_w + gx;
                        if constexpr (VEC) {
                            if (a.x0_hat) *reinterpret_cast<float4 *>(a.x0_hat + po) = make_float4(val[0], val[1], val[2], val[3]);
                            *reinterpret_cast<float4 *>(a.sample + po) = make_float4(sm[0], sm[1], sm[2], sm[3]);
                            *reinterpret_cast<uchar4 *>(a.inside_w + po) = make_uchar4(ins[0], ins[1], ins[2], ins[3]);
                        } else {
                            if (a.x0_hat) a.x0_hat[po] = val[0];
                            a.sample[po] = sm[0];
                            a.inside_w[po] = ins[0];
                        }
                    }
                }
                float *dst = s_in + rr * iws;
                if constexpr (VEC) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[e * q4 + cu] = val[e];      // gx = 4 cu: residue e, quotient cu
                } else {
                    dst[(gx & 3) * q4 + (gx >> 2)] = val[0];
                }
            }
        }
    }
    __syncthreads();
    // ---- stage B: W pass  tmp[r][o] = sum_k w_w[k,o] * in[r][i_w[k,o]]
    for (int it = threadIdx.x; it < cnt * d.out_w; it += RT) {
        const int rr = it / d.out_w, o = it - rr * d.out_w;
        const float *row = s_in + rr * iws;
        float acc = 0.0f;
        for (int k = 0; k < d.taps_w; ++k)
            acc = fmaf(s_ww[k * d.out_w + o], row[s_iw[k * d.out_w + o]], acc);
        s_tmp[rr * d.out_w + o] = acc;
    }
    __syncthreads();
    // ---- stage C: H pass  out[p][o] = sum_k w_h[k,p] * tmp[i_h[k,p] - lo][o]
    // one output element: the H pass, the store, and (RESID) the term it adds to the block's sum
    auto h_out = [&](int it, float run) -> float {
        const int pl = it / d.out_w, p = p0 + pl, o = it % d.out_w;
        float acc = 0.0f;
        for (int k = 0; k < d.taps_h; ++k)
            acc = fmaf(s_wh[k * d.tp + pl], s_tmp[s_ih[k * d.tp + pl] * d.out_w + o], acc);
        const int64_t oo = (int64_t)p * d.out_w + o;
        if constexpr (RESID) {
            const float yv = a.y[((int64_t)(a.y_n == 1 ? 0 : n) * a.c + ch) * ohw + oo];
            const float r = yv - acc;
            if (a.out) a.out[(int64_t)plane * ohw + oo] = r;
            if constexpr (POST) return fmaf(r, r, run);
            else return a.l1 ? run + fabsf(r) : fmaf(r, r, run);
        } else {
            a.out[(int64_t)plane * ohw + oo] = acc;
            return 0.0f;
        }
    };
    resize_h_pass<RESID>(a, d, h_out, (p1 - p0) * d.out_w, plane, nblk, blk, s_red);
}

// ---------------------------------------------------------------------------------------------------------
// Row-streaming forward (the common geometries: in_w = 4 wu with wu | 64, out_w <= wu i.e. scale >= 4, <= 32 taps).
// One wave-instruction loads 64 float4 units = 64 / wu WHOLE rows, so the W pass of those rows runs right away out
// of a 1 KiB per-wave row buffer and only its result (out_w floats per row) is kept for the H pass: LDS per block
// drops from fwd_rows * (in_w + out_w) floats (65 KB at 256 -> 64, two blocks per CU) to ~17 KB, lane q owns one
// W output column for the whole block so its taps and (residue-major) indices live in registers, and the blocks
// of a plane share an XCD so the input rows neighbouring bands both stage are L2 hits.
// LDS: s_buf[4 waves][256] | s_tmp[fwd_rows][out_w] | s_red[16] | s_wh[taps_h][tp] | s_ih[taps_h][tp]
__device__ float g_zero_unit_rz[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // never written; non-const keeps it global-space

template <bool POST, bool RESID, int TWN, int B>
__global__ __launch_bounds__(RT, 3) void k_resize_fwd_rows(ResizeArgs a, ResizeDev d)
{
    extern __shared__ __align__(16) float lds[];
    const int wu = d.in_w >> 2, rps = 64 / wu, iws = d.in_w;
    const int nblk = (d.out_h + d.tp - 1) / d.tp;
    const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
    const int plane = (kq / nblk) * 8 + xcd, blk = kq % nblk;
    if (plane >= a.planes) return;
    float *s_buf = lds, *s_tmp = lds + 4 * 256, *s_red = s_tmp + d.fwd_rows * d.out_w, *s_wh = s_red + 16;
    int *s_ih = reinterpret_cast<int *>(s_wh + d.taps_h * d.tp);
    const int lo = d.blk_lo[blk], cnt = d.blk_cnt[blk];
    const unsigned ihw = (unsigned)(d.in_h * d.in_w), ohw = (unsigned)(d.out_h * d.out_w);
    const int n = plane / a.c, ch = plane % a.c;
    const int p0 = blk * d.tp, p1 = min(d.out_h, p0 + d.tp);
    for (int i = threadIdx.x; i < d.taps_h * d.tp; i += RT) {
        const int k = i / d.tp, p = p0 + i % d.tp;
        s_wh[i] = p < d.out_h ? d.w_h[k * d.out_h + p] : 0.0f;
        s_ih[i] = p < d.out_h ? d.i_h[k * d.out_h + p] - lo : 0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- this lane's W output: row rq of the wave-step, column oq; taps and remapped indices in registers
    const int nq = rps * d.out_w;
    const int rq = lane / d.out_w, oq = lane - rq * d.out_w;
    float ww[TWN];
    int iw[TWN];
    {   // unconditional loads from clamped addresses, all issued before the first is used (a load under a divergent
        // branch, or one the next address depends on, costs a serialised round trip each); lanes without an output
        // and taps beyond taps_w get weight 0 and a valid index
        const bool has_out = lane < nq;
        const int rqc = has_out ? rq : 0, oqc = has_out ? oq : 0;
        int jj[TWN];
#pragma unroll
        for (int k = 0; k < TWN; ++k) {
            const int kk = min(k, d.taps_w - 1);
            ww[k] = d.w_w[kk * d.out_w + oqc];
            jj[k] = d.i_w[kk * d.out_w + oqc];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < TWN; ++k) {
            ww[k] = (has_out && k < d.taps_w) ? ww[k] : 0.0f;
            iw[k] = rqc * iws + (jj[k] & 3) * wu + (jj[k] >> 2);
        }
    }
    const float *src, *eps = nullptr, *vv = nullptr, *zz = nullptr;
    int olo = 0, ohi = 0;
    if constexpr (POST) {
        src = a.x_t + (int64_t)plane * ihw;
        eps = a.model_out + ((int64_t)n * 2 * a.c + ch) * ihw;
        vv = eps + (int64_t)a.c * ihw;
        zz = a.noise + (int64_t)plane * ihw;
        olo = d.own_lo[blk];
        ohi = d.own_lo[blk + 1];
    } else {
        src = a.x + (int64_t)plane * ihw;
    }
    float *mybuf = s_buf + wave * 256;
    const int rl = lane / wu, cu = lane - rl * wu;
    const int nsteps = (cnt + rps - 1) / rps;
    const bool noisy = POST && (a.k.add_noise & 1);
    for (int s0 = wave; s0 < nsteps; s0 += 4 * B) {
        float4 xv[B], ev[B], vq[B], zq[B];
#pragma unroll
        for (int bb = 0; bb < B; ++bb) {
            const int rr = min((s0 + 4 * bb) * rps + rl, cnt - 1);      // surplus lanes / steps re-read the last row
            const int gy = lo + rr;
            const unsigned o = (unsigned)(gy * d.in_w + 4 * cu);
            xv[bb] = *reinterpret_cast<const float4 *>(src + o);
            if constexpr (POST) {
                ev[bb] = *reinterpret_cast<const float4 *>(eps + o);
                // rows this block does not own read a block of zeros: the loads themselves stay unconditional
                const bool own = noisy && gy >= olo && gy < ohi;
                vq[bb] = *reinterpret_cast<const float4 *>(own ? vv + o : g_zero_unit_rz);
                zq[bb] = *reinterpret_cast<const float4 *>(own ? zz + o : g_zero_unit_rz);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int bb = 0; bb < B; ++bb) {
            const int st = s0 + 4 * bb;
            if (st >= nsteps) break;                                     // wave-uniform
            const int rr = st * rps + rl;
            const int gy = lo + rr;
            float val[4] = {xv[bb].x, xv[bb].y, xv[bb].z, xv[bb].w};
            if constexpr (POST) {
                float4 x0;
                if (rr < cnt && gy >= olo && gy < ohi) {       // a row this block owns: the whole of S1 (packed: common.h)
                    float4 sm;
                    uchar4 gate;
                    post_unit(xv[bb], ev[bb], vq[bb], zq[bb], a.k, x0, sm, gate);
                    const int64_t po = (int64_t)plane * ihw + (unsigned)(gy * d.in_w + 4 * cu);
                    if (a.x0_hat) *reinterpret_cast<float4 *>(a.x0_hat + po) = x0;
                    *reinterpret_cast<float4 *>(a.sample + po) = sm;
                    *reinterpret_cast<uchar4 *>(a.inside_w + po) = gate;
                } else {
                    x0 = post_x0_unit(xv[bb], ev[bb], a.k);
                }
                val[0] = x0.x; val[1] = x0.y; val[2] = x0.z; val[3] = x0.w;
            }
            // residue-major rows of this step -> the wave's buffer; DS operations of one wave execute in order, so the
            // gathers below see these writes and the next step's writes cannot pass this step's reads
            float *dst = mybuf + rl * iws + cu;
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e * wu] = val[e];
            if (lane < nq && st * rps + rq < cnt) {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < TWN; ++k)
                    if (k < d.taps_w) acc = fmaf(ww[k], mybuf[iw[k]], acc);
                s_tmp[(st * rps + rq) * d.out_w + oq] = acc;
            }
        }
    }
    __syncthreads();
    // ---- H pass  out[p][o] = sum_k w_h[k,p] * tmp[i_h[k,p] - lo][o]   (as in k_resize_fwd)
    auto h_out = [&](int it, float run) -> float {
        const int pl = it / d.out_w, p = p0 + pl, o = it % d.out_w;
        float acc = 0.0f;
        for (int k = 0; k < d.taps_h; ++k)
            acc = fmaf(s_wh[k * d.tp + pl], s_tmp[s_ih[k * d.tp + pl] * d.out_w + o], acc);
        const unsigned oo = (unsigned)(p * d.out_w + o);
        if constexpr (RESID) {
            const float yv = a.y[((int64_t)(a.y_n == 1 ? 0 : n) * a.c + ch) * ohw + oo];
            const float r = yv - acc;
            if (a.out) a.out[(int64_t)plane * ohw + oo] = r;
            if constexpr (POST) return fmaf(r, r, run);
            else return a.l1 ? run + fabsf(r) : fmaf(r, r, run);
        } else {
            a.out[(int64_t)plane * ohw + oo] = acc;
            return 0.0f;
        }
    };
    resize_h_pass<RESID>(a, d, h_out, (p1 - p0) * d.out_w, plane, nblk, blk, s_red);
}

// LDS: s_u[adj_rows][out_w] | s_t[ti][out_w]
template <bool EPI, bool VEC>
__global__ __launch_bounds__(RT) void k_resize_adj(ResizeArgs a, ResizeDev d)
{
    extern __shared__ __align__(16) float lds[];
    float *s_u = lds, *s_t = lds + d.adj_rows * d.out_w;
    const int elln = d.ell_w * d.in_w;
    float *s_wv = s_t + d.ti * d.out_w, *s_hv = s_wv + elln;
    int *s_wi = reinterpret_cast<int *>(s_hv + d.max_he), *s_hp = s_wi + elln, *s_hi = s_hp + d.ti + 1;
    const int nblk = (d.in_h + d.ti - 1) / d.ti;
    const int plane = blockIdx.x / nblk, blk = blockIdx.x % nblk;
    const int lo = d.ablk_lo[blk], cnt = d.ablk_cnt[blk];
    const int64_t ihw = (int64_t)d.in_h * d.in_w, ohw = (int64_t)d.out_h * d.out_w;
    const int i0 = blk * d.ti, i1 = min(d.in_h, i0 + d.ti);
    const float *up = a.x + (int64_t)plane * ohw + (int64_t)lo * d.out_w;
    // staging loads first (the cotangent rows, then the tables below): each of these loops used to be a chain of dependent
    // round trips (load -> LDS store per iteration, trip counts unknown to the compiler) at the head of every workgroup
    const int nu = cnt * d.out_w;
    constexpr int UQ = 2;
    const bool u4 = VEC && (d.out_w % 4) == 0 && (reinterpret_cast<uintptr_t>(up) & 15u) == 0;
    float4 ureg0 = make_float4(0, 0, 0, 0), ureg1 = ureg0;
    if (u4) {
        ureg0 = *reinterpret_cast<const float4 *>(up + 4 * min((int)threadIdx.x, nu / 4 - 1));
        ureg1 = *reinterpret_cast<const float4 *>(up + 4 * min((int)threadIdx.x + RT, nu / 4 - 1));
    }
    // inverse tables -> LDS: the whole W inverse, and the H-inverse rows of this block (rebased to 0)
    // RT % (in_w / 4) == 0: a lane keeps its column group over all its rows, so the (<= 8 deep) ELL entries of its
    // four columns go from global memory (16-byte loads, L2-resident) straight to registers, once; otherwise the W
    // inverse is staged in LDS and read per element
    constexpr int EW = 8;
    const bool regtab = VEC && RT % (d.in_w / 4) == 0 && d.ell_w <= EW;
    float4 tw[EW];
    int4 ti4[EW];
    if (regtab) {
        const int j0 = (threadIdx.x % (d.in_w / 4)) * 4;
#pragma unroll
        for (int k = 0; k < EW; ++k) {
            const int kk = min(k, d.ell_w - 1);
            tw[k] = *reinterpret_cast<const float4 *>(d.ell_w_w + kk * d.in_w + j0);
            ti4[k] = *reinterpret_cast<const int4 *>(d.ell_w_idx + kk * d.in_w + j0);
            if (k >= d.ell_w) tw[k] = make_float4(0, 0, 0, 0);          // weight 0, any valid index
        }
    } else {
        for (int i = threadIdx.x; i < elln; i += RT) {      // [ell_w][in_w]: lane j reads word k*in_w + j, conflict-free
            s_wi[i] = d.ell_w_idx[i];
            s_wv[i] = d.ell_w_w[i];
        }
    }
    // H inverse of this block's rows: fixed-width (ELL) rows when they fit the CSR form's LDS slots -- [EH][ti] indices
    // (rebased to the staged rows) and weights, ONE independent load per thread, no pointer chase
    constexpr int EH = 4;
    static_assert(EH * 64 <= RT, "one ELL-H entry per thread");
    const int nrows = i1 - i0;
    const bool ellh = d.ell_h <= EH && EH * d.ti <= d.max_he && EH * d.ti <= RT && !d.no_ellh;     // launch-uniform
    if (ellh) {
        const int t = threadIdx.x, kk = t / d.ti, r = t - kk * d.ti;
        const bool live = kk < EH && r < nrows;
        const int src = min(kk, d.ell_h - 1) * d.in_h + i0 + min(r, nrows - 1);
        const int hi0 = d.ell_h_idx[src];
        const float hv0 = d.ell_h_w[src];
        if (u4) {
            if (t < nu / 4) *reinterpret_cast<float4 *>(s_u + 4 * t) = ureg0;
            if (t + RT < nu / 4) *reinterpret_cast<float4 *>(s_u + 4 * (t + RT)) = ureg1;
            for (int u = 4 * UQ * RT + t; u < nu; u += RT) s_u[u] = up[u];
        } else {
            for (int u = t; u < nu; u += RT) s_u[u] = up[u];
        }
        if (live) {
            s_hi[kk * d.ti + r] = max(hi0 - lo, 0);
            s_hv[kk * d.ti + r] = kk < d.ell_h ? hv0 : 0.0f;
        }
    } else {
    const int he0 = d.inv_h_ptr[i0], he1 = d.inv_h_ptr[i1];
    {   // first RT entries of each table in one shot (registers), leftovers by the loops
        const int nhe = he1 - he0, t = threadIdx.x;
        const int hp0 = d.inv_h_ptr[i0 + min(t, i1 - i0)];
        const int hi0 = d.inv_h_idx[he0 + min(t, max(nhe - 1, 0))];
        const float hv0 = d.inv_h_w[he0 + min(t, max(nhe - 1, 0))];
        if (u4) {
            if (t < nu / 4) *reinterpret_cast<float4 *>(s_u + 4 * t) = ureg0;
            if (t + RT < nu / 4) *reinterpret_cast<float4 *>(s_u + 4 * (t + RT)) = ureg1;
            for (int u = 4 * UQ * RT + t; u < nu; u += RT) s_u[u] = up[u];
        } else {
            for (int u = t; u < nu; u += RT) s_u[u] = up[u];
        }
        if (t <= i1 - i0) s_hp[t] = hp0 - he0;
        if (t < nhe) { s_hi[t] = hi0 - lo; s_hv[t] = hv0; }
        for (int i = RT + t; i <= i1 - i0; i += RT) s_hp[i] = d.inv_h_ptr[i0 + i] - he0;
        for (int i = RT + t; i < nhe; i += RT) {
            s_hi[i] = d.inv_h_idx[he0 + i] - lo;
            s_hv[i] = d.inv_h_w[he0 + i];
        }
    }
    }
    float *s_nrm = reinterpret_cast<float *>(s_hi + d.max_he);
    if constexpr (EPI) { if (!a.norm_in) particle_norm_to_lds(a.norm_partials, a.norm_parts, plane / a.c, s_nrm); }
    __syncthreads();
    // H adjoint: T[i][o] = sum_{e in inv_h[i]} w_e * u[p_e][o]
    if (ellh) {
        // fixed entry count: the gathers of an item do not wait for a trip count, and items of successive trips overlap
        const bool inc = RT % d.out_w == 0;          // a lane keeps its column, its row advances by RT / out_w per trip
        const int o_inc = threadIdx.x % d.out_w, dii_h = RT / d.out_w;
        int ii_inc = threadIdx.x / d.out_w - dii_h;
#pragma unroll 4
        for (int it = threadIdx.x; it < nrows * d.out_w; it += RT) {
            ii_inc += dii_h;
            const int ii = inc ? ii_inc : it / d.out_w, o = inc ? o_inc : it - ii * d.out_w;
            float wq[EH];
            int iq[EH];
#pragma unroll
            for (int k = 0; k < EH; ++k) { wq[k] = s_hv[k * d.ti + ii]; iq[k] = s_hi[k * d.ti + ii]; }
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < EH; ++k) acc = fmaf(wq[k], s_u[iq[k] * d.out_w + o], acc);    // CSR entry order: same bits
            s_t[ii * d.out_w + o] = acc;
        }
    } else
    for (int it = threadIdx.x; it < (i1 - i0) * d.out_w; it += RT) {
        const int ii = it / d.out_w, o = it - ii * d.out_w;
        const int e0 = s_hp[ii], e1 = s_hp[ii + 1];
        float acc = 0.0f;
        for (int e = e0; e < e1; ++e) acc = fmaf(s_hv[e], s_u[s_hi[e] * d.out_w + o], acc);
        s_t[ii * d.out_w + o] = acc;
    }
    __syncthreads();
    float coef = 0.0f;
    const int n = plane / a.c, ch = plane % a.c;
    if constexpr (EPI) {
        const float nv = a.norm_in ? a.norm_in[n] : s_nrm[0];
        coef = norm_coef_r(nv, a.scale, a.power);
        if (!a.norm_in && a.norm_out && threadIdx.x == 0 && blk == 0 && ch == 0) a.norm_out[n] = nv;
    }
    // W adjoint: g[i][j] = sum_{e in inv_w[j]} w_e * T[i][o_e]
    constexpr int U = VEC ? 4 : 1;
    const int wu = d.in_w / U;
    const int total = (i1 - i0) * wu;
    const float mb = -a.k.b;
    // plane bases once; a unit's offset inside the plane fits 32 bits
    const uint8_t *ipb = EPI ? a.inside_r + (int64_t)plane * ihw : nullptr;
    float *gpb = EPI ? a.g_model_out + ((int64_t)n * 2 * a.c + ch) * ihw : a.out + (int64_t)plane * ihw;
    const float *epb = (EPI && a.g_extra) ? a.g_extra + (int64_t)plane * ihw : nullptr;
    auto unit = [&](const int ii, const int j0, const uchar4 gate4, const bool have_gate) {
        float g[U];
        if (VEC && regtab) {
            if constexpr (VEC) {
                const float *trow = s_t + ii * d.out_w;
                g[0] = g[1] = g[2] = g[3] = 0.0f;
#pragma unroll
                for (int k = 0; k < EW; ++k)
                    if (k < d.ell_w) {          // same entry order as the LDS form: bit-identical sums
                        g[0] = fmaf(tw[k].x, trow[ti4[k].x], g[0]);
                        g[1] = fmaf(tw[k].y, trow[ti4[k].y], g[1]);
                        g[2] = fmaf(tw[k].z, trow[ti4[k].z], g[2]);
                        g[3] = fmaf(tw[k].w, trow[ti4[k].w], g[3]);
                    }
            }
        } else {
#pragma unroll
            for (int q = 0; q < U; ++q) {
                float acc = 0.0f;
                for (int k = 0; k < d.ell_w; ++k)
                    acc = fmaf(s_wv[k * d.in_w + j0 + q], s_t[ii * d.out_w + s_wi[k * d.in_w + j0 + q]], acc);
                g[q] = acc;
            }
        }
        const unsigned o = (unsigned)((i0 + ii) * d.in_w + j0);
        if constexpr (EPI) {
            const uint8_t *ip = ipb + o;
            float *gp = gpb + o;
            const float *ep = epb ? epb + o : nullptr;
            if constexpr (VEC) {
                const uchar4 in = have_gate ? gate4 : *reinterpret_cast<const uchar4 *>(ip);
                float4 ex = make_float4(0, 0, 0, 0);
                if (ep) ex = *reinterpret_cast<const float4 *>(ep);
                float4 r;
                r.x = in.x ? mb * (coef * g[0] + ex.x) : 0.0f;
                r.y = in.y ? mb * (coef * g[1] + ex.y) : 0.0f;
                r.z = in.z ? mb * (coef * g[2] + ex.z) : 0.0f;
                r.w = in.w ? mb * (coef * g[3] + ex.w) : 0.0f;
                *reinterpret_cast<float4 *>(gp) = r;
            } else {
                gp[0] = ip[0] ? mb * (coef * g[0] + (ep ? ep[0] : 0.0f)) : 0.0f;
            }
        } else {
            float *gp = gpb + o;
            if constexpr (VEC) *reinterpret_cast<float4 *>(gp) = make_float4(g[0], g[1], g[2], g[3]);
            else gp[0] = g[0];
        }
    };
    // The clamp gate of a unit is one dependent global load in front of its store: fetched inside the loop, a lane's 4-16
    // units paid 4-16 exposed round trips (the launch moved 67 MB in 33-37 us at N = 64).  With at most GP units per lane
    // (every shipped geometry) all gate words are issued together, ahead of the loop, one register each.
    constexpr int GP = 16;
    if (EPI && VEC && total <= GP * RT) {          // block-uniform
        uchar4 gates[GP];
#pragma unroll
        for (int q = 0; q < GP; ++q) {
            int ii, j0;
            if (regtab) {       // (same incremental indices as the unit loop below)
                const int ii0 = threadIdx.x / wu;
                ii = min(ii0 + q * (RT / wu), i1 - i0 - 1);
                j0 = (threadIdx.x - ii0 * wu) * U;
            } else {
                const int it = min((int)threadIdx.x + q * RT, total - 1);
                ii = it / wu;
                j0 = (it - ii * wu) * U;
            }
            gates[q] = *reinterpret_cast<const uchar4 *>(ipb + (unsigned)((i0 + ii) * d.in_w + j0));
        }
        if (regtab) {           // RT % wu == 0: a lane keeps its column group, its row advances by RT / wu per unit -- no division
            const int ii0 = threadIdx.x / wu, j0 = (threadIdx.x - ii0 * wu) * U, dii = RT / wu;
#pragma unroll
            for (int q = 0; q < GP; ++q)
                if (ii0 + q * dii < i1 - i0) unit(ii0 + q * dii, j0, gates[q], true);
        } else {
#pragma unroll
            for (int q = 0; q < GP; ++q) {
                const int it = threadIdx.x + q * RT;
                if (it < total) unit(it / wu, (it % wu) * U, gates[q], true);
            }
        }
    } else {
        for (int it = threadIdx.x; it < total; it += RT) unit(it / wu, (it % wu) * U, make_uchar4(0, 0, 0, 0), false);
    }
}

// ---------------------------------------------------------------- host
struct ResizeHost {
    ResizeDev dev{};
    std::vector<void *> allocs;
};

template <typename T>
static int upload(ResizeHost *h, const std::vector<T> &v, const T **out)
{
    void *p = nullptr;
    const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    DPSX_HIP_TRY(hipMalloc(&p, bytes));
    h->allocs.push_back(p);
    if (!v.empty()) DPSX_HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T *>(p);
    return DPSX_OK;
}

static void build_inverse(const float *w, const int64_t *idx, int64_t taps, int64_t n_out, int64_t n_in,
                          std::vector<int> &ptr, std::vector<int> &oidx, std::vector<float> &ow)
{
    // CSR by input index; entries ordered by (output index, tap) -> deterministic summation order
    std::vector<std::vector<std::pair<int, float>>> rows((size_t)n_in);
    for (int64_t o = 0; o < n_out; ++o)
        for (int64_t k = 0; k < taps; ++k) {
            const float wt = w[k * n_out + o];
            if (wt == 0.0f) continue;
            rows[(size_t)idx[k * n_out + o]].push_back({(int)o, wt});
        }
    ptr.assign((size_t)n_in + 1, 0);
    for (int64_t i = 0; i < n_in; ++i) ptr[(size_t)i + 1] = ptr[(size_t)i] + (int)rows[(size_t)i].size();
    oidx.clear();
    ow.clear();
    for (auto &r : rows)
        for (auto &e : r) {
            oidx.push_back(e.first);
            ow.push_back(e.second);
        }
}

constexpr size_t kLdsBudget = 96 * 1024;

int resize_create(dpsx_op *op, const float *w_h, const int64_t *i_h, const float *w_w, const int64_t *i_w)
{
    auto *h = new ResizeHost();
    ResizeDev &d = h->dev;
    d.in_h = (int)op->in_h; d.in_w = (int)op->in_w; d.out_h = (int)op->out_h; d.out_w = (int)op->out_w;
    d.taps_h = (int)op->taps_h; d.taps_w = (int)op->taps_w;
    for (int64_t i = 0; i < op->taps_h * op->out_h; ++i)
        if (i_h[i] < 0 || i_h[i] >= op->in_h) { delete h; return DPSX_EINVAL; }
    for (int64_t i = 0; i < op->taps_w * op->out_w; ++i)
        if (i_w[i] < 0 || i_w[i] >= op->in_w) { delete h; return DPSX_EINVAL; }
    std::vector<float> vwh(w_h, w_h + op->taps_h * op->out_h), vww(w_w, w_w + op->taps_w * op->out_w);
    std::vector<int> vih((size_t)(op->taps_h * op->out_h)), viw((size_t)(op->taps_w * op->out_w));
    for (size_t i = 0; i < vih.size(); ++i) vih[i] = (int)i_h[i];
    for (size_t i = 0; i < viw.size(); ++i) viw[i] = (int)i_w[i];
    int rc;
#define UP(vec, field) if ((rc = upload(h, vec, &d.field)) != DPSX_OK) { for (void *p : h->allocs) (void)hipFree(p); delete h; return rc; }
    UP(vwh, w_h) UP(vww, w_w) UP(vih, i_h) UP(viw, i_w)
    std::vector<int> ptr, oi;
    std::vector<float> ow;
    build_inverse(w_h, i_h, op->taps_h, op->out_h, op->in_h, ptr, oi, ow);
    std::vector<int> hp = ptr, hi = oi;
    std::vector<float> hw = ow;
    UP(hp, inv_h_ptr) UP(hi, inv_h_idx) UP(hw, inv_h_w)
    build_inverse(w_w, i_w, op->taps_w, op->out_w, op->in_w, ptr, oi, ow);
    UP(ptr, inv_w_ptr) UP(oi, inv_w_idx) UP(ow, inv_w_w)
    d.nnz_w = (int)ow.size();
    {   // ELL copy of the W inverse: same entry order per column (deterministic), zero padded
        int ell = 1;
        for (int j = 0; j < d.in_w; ++j) ell = std::max(ell, ptr[(size_t)j + 1] - ptr[(size_t)j]);
        std::vector<int> ei((size_t)ell * d.in_w, 0);
        std::vector<float> ew((size_t)ell * d.in_w, 0.0f);
        for (int j = 0; j < d.in_w; ++j)
            for (int e = ptr[(size_t)j]; e < ptr[(size_t)j + 1]; ++e) {
                const int k = e - ptr[(size_t)j];
                ei[(size_t)k * d.in_w + j] = oi[(size_t)e];
                ew[(size_t)k * d.in_w + j] = ow[(size_t)e];
            }
        d.ell_w = ell;
        UP(ei, ell_w_idx) UP(ew, ell_w_w)
    }
    {   // ELL copy of the H inverse (hp / hi / hw: CSR by input row): fixed-width rows let the adjoint's H pass run with a
        // compile-time entry count -- no per-row trip counts, no dependent pointer loads at the head of every workgroup
        int ell = 1;
        for (int i = 0; i < d.in_h; ++i) ell = std::max(ell, hp[(size_t)i + 1] - hp[(size_t)i]);
        std::vector<int> ei((size_t)ell * d.in_h, 0);
        std::vector<float> ew((size_t)ell * d.in_h, 0.0f);
        for (int i = 0; i < d.in_h; ++i) {
            const int e0 = hp[(size_t)i], e1 = hp[(size_t)i + 1];
            for (int k = 0; k < ell; ++k) {
                const bool real = e0 + k < e1;
                ei[(size_t)k * d.in_h + i] = real ? hi[(size_t)(e0 + k)] : (e1 > e0 ? hi[(size_t)e0] : -1);
                ew[(size_t)k * d.in_h + i] = real ? hw[(size_t)(e0 + k)] : 0.0f;
            }
        }
        d.ell_h = ell;
        UP(ei, ell_h_idx) UP(ew, ell_h_w)
    }

    // ---- forward blocking: largest tp (<= 16) whose staged rows fit the LDS budget
    auto fwd_blocking = [&](int tp, std::vector<int> &lo, std::vector<int> &cnt, std::vector<int> &ow_, int &maxrows) {
        const int nblk = (d.out_h + tp - 1) / tp;
        lo.assign((size_t)nblk, 0); cnt.assign((size_t)nblk, 0); ow_.assign((size_t)nblk + 1, 0);
        maxrows = 0;
        for (int b = 0; b <= nblk; ++b) ow_[(size_t)b] = (int)((int64_t)b * d.in_h / nblk);
        for (int b = 0; b < nblk; ++b) {
            int mn = ow_[(size_t)b], mx = ow_[(size_t)b + 1] - 1;  // staged range covers the owned rows too
            for (int p = b * tp; p < std::min(d.out_h, (b + 1) * tp); ++p)
                for (int k = 0; k < d.taps_h; ++k) {
                    const int v = vih[(size_t)k * d.out_h + p];
                    mn = std::min(mn, v);
                    mx = std::max(mx, v);
                }
            lo[(size_t)b] = mn;
            cnt[(size_t)b] = mx - mn + 1;
            maxrows = std::max(maxrows, mx - mn + 1);
        }
        return ((size_t)maxrows * ((d.in_w + 3) / 4 * 4 + d.out_w) + 32 + 2 * (size_t)(d.taps_w * d.out_w + d.taps_h * tp)) * 4;
    };
    int best_tp = 0;
    std::vector<int> blo, bcnt, own;
    for (int tp = 16; tp >= 1; tp >>= 1) {
        int maxrows = 0;
        const size_t lds = fwd_blocking(tp, blo, bcnt, own, maxrows);
        if (lds <= kLdsBudget || tp == 1) {
            best_tp = tp; d.fwd_rows = maxrows;
            if (lds > 150 * 1024) { for (void *p : h->allocs) (void)hipFree(p); delete h; return DPSX_EUNSUPPORTED; }
            break;
        }
    }
    d.tp = best_tp;
    UP(blo, blk_lo) UP(bcnt, blk_cnt) UP(own, own_lo)
    // the finer forward blocking (see ResizeDev): blocks of exactly RT outputs, when the coarse ones are 2 or 4 of them
    d.gparts = 1; d.tp2 = d.tp; d.fwd_rows2 = d.fwd_rows; d.blk_lo2 = d.blk_lo; d.blk_cnt2 = d.blk_cnt; d.own_lo2 = d.own_lo;
    if (RT % d.out_w == 0 && d.out_h % d.tp == 0) {
        const int tp2 = RT / d.out_w;
        if (tp2 >= 1 && d.tp % tp2 == 0 && (d.tp / tp2 == 2 || d.tp / tp2 == 4)) {
            std::vector<int> blo2, bcnt2, own2;
            int maxrows2 = 0;
            (void)fwd_blocking(tp2, blo2, bcnt2, own2, maxrows2);        // fewer staged rows than the coarse one: fits
            d.gparts = d.tp / tp2; d.tp2 = tp2; d.fwd_rows2 = maxrows2;
            UP(blo2, blk_lo2) UP(bcnt2, blk_cnt2) UP(own2, own_lo2)
        }
    }
    // ---- adjoint blocking: the largest ti (<= ti_max) whose staged rows fit the LDS budget
    auto adj_blocking = [&](int ti_max, int &best_ti, int &rows, int &he, std::vector<int> &alo, std::vector<int> &acnt) {
        for (int ti = ti_max; ti >= 1; ti >>= 1) {
            const int nblk = (d.in_h + ti - 1) / ti;
            std::vector<int> lo((size_t)nblk), cnt((size_t)nblk);
            int maxrows = 1;
            for (int b = 0; b < nblk; ++b) {
                int mn = d.out_h, mx = -1;
                for (int i = b * ti; i < std::min(d.in_h, (b + 1) * ti); ++i)
                    for (int e = hp[(size_t)i]; e < hp[(size_t)i + 1]; ++e) {
                        mn = std::min(mn, hi[(size_t)e]);
                        mx = std::max(mx, hi[(size_t)e]);
                    }
                if (mx < mn) { mn = 0; mx = 0; }
                lo[(size_t)b] = mn;
                cnt[(size_t)b] = mx - mn + 1;
                maxrows = std::max(maxrows, mx - mn + 1);
            }
            int max_he = 1;
            for (int b = 0; b < nblk; ++b)
                max_he = std::max(max_he, hp[(size_t)std::min(d.in_h, (b + 1) * ti)] - hp[(size_t)(b * ti)]);
            const size_t lds = ((size_t)(maxrows + ti) * d.out_w + 2 * (size_t)(d.ell_w * d.in_w + max_he) + ti + 8) * 4;
            if (lds <= kLdsBudget || ti == 1) {
                best_ti = ti; rows = maxrows; alo = lo; acnt = cnt; he = max_he;
                return lds <= 150 * 1024;
            }
        }
        return false;
    };
    std::vector<int> alo, acnt, alo2, acnt2;
    if (!adj_blocking(64, d.ti, d.adj_rows, d.max_he, alo, acnt) ||
        !adj_blocking(std::max(1, std::min(16, d.ti / 4)), d.ti2, d.adj_rows2, d.max_he2, alo2, acnt2)) {
        for (void *p : h->allocs) (void)hipFree(p);
        delete h;
        return DPSX_EUNSUPPORTED;
    }
    UP(alo, ablk_lo) UP(acnt, ablk_cnt) UP(alo2, ablk_lo2) UP(acnt2, ablk_cnt2)
#undef UP
    op->d_w_h = reinterpret_cast<float *>(h);  // opaque owner pointer (see resize_destroy)
    return DPSX_OK;
}

void resize_destroy(dpsx_op *op)
{
    auto *h = reinterpret_cast<ResizeHost *>(op->d_w_h);
    if (!h) return;
    for (void *p : h->allocs) (void)hipFree(p);
    delete h;
    op->d_w_h = nullptr;
}

static const ResizeDev &dev_of(const dpsx_op *op) { return reinterpret_cast<const ResizeHost *>(op->d_w_h)->dev; }

int64_t resize_parts_per_particle(const dpsx_op *op, int64_t c)
{
    const ResizeDev &d = dev_of(op);
    return c * ((d.out_h + d.tp - 1) / d.tp) * d.gparts;        // the same slots under either forward blocking
}

// workgroups per particle of the forward launch for `planes` planes (the in-launch tail counts arrivals)
static bool fwd_fine(const ResizeDev &d, int64_t planes)
{
    static const char *force = getenv("DPSX_RESIZE_FWD_BLOCKING");     // A/B switch for tools/kbench.py: coarse | fine
    if (d.gparts == 1) return false;
    if (force && force[0] == 'c') return false;
    if (force && force[0] == 'f') return true;
    // few planes: the coarse blocking would leave CUs without a workgroup (N = 16 at 256 x 256: 192 workgroups on 256
    // CUs).  Measured on MI355X (tools/ab_resize_fwd.sh, x4 at 256 x 256, fused forward, coarse / fine in us): N = 16
    // 22.0 / 20.4 (rocprof), N = 32 32.7 / 35.9, N = 40 45.0 / 44.3, N = 48 49.2 / 52.9 -- the fine blocks re-stage 1.75x
    // the rows, which only pays while the coarse grid is smaller than the chip
    return planes * (d.out_h / d.tp) <= 256;
}

static int64_t resize_fwd_blocks_per_particle(const dpsx_op *op, int64_t c, int64_t planes)
{
    const ResizeDev &d = dev_of(op);
    return c * ((d.out_h + d.tp - 1) / d.tp) * (fwd_fine(d, planes) ? d.gparts : 1);
}

template <typename K>
static int allow_lds(K kernel, bool &done)
{
    if (!done) {
        DPSX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)(144 * 1024)));
        done = true;
    }
    return DPSX_OK;
}

#define RZ_LAUNCH(KERNEL, GRID, LDS, STREAM, ...)                                   \
    do {                                                                             \
        static bool s_done = false;                                                  \
        int rc_ = allow_lds(&KERNEL, s_done);                                        \
        if (rc_ != DPSX_OK) return rc_;                                              \
        hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(RT), LDS, STREAM, __VA_ARGS__);  \
        return check_launch();                                                       \
    } while (0)

template <bool POST, bool RESID>
static int launch_fwd(const dpsx_op *op, const ResizeArgs &a, bool vec, hipStream_t s)
{
    ResizeDev d = dev_of(op);
    if (fwd_fine(d, a.planes)) {        // one partial per block, in the slots the coarse blocks would have filled
        d.tp = d.tp2; d.fwd_rows = d.fwd_rows2; d.blk_lo = d.blk_lo2; d.blk_cnt = d.blk_cnt2; d.own_lo = d.own_lo2;
        d.gparts = 1;
    }
    const unsigned grid = (unsigned)(a.planes * ((d.out_h + d.tp - 1) / d.tp));
    const size_t lds = ((size_t)d.fwd_rows * ((d.in_w + 3) / 4 * 4 + d.out_w) + 32 + 2 * (size_t)(d.taps_w * d.out_w + d.taps_h * d.tp)) * 4;
    const int wu = d.in_w / 4;
    static const bool no_rows = getenv("DPSX_RESIZE_NO_ROWS") != nullptr;        // A/B switch for tools/kbench.py
    if (vec && !no_rows && d.in_w % 4 == 0 && wu >= 1 && wu <= 64 && 64 % wu == 0 && d.out_w <= wu && d.taps_w <= 32) {
        const unsigned grid_x = (unsigned)((a.planes + 7) / 8 * 8 * ((d.out_h + d.tp - 1) / d.tp));
        const size_t lds_r = ((size_t)4 * 256 + (size_t)d.fwd_rows * d.out_w + 16 + 2 * (size_t)(d.taps_h * d.tp)) * 4;
        if (d.taps_w <= 16) RZ_LAUNCH((k_resize_fwd_rows<POST, RESID, 16, 4>), grid_x, lds_r, s, a, d);
        RZ_LAUNCH((k_resize_fwd_rows<POST, RESID, 32, 2>), grid_x, lds_r, s, a, d);
    }
    if (vec) RZ_LAUNCH((k_resize_fwd<POST, RESID, true>), grid, lds, s, a, d);
    RZ_LAUNCH((k_resize_fwd<POST, RESID, false>), grid, lds, s, a, d);
}

template <bool EPI>
static int launch_adj(const dpsx_op *op, const ResizeArgs &a, bool vec, hipStream_t s)
{
    ResizeDev d = dev_of(op);
    // few planes: the coarse blocking would leave most of the 256 CUs without a workgroup (N = 16 at 256 x 256: 192 of
    // them) -- take the fine one
    static const char *force = getenv("DPSX_RESIZE_ADJ_BLOCKING");     // A/B switch for tools/kbench.py: coarse | fine
    static const bool csr = getenv("DPSX_RESIZE_ADJ_CSR") != nullptr;
    d.no_ellh = csr ? 1 : 0;
    const bool fine = force ? force[0] == 'f' : (int64_t)a.planes * ((d.in_h + d.ti - 1) / d.ti) < 2 * 256;
    if (fine && d.ti2 < d.ti) {
        d.ti = d.ti2; d.adj_rows = d.adj_rows2; d.max_he = d.max_he2; d.ablk_lo = d.ablk_lo2; d.ablk_cnt = d.ablk_cnt2;
    }
    const unsigned grid = (unsigned)(a.planes * ((d.in_h + d.ti - 1) / d.ti));
    const size_t lds = ((size_t)(d.adj_rows + d.ti) * d.out_w + 2 * (size_t)(d.ell_w * d.in_w + d.max_he) + d.ti + 8) * 4;
    if (vec) RZ_LAUNCH((k_resize_adj<EPI, true>), grid, lds, s, a, d);
    RZ_LAUNCH((k_resize_adj<EPI, false>), grid, lds, s, a, d);
}

static bool rz_vec(const dpsx_op *op, std::initializer_list<const void *> ptrs)
{
    if (op->in_w % 4 != 0) return false;
    for (const void *p : ptrs)
        if (p && !aligned16(p)) return false;
    return true;
}

int resize_forward(const dpsx_op *op, const float *x, float *y, int64_t planes, hipStream_t s)
{
    if (planes == 0) return DPSX_OK;
    ResizeArgs a{};
    a.x = x; a.out = y; a.c = 1; a.planes = (int)planes;
    return launch_fwd<false, false>(op, a, rz_vec(op, {x}), s);
}

int resize_adjoint(const dpsx_op *op, const float *u, float *g, int64_t planes, hipStream_t s)
{
    if (planes == 0) return DPSX_OK;
    ResizeArgs a{};
    a.x = u; a.out = g; a.c = 1; a.planes = (int)planes;
    return launch_adj<false>(op, a, rz_vec(op, {g}), s);
}

int resize_step_fwd(const dpsx_op *op, const StepFwdArgs &f, hipStream_t s)
{
    if (f.n == 0) return DPSX_OK;
    ResizeArgs a{};
    a.x_t = f.x_t; a.model_out = f.model_out; a.noise = f.noise; a.x0_hat = f.x0_hat; a.sample = f.sample;
    a.inside_w = f.inside; a.y = f.y; a.y_n = (int)f.y_n; a.out = f.resid; a.partials = f.partials;
    a.c = (int)f.c; a.planes = (int)(f.n * f.c); a.k = f.k;
    a.tail = f.tail;
    a.tail.blocks_per_particle = (int)resize_fwd_blocks_per_particle(op, f.c, f.n * f.c);
    const bool vec = rz_vec(op, {f.x_t, f.model_out, f.noise, f.x0_hat, f.sample}) &&
                     (reinterpret_cast<uintptr_t>(f.inside) & 3u) == 0;
    return launch_fwd<true, true>(op, a, vec, s);
}

int resize_step_bwd(const dpsx_op *op, const StepBwdArgs &b, hipStream_t s)
{
    if (b.n == 0) return DPSX_OK;
    ResizeArgs a{};
    a.x = b.resid; a.norm_in = b.norm; a.norm_partials = b.partials; a.norm_parts = b.parts; a.norm_out = b.norm_out;
    a.inside_r = b.inside; a.g_extra = b.g_extra; a.g_model_out = b.g_model_out;
    a.scale = b.scale; a.power = b.power; a.c = (int)b.c; a.planes = (int)(b.n * b.c); a.k = b.k;
    const bool vec = rz_vec(op, {b.g_model_out, b.g_extra}) && (reinterpret_cast<uintptr_t>(b.inside) & 3u) == 0;
    return launch_adj<true>(op, a, vec, s);
}

int resize_score(const dpsx_op *op, const float *x, const float *y, int64_t y_n, float *partials, int64_t n,
                 int64_t c, int l1, const Tail &tail, hipStream_t s)
{
    if (n == 0) return DPSX_OK;
    ResizeArgs a{};
    a.x = x; a.y = y; a.y_n = (int)y_n; a.out = nullptr; a.partials = partials; a.c = (int)c;
    a.planes = (int)(n * c);
    a.l1 = l1;
    a.tail = tail;
    a.tail.blocks_per_particle = (int)resize_fwd_blocks_per_particle(op, c, n * c);
    return launch_fwd<false, true>(op, a, rz_vec(op, {x}), s);
}

}  // namespace dpsx
