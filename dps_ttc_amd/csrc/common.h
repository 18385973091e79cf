// dpsx internal helpers (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <initializer_list>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dpsx.h"

namespace dpsx {

constexpr int kWave = 64;

// -------------------------------------------------------------------- errors
void set_last_hip_error(hipError_t e);

inline int check_launch()
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_hip_error(e);
        return DPSX_ELAUNCH;
    }
    return DPSX_OK;
}

#define DPSX_HIP_TRY(expr)                                   \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) {                              \
            ::dpsx::set_last_hip_error(_e);                  \
            return _e == hipErrorOutOfMemory ? DPSX_ENOMEM : DPSX_ELAUNCH; \
        }                                                    \
    } while (0)

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// -------------------------------------------------------------------- S1 point math
// Reference op order, one rounding per ATen op (no FMA contraction), so the
// element-wise chain is bit-identical to torch eager:
//   posterior_mean_variance.py:120-123 (x0 from eps), :43-44 (clamp),
//   :116-118 (mean), :239-240 (log-variance), gaussian_diffusion.py:472-474.
struct Coefs {
    float a, b, c1, c2, min_log, max_log;
    int add_noise;
};

inline Coefs to_coefs(const dpsx_coefs *c)
{
    return Coefs{c->a, c->b, c->c1, c->c2, c->min_log, c->max_log, c->add_noise};
}

__device__ __forceinline__ float post_x0(float x, float e, const Coefs &c, bool &inside)
{
    float pre = __fsub_rn(__fmul_rn(c.a, x), __fmul_rn(c.b, e));
    inside = (pre >= -1.0f) && (pre <= 1.0f);
    return pre < -1.0f ? -1.0f : (pre > 1.0f ? 1.0f : pre);
}

__device__ __forceinline__ float post_logvar(float v, const Coefs &c)
{
    float frac = __fmul_rn(__fadd_rn(v, 1.0f), 0.5f);   // == (v + 1) / 2 bit for bit (exact scaling by 2^-1)
    return __fadd_rn(__fmul_rn(frac, c.max_log), __fmul_rn(__fsub_rn(1.0f, frac), c.min_log));
}

// add_noise: bit 0 = add the noise term (t != 0); bit 1 = DDIM step (gaussian_diffusion.py:479-509) with
// c1 = sqrt(abar_prev), c2 = sqrt(1 - abar_prev - sigma^2), min_log = sigma; eps is re-derived from the
// clamped x0_hat as predict_eps_from_x_start does (:506-509), one rounding per reference op.
__device__ __forceinline__ float post_sample(float x, float x0, float v, float z, const Coefs &c)
{
    if (c.add_noise & 2) {
        const float eps = __fdiv_rn(__fsub_rn(__fmul_rn(c.a, x), x0), c.b);
        const float mean = __fadd_rn(__fmul_rn(x0, c.c1), __fmul_rn(c.c2, eps));
        return (c.add_noise & 1) ? __fadd_rn(mean, __fmul_rn(c.min_log, z)) : mean;
    }
    float mean = __fadd_rn(__fmul_rn(c.c1, x0), __fmul_rn(c.c2, x));
    if (!c.add_noise) return mean;
    // exp(x) = 2^(x log2 e) on the transcendental unit (v_exp_f32); |x| <= ~10 here, error ~4e-7 relative
    float sd = __expf(__fmul_rn(0.5f, post_logvar(v, c)));
    return __fadd_rn(mean, __fmul_rn(sd, z));
}

// The same arithmetic on one float4 unit, two lanes of data per instruction: v_pk_mul_f32 / v_pk_add_f32 round each
// operation exactly as their scalar forms do (no contraction: -ffp-contract=off; scalar x vector broadcasts), so x0_hat,
// the clamp gate and the sample come out bit for bit as from post_x0 / post_sample -- at about half the vector
// instructions (the tap-list forward launch spent more of them on S1 and its loader than on its tap loop:
// profiles/r03_valu_taps.txt).  The gate test is |pre| <= 1: one compare with the abs modifier, false for NaN like the
// closed-interval pair.  DDIM records (a division per element) take the scalar forms.
typedef float s1v2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float s1_clamp(float pre) { return pre < -1.0f ? -1.0f : (pre > 1.0f ? 1.0f : pre); }

// (the coefficients are splat into explicit pairs from by-value scalars: taking them through the Coefs reference made the
// compiler keep the struct in scratch memory and reload it -- behind a vmcnt(0) that also drained the tile's loads)
__device__ __forceinline__ s1v2 s1_splat(float v) { return s1v2{v, v}; }
// a launch-uniform scalar, made opaque: without this the compiler turns a pair built from two adjacent Coefs fields into ONE
// 8-byte load of the struct -- which forces the by-value kernel argument into scratch memory (28 bytes per lane, reloaded
// behind s_waitcnt vmcnt(0), which also drains the tile's loads in flight: +11 us on the tap-list forward launch)
__device__ __forceinline__ float s1_uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ s1v2 s1_lo(s1v2 p) { return __builtin_shufflevector(p, p, 0, 0); }   // op_sel: no instruction
__device__ __forceinline__ s1v2 s1_hi(s1v2 p) { return __builtin_shufflevector(p, p, 1, 1); }

// x0_hat of a unit (halo units: no sample, no gate)
__device__ __forceinline__ float4 post_x0_unit(const float4 &x, const float4 &e, const float ca, const float cb)
{
    const s1v2 AB{s1_uniform(ca), s1_uniform(cb)};
    const s1v2 p0 = s1_lo(AB) * s1v2{x.x, x.y} - s1_hi(AB) * s1v2{e.x, e.y}, p1 = s1_lo(AB) * s1v2{x.z, x.w} - s1_hi(AB) * s1v2{e.z, e.w};
    return make_float4(s1_clamp(p0.x), s1_clamp(p0.y), s1_clamp(p1.x), s1_clamp(p1.y));
}
__device__ __forceinline__ float4 post_x0_unit(const float4 &x, const float4 &e, const Coefs &c)
{
    return post_x0_unit(x, e, c.a, c.b);
}

__device__ __forceinline__ void post_unit(const float4 &x, const float4 &e, const float4 &v, const float4 &z, const Coefs &c,
                                          float4 &x0, float4 &sm, uchar4 &gate)
{
    // the six coefficients as three register pairs; a packed instruction broadcasts either half of a pair (op_sel)
    const s1v2 AB{s1_uniform(c.a), s1_uniform(c.b)}, CC{s1_uniform(c.c1), s1_uniform(c.c2)}, LL{s1_uniform(c.max_log), s1_uniform(c.min_log)};
    const int mode = c.add_noise;
    const s1v2 xa{x.x, x.y}, xb{x.z, x.w};
    const s1v2 p0 = s1_lo(AB) * xa - s1_hi(AB) * s1v2{e.x, e.y}, p1 = s1_lo(AB) * xb - s1_hi(AB) * s1v2{e.z, e.w};
    gate = make_uchar4(fabsf(p0.x) <= 1.0f, fabsf(p0.y) <= 1.0f, fabsf(p1.x) <= 1.0f, fabsf(p1.y) <= 1.0f);
    x0 = make_float4(s1_clamp(p0.x), s1_clamp(p0.y), s1_clamp(p1.x), s1_clamp(p1.y));
    if (mode & 2) {                 // DDIM: launch-uniform
        sm = make_float4(post_sample(x.x, x0.x, v.x, z.x, c), post_sample(x.y, x0.y, v.y, z.y, c),
                         post_sample(x.z, x0.z, v.z, z.z, c), post_sample(x.w, x0.w, v.w, z.w, c));
        return;
    }
    const s1v2 m0 = s1_lo(CC) * s1v2{x0.x, x0.y} + s1_hi(CC) * xa, m1 = s1_lo(CC) * s1v2{x0.z, x0.w} + s1_hi(CC) * xb;
    if (!mode) {
        sm = make_float4(m0.x, m0.y, m1.x, m1.y);
        return;
    }
    const s1v2 one = s1_splat(1.0f), hlf = s1_splat(0.5f);
    const s1v2 f0 = (s1v2{v.x, v.y} + one) * hlf, f1 = (s1v2{v.z, v.w} + one) * hlf;
    const s1v2 h0 = hlf * (f0 * s1_lo(LL) + (one - f0) * s1_hi(LL)), h1 = hlf * (f1 * s1_lo(LL) + (one - f1) * s1_hi(LL));
    const s1v2 s0 = m0 + s1v2{__expf(h0.x), __expf(h0.y)} * s1v2{z.x, z.y};
    const s1v2 s1 = m1 + s1v2{__expf(h1.x), __expf(h1.y)} * s1v2{z.z, z.w};
    sm = make_float4(s0.x, s0.y, s1.x, s1.y);
}

// -------------------------------------------------------------------- reductions
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    return v;
}

// Deterministic block sum (fixed shuffle tree, waves added in index order).
// Result valid in thread 0.  `scratch` holds >= blockDim.x/64 floats.
__device__ __forceinline__ float block_sum(float v, float *scratch)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[wv] = v;
    __syncthreads();
    float t = 0.0f;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) t += scratch[i];
    return t;
}

// Per-particle norm from the per-block partial sums of squares the forward half left behind.  Same
// order as k_finalize_norm (lane-strided sums, fixed shuffle tree, double), so every block of a particle
// -- and the stand-alone finalisation -- get the bit-identical value.  Called by all threads; the result is
// valid in *slot after the next __syncthreads().
__device__ __forceinline__ void particle_norm_to_lds(const float *partials, int parts, int64_t n, float *slot)
{
    if (threadIdx.x < kWave) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < parts; i += kWave) acc += (double)partials[n * parts + i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, kWave);
        if (threadIdx.x == 0) *slot = (float)sqrt(acc);
    }
}

// The same value in two halves, so the partial loads fly together with the caller's tile loads instead of
// costing a memory round trip at the top of every block: *_issue starts them (wave 0, <= 4 per lane, fixed
// slots), *_reduce adds them in the order of the loop above (adding the +0.0 of an absent slot is exact).
// Only for parts <= 4 * kWave (block-uniform test by the caller).
struct NormPartials { float v[4]; };
__device__ __forceinline__ NormPartials particle_norm_issue(const float *partials, int parts, int64_t n)
{
    NormPartials p;
    const float *base = partials + n * parts;
#pragma unroll
    for (int j = 0; j < 4; ++j)      // every lane loads (clamped address): no branch, nothing waits here
        p.v[j] = base[min((int)(threadIdx.x & (kWave - 1)) + j * kWave, parts - 1)];
    return p;
}
__device__ __forceinline__ void particle_norm_reduce(const NormPartials &p, int parts, float *slot)
{
    if (threadIdx.x < kWave) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += (int)threadIdx.x + j * kWave < parts ? (double)p.v[j] : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, kWave);
        if (threadIdx.x == 0) *slot = (float)sqrt(acc);
    }
}

// -------------------------------------------------------------------- select (torch.argmin order)
// (value, index) pairs under a total order -- NaN before everything, then the smaller value, then the smaller
// index -- so the result does not depend on the reduction shape.
struct ArgMin {
    float v;
    int64_t i;   // -1: empty
};

__device__ __forceinline__ bool argmin_better(const ArgMin &a, const ArgMin &b)   // a strictly before b
{
    if (a.i < 0) return false;
    if (b.i < 0) return true;
    const bool an = a.v != a.v, bn = b.v != b.v;
    if (an != bn) return an;
    if (an) return a.i < b.i;
    return a.v < b.v || (a.v == b.v && a.i < b.i);
}

// -------------------------------------------------------------------- "last block done" tail
// A launch whose blocks each leave one partial sum per (particle, slot) can finish the per-particle reduction
// itself: every partial-writing block arrives at its particle's counter; the last one re-sums ALL of that particle's
// partials in the fixed order of k_finalize_norm (so the value is bit-identical to the stand-alone finalisation and
// independent of which block came last) and writes the per-particle value; optionally the last particle to finish
// runs the argmin over all values.  Nothing waits on anything: a block that is not last just leaves.  Counters are
// zero between launches (the last arriver resets its counter; the host wrapper clears them if a launch fails).  One launch
// at a time per counter array (= per dpsx_op): an operator handle serves ONE stream at a time, as its workspace does.
enum { TAIL_L2 = 0,      // value = sqrt(sum of squares)                      ||y - A x||_2
       TAIL_L1SQ = 1 };  // value = (sum of |.|)^2 * l1_scale                 ||y - A x||_1^2 / (C H W)
enum { POT_NONE = 0, POT_MEAN = 1, POT_MIN = 2, POT_DIFF = 3, POT_CURR = 4 };   // SearchDDPM.resample_update :565-585

struct Tail {
    unsigned *counters = nullptr;   // [1 + n]: [0] particles finished, [1 + p] blocks of particle p arrived; null: no tail
    int blocks_per_particle = 0;
    const float *partials = nullptr;
    int parts = 0;
    int mode = TAIL_L2;
    float l1_scale = 0.0f;
    const float *prev = nullptr;    // [n] previous costs (nullable) and how to combine them with the new value
    int potential = POT_NONE;
    float *raw_out = nullptr;       // [n] the uncombined value (nullable)
    float *out = nullptr;           // [n]
    int64_t *best_idx = nullptr;    // argmin over out[0..n) (nullable)
    float *best_val = nullptr;      // out[argmin] (nullable)
    int n = 0;
};

constexpr int kTailMaxParticles = 1 << 16;    // counters allocated per operator handle

// Visibility across the 8 XCDs (one L2 each) WITHOUT a device-scope release fence: on gfx950 that fence is a write-back
// of the whole L2 (`buffer_wbl2`), and with megabytes of freshly written x0_hat / sample lines dirty in it, one fence per
// block cost 535 us per launch (measured, N = 64).  The hand-off is instead the write-through form MI355X_MICROARCH.md
// lists as valid and measured ("Valid forms", first table row): every handed-off word -- the partial sums, the finished
// values -- is stored `sc1` (agent-scope atomic store: write-through past the XCD's L2) by ONE lane, which drains its
// stores with an `asm volatile("s_waitcnt vmcnt(0)" ::: "memory")` -- the asm form, because the compiler may drop or move a
// builtin wait, and its "memory" clobber keeps the store and the counter add in program order -- before ONE agent-scope
// atomic add signals for the block; the block whose add returned last (N blocks per launch) takes an agent-scope acquire
// (an L2 invalidate, no write-back) and reads every word back with `sc1` loads (agent-scope atomic loads).
// Not the C++ memory model's release / acquire pair (there is no release on the writer side by design: it IS the 535 us
// write-back); it relies on the documented gfx950 behaviour of sc1 stores, which is why this library targets gfx950 only.
// The default loops do not take this path (the norm is finalised in the backward launch's prologue, the costs by
// k_finalize_select); it serves `dpsx_step_fwd_f32(norm != NULL)`.
__device__ __forceinline__ float tail_ld(const float *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a partial sum: read by a later launch (plain store: launch boundaries order it) or, with an in-launch tail, by the last
// block of its particle (write-through store)
__device__ __forceinline__ void tail_publish(float *p, float v, bool in_launch_reader)
{
    if (in_launch_reader) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ __forceinline__ void tail_drain_stores()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Called by ALL threads of every block that wrote a partial of `particle`, after the write (block-uniform call site).
__device__ __forceinline__ void tail_arrive(const Tail &t, int particle)
{
    if (!t.counters) return;                                    // launch-uniform
    __shared__ __attribute__((aligned(16))) int s_tail[52];     // flag + 16 x (value, index lo, index hi); 16-byte multiple (G17)
    int *s_flag = s_tail;
    if (threadIdx.x == 0) {                                     // the thread that published the partial
        tail_drain_stores();                                    // ... whose write-through store has completed
        const unsigned prev = __hip_atomic_fetch_add(&t.counters[1 + particle], 1u, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == (unsigned)t.blocks_per_particle - 1u;
        if (last) __hip_atomic_store(&t.counters[1 + particle], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = last;
    }
    __syncthreads();
    if (!*s_flag) return;                                       // block-uniform
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // invalidate, no write-back; N blocks per launch
    if (threadIdx.x < kWave) {
        double acc = 0.0;
        const float *pp = t.partials + (int64_t)particle * t.parts;
        for (int i = threadIdx.x; i < t.parts; i += kWave) acc += (double)tail_ld(pp + i);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, kWave);
        if (threadIdx.x == 0) {
            float v = t.mode == TAIL_L1SQ ? (float)(acc * acc * (double)t.l1_scale) : (float)sqrt(acc);
            if (t.raw_out) t.raw_out[particle] = v;
            if (t.prev) {
                const float q = t.prev[particle];
                if (t.potential == POT_MEAN) v = v + q;
                else if (t.potential == POT_MIN) v = (v != v || q != q) ? __builtin_nanf("") : fminf(v, q);   // torch.min propagates NaN
                else if (t.potential == POT_DIFF) v = v - q;
            }
            __hip_atomic_store(t.out + particle, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!t.best_idx) return;                                    // launch-uniform
    __syncthreads();
    if (threadIdx.x == 0) {                                     // the thread that stored out[particle]
        tail_drain_stores();
        const unsigned prev = __hip_atomic_fetch_add(&t.counters[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == (unsigned)t.n - 1u;
        if (last) __hip_atomic_store(&t.counters[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = last;
    }
    __syncthreads();
    if (!*s_flag) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // argmin over out[0..n): per-lane scan, wave shuffles, then thread 0 over the waves' winners (one LDS hop)
    ArgMin best{0.0f, -1};
    for (int64_t i = threadIdx.x; i < t.n; i += blockDim.x) {
        const ArgMin c{tail_ld(t.out + i), i};
        if (argmin_better(c, best)) best = c;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        ArgMin c;
        c.v = __shfl_down(best.v, o, kWave);
        c.i = __shfl_down(best.i, o, kWave);
        if (argmin_better(c, best)) best = c;
    }
    float *s_v = reinterpret_cast<float *>(s_flag + 1);
    int *s_lo = s_flag + 17, *s_hi = s_flag + 33;
    const int wave = threadIdx.x / kWave, nw = (blockDim.x + kWave - 1) / kWave;
    __syncthreads();
    if (threadIdx.x % kWave == 0) {
        s_v[wave] = best.v;
        s_lo[wave] = (int)(best.i & 0xffffffff);
        s_hi[wave] = (int)(best.i >> 32);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) {
            const ArgMin c{s_v[w], ((int64_t)s_hi[w] << 32) | (uint32_t)s_lo[w]};
            if (argmin_better(c, best)) best = c;
        }
        *t.best_idx = best.i < 0 ? 0 : best.i;
        if (t.best_val) *t.best_val = best.v;
    }
}

// ReflectionPad2d index map (no edge repeat); valid while |overhang| < n.
__device__ __forceinline__ int reflect_idx(int i, int n)
{
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

}  // namespace dpsx

// -------------------------------------------------------------------- op object
enum { OP_TAPS = 0, OP_SEP = 1, OP_RESIZE = 2, OP_MASK = 3, OP_IDENT = 4, OP_PHASE = 5 };

constexpr int kMaxRadius = 32;  // kernel side <= 65

// L (4 or 2) vertically consecutive taps of one kernel column (zero padded): out[r][c] += sum_j w[j] * in[r + dy0 + j][c + dx].
// A lane that owns 8 rows x 2 adjacent columns reads the (8 + L - 1) x 2 inputs of a run once, as 8-byte LDS reads,
// and issues 8 L packed FMAs on them (blur.hip: tap_runs_pk).  dy0 is shifted so that dy0 .. dy0 + L - 1 stays inside
// the staged halo.  Runs are stored grouped by class: [even dx, L=4 | even dx, L=2 | odd dx, L=4 | odd dx, L=2].
struct TapRun {
    int dy0, dx, pad0, pad1;
    float w[4];
};

struct SepTaps {  // passed by value -> SGPRs
    float h[2 * kMaxRadius + 1];
    float v[2 * kMaxRadius + 1];
};

struct dpsx_op {
    int kind = OP_IDENT;
    // ---- blur
    int ks = 0, radius = 0;               // radius = ks/2 = the reflection pad
    int reach = 0, radius4 = 0;           // reach: largest |offset| of a non-zero tap; radius4: reach rounded up to 4
    SepTaps sep{};                        // centred at index radius4 (zero padded)
    int nnz = 0;
    int *d_tap_dy = nullptr, *d_tap_dx = nullptr;  // sparse non-zero taps, row-major order
    float *d_tap_w = nullptr;
    // the same taps as vertical runs of <= 4 (TapRun records): what the tap-list kernels iterate over
    void *d_runs_fwd = nullptr, *d_runs_adj = nullptr;
    int nruns = 0;
    int nrun[4] = {0, 0, 0, 0};           // runs per class (even/odd dx) x (4/2 taps), in table order
    // the ADJOINT table is sorted by dx, descending, inside each class: its first nrun_adj_pos[c] runs have dx >= 1 (the only
    // ones a left-border strip can use), its last nrun_adj_neg[c] have dx <= -1 (right border)
    int nrun_adj_pos[4] = {0, 0, 0, 0}, nrun_adj_neg[4] = {0, 0, 0, 0};
    // halo the tap list actually needs on each side of a tile (rows as they are, columns rounded up to 4): a motion
    // path usually leaves the centre in one direction, so this is about half of radius4 per axis
    int halo_t = 0, halo_b = 0, halo_l = 0, halo_r = 0;
    // ---- resize
    int64_t in_h = 0, in_w = 0, out_h = 0, out_w = 0, taps_h = 0, taps_w = 0;
    float *d_w_h = nullptr;  // resize.hip keeps its table owner (ResizeHost*) here
    // ---- mask
    const float *mask = nullptr;
    // ---- "last block done" arrival counters (see Tail): [1 + kTailMaxParticles], zero between launches
    unsigned *d_counters = nullptr;
    // ---- phase
    int64_t pr_h = 0, pr_pad = 0, pr_planes = 0;
    void *fft_plan = nullptr;  // hipfftHandle stored as integer
    bool has_plan = false;
};

// kernels implemented across the .hip files (all enqueue on `s`, return a DPSX_* code)
namespace dpsx {

struct StepFwdArgs {
    const float *x_t, *model_out, *noise, *y;
    int64_t y_n;
    float *x0_hat, *sample;
    uint8_t *inside;
    float *resid;
    float *partials;  // [n * parts_per_particle] sums of squares, finalized into norm by the caller
    int64_t n, c, h, w;
    Coefs k;
    Tail tail{};      // per-particle finalisation inside the launch (tail.counters == nullptr: not requested / not supported)
};

struct StepBwdArgs {
    const float *resid, *norm;   // norm == nullptr: derive it from `partials` (parts per particle) and write norm_out
    const float *partials;
    int parts;
    float *norm_out;
    const uint8_t *inside;
    const float *x0_hat, *y;
    int64_t y_n;
    float scale;
    int power;
    float *g_model_out;
    int64_t n, c, h, w;
    Coefs k;
    const float *g_extra = nullptr;   // optional extra cotangent on x0_hat [n, c, h, w], added before the clamp gate
};

// blur.hip
int blur_forward(const dpsx_op *op, const float *x, float *y, int64_t planes, int64_t h, int64_t w, hipStream_t s);
int blur_adjoint(const dpsx_op *op, const float *u, float *g, int64_t planes, int64_t h, int64_t w, float *scratch,
                 int64_t scratch_bytes, hipStream_t s);
int64_t blur_adjoint_scratch_bytes(const dpsx_op *op, int64_t planes, int64_t h, int64_t w);
int blur_step_fwd(const dpsx_op *op, const StepFwdArgs &a, hipStream_t s);
int blur_step_bwd(const dpsx_op *op, const StepBwdArgs &a, float *scratch, int64_t scratch_bytes, hipStream_t s);
// l1: partials are sums of |r| instead of r^2;  tail: finish the reduction inside the launch (see Tail)
int blur_score(const dpsx_op *op, const float *x, const float *y, int64_t y_n, float *partials,
               int64_t n, int64_t c, int64_t h, int64_t w, int l1, const Tail &tail, hipStream_t s);
int64_t blur_parts_per_particle(const dpsx_op *op, int64_t c, int64_t h, int64_t w);


// resize.hip
int resize_create(dpsx_op *op, const float *w_h, const int64_t *i_h, const float *w_w, const int64_t *i_w);
void resize_destroy(dpsx_op *op);
int64_t resize_parts_per_particle(const dpsx_op *op, int64_t c);
int resize_forward(const dpsx_op *op, const float *x, float *y, int64_t planes, hipStream_t s);
int resize_adjoint(const dpsx_op *op, const float *u, float *g, int64_t planes, hipStream_t s);
int resize_step_fwd(const dpsx_op *op, const StepFwdArgs &a, hipStream_t s);
int resize_step_bwd(const dpsx_op *op, const StepBwdArgs &a, hipStream_t s);
int resize_score(const dpsx_op *op, const float *x, const float *y, int64_t y_n, float *partials,
                 int64_t n, int64_t c, int l1, const Tail &tail, hipStream_t s);

// elementwise.hip
// one_state: x [1, chw] and mo [1, 2 chw] feed all n particles (z, x0, sample, inside stay per particle)
int posterior_fwd(const float *x, const float *mo, const float *z, float *x0, float *sample, uint8_t *inside,
                  int64_t n, int64_t chw, const Coefs &k, hipStream_t s, bool one_state = false);
int posterior_bwd(const float *g_x0, const float *g_s, const float *x, const float *mo, const float *z,
                  float *g_x, float *g_mo, int64_t n, int64_t chw, const Coefs &k, hipStream_t s);
int mask_mul(const float *x, const float *mask, float *y, int64_t planes, int64_t hw, hipStream_t s);
// sums of squares of (y - ax) per particle in `parts` chunks -> partials[n*parts]; r optional
int residual_partials(const float *y, int64_t y_n, const float *ax, float *r, float *partials,
                      int64_t n, int64_t m, int parts, hipStream_t s, int l1 = 0, const Tail &tail = Tail{},
                      const float *mask = nullptr, int64_t hw = 0);
int finalize_norm(const float *partials, int parts, float *norm, int64_t n, hipStream_t s);
// one small launch: per-particle values from the partials (t.partials / parts / mode / prev / potential -> raw_out, out)
// and, if t.best_idx, the torch.argmin-order select over them (t.counters is not used)
int finalize_select(const Tail &t, hipStream_t s);
int finalize_select_copy(const Tail &t, const float *src, float *dst, int64_t chw, hipStream_t s);
int norm_bwd(const float *r, const float *norm, const float *g_norm, int power, float *g_ax,
             int64_t n, int64_t m, hipStream_t s);
// g_model_out[:, :c] = -b * (inside ? coef_p * g_x0 : 0),  g_x0 = A^T r
int clamp_scale_to_eps(const float *g_x0, const float *norm, const uint8_t *inside, float scale, int power,
                       float *g_model_out, int64_t n, int64_t chw, const Coefs &k, hipStream_t s,
                       const float *g_extra = nullptr);
int step_update(const float *sample, const float *g_mo, const float *g_unet, float *x_next,
                int64_t n, int64_t chw, const Coefs &k, hipStream_t s);
int plain_update(const float *sample, const float *ga, const float *gb, float *out, int64_t count, hipStream_t s);
int mask_step_fwd(const dpsx_op *op, const StepFwdArgs &a, int parts, hipStream_t s);
int mask_step_bwd(const dpsx_op *op, const StepBwdArgs &a, hipStream_t s);
int argmin_f32(const float *v, int64_t n, int64_t *idx, float *val, hipStream_t s);
int gather_f32(const float *src, const int64_t *ids, float *dst, int64_t n_out, int64_t n_src, int64_t chw,
               bool replicate, hipStream_t s);
int pack_champion(const float *particles, const float *costs, const int64_t *best, const float *val, float *out, int64_t n,
                  int64_t chw, hipStream_t s);
int select_champion(const float *table, int world, int64_t chw, float *dst, int64_t n_out, int64_t *win_rank,
                    int64_t *win_local, hipStream_t s);

// phase.hip
int phase_create(dpsx_op *op);
void phase_destroy(dpsx_op *op);
int64_t phase_workspace_bytes(const dpsx_op *op, int64_t planes);
// amp = |F(pad(x))| ; spec (optional) receives the centred complex spectrum, interleaved re/im
int phase_forward(dpsx_op *op, const float *x, float *amp, float *spec, int64_t planes,
                  void *ws, int64_t ws_bytes, hipStream_t s);
// g = Re crop F^H (u * z/|z|), z = spectrum at x (recomputed)
int phase_adjoint(dpsx_op *op, const float *u, const float *x, float *g, int64_t planes,
                  void *ws, int64_t ws_bytes, hipStream_t s);
// fused step halves: fwd leaves w = (y-|z|) z/|z| (complex) in resid_c and the sums of squares in partials;
// bwd turns resid_c into g_x0 = Re crop F^H w (unit cotangent) -- caller applies coef / clamp / -b
int phase_step_fwd(dpsx_op *op, const StepFwdArgs &f, float *resid_c, hipStream_t s);   // S1 included
bool phase_vec4_ok(const dpsx_op *op);
bool phase_is_spectral(const dpsx_op *op);
bool phase_norm_in_bwd(const dpsx_op *op);   // the backward launch finalises the norm from the partial sums itself
int phase_step_bwd_fused(dpsx_op *op, float *resid_c, const StepBwdArgs &b, hipStream_t s);
int phase_step_bwd(dpsx_op *op, float *resid_c, float *g_x0, int64_t planes, hipStream_t s);
int64_t phase_parts_per_particle(const dpsx_op *op, int64_t c);
int64_t phase_step_resid_bytes(const dpsx_op *op, int64_t planes);

}  // namespace dpsx
