// Hand-written spectral step of phase retrieval for s = 384 (256 x 256 images, oversample 2.0): three passes instead
// of S1 + staging + a four-pass library R2C + pointwise + a four-pass library C2R + crop.
//
//   A  k_pr_rows_fwd : per image row: S1 (x0_hat, sample, gate out), zero-pad + ifftshift, 384-point FFT of the row,
//                      bins 0..192 out (Hermitian half: the row is real).  Only the 256 non-zero rows exist.
//   B  k_pr_cols     : per (plane, 16-column tile) in LDS: 384-point FFT down the columns, modulus / residual against
//                      y for the frequency and its mirror / sums of squares / Hermitian part of the cotangent, then
//                      the inverse transform of the columns in place.  The per-particle coefficient -scale/norm is a
//                      scalar, so it moves past the (linear) inverse transforms into pass C.
//   C  k_pr_rows_inv : per image row: Hermitian extension of the 193 bins, inverse 384-point FFT, the 256 image
//                      columns, x coef / s, + optional extra cotangent, clamp gate, -b  ->  g_model_out.
//
// The FFT is an in-place decimation-in-frequency network with radices 4 4 4 2 3; its output sits in digit-reversed
// order, which pass B never undoes: the pointwise stage maps a position to its frequency, and the inverse runs the
// exact mirror of the network (conjugate twiddles before each inverse butterfly, stages reversed), which takes the
// digit-reversed order back to natural.  Passes A and C move between natural and digit-reversed order while copying
// between LDS and global memory.  Twiddles: one table of exp(-2 pi i t / 384), computed in double on the host.
#pragma once

namespace prfft {

constexpr int N = 384, HS = 193, IMG = 256, PADW = 64, HALF = 192, CT = 16;

// complex values as 2-vectors: additions are one v_pk_add_f32, products a v_pk_mul_f32 + v_pk_fma_f32 pair
typedef float cf __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf cmul(cf a, cf b)    // a * b
{
    return __builtin_elementwise_fma(cf{a.y, a.y}, cf{-b.y, b.x}, cf{a.x, a.x} * b);
}
__device__ __forceinline__ cf cmulc(cf a, cf b)   // a * conj(b)
{
    return __builtin_elementwise_fma(cf{a.y, a.y}, cf{b.y, b.x}, cf{a.x, a.x} * cf{b.x, -b.y});
}
__device__ __forceinline__ cf mul_mi(cf a) { return cf{a.y, -a.x}; }   // a * (-i)
__device__ __forceinline__ cf mul_pi(cf a) { return cf{-a.y, a.x}; }   // a * (+i)
__device__ __forceinline__ cf ld(const float2 *p) { const float2 v = *p; return cf{v.x, v.y}; }
__device__ __forceinline__ void st(float2 *p, cf v) { *p = make_float2(v.x, v.y); }

__device__ __forceinline__ int freq_of_pos(int p)
{
    const int q1 = p / 96, r1 = p - q1 * 96, q2 = r1 / 24, r2 = r1 - q2 * 24, q3 = r2 / 6, r3 = r2 - q3 * 6;
    const int q4 = r3 / 3, q5 = r3 - q4 * 3;
    return q1 + 4 * q2 + 16 * q3 + 64 * q4 + 128 * q5;
}
__device__ __forceinline__ int pos_of_freq(int k)
{
    const int q1 = k & 3, q2 = (k >> 2) & 3, q3 = (k >> 4) & 3, q4 = (k >> 6) & 1, q5 = k >> 7;
    return q1 * 96 + q2 * 24 + q3 * 6 + q4 * 3 + q5;
}

// One stage over all butterflies bf = first, first + step, ... < N / R of ONE transform whose element p lives at
// d[p * STRIDE].  FWD: y = DFT_R(x), y_q *= W_L^{jq}.  !FWD: the exact inverse (without the 1/R).
template <int R, int STRIDE, bool FWD>
__device__ __forceinline__ void stage(float2 *d, const float2 *tw, const int L, const int first, const int step)
{
    const int sub = L / R, tmul = N / L;
    for (int bf = first; bf < N / R; bf += step) {
        const int blk = bf / sub, j = bf - blk * sub;
        float2 *e = d + (blk * L + j) * STRIDE;
        cf v[R];
#pragma unroll
        for (int m = 0; m < R; ++m) v[m] = ld(e + m * sub * STRIDE);
        const int t1 = tmul * j;       // twiddle W_L^{jq} = tw[t1 * q]; t1 * q < N * (R - 1) / R: no wrap-around
        if constexpr (!FWD) {
#pragma unroll
            for (int q = 1; q < R; ++q) v[q] = cmulc(v[q], ld(tw + t1 * q));
        }
        cf y[R];
        if constexpr (R == 4) {
            const cf s02 = v[0] + v[2], d02 = v[0] - v[2], s13 = v[1] + v[3], d13 = v[1] - v[3];
            y[0] = s02 + s13;
            y[2] = s02 - s13;
            if constexpr (FWD) { y[1] = d02 + mul_mi(d13); y[3] = d02 + mul_pi(d13); }
            else { y[1] = d02 + mul_pi(d13); y[3] = d02 + mul_mi(d13); }
        } else if constexpr (R == 2) {
            y[0] = v[0] + v[1];
            y[1] = v[0] - v[1];
        } else {   // R == 3: w = exp(-+ 2 pi i / 3) = -1/2 -+ i sqrt(3)/2
            const cf s12 = v[1] + v[2], d12 = v[1] - v[2];
            const cf m = v[0] - 0.5f * s12;
            const float h = 0.86602540378443864676f;
            const cf r = FWD ? h * mul_mi(d12) : h * mul_pi(d12);      // -+ i h d12
            y[0] = v[0] + s12;
            y[1] = m + r;
            y[2] = m - r;
        }
        if constexpr (FWD) {
#pragma unroll
            for (int q = 1; q < R; ++q) y[q] = cmul(y[q], ld(tw + t1 * q));
        }
#pragma unroll
        for (int q = 0; q < R; ++q) st(e + q * sub * STRIDE, y[q]);
    }
}

// all threads of the block call these together (block-wide barriers between stages)
template <int STRIDE>
__device__ __forceinline__ void fft_fwd(float2 *d, const float2 *tw, int first, int step)
{
    stage<4, STRIDE, true>(d, tw, 384, first, step); __syncthreads();
    stage<4, STRIDE, true>(d, tw, 96, first, step);  __syncthreads();
    stage<4, STRIDE, true>(d, tw, 24, first, step);  __syncthreads();
    stage<2, STRIDE, true>(d, tw, 6, first, step);   __syncthreads();
    stage<3, STRIDE, true>(d, tw, 3, first, step);   __syncthreads();
}
template <int STRIDE>
__device__ __forceinline__ void fft_inv(float2 *d, const float2 *tw, int first, int step)
{
    stage<3, STRIDE, false>(d, tw, 3, first, step);   __syncthreads();
    stage<2, STRIDE, false>(d, tw, 6, first, step);   __syncthreads();
    stage<4, STRIDE, false>(d, tw, 24, first, step);  __syncthreads();
    stage<4, STRIDE, false>(d, tw, 96, first, step);  __syncthreads();
    stage<4, STRIDE, false>(d, tw, 384, first, step); __syncthreads();
}

// image row a <-> padded, ifftshifted row u;  image column b <-> v   (both: (a + 64 + 192) mod 384)
__device__ __forceinline__ int shifted(int a) { const int u = a + PADW + HALF; return u >= N ? u - N : u; }

// ---- pass A: 4 image rows per block (one per wave); half: [planes][256][193]
__global__ __launch_bounds__(256) void k_pr_rows_fwd(const float *__restrict__ x, const float *__restrict__ mo,
                                                     const float *__restrict__ z, float *__restrict__ x0o,
                                                     float *__restrict__ so, uint8_t *__restrict__ ins,
                                                     float2 *__restrict__ half, const float2 *__restrict__ tw_g,
                                                     int c, dpsx::Coefs k)
{
    __shared__ float2 s_tw[N];
    __shared__ float2 s_row[4][N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t plane = blockIdx.y;
    const int a = blockIdx.x * 4 + wave;                 // image row
    for (int i = threadIdx.x; i < N; i += 256) s_tw[i] = tw_g[i];
    const int64_t hw = (int64_t)IMG * IMG, o = plane * hw + (int64_t)a * IMG + 4 * lane;
    const int64_t n = plane / c, ch = plane % c, e = (n * 2 * c + ch) * hw + (int64_t)a * IMG + 4 * lane;
    const float4 xv = *reinterpret_cast<const float4 *>(x + o);
    const float4 ev = *reinterpret_cast<const float4 *>(mo + e);
    float4 vv = make_float4(0, 0, 0, 0), zv = vv;
    if (k.add_noise & 1) {
        vv = *reinterpret_cast<const float4 *>(mo + e + (int64_t)c * hw);
        zv = *reinterpret_cast<const float4 *>(z + o);
    }
    bool b0, b1, b2, b3;
    float4 x0, sm;
    x0.x = dpsx::post_x0(xv.x, ev.x, k, b0);
    x0.y = dpsx::post_x0(xv.y, ev.y, k, b1);
    x0.z = dpsx::post_x0(xv.z, ev.z, k, b2);
    x0.w = dpsx::post_x0(xv.w, ev.w, k, b3);
    sm.x = dpsx::post_sample(xv.x, x0.x, vv.x, zv.x, k);
    sm.y = dpsx::post_sample(xv.y, x0.y, vv.y, zv.y, k);
    sm.z = dpsx::post_sample(xv.z, x0.z, vv.z, zv.z, k);
    sm.w = dpsx::post_sample(xv.w, x0.w, vv.w, zv.w, k);
    *reinterpret_cast<float4 *>(x0o + o) = x0;
    *reinterpret_cast<float4 *>(so + o) = sm;
    *reinterpret_cast<uchar4 *>(ins + o) = make_uchar4(b0, b1, b2, b3);
    // the padded, shifted row: image columns 4 lane .. 4 lane + 3 sit at v0 .. v0 + 3; columns 128..255 of v are zero
    float2 *row = s_row[wave];
    const int v0 = shifted(4 * lane);
    row[v0] = make_float2(x0.x, 0.0f);
    row[v0 + 1] = make_float2(x0.y, 0.0f);
    row[v0 + 2] = make_float2(x0.z, 0.0f);
    row[v0 + 3] = make_float2(x0.w, 0.0f);
    row[128 + 2 * lane] = make_float2(0.0f, 0.0f);
    row[129 + 2 * lane] = make_float2(0.0f, 0.0f);
    __syncthreads();
    fft_fwd<1>(row, s_tw, lane, 64);
    float2 *out = half + (plane * IMG + a) * HS;
    for (int kx = lane; kx < HS; kx += 64) out[kx] = row[pos_of_freq(kx)];
}

// ---- pass B: one (plane, 16-column tile) per block
__global__ __launch_bounds__(256) void k_pr_cols(float2 *__restrict__ half, const float *__restrict__ y, int y_n, int c,
                                                 float *__restrict__ partials, const float2 *__restrict__ tw_g)
{
    extern __shared__ __align__(16) float2 s_dyn[];
    float2 *s_tw = s_dyn, *s_d = s_dyn + N;                 // s_d[384][CT]
    __shared__ float scratch[256 / dpsx::kWave];
    const int64_t plane = blockIdx.y;
    const int tile = blockIdx.x, cl = threadIdx.x & (CT - 1), g = threadIdx.x / CT;   // 16 row-lanes g per column
    const int kx = tile * CT + cl;
    const bool colok = kx < HS;
    for (int i = threadIdx.x; i < N; i += 256) s_tw[i] = tw_g[i];
    float2 *hp = half + plane * IMG * HS;
    {
        constexpr int RPL = IMG / (256 / CT);      // 16 image rows per lane, loads first
        float2 t[RPL];
        const int kxc = colok ? kx : HS - 1;
#pragma unroll
        for (int i = 0; i < RPL; ++i) t[i] = hp[(int64_t)(g + i * (256 / CT)) * HS + kxc];
#pragma unroll
        for (int i = 0; i < RPL; ++i) s_d[shifted(g + i * (256 / CT)) * CT + cl] = colok ? t[i] : make_float2(0.0f, 0.0f);
    }
    for (int u = 128 + g; u < 256; u += 256 / CT) s_d[u * CT + cl] = make_float2(0.0f, 0.0f);   // the zero rows
    __syncthreads();
    fft_fwd<CT>(s_d + cl, s_tw, g, 256 / CT);
    // pointwise on digit-reversed positions
    const int n = (int)(plane / c), ch = (int)(plane % c);
    const float *yp = y + ((int64_t)(y_n == 1 ? 0 : n) * c + ch) * N * N;
    const float inv = 1.0f / (float)N;
    float acc = 0.0f;
    if (colok) {
        const int mx = kx == 0 ? 0 : N - kx;
        int ox = kx + HALF, px = mx + HALF;
        ox = ox >= N ? ox - N : ox;
        px = px >= N ? px - N : px;
        constexpr int PER = N / (256 / CT);        // 24 positions per lane: all 48 measurement loads issue first
        float y1[PER], y2[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int ky = freq_of_pos(g + i * (256 / CT)), my = ky == 0 ? 0 : N - ky;
            int oy = ky + HALF, py = my + HALF;
            oy = oy >= N ? oy - N : oy;
            py = py >= N ? py - N : py;
            y1[i] = yp[oy * N + ox];
            y2[i] = yp[py * N + px];
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int p = g + i * (256 / CT);
            float2 zz = s_d[p * CT + cl];
            zz.x *= inv;
            zz.y *= inv;
            const float mag = sqrtf(zz.x * zz.x + zz.y * zz.y);
            const float r1 = y1[i] - mag, r2 = y2[i] - mag;
            acc = fmaf(r1, r1, acc);
            if (kx != 0 && kx != HALF) acc = fmaf(r2, r2, acc);
            const float f = mag == 0.0f ? 0.0f : 0.5f * (r1 + r2) / mag;
            s_d[p * CT + cl] = make_float2(f * zz.x, f * zz.y);
        }
    }
    __syncthreads();
    fft_inv<CT>(s_d + cl, s_tw, g, 256 / CT);
    if (colok)
        for (int a = g; a < IMG; a += 256 / CT) hp[(int64_t)a * HS + kx] = s_d[shifted(a) * CT + cl];
    const float t = dpsx::block_sum(acc, scratch);
    if (threadIdx.x == 0) partials[((int64_t)n * c + ch) * gridDim.x + blockIdx.x] = t;
}

// ---- pass C: 4 image rows per block
__global__ __launch_bounds__(256) void k_pr_rows_inv(const float2 *__restrict__ half, const float *__restrict__ norm,
                                                     const uint8_t *__restrict__ ins, const float *__restrict__ g_extra,
                                                     float scale, int power, float neg_b, float *__restrict__ g_mo,
                                                     const float2 *__restrict__ tw_g, int c)
{
    __shared__ float2 s_tw[N];
    __shared__ float2 s_row[4][N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t plane = blockIdx.y;
    const int a = blockIdx.x * 4 + wave;
    for (int i = threadIdx.x; i < N; i += 256) s_tw[i] = tw_g[i];
    float2 *row = s_row[wave];
    const float2 *in = half + (plane * IMG + a) * HS;
    for (int kx = lane; kx < HS; kx += 64) {
        const float2 v = in[kx];
        row[pos_of_freq(kx)] = v;
        if (kx != 0 && kx != HALF) row[pos_of_freq(N - kx)] = make_float2(v.x, -v.y);   // Hermitian extension
    }
    __syncthreads();
    fft_inv<1>(row, s_tw, lane, 64);
    const int64_t n = plane / c, ch = plane % c, hw = (int64_t)IMG * IMG, o = plane * hw + (int64_t)a * IMG + 4 * lane;
    const float nv = norm[n];
    const float coef = (power == 2 ? -2.0f * scale : (nv == 0.0f ? 0.0f : -scale / nv)) * (1.0f / (float)N);
    const int v0 = shifted(4 * lane);
    const uchar4 gate = *reinterpret_cast<const uchar4 *>(ins + o);
    float4 ex = make_float4(0, 0, 0, 0);
    if (g_extra) ex = *reinterpret_cast<const float4 *>(g_extra + o);
    float4 g;
    g.x = gate.x ? neg_b * (coef * row[v0].x + ex.x) : 0.0f;
    g.y = gate.y ? neg_b * (coef * row[v0 + 1].x + ex.y) : 0.0f;
    g.z = gate.z ? neg_b * (coef * row[v0 + 2].x + ex.z) : 0.0f;
    g.w = gate.w ? neg_b * (coef * row[v0 + 3].x + ex.w) : 0.0f;
    *reinterpret_cast<float4 *>(g_mo + (n * 2 * c + ch) * hw + (int64_t)a * IMG + 4 * lane) = g;
}

}  // namespace prfft
