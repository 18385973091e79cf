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
// row pitch of the half-spectrum scratch [planes][256][HP]: 208 complex = 13 x 128 B, so the 16-column (128 B) tiles of pass B
// are whole cache lines (with the natural pitch of 193 a tile row straddled two lines: the PMC pass counted 1.5 x the array)
constexpr int HP = 208;

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

// ---- two-level in-place FFT, 384 = 16 x 24, two LDS passes, the small transforms entirely in registers
// level 1 (stride 24):  y_q = DFT16_m(x[j + 24 m]) * W_384^(j q)   -> d[j + 24 q]           j in [0, 24)
// level 2 (blocks)   :  X[q + 16 k'] = DFT24_j'(d[24 q + j'])      -> d[24 q + k']          q in [0, 16)
// so position p holds frequency p / 24 + 16 (p mod 24).  The inverse is the exact mirror (level 2 first, conjugate
// twiddles before the inverse DFT16), unnormalised.  W_16 / W_24 / W_6 twiddles are compile-time constants.
__device__ __forceinline__ int freq_of_pos(int p) { const int q = p / 24; return q + 16 * (p - 24 * q); }
__device__ __forceinline__ int pos_of_freq(int k) { return 24 * (k & 15) + (k >> 4); }

constexpr float kC16[16] = {1.0f, 0.923879533f, 0.707106781f, 0.382683432f, 6.123234e-17f, -0.382683432f, -0.707106781f, -0.923879533f, -1.0f, -0.923879533f, -0.707106781f, -0.382683432f, -1.8369702e-16f, 0.382683432f, 0.707106781f, 0.923879533f};
constexpr float kS16[16] = {0.0f, 0.382683432f, 0.707106781f, 0.923879533f, 1.0f, 0.923879533f, 0.707106781f, 0.382683432f, 1.2246468e-16f, -0.382683432f, -0.707106781f, -0.923879533f, -1.0f, -0.923879533f, -0.707106781f, -0.382683432f};
constexpr float kC24[24] = {1.0f, 0.965925826f, 0.866025404f, 0.707106781f, 0.5f, 0.258819045f, 6.123234e-17f, -0.258819045f, -0.5f, -0.707106781f, -0.866025404f, -0.965925826f, -1.0f, -0.965925826f, -0.866025404f, -0.707106781f, -0.5f, -0.258819045f, -1.8369702e-16f, 0.258819045f, 0.5f, 0.707106781f, 0.866025404f, 0.965925826f};
constexpr float kS24[24] = {0.0f, 0.258819045f, 0.5f, 0.707106781f, 0.866025404f, 0.965925826f, 1.0f, 0.965925826f, 0.866025404f, 0.707106781f, 0.5f, 0.258819045f, 1.2246468e-16f, -0.258819045f, -0.5f, -0.707106781f, -0.866025404f, -0.965925826f, -1.0f, -0.965925826f, -0.866025404f, -0.707106781f, -0.5f, -0.258819045f};
constexpr float kC6[6] = {1.0f, 0.5f, -0.5f, -1.0f, -0.5f, 0.5f};
constexpr float kS6[6] = {0.0f, 0.866025404f, 0.866025404f, 1.2246468e-16f, -0.866025404f, -0.866025404f};

// v * exp(-+ i theta) given cos / sin of theta (FWD: minus)
template <bool FWD>
__device__ __forceinline__ cf rot(cf v, float c, float s)
{
    return FWD ? cmul(v, cf{c, -s}) : cmul(v, cf{c, s});
}
template <bool FWD>
__device__ __forceinline__ void dft2(cf &a, cf &b) { const cf t = a - b; a = a + b; b = t; }
template <bool FWD>
__device__ __forceinline__ void dft3(cf &a, cf &b, cf &c)
{
    const cf s12 = b + c, d12 = b - c, m = a - 0.5f * s12;
    const float h = 0.86602540378443864676f;
    const cf r = FWD ? h * mul_mi(d12) : h * mul_pi(d12);
    a = a + s12; b = m + r; c = m - r;
}
template <bool FWD>
__device__ __forceinline__ void dft4(cf &a, cf &b, cf &c, cf &d)
{
    const cf s02 = a + c, d02 = a - c, s13 = b + d, d13 = b - d;
    a = s02 + s13; c = s02 - s13;
    if (FWD) { b = d02 + mul_mi(d13); d = d02 + mul_pi(d13); }
    else { b = d02 + mul_pi(d13); d = d02 + mul_mi(d13); }
}
// natural order in, natural order out
template <bool FWD>
__device__ __forceinline__ void dft6(cf (&x)[6])
{
    // m = a + 2 b, q = 3 c + d:  X[3c + d] = sum_a W_6^(a d) W_2^(a c) sum_b x[a + 2b] W_3^(b d)
    cf e0 = x[0], e1 = x[2], e2 = x[4], o0 = x[1], o1 = x[3], o2 = x[5];
    dft3<FWD>(e0, e1, e2);
    dft3<FWD>(o0, o1, o2);
    o1 = rot<FWD>(o1, kC6[1], kS6[1]);
    o2 = rot<FWD>(o2, kC6[2], kS6[2]);
    x[0] = e0 + o0; x[3] = e0 - o0;
    x[1] = e1 + o1; x[4] = e1 - o1;
    x[2] = e2 + o2; x[5] = e2 - o2;
}
template <bool FWD>
__device__ __forceinline__ void dft16(cf (&x)[16])
{
    // m = m1 + 4 m2, q = 4 q1 + q2:  X[4 q1 + q2] = sum_m1 W_16^(m1 q2) W_4^(m1 q1) sum_m2 x[m1 + 4 m2] W_4^(m2 q2)
#pragma unroll
    for (int m1 = 0; m1 < 4; ++m1) dft4<FWD>(x[m1], x[m1 + 4], x[m1 + 8], x[m1 + 12]);    // x[m1 + 4 q2] = A[m1][q2]
#pragma unroll
    for (int m1 = 1; m1 < 4; ++m1)
#pragma unroll
        for (int q2 = 1; q2 < 4; ++q2) x[m1 + 4 * q2] = rot<FWD>(x[m1 + 4 * q2], kC16[m1 * q2], kS16[m1 * q2]);
    cf y[16];
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) {
        cf t0 = x[0 + 4 * q2], t1 = x[1 + 4 * q2], t2 = x[2 + 4 * q2], t3 = x[3 + 4 * q2];
        dft4<FWD>(t0, t1, t2, t3);                                                      // index q1
        y[q2] = t0; y[4 + q2] = t1; y[8 + q2] = t2; y[12 + q2] = t3;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = y[i];
}
template <bool FWD>
__device__ __forceinline__ void dft24(cf (&x)[24])
{
    // m = m1 + 4 m2 (m2 < 6), q = 6 q1 + q2 (q2 < 6):
    // X[6 q1 + q2] = sum_m1 W_24^(m1 q2) W_4^(m1 q1) sum_m2 x[m1 + 4 m2] W_6^(m2 q2)
    cf B[4][6];
#pragma unroll
    for (int m1 = 0; m1 < 4; ++m1) {
        cf t[6];
#pragma unroll
        for (int m2 = 0; m2 < 6; ++m2) t[m2] = x[m1 + 4 * m2];
        dft6<FWD>(t);
#pragma unroll
        for (int q2 = 0; q2 < 6; ++q2) B[m1][q2] = (m1 == 0 || q2 == 0) ? t[q2] : rot<FWD>(t[q2], kC24[m1 * q2], kS24[m1 * q2]);
    }
#pragma unroll
    for (int q2 = 0; q2 < 6; ++q2) {
        cf t0 = B[0][q2], t1 = B[1][q2], t2 = B[2][q2], t3 = B[3][q2];
        dft4<FWD>(t0, t1, t2, t3);
        x[q2] = t0; x[6 + q2] = t1; x[12 + q2] = t2; x[18 + q2] = t3;
    }
}

// element p of the transform lives at d[p * STRIDE]; `first`, `step`: this lane's share of the 24 level-1 butterflies
// and of the 16 level-2 blocks.  All threads of the block call these together.
template <int STRIDE>
__device__ __forceinline__ void fft_fwd(float2 *d, const float2 *tw, int first, int step)
{
    for (int j = first; j < 24; j += step) {
        cf x[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) x[m] = ld(d + (j + 24 * m) * STRIDE);
        dft16<true>(x);
#pragma unroll
        for (int q = 1; q < 16; ++q) x[q] = cmul(x[q], ld(tw + j * q));
#pragma unroll
        for (int q = 0; q < 16; ++q) st(d + (j + 24 * q) * STRIDE, x[q]);
    }
    __syncthreads();
    for (int q = first; q < 16; q += step) {
        cf x[24];
#pragma unroll
        for (int j = 0; j < 24; ++j) x[j] = ld(d + (24 * q + j) * STRIDE);
        dft24<true>(x);
#pragma unroll
        for (int k = 0; k < 24; ++k) st(d + (24 * q + k) * STRIDE, x[k]);
    }
    __syncthreads();
}
template <int STRIDE>
__device__ __forceinline__ void fft_inv(float2 *d, const float2 *tw, int first, int step)
{
    for (int q = first; q < 16; q += step) {
        cf x[24];
#pragma unroll
        for (int k = 0; k < 24; ++k) x[k] = ld(d + (24 * q + k) * STRIDE);
        dft24<false>(x);
#pragma unroll
        for (int j = 0; j < 24; ++j) st(d + (24 * q + j) * STRIDE, x[j]);
    }
    __syncthreads();
    for (int j = first; j < 24; j += step) {
        cf x[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = ld(d + (j + 24 * q) * STRIDE);
#pragma unroll
        for (int q = 1; q < 16; ++q) x[q] = cmulc(x[q], ld(tw + j * q));
        dft16<false>(x);
#pragma unroll
        for (int m = 0; m < 16; ++m) st(d + (j + 24 * m) * STRIDE, x[m]);
    }
    __syncthreads();
}

// image row a <-> padded, ifftshifted row u;  image column b <-> v   (both: (a + 64 + 192) mod 384)
__device__ __forceinline__ int shifted(int a) { const int u = a + PADW + HALF; return u >= N ? u - N : u; }

// ---- pass A: RPW image rows per wave (64 / RPW lanes each), 4 waves per block; half: [planes][256][193]
constexpr int RPW = 2, RPB = 4 * RPW, LPR = 64 / RPW;       // rows per wave / per block, lanes per row

__global__ __launch_bounds__(256) void k_pr_rows_fwd(const float *__restrict__ x, const float *__restrict__ mo,
                                                     const float *__restrict__ z, float *__restrict__ x0o,
                                                     float *__restrict__ so, uint8_t *__restrict__ ins,
                                                     float2 *__restrict__ half, const float2 *__restrict__ tw_g,
                                                     int c, dpsx::Coefs k)
{
    __shared__ float2 s_tw[N];
    __shared__ float2 s_row[RPB][N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t plane = blockIdx.y;
    const int a0 = blockIdx.x * RPB + wave * RPW;        // first image row of this wave
    for (int i = threadIdx.x; i < N; i += 256) s_tw[i] = tw_g[i];
    const int64_t hw = (int64_t)IMG * IMG;
    const int64_t n = plane / c, ch = plane % c;
    float4 xv[RPW], ev[RPW], vv[RPW], zv[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {                      // one float4 unit of each of the wave's rows per lane
        const int64_t o = plane * hw + (int64_t)(a0 + r) * IMG + 4 * lane;
        const int64_t e = (n * 2 * c + ch) * hw + (int64_t)(a0 + r) * IMG + 4 * lane;
        xv[r] = *reinterpret_cast<const float4 *>(x + o);
        ev[r] = *reinterpret_cast<const float4 *>(mo + e);
        vv[r] = zv[r] = make_float4(0, 0, 0, 0);
        if (k.add_noise & 1) {
            vv[r] = *reinterpret_cast<const float4 *>(mo + e + (int64_t)c * hw);
            zv[r] = *reinterpret_cast<const float4 *>(z + o);
        }
    }
    const int v0 = shifted(4 * lane);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int64_t o = plane * hw + (int64_t)(a0 + r) * IMG + 4 * lane;
        bool b0, b1, b2, b3;
        float4 x0, sm;
        x0.x = dpsx::post_x0(xv[r].x, ev[r].x, k, b0);
        x0.y = dpsx::post_x0(xv[r].y, ev[r].y, k, b1);
        x0.z = dpsx::post_x0(xv[r].z, ev[r].z, k, b2);
        x0.w = dpsx::post_x0(xv[r].w, ev[r].w, k, b3);
        sm.x = dpsx::post_sample(xv[r].x, x0.x, vv[r].x, zv[r].x, k);
        sm.y = dpsx::post_sample(xv[r].y, x0.y, vv[r].y, zv[r].y, k);
        sm.z = dpsx::post_sample(xv[r].z, x0.z, vv[r].z, zv[r].z, k);
        sm.w = dpsx::post_sample(xv[r].w, x0.w, vv[r].w, zv[r].w, k);
        if (x0o) *reinterpret_cast<float4 *>(x0o + o) = x0;     // launch-uniform: the image is an optional output
        *reinterpret_cast<float4 *>(so + o) = sm;
        *reinterpret_cast<uchar4 *>(ins + o) = make_uchar4(b0, b1, b2, b3);
        // the padded, shifted row: image columns 4 lane .. + 3 sit at v0 .. v0 + 3; columns 128..255 of v are zero
        float2 *row = s_row[wave * RPW + r];
        row[v0] = make_float2(x0.x, 0.0f);
        row[v0 + 1] = make_float2(x0.y, 0.0f);
        row[v0 + 2] = make_float2(x0.z, 0.0f);
        row[v0 + 3] = make_float2(x0.w, 0.0f);
        row[128 + 2 * lane] = make_float2(0.0f, 0.0f);
        row[129 + 2 * lane] = make_float2(0.0f, 0.0f);
    }
    __syncthreads();
    const int sub = lane / LPR, l = lane - sub * LPR;      // this lane's row within the wave, lane within the row
    float2 *row = s_row[wave * RPW + sub];
    fft_fwd<1>(row, s_tw, l, LPR);
    float2 *out = half + (plane * IMG + a0 + sub) * HP;
    for (int kx = l; kx < HS; kx += LPR) out[kx] = row[pos_of_freq(kx)];
}

// ---- pass B: one (plane, 16-column tile) per block
constexpr int BT = 512, TPC = BT / CT;      // threads of pass B, threads per column

__global__ __launch_bounds__(BT) void k_pr_cols(float2 *__restrict__ half, const float *__restrict__ y, int y_n, int c,
                                                 float *__restrict__ partials, const float2 *__restrict__ tw_g)
{
    extern __shared__ __align__(16) float2 s_dyn[];
    float2 *s_tw = s_dyn, *s_d = s_dyn + N;                 // s_d[384][CT]
    __shared__ float scratch[BT / dpsx::kWave];
    const int64_t plane = blockIdx.y;
    const int tile = blockIdx.x, cl = threadIdx.x & (CT - 1), g = threadIdx.x / CT;   // TPC row-lanes g per column
    const int kx = tile * CT + cl;
    const bool colok = kx < HS;
    for (int i = threadIdx.x; i < N; i += BT) s_tw[i] = tw_g[i];
    float2 *hp = half + plane * IMG * HP;
    {
        constexpr int RPL = IMG / (TPC);      // 16 image rows per lane, loads first
        float2 t[RPL];
        const int kxc = colok ? kx : HS - 1;
#pragma unroll
        for (int i = 0; i < RPL; ++i) t[i] = hp[(int64_t)(g + i * (TPC)) * HP + kxc];
#pragma unroll
        for (int i = 0; i < RPL; ++i) s_d[shifted(g + i * (TPC)) * CT + cl] = colok ? t[i] : make_float2(0.0f, 0.0f);
    }
    for (int u = 128 + g; u < 256; u += TPC) s_d[u * CT + cl] = make_float2(0.0f, 0.0f);   // the zero rows
    __syncthreads();
    fft_fwd<CT>(s_d + cl, s_tw, g, TPC);
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
        constexpr int PER = N / (TPC);        // 24 positions per lane: all 48 measurement loads issue first
        float y1[PER], y2[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int ky = freq_of_pos(g + i * (TPC)), my = ky == 0 ? 0 : N - ky;
            int oy = ky + HALF, py = my + HALF;
            oy = oy >= N ? oy - N : oy;
            py = py >= N ? py - N : py;
            y1[i] = yp[oy * N + ox];
            y2[i] = yp[py * N + px];
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int p = g + i * (TPC);
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
    fft_inv<CT>(s_d + cl, s_tw, g, TPC);
    if (colok)
        for (int a = g; a < IMG; a += TPC) hp[(int64_t)a * HP + kx] = s_d[shifted(a) * CT + cl];
    const float t = dpsx::block_sum(acc, scratch);
    if (threadIdx.x == 0) partials[((int64_t)n * c + ch) * gridDim.x + blockIdx.x] = t;
}

// ---- pass C: RPW image rows per wave
__global__ __launch_bounds__(256) void k_pr_rows_inv(const float2 *__restrict__ half, const float *__restrict__ norm,
                                                     const uint8_t *__restrict__ ins, const float *__restrict__ g_extra,
                                                     float scale, int power, float neg_b, float *__restrict__ g_mo,
                                                     const float2 *__restrict__ tw_g, int c)
{
    __shared__ float2 s_tw[N];
    __shared__ float2 s_row[RPB][N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t plane = blockIdx.y;
    const int a0 = blockIdx.x * RPB + wave * RPW;
    for (int i = threadIdx.x; i < N; i += 256) s_tw[i] = tw_g[i];
    {
        const int sub = lane / LPR, l = lane - sub * LPR;
        float2 *row = s_row[wave * RPW + sub];
        const float2 *in = half + (plane * IMG + a0 + sub) * HP;
        for (int kx = l; kx < HS; kx += LPR) {
            const float2 v = in[kx];
            row[pos_of_freq(kx)] = v;
            if (kx != 0 && kx != HALF) row[pos_of_freq(N - kx)] = make_float2(v.x, -v.y);   // Hermitian extension
        }
        __syncthreads();
        fft_inv<1>(row, s_tw, l, LPR);
    }
    const int64_t n = plane / c, ch = plane % c, hw = (int64_t)IMG * IMG;
    const float nv = norm[n];
    const float coef = (power == 2 ? -2.0f * scale : (nv == 0.0f ? 0.0f : -scale / nv)) * (1.0f / (float)N);
    const int v0 = shifted(4 * lane);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const float2 *row = s_row[wave * RPW + r];
        const int64_t o = plane * hw + (int64_t)(a0 + r) * IMG + 4 * lane;
        const uchar4 gate = *reinterpret_cast<const uchar4 *>(ins + o);
        float4 ex = make_float4(0, 0, 0, 0);
        if (g_extra) ex = *reinterpret_cast<const float4 *>(g_extra + o);
        float4 g;
        g.x = gate.x ? neg_b * (coef * row[v0].x + ex.x) : 0.0f;
        g.y = gate.y ? neg_b * (coef * row[v0 + 1].x + ex.y) : 0.0f;
        g.z = gate.z ? neg_b * (coef * row[v0 + 2].x + ex.z) : 0.0f;
        g.w = gate.w ? neg_b * (coef * row[v0 + 3].x + ex.w) : 0.0f;
        *reinterpret_cast<float4 *>(g_mo + (n * 2 * c + ch) * hw + (int64_t)(a0 + r) * IMG + 4 * lane) = g;
    }
}

// =====================================================================================================================
// Round 3: passes B and C rebuilt around REGISTERS instead of LDS passes.
//
// What the round-2 kernels above paid for (SQ counters, profiles/r03_sq_phase_retrieval_*): every FFT level was an
// LDS read + an LDS write of the whole array behind a workgroup barrier (pass B: six writes and six reads per element,
// seven barriers), and the 32 lanes that shared one transform were 75 % (24 radix-16 butterflies) and 50 % (16 radix-24
// blocks) busy.  Here a transform belongs to SIXTEEN lanes:
//   level 2 (16 radix-24 blocks)      one block per lane -- and everything that happens to a frequency happens in that
//                                     lane's registers: forward DFT24, modulus / residual / cotangent, inverse DFT24;
//   level 1 (24 radix-16 butterflies) two rounds (16 + 8): inputs come straight from global memory (forward) and the
//                                     outputs go straight back (inverse) -- the stride-24 gather IS the global access,
//                                     the known-zero inputs (the 128 padding rows / columns) are never loaded;
// so the array crosses LDS only for the transposition between the two levels: two writes and two reads per element, two
// barriers.  Pass C also puts TWO image rows into one complex transform (both spectra are Hermitian, so row r1 comes
// out as the real part and r2 as the imaginary part of the inverse of X1 + i X2): half the transforms.
// =====================================================================================================================
constexpr int G16 = 16;                     // lanes per transform
// LDS index of position p = 24 q + k = j + 24 q in the row kernels (one transform = 384 contiguous float2): element k of
// block q sits at 24 q + (k + q) mod 24.  The 16 lanes of a transform read "their" k of blocks q = 0..15 together: without
// the rotation they are 48 words apart (four lanes per bank pair -- the round-2 row kernels spent 60 % of their LDS cycles
// in bank conflicts, profiles/r03_sq_phase_retrieval_bwd_summary.txt); with it 50 words, all distinct except where the
// rotation wraps.  No padding: 3 KB per transform, three 16-transform workgroups per CU.
__device__ __forceinline__ int rot24(int q, int k) { const int r = k + q; return 24 * q + (r >= 24 ? r - 24 : r); }

// rows of the shifted (ifftshift + zero padding) axis: u in [0, 128) is image index u + 128, u in [256, 384) is u - 256,
// u in [128, 256) is padding.  For a level-1 butterfly j its inputs u = j + 24 m are therefore: m <= 4 image, m = 5 image
// iff j < 8, m = 6..9 padding, m = 10 image iff j >= 16, m >= 11 image -- compile-time per m except for those two.
__device__ __forceinline__ bool lvl1_is_image(int j, int m) { return m <= 4 || (m == 5 && j < 8) || (m == 10 && j >= 16) || m >= 11; }
__device__ __forceinline__ int lvl1_image_index(int j, int m) { const int u = j + 24 * m; return u < 128 ? u + 128 : u - 256; }

// ---- pass B: one (plane, 16-column tile) per 256-thread block; lane (cl, g): column cl of the tile, level-2 block g,
// level-1 butterflies g and (waves 0, 1 only) g + 16.  LDS: twiddles | s_d[384][16] (row swizzle: b2_idx1 / b2_idx2).
constexpr int B2T = 256, B2LDS = N + N * CT;     // float2 elements of dynamic LDS: twiddles | 384 rows x 16 columns

// position p = j + 24 q (level 1: butterfly j, output q) = 24 q + k (level 2: block q, element k) sits in LDS row
// 24 q + (k ^ (q & 1)): a row of 16 columns is 32 words, so rows of equal parity share their banks -- and the two blocks
// q, q + 1 that share a ds_read_b64 half-wave in level 2 read the same k.  Swapping the row pairs of the odd blocks puts
// them on opposite halves of the banks without a padding row: 52,224 B with the twiddles, three workgroups per CU (LDS is
// granted in 512-byte granules: the padded 54,288 B image admitted only two -- 91 us instead of 66).
__device__ __forceinline__ int b2_idx1(int j, int q, int cl) { return (24 * q + (j ^ (q & 1))) * CT + cl; }
// (level 2, block q runtime, k compile-time: k ^ b = k + b for even k, k - b for odd k -- two base registers + immediates)
__device__ __forceinline__ int b2_idx2(int be, int bo, int k) { return ((k & 1) ? bo : be) + k * CT; }

// Workgroup -> work: 193 columns are twelve 16-column tiles and ONE column (the Nyquist bin kx = 192).  A thirteenth tile per
// plane would spend a whole workgroup on that column (2,496 workgroups for 768 slots: 3.25 generations, 7 % of the lanes'
// work wasted); instead the first ceil(planes / 16) workgroups of the grid each take the Nyquist column of SIXTEEN planes
// (lane cl <-> plane 16 b + cl), and the other 12 x planes workgroups a 16-column tile of one plane: 2,316 workgroups, three
// generations.  Per-lane plane: the base pointers are per-lane values in both forms.
__global__ __launch_bounds__(B2T, 3) void k_pr_cols2(float2 *__restrict__ half, const float *__restrict__ y, int y_n, int c,
                                                      float *__restrict__ partials, const float2 *__restrict__ tw_g,
                                                      int planes, int nyq_blocks)
{
    extern __shared__ __align__(16) float2 s_dyn[];
    float2 *s_tw = s_dyn, *s_d = s_dyn + N;
    __shared__ float scratch[B2T / dpsx::kWave];
    constexpr int TILES = HS / CT;                  // 12 full tiles; partial slot TILES of a plane is its Nyquist column
    const int cl = threadIdx.x & (CT - 1), g = threadIdx.x / CT;
    const bool nyq = (int)blockIdx.x < nyq_blocks;  // block-uniform
    const int vb = (int)blockIdx.x - nyq_blocks;
    const int plane_raw = nyq ? (int)blockIdx.x * CT + cl : vb / TILES;
    const int tile = nyq ? TILES : vb - (vb / TILES) * TILES;
    const bool colok = plane_raw < planes;          // (only the last Nyquist workgroup can hold surplus lanes)
    const int64_t plane = colok ? plane_raw : planes - 1;      // surplus lanes shadow the last plane (never stored)
    const int kx = nyq ? HALF : tile * CT + cl;
    const int kxc = kx;
    for (int i = threadIdx.x; i < N; i += B2T) s_tw[i] = tw_g[i];
    float2 *hp = half + plane * IMG * HP + kxc;
    __syncthreads();                                 // twiddles
    // ---- forward level 1, inputs from global memory
    for (int j = g; j < 24; j += G16) {              // second round: waves 0 and 1 only (wave-uniform)
        cf x[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            x[m] = cf{0.0f, 0.0f};
            if (m <= 4 || m >= 11) x[m] = ld(hp + (int64_t)lvl1_image_index(j, m) * HP);
            else if (m == 5 || m == 10) {
                // (both index forms stay inside the plane: the load is unconditional, the value selected)
                const cf v = ld(hp + (int64_t)(m == 5 ? (j + 120 + 128) & 255 : (j + 240 - 256) & 255) * HP);
                x[m] = lvl1_is_image(j, m) ? v : cf{0.0f, 0.0f};
            }
        }
        dft16<true>(x);
#pragma unroll
        for (int q = 1; q < 16; ++q) x[q] = cmul(x[q], ld(s_tw + j * q));
#pragma unroll
        for (int q = 0; q < 16; ++q) st(s_d + b2_idx1(j, q, cl), x[q]);
    }
    const int n = (int)(plane / c), ch = (int)(plane % c);
    const float *yp = y + ((int64_t)(y_n == 1 ? 0 : n) * c + ch) * N * N;       // (per lane in a Nyquist workgroup)
    // the measurement values of this lane's 24 frequencies and of their mirrors (an L2-resident table: 1.8 MB for a broadcast
    // measurement): issued ahead of the barrier, so that their latency runs under the wait and the forward DFT24
    float y1[24], y2[24];
    {
        const int mx = kxc == 0 ? 0 : N - kxc;
        int ox = kxc + HALF, px = mx + HALF;
        ox = ox >= N ? ox - N : ox;
        px = px >= N ? px - N : px;
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const int ky = g + 16 * k, my = ky == 0 ? 0 : N - ky;
            int oy = ky + HALF, py = my + HALF;
            oy = oy >= N ? oy - N : oy;
            py = py >= N ? py - N : py;
            y1[k] = yp[oy * N + ox];
            y2[k] = yp[py * N + px];
        }
    }
    __syncthreads();
    // ---- level 2 forward, pointwise, level 2 inverse: block g, frequencies g + 16 k, all in registers
    float acc = 0.0f;
    {
        cf x[24];
        const int be = (24 * g + (g & 1)) * CT + cl, bo = (24 * g - (g & 1)) * CT + cl;
#pragma unroll
        for (int k = 0; k < 24; ++k) x[k] = ld(s_d + b2_idx2(be, bo, k));
        dft24<true>(x);
        // |z| / s, the residuals and the cotangent (r1 + r2) / 2 * z / |z| from ONE transcendental: rs = 1 / |X| (v_rsq_f32,
        // 1 ulp), |X| = |X|^2 rs.  The correctly rounded sqrtf and division of the round-2 pass were 40 % of its vector
        // instructions (24 frequencies x ~22 instructions of Newton fix-ups); the test tolerance is 1e-5.
        const float inv = 1.0f / (float)N;
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const float m2 = fmaf(x[k].x, x[k].x, x[k].y * x[k].y);
            const float rs = m2 == 0.0f ? 0.0f : __builtin_amdgcn_rsqf(m2);     // torch: d|z| = 0 at z = 0
            const float mag = m2 * rs * inv;
            const float r1 = y1[k] - mag, r2 = y2[k] - mag;
            const float r2m = (kx != 0 && kx != HALF) ? r2 : 0.0f;      // (a select, not a branch per frequency)
            acc = fmaf(r1, r1, acc);
            acc = fmaf(r2m, r2m, acc);
            x[k] = (0.5f * (r1 + r2) * rs) * x[k];                       // = f * z / s with f = (r1 + r2) / 2 / (|z| / s)
        }
        if (!colok) acc = 0.0f;
        dft24<false>(x);
#pragma unroll
        for (int k = 0; k < 24; ++k) st(s_d + b2_idx2(be, bo, k), x[k]);
    }
    __syncthreads();
    // ---- inverse level 1, outputs to global memory (image rows only)
    for (int j = g; j < 24; j += G16) {
        cf x[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = ld(s_d + b2_idx1(j, q, cl));
#pragma unroll
        for (int q = 1; q < 16; ++q) x[q] = cmulc(x[q], ld(s_tw + j * q));
        dft16<false>(x);
        if (colok) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (m >= 6 && m <= 9) continue;
                if (lvl1_is_image(j, m)) st(hp + (int64_t)lvl1_image_index(j, m) * HP, x[m]);
            }
        }
    }
    if (nyq) {
        // sixteen planes in this workgroup: one sum per column, over its 16 lanes g in index order (fixed order, no atomics)
        __syncthreads();                             // every lane has read its LDS inputs: the array is free
        float *s_acc = reinterpret_cast<float *>(s_d);
        s_acc[g * CT + cl] = acc;
        __syncthreads();
        if (threadIdx.x < CT && (int)blockIdx.x * CT + (int)threadIdx.x < planes) {
            float t = 0.0f;
#pragma unroll
            for (int q = 0; q < G16; ++q) t += s_acc[q * CT + threadIdx.x];
            partials[((int64_t)blockIdx.x * CT + threadIdx.x) * (TILES + 1) + TILES] = t;
        }
        return;
    }
    const float t = dpsx::block_sum(acc, scratch);
    if (threadIdx.x == 0) partials[(int64_t)plane * (TILES + 1) + tile] = t;
}

// ---- pass C: 32 image rows per 256-thread block, one complex transform per PAIR of rows and 16 lanes.
// norm == nullptr: the per-particle norm is finalised in the prologue from the forward half's partial sums (the order of
// k_finalize_norm: same bits in every block) and written to norm_out -- no separate finalisation launch.
constexpr int C2T = 256, C2ROWS = 2 * (C2T / G16);      // 32 rows per block
constexpr int C2ROWLDS = N;                             // float2 per transform

__global__ __launch_bounds__(C2T, 3) void k_pr_rows_inv2(const float2 *__restrict__ half, const float *__restrict__ norm,
                                                          const float *__restrict__ norm_partials, int norm_parts,
                                                          float *__restrict__ norm_out,
                                                          const uint8_t *__restrict__ ins, const float *__restrict__ g_extra,
                                                          float scale, int power, float neg_b, float *__restrict__ g_mo,
                                                          const float2 *__restrict__ tw_g, int c)
{
    __shared__ float2 s_tw[N];
    __shared__ __align__(16) float2 s_d[C2T / G16][C2ROWLDS];
    __shared__ float s_nrm[1];
    const int q = threadIdx.x & (G16 - 1), f = threadIdx.x / G16;
    const int64_t plane = blockIdx.y;
    const int a0 = blockIdx.x * C2ROWS;
    const int64_t n = plane / c, ch = plane % c, hw = (int64_t)IMG * IMG;
    for (int i = threadIdx.x; i < N; i += C2T) s_tw[i] = tw_g[i];
    if (!norm) dpsx::particle_norm_to_lds(norm_partials, norm_parts, n, s_nrm);
    // the clamp gate of the eight float4 units this lane finishes in the epilogue: fetched with everything else
    uchar4 gate[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = threadIdx.x + C2T * i, row = idx >> 6, l = idx & 63;
        gate[i] = *reinterpret_cast<const uchar4 *>(ins + plane * hw + (int64_t)(a0 + row) * IMG + 4 * l);
    }
    // ---- Z = X1 + i X2 on this lane's 24 frequencies ky = q + 16 k (Hermitian extension for ky > 192), inverse level 2
    {
        const float2 *in1 = half + (plane * IMG + a0 + 2 * f) * HP, *in2 = in1 + HP;
        cf x[24];
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const int ky = q + 16 * k;
            const bool direct = ky <= HALF;
            const int kx = direct ? ky : N - ky;
            const cf a = ld(in1 + kx), b = ld(in2 + kx);
            // direct: (a.re - b.im, a.im + b.re);  mirrored: conj(a) + i conj(b) = (a.re + b.im, b.re - a.im)
            x[k] = direct ? cf{a.x - b.y, a.y + b.x} : cf{a.x + b.y, b.x - a.y};
            // the self-conjugate bins carry only their real parts into a real row (as the one-row form, which drops the
            // imaginary part of its output): rounding-level imaginary parts must not cross into the partner row
            if (ky == 0 || ky == HALF) x[k] = cf{a.x, b.x};
        }
        dft24<false>(x);
#pragma unroll
        for (int k = 0; k < 24; ++k) st(&s_d[f][rot24(q, k)], x[k]);
    }
    __syncthreads();
    // ---- inverse level 1: butterflies q and (q < 8) q + 16; row r1 = real parts, r2 = imaginary parts
    cf o1[16], o2[16];
    {
#pragma unroll
        for (int m = 0; m < 16; ++m) o1[m] = ld(&s_d[f][rot24(m, q)]);
#pragma unroll
        for (int m = 1; m < 16; ++m) o1[m] = cmulc(o1[m], ld(s_tw + q * m));
        dft16<false>(o1);
        if (q < 8) {
#pragma unroll
            for (int m = 0; m < 16; ++m) o2[m] = ld(&s_d[f][rot24(m, q + 16)]);
#pragma unroll
            for (int m = 1; m < 16; ++m) o2[m] = cmulc(o2[m], ld(s_tw + (q + 16) * m));
            dft16<false>(o2);
        }
    }
    __syncthreads();                     // every lane has read its inputs: the transform's LDS becomes its two image rows
    {
        float *r1 = reinterpret_cast<float *>(&s_d[f][0]), *r2 = r1 + IMG;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (m >= 6 && m <= 9) continue;
            if (lvl1_is_image(q, m)) {
                const int b = lvl1_image_index(q, m);
                r1[b] = o1[m].x;
                r2[b] = o1[m].y;
            }
            if (q < 8 && lvl1_is_image(q + 16, m)) {
                const int b = lvl1_image_index(q + 16, m);
                r1[b] = o2[m].x;
                r2[b] = o2[m].y;
            }
        }
    }
    __syncthreads();
    const float nv = norm ? norm[n] : s_nrm[0];
    if (!norm && norm_out && blockIdx.x == 0 && ch == 0 && threadIdx.x == 0) norm_out[n] = nv;
    const float coef = (power == 2 ? -2.0f * scale : (nv == 0.0f ? 0.0f : -scale / nv)) * (1.0f / (float)N);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = threadIdx.x + C2T * i, row = idx >> 6, l = idx & 63;
        const float *rr = reinterpret_cast<const float *>(&s_d[row >> 1][0]) + (row & 1) * IMG + 4 * l;
        const float4 v = *reinterpret_cast<const float4 *>(rr);
        const int64_t o = plane * hw + (int64_t)(a0 + row) * IMG + 4 * l;
        float4 ex = make_float4(0, 0, 0, 0);
        if (g_extra) ex = *reinterpret_cast<const float4 *>(g_extra + o);
        float4 gq;
        gq.x = gate[i].x ? neg_b * (coef * v.x + ex.x) : 0.0f;
        gq.y = gate[i].y ? neg_b * (coef * v.y + ex.y) : 0.0f;
        gq.z = gate[i].z ? neg_b * (coef * v.z + ex.z) : 0.0f;
        gq.w = gate[i].w ? neg_b * (coef * v.w + ex.w) : 0.0f;
        *reinterpret_cast<float4 *>(g_mo + (n * 2 * c + ch) * hw + (int64_t)(a0 + row) * IMG + 4 * l) = gq;
    }
}

}  // namespace prfft
