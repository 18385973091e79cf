// Micro-benchmark: HBM efficiency of tiled vs linear access for the fused step's stream mix
// (4 fp32 input streams + 2 fp32 + 1 u8 output streams over [192 planes, 256, 256]).
//   hipcc --offload-arch=gfx950 -O3 tools/tilecopy.hip -o gpurun_out/tilecopy && gpurun_out/tilecopy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int H = 256, W = 256, PLANES = 192;

template <int TH, int TW>   // tile of TH rows x TW floats, 256 threads, all loads issued first
__global__ __launch_bounds__(256) void k_tiled(const float *a, const float *b, const float *c, const float *d,
                                               float *o1, float *o2, unsigned char *o3)
{
    constexpr int UPT = TH * TW / 4 / 256;           // float4 units per thread
    constexpr int TX = W / TW, TY = H / TH, TILES = TX * TY;
    extern __shared__ float dyn_lds[];
    if (o3 == nullptr) dyn_lds[threadIdx.x] = 1.0f;     // keep the allocation alive
    const int blk = blockIdx.x, xcd = blk & 7, k = blk >> 3;
    const int plane = (k / TILES) * 8 + xcd, t = k % TILES;
    const int h0 = (t / TX) * TH, w0 = (t % TX) * TW;
    float4 va[UPT], vb[UPT], vc[UPT], vd[UPT];
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
        const int i = threadIdx.x + u * 256, row = i / (TW / 4), cu = i % (TW / 4);
        const size_t o = (size_t)plane * H * W + (size_t)(h0 + row) * W + w0 + 4 * cu;
        va[u] = *(const float4 *)(a + o); vb[u] = *(const float4 *)(b + o);
        vc[u] = *(const float4 *)(c + o); vd[u] = *(const float4 *)(d + o);
    }
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
        const int i = threadIdx.x + u * 256, row = i / (TW / 4), cu = i % (TW / 4);
        const size_t o = (size_t)plane * H * W + (size_t)(h0 + row) * W + w0 + 4 * cu;
        float4 r1, r2;
        r1.x = va[u].x * 2.f - vb[u].x; r1.y = va[u].y * 2.f - vb[u].y; r1.z = va[u].z * 2.f - vb[u].z; r1.w = va[u].w * 2.f - vb[u].w;
        r2.x = r1.x + vc[u].x * vd[u].x; r2.y = r1.y + vc[u].y * vd[u].y; r2.z = r1.z + vc[u].z * vd[u].z; r2.w = r1.w + vc[u].w * vd[u].w;
        *(float4 *)(o1 + o) = r1; *(float4 *)(o2 + o) = r2;
        *(uchar4 *)(o3 + o) = make_uchar4(r1.x > 0, r1.y > 0, r1.z > 0, r1.w > 0);
    }
}

__global__ __launch_bounds__(256) void k_linear(const float *a, const float *b, const float *c, const float *d,
                                                float *o1, float *o2, unsigned char *o3, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const size_t o = i * 4;
    const float4 va = *(const float4 *)(a + o), vb = *(const float4 *)(b + o), vc = *(const float4 *)(c + o),
                 vd = *(const float4 *)(d + o);
    float4 r1, r2;
    r1.x = va.x * 2.f - vb.x; r1.y = va.y * 2.f - vb.y; r1.z = va.z * 2.f - vb.z; r1.w = va.w * 2.f - vb.w;
    r2.x = r1.x + vc.x * vd.x; r2.y = r1.y + vc.y * vd.y; r2.z = r1.z + vc.z * vd.z; r2.w = r1.w + vc.w * vd.w;
    *(float4 *)(o1 + o) = r1; *(float4 *)(o2 + o) = r2;
    *(uchar4 *)(o3 + o) = make_uchar4(r1.x > 0, r1.y > 0, r1.z > 0, r1.w > 0);
}

template <typename F>
static float timeit(F f, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main()
{
    const size_t n = (size_t)PLANES * H * W;
    float *in[8], *o1, *o2; unsigned char *o3;
    for (int i = 0; i < 8; ++i) { CK(hipMalloc(&in[i], n * 4)); CK(hipMemset(in[i], 0x3c, n * 4)); }
    CK(hipMalloc(&o1, n * 4)); CK(hipMalloc(&o2, n * 4)); CK(hipMalloc(&o3, n));
    const double bytes = (double)n * (4 * 4 + 2 * 4 + 1);
    int flip = 0;
    auto lin = [&] { flip ^= 4; hipLaunchKernelGGL(k_linear, dim3((n / 4 + 255) / 256), dim3(256), 0, 0, in[flip], in[flip+1], in[flip+2], in[flip+3], o1, o2, o3, n / 4); };
    float t = timeit(lin, 20);
    printf("linear           %7.1f us  %6.0f GB/s\n", t, bytes / t / 1e3);
#define TILED(TH_, TW_, LDS_) { auto f = [&] { flip ^= 4; hipLaunchKernelGGL((k_tiled<TH_, TW_>), dim3(PLANES * (H / TH_) * (W / TW_)), dim3(256), LDS_, 0, in[flip], in[flip+1], in[flip+2], in[flip+3], o1, o2, o3); }; \
        float tt = timeit(f, 20); printf("tiled %3d x %3d  lds %6d  %7.1f us  %6.0f GB/s\n", TH_, TW_, LDS_, tt, bytes / tt / 1e3); }
    TILED(64, 64, 0) TILED(64, 64, 20000) TILED(64, 64, 26000) TILED(64, 64, 32000) TILED(64, 64, 34816) TILED(64, 64, 50000) TILED(64, 64, 80000)
    TILED(32, 128, 0) TILED(16, 256, 0) TILED(32, 64, 0) TILED(32, 64, 17000) TILED(64, 128, 0)
    CK(hipDeviceSynchronize());
    return 0;
}
