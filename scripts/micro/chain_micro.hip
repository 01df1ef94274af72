// Micro-benchmark: what does one dependent kernel in a chain cost on MI355X?
// Build: hipcc -O3 --offload-arch=gfx950 chain_micro.hip -o chain_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1;} } while(0)

__global__ void k_empty(float* p) {}
__global__ void k_touch(const float* __restrict__ in, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    out[i] = in[i] + 1.0f;
}
// stream `bytes` of weights: each wave reads rows of 4KB (4 x float4 per lane), ROWS rows per wave, sums, writes 1 float per row
template <int ROWS>
__global__ __launch_bounds__(256) void k_gemv(const float* __restrict__ W, const float* __restrict__ x, float* __restrict__ y, int nrows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * (blockDim.x >> 6) + wave;
    const int row0 = gw * ROWS;
    if (row0 >= nrows) return;
    float4 w[ROWS][4];
#pragma unroll
    for (int j = 0; j < ROWS; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) w[j][q] = *reinterpret_cast<const float4*>(W + (size_t)(row0 + j) * 1024 + q * 256 + lane * 4);
    float4 xv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) xv[q] = *reinterpret_cast<const float4*>(x + q * 256 + lane * 4);
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) s += w[j][q].x * xv[q].x + w[j][q].y * xv[q].y + w[j][q].z * xv[q].z + w[j][q].w * xv[q].w;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) y[row0 + j] = s;
    }
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t WB = 512ull << 20;  // 512 MB of weights so every pass misses the caches
    float *W, *x, *y; CK(hipMalloc(&W, WB)); CK(hipMalloc(&x, 1 << 20)); CK(hipMalloc(&y, 64 << 20));
    CK(hipMemset(W, 0, WB)); CK(hipMemset(x, 0, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch, int n, double bytes_per) {
        for (int i = 0; i < 20; ++i) launch(i);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int i = 0; i < n; ++i) launch(i);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double us = ms * 1e3 / n;
        printf("%-46s %8.2f us/kernel", name, us);
        if (bytes_per > 0) printf("  %8.1f GB/s", bytes_per / us / 1e3);
        printf("\n");
    };
    timeit("empty 256x256", [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, y); }, 2000, 0);
    timeit("empty 1x64", [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, y); }, 2000, 0);
    timeit("touch 256x256 (dependent chain x->y->x)", [&](int i) { hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, st, (i & 1) ? y : x, (i & 1) ? x : y); }, 2000, 0);
    // gemv variants: rows of 4KB; matrices of 4MB (1024 rows), 12.6MB (3072), 33.5MB (8192); rotate through the 512MB buffer
    struct Cfg { int nrows; const char* nm; } cfgs[] = {{1024, "4MB"}, {3072, "12.6MB"}, {8192, "33.5MB"}};
    for (auto c : cfgs) {
        const size_t mat = (size_t)c.nrows * 1024;  // floats
        const int nmat = (int)((WB / 4) / mat);
        char nm[128];
#define RUN(ROWS, NW)                                                                                   \
        {                                                                                               \
            int waves = (c.nrows + ROWS - 1) / ROWS; int blocks = (waves + NW - 1) / NW;                \
            snprintf(nm, sizeof nm, "gemv %s rows/wave=%d waves/blk=%d blocks=%d", c.nm, ROWS, NW, blocks); \
            timeit(nm, [&](int i) { hipLaunchKernelGGL(k_gemv<ROWS>, dim3(blocks), dim3(NW * 64), 0, st, W + (size_t)(i % nmat) * mat, x, y, c.nrows); }, 600, mat * 4.0); \
        }
        RUN(1, 4) RUN(2, 4) RUN(4, 4) RUN(4, 1) RUN(2, 1) RUN(1, 1) RUN(8, 4) RUN(4, 8)
    }
    // same but L2/MALL resident: always the same 4MB matrix
    {
        const int nrows = 1024; char nm[128];
        int blocks = 256;
        snprintf(nm, sizeof nm, "gemv 4MB cache-resident rows/wave=1 blocks=256");
        timeit(nm, [&](int) { hipLaunchKernelGGL(k_gemv<1>, dim3(blocks), dim3(256), 0, st, W, x, y, nrows); }, 600, 4.0 * 1024 * 1024);
    }
    return 0;
}
