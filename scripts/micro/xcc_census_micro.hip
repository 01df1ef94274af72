// Micro-benchmark: which hardware XCD does block b of a launch run on, for several grid / block / LDS shapes?
// Build: hipcc -O3 --offload-arch=gfx950 xcc_census_micro.hip -o xcc_census_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_census(int* out, int spin) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        out[lin] = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 15u);
        lds[0] = 1.f;
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(10);
}

int main() {
    int* d;
    CK(hipMalloc(&d, 1 << 20));
    struct { dim3 grid; int threads; int lds; int spin; } cases[] = {
        {dim3(256), 256, 0, 0},   {dim3(256), 256, 0, 50},  {dim3(192), 512, 9000, 50}, {dim3(512), 512, 9000, 50},
        {dim3(256), 512, 17000, 50}, {dim3(513), 512, 9000, 50}, {dim3(16, 2, 4), 256, 4300, 50}, {dim3(1), 1024, 100000, 10},
        {dim3(256), 256, 0, 0},
    };
    for (auto& c : cases) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_census), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(k_census, c.grid, dim3(c.threads), c.lds, 0, d, c.spin);
        CK(hipDeviceSynchronize());
        const int n = c.grid.x * c.grid.y * c.grid.z;
        std::vector<int> h(n);
        CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int b = 0; b < n; ++b)
            if (h[b] != h[b % 8]) bad++;
        printf("grid (%d,%d,%d) x %d threads, lds %d: first 16 =", c.grid.x, c.grid.y, c.grid.z, c.threads, c.lds);
        for (int b = 0; b < 16 && b < n; ++b) printf(" %d", h[b]);
        printf("   blocks off the b%%8 pattern: %d of %d\n", bad, n);
    }
    return 0;
}
