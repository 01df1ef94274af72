// Micro-benchmark: does a kernel that keeps loads in flight on ANOTHER stream make the dependent-kernel boundary of
// a chain more expensive?  (Decides whether a side-stream L2 prefetcher can pay for itself.)
// Build: hipcc -O3 --offload-arch=gfx950 bg_boundary_micro.hip -o bg_boundary_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_touch(const float* __restrict__ in, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    out[i] = in[i] + 1.0f;
}

// background: `blocks` x 256 threads; mode 0 = sleep only, 1 = one 8 KB wave-load in flight per wave over a big buffer,
// 2 = poll one word with agent-scope atomic loads.  Runs until *stop != 0 or iters exhausted.
__global__ void k_bg(const float* __restrict__ buf, size_t n_lines, int mode, int iters, const int* stop, float* sink) {
    float acc = 0.f;
    size_t line = ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
    for (int it = 0; it < iters; ++it) {
        if (mode == 1) {
            acc += buf[(line % n_lines) * 32];
            line += (size_t)gridDim.x * blockDim.x;
        } else if (mode == 2) {
            if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            __builtin_amdgcn_s_sleep(20);
        } else {
            __builtin_amdgcn_s_sleep(100);
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    float *x, *y, *big, *sink;
    int* stop;
    const size_t BIG = 1ull << 30;
    CK(hipMalloc(&x, 1 << 20)); CK(hipMalloc(&y, 1 << 20)); CK(hipMalloc(&big, BIG)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&stop, 64));
    CK(hipMemset(x, 0, 1 << 20)); CK(hipMemset(big, 0, BIG)); CK(hipMemset(stop, 0, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 3000;
    // the chain is a captured graph of 500 dependent kernels (device-side boundary cost, not host launch cost)
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(a, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, a, (i & 1) ? y : x, (i & 1) ? x : y);
    CK(hipStreamEndCapture(a, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    auto chain = [&](const char* name) -> int {
        CK(hipGraphLaunch(ge, a));
        CK(hipEventRecord(e0, a));
        for (int i = 0; i < N / 500; ++i) CK(hipGraphLaunch(ge, a));
        CK(hipEventRecord(e1, a));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-62s %7.3f us per dependent kernel\n", name, ms * 1e3 / N);
        return 0;
    };
    if (chain("chain alone")) return 1;
    struct { const char* name; int blocks, mode, iters; } bg[] = {
        {"beside 8 blocks sleeping", 8, 0, 4000},
        {"beside 8 blocks polling a word", 8, 2, 100000},
        {"beside 8 blocks, one 8 KB load in flight per wave", 8, 1, 20000},
        {"beside 64 blocks, one 8 KB load in flight per wave", 64, 1, 20000},
        {"beside 256 blocks, one 8 KB load in flight per wave", 256, 1, 20000},
        {"beside 32 blocks, one 8 KB load in flight per wave", 32, 1, 20000},
        {"beside 8 blocks (1 wave each busy), 8 KB loads", 8, 1, 20000},
    };
    for (auto& g : bg) {
        hipLaunchKernelGGL(k_bg, dim3(g.blocks), dim3(256), 0, b, (const float*)big, BIG / 128, g.mode, g.iters, (const int*)stop, sink);
        if (chain(g.name)) return 1;
        hipEvent_t eb;
        CK(hipEventCreate(&eb));
        CK(hipEventRecord(eb, b));
        const bool still = hipEventQuery(eb) == hipErrorNotReady;
        printf("    (background kernel %s when the chain ended)\n", still ? "still running" : "had finished");
        CK(hipStreamSynchronize(b));
    }
    printf("done\n");
    return 0;
}
