// Micro-benchmark of dense_gemm_kernel (chatterbox_amd/csrc/dense_gemm.hip) on the CFM estimator's shapes: every tile x ring
// depth per shape, back to back in a chain that alternates two output buffers (the next launch reads what another XCD wrote,
// as in the estimator), HIP events around 200 launches.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I include -I chatterbox_amd/csrc scripts/micro/dense_micro.hip -o scripts/micro/dense_micro
#include "../../chatterbox_amd/csrc/dense_gemm.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace t3k;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int BM, int BN, int EPI, bool LN, int D>
float time_one(const DenseGemmArgs& a0, float* y2, int reps) {
    if (a0.N % BN || (a0.K / 64) % D) return -1.f;
    DenseGemmArgs a = a0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK((launch_dense_d<BM, BN, EPI, LN, D>(a, 0)));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) {
        a.y = (i & 1) ? y2 : a0.y;
        if (EPI == EPI_RESID) a.resid = a.y;
        CK((launch_dense_d<BM, BN, EPI, LN, D>(a, 0)));
    }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

template <int EPI, bool LN>
void sweep(const char* name, DenseGemmArgs a, float* y2) {
    const double gf = 2.0 * a.M * a.N * a.K * 1e-9;
    printf("%-8s M=%d N=%d K=%d (%.2f GFLOP; %.1f us at 157 TFLOP/s)\n", name, a.M, a.N, a.K, gf, gf / 157.3 * 1e3);
#define ROW(BM, BN) printf("   %dx%d  D1 %6.2f  D2 %6.2f  D4 %6.2f  D8 %6.2f us\n", BM, BN, time_one<BM, BN, EPI, LN, 1>(a, y2, 200), \
        time_one<BM, BN, EPI, LN, 2>(a, y2, 200), time_one<BM, BN, EPI, LN, 4>(a, y2, 200), BM + BN <= 64 ? time_one<BM, BN, EPI, LN, 8>(a, y2, 200) : -1.f)
    ROW(32, 32); ROW(32, 64);
#undef ROW
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 1000;
    const size_t big = (size_t)4096 * 2048;
    float *A, *W, *y, *y2, *b, *lw, *lb;
    CK(hipMalloc(&A, big * 4)); CK(hipMalloc(&W, big * 4)); CK(hipMalloc(&y, big * 4)); CK(hipMalloc(&y2, big * 4));
    CK(hipMalloc(&b, 4096 * 4)); CK(hipMalloc(&lw, 4096 * 4)); CK(hipMalloc(&lb, 4096 * 4));
    std::vector<float> h(big);
    for (size_t i = 0; i < big; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.f - 0.5f;
    CK(hipMemcpy(A, h.data(), big * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), big * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(lw, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(lb, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMemset(y, 0, big * 4)); CK(hipMemset(y2, 0, big * 4));
    DenseGemmArgs a;
    a.A = A; a.W = W; a.bias = b; a.y = y; a.M = M;
    auto shape = [&](int N, int K, int epi, bool ln) {
        a.N = N; a.K = K; a.lda = K; a.ldy = N; a.epi = epi; a.resid = epi == EPI_RESID ? y : nullptr;
        a.ln_colsum = ln ? lw : nullptr;
    };
    shape(768, 256, EPI_STORE, false);  sweep<EPI_STORE, false>("warm-up", a, y2);
    shape(1536, 256, EPI_STORE, true);   sweep<EPI_STORE, true>("qkv+LN", a, y2);
    shape(1536, 256, EPI_STORE, false);  sweep<EPI_STORE, false>("qkv", a, y2);
    shape(1024, 256, EPI_GELU, true);   sweep<EPI_GELU, true>("ff1+LN", a, y2);
    shape(1024, 256, EPI_GELU, false);  sweep<EPI_GELU, false>("ff1", a, y2);
    shape(256, 512, EPI_RESID, false);  sweep<EPI_RESID, false>("to_out", a, y2);
    shape(256, 1024, EPI_RESID, false); sweep<EPI_RESID, false>("ff2", a, y2);
    shape(256, 768, EPI_STORE, false);  sweep<EPI_STORE, false>("conv3", a, y2);
    shape(256, 1536, EPI_STORE, false); sweep<EPI_STORE, false>("conv3x2", a, y2);
    return 0;
}
