// Micro-benchmark of the many-row bf16 GEMM (batch_kernels.hip) on the decode shapes at M = 128, cold weights.
// Build (from the repo root, after `make -C chatterbox_amd/csrc`):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -x hip scripts/micro/gemm_micro.cpp chatterbox_amd/csrc/batch_kernels.o -o scripts/micro/gemm_micro
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#include "../../chatterbox_amd/csrc/kernels.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 128;
    const int d = 1024, F = 4096, H = 16, max_len = 512;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int NCOPY = 24;   // rotate through copies so the weights are never cached
    const size_t wmax = (size_t)8448 * d;    // elements of the largest matrix (head: 8194 rows; gate_up: 8192), padded
    uint16_t *W, *A, *yact;
    float *y, *q, *kc, *vc, *part, *cosv, *sinv;
    int* pos;
    unsigned long long* dbg;
    CK(hipMalloc(&W, wmax * 2 * NCOPY)); CK(hipMemset(W, 0, wmax * 2 * NCOPY));
    CK(hipMalloc(&A, (size_t)M * F * 2)); CK(hipMemset(A, 0, (size_t)M * F * 2));
    CK(hipMalloc(&yact, (size_t)M * F * 2));
    CK(hipMalloc(&y, (size_t)M * 8200 * 4)); CK(hipMemset(y, 0, (size_t)M * 8200 * 4));
    CK(hipMalloc(&q, (size_t)M * d * 4));
    CK(hipMalloc(&kc, (size_t)M * H * max_len * 64 * 4)); CK(hipMalloc(&vc, (size_t)M * H * max_len * 64 * 4));
    CK(hipMalloc(&part, (size_t)8 * M * d * 4));
    CK(hipMalloc(&cosv, max_len * 64 * 4)); CK(hipMalloc(&sinv, max_len * 64 * 4));
    CK(hipMemset(cosv, 0, max_len * 64 * 4)); CK(hipMemset(sinv, 0, max_len * 64 * 4));
    CK(hipMalloc(&pos, M * 4));
    std::vector<int> hp(M);
    for (int i = 0; i < M; ++i) hp[i] = 100 + (i % 50);
    CK(hipMemcpy(pos, hp.data(), M * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dbg, 4096 * 4 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; int epi, N, K; };
    const Case cases[] = {{"qkv_rope_kv", t3k::EPI_ROPE_KV, 3 * d, d}, {"o_proj_resid", t3k::EPI_RESID, d, d},
                          {"gate_up_swiglu", t3k::EPI_SWIGLU, 2 * F, d}, {"down_resid", t3k::EPI_RESID, d, F},
                          {"head_store", t3k::EPI_STORE, 8194, d}};
    for (const Case& c : cases) {
        auto launch = [&](int i, unsigned long long* dg) {
            t3k::GemmArgs a;
            a.W = W + (size_t)(i % NCOPY) * wmax; a.A = A; a.M = M; a.N = c.N; a.K = c.K;
            a.y = y; a.resid = y; a.ldy = c.epi == t3k::EPI_STORE ? c.N : d; a.yact = yact;
            a.posmap = pos; a.rope_cos = cosv; a.rope_sin = sinv; a.q = q; a.kcache = kc; a.vcache = vc;
            a.n_heads = H; a.max_len = max_len; a.part = part; a.part_floats = (size_t)8 * M * d; a.dbg = dg;
            return t3k::launch_gemm(a, c.epi, 1, st);
        };
        for (int i = 0; i < 30; ++i) CK(launch(i, nullptr));
        CK(hipStreamSynchronize(st));
        const int n = 240;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < n; ++i) CK(launch(i, nullptr));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemset(dbg, 0, 4096 * 4 * 8));
        CK(launch(7, dbg));
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> h(4096 * 4);
        CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, tend = 0;
        std::vector<double> p1, p2, p3;
        int nb = 0;
        for (int b = 0; b < 4096; ++b) {
            if (!h[4 * b]) continue;
            nb++;
            t0 = std::min(t0, h[4 * b]);
            tend = std::max(tend, std::max(h[4 * b + 2], h[4 * b + 3]));
        }
        for (int b = 0; b < 4096; ++b) {
            if (!h[4 * b]) continue;
            p1.push_back((h[4 * b + 1] - t0) * 0.01);
            p2.push_back((h[4 * b + 2] - t0) * 0.01);
            if (h[4 * b + 3]) p3.push_back((h[4 * b + 3] - t0) * 0.01);
        }
        auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        const double wmb = (double)c.N * c.K * 2 / 1e6;
        printf("%-16s M=%d N=%d K=%d  %7.2f us/launch (back to back, incl. fold)  W %.1f MB -> %.0f GB/s | blocks %d: first tile in LDS %.2f, loop done %.2f, "
               "epilogue done %.2f (median us after first block start), last block end %.2f\n",
               c.name, M, c.N, c.K, ms * 1e3 / n, wmb, wmb / (ms * 1e3 / n) * 1e3, nb, med(p1), med(p2), med(p3), (tend - t0) * 0.01);
    }
    return 0;
}
