// Micro-benchmark / prototype: ONE persistent launch for a whole chain of dependent GEMV phases (the batch-1 decode
// layer: qkv 12.6 MB -> attention 4.2 MB -> o_proj 4.2 MB -> gate 16.8 MB + up 16.8 MB -> down 16.8 MB, 30 layers,
// two activation rows, 5 dependency edges per layer), with the weight stream running ACROSS the edges.
//
//   * 256 workgroups (one per CU), 8 compute waves + 1 communication wave each.
//   * Compute waves w and w+4 share the rows of row group w&3 and take one half of K each; a wave keeps its weights
//     in REGISTERS: a ring of four 32-VGPR slabs (qkv | attention+o_proj | gate | up | down, five slabs per layer, so
//     the ring turns by one per layer and the layer body is instantiated four times).  A slab is re-issued as soon as
//     the phase that used it has been published, so up to three slabs per wave are in flight while the chain waits at
//     an edge.  Loads return in issue order and are consumed in issue order: counted vmcnt waits suffice.
//   * A compute wave with prefetch loads in flight cannot get a small load back quickly (in-order return) and cannot
//     drain a store without draining its prefetch, so ALL hand-off traffic goes through the communication wave.
//   * Hand-off without flags: the data is its own flag.  Every hand-off buffer is poisoned (0xFFFFFFFF); a producer
//     publishes with sc1 stores; a consumer fetches the whole vector with 16-byte sc1 buffer loads and repeats the
//     batch until no word is poison (MI355X_MICROARCH.md, 'Valid forms': sc1 store / sc1 load at agent scope; each
//     dword is single-copy atomic and is checked on its own, so no ordering between words is needed).  A workgroup
//     that has fetched the complete input of phase e knows every workgroup has finished reading the input of phase
//     e-1, and re-poisons its own slice of that buffer; s_waitcnt vmcnt(0) before its next publish orders the poison
//     before anything that could let another workgroup reach the buffer's next use.
//   * Every spin is bounded; a timeout sets an error word and the kernel runs to its end without further waiting.
// Check: all weights are 1/K and the vector starts at 1.0, so every output of every phase must be exactly 1.0.
// First version (4 compute waves, 3 x 64-VGPR slabs, counter + separate fetch per edge): 59.6 us per layer, the
// edge being five serial memory round trips and the 32-load fetch of the down phase issued one load at a time.
// Build: hipcc -O3 --offload-arch=gfx950 persist_micro.hip -o persist_micro
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int kWG = 256, kCW = 8, kThreads = (kCW + 1) * 64, kGroups = kWG * 4;   // 1024 row groups, 2 waves each
constexpr int kLogWG = 100;
constexpr unsigned kPoison = 0xFFFFFFFFu;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

struct Bufs {
    const float* W;        // all layers, phase after phase, row-major [rows][K]
    float* xA;             // [2][1024]  layer input / down output
    float* q;              // [2][3072]
    float* a;              // [2][1024]
    float* xB;             // [2][1024]
    float* h;              // [2][8192]
    int* err;
    unsigned long long* stamps;   // [kWG][2]
    unsigned long long* log;      // [layers][5][3]
    int layers;
    int max_polls;
};

// phase: RPG rows per row group, row length K
template <int RPG_, int K_>
struct Ph {
    static constexpr int RPG = RPG_, K = K_, Q = K_ / 512;   // Q = float4 per lane per row for one K half
    static constexpr size_t floats = (size_t)kGroups * RPG_ * K_;
};
using P0 = Ph<3, 1024>;   // qkv
using P1 = Ph<1, 1024>;   // attention (KV bytes)
using P2 = Ph<1, 1024>;   // o_proj
using P3 = Ph<4, 1024>;   // gate
using P4 = Ph<4, 1024>;   // up
using P5 = Ph<1, 4096>;   // down
constexpr size_t kO1 = P0::floats, kO2 = kO1 + P1::floats, kO3 = kO2 + P2::floats, kO4 = kO3 + P3::floats, kO5 = kO4 + P4::floats;
constexpr size_t kLayerFloats = kO5 + P5::floats;

struct Ctx {
    Bufs bf;
    float* x_lds;      // [2][4096]
    float* part;       // [2][8 waves][8]
    int b, lane, wave, grp, half;
    __amdgpu_buffer_rsrc_t wrs;   // the weight stream
    bool comm;
    int err;
};

// weight loads: buffer loads with one shared lane offset register, the wave-uniform part of the address in soffset
// (base = float index of the phase's first weight, counted from W; the whole stream is below 4 GB)
template <typename P, int OFF>
__device__ __forceinline__ void issue(f4 (&R)[8], size_t base, const Ctx& c) {
#pragma unroll
    for (int r = 0; r < P::RPG; ++r)
#pragma unroll
        for (int qq = 0; qq < P::Q; ++qq) {
            const unsigned so = (unsigned)((base + ((size_t)c.grp * P::RPG + r) * P::K + c.half * (P::K / 2) + qq * 256) * 4);
            const u4 d = __builtin_amdgcn_raw_buffer_load_b128(c.wrs, c.lane * 16, so, 2 /* nt */);
            R[OFF + r * P::Q + qq] = __builtin_bit_cast(f4, d);
        }
}

// partial dot products (this wave's K half) of the group's rows with the two activation rows in LDS
template <typename P, int OFF>
__device__ __forceinline__ void consume(const f4 (&R)[8], const Ctx& c, int set) {
    float acc[P::RPG][2];
#pragma unroll
    for (int r = 0; r < P::RPG; ++r) acc[r][0] = acc[r][1] = 0.f;
    const float* x = c.x_lds + c.half * (P::K / 2) + c.lane * 4;
#pragma unroll
    for (int qq = 0; qq < P::Q; ++qq) {
        const f4 x0 = *reinterpret_cast<const f4*>(x + qq * 256);
        const f4 x1 = *reinterpret_cast<const f4*>(x + 4096 + qq * 256);
#pragma unroll
        for (int r = 0; r < P::RPG; ++r) {
            const f4 w = R[OFF + r * P::Q + qq];
            acc[r][0] += w.x * x0.x + w.y * x0.y + w.z * x0.z + w.w * x0.w;
            acc[r][1] += w.x * x1.x + w.y * x1.y + w.z * x1.z + w.w * x1.w;
        }
    }
    float* out = c.part + (set * kCW + c.wave) * 8;
#pragma unroll
    for (int r = 0; r < P::RPG; ++r)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            float s = acc[r][v];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (c.lane == 0) out[r * 2 + v] = s;
        }
}

// communication wave: fetch both rows of a K-long vector (row stride gstride floats) with NL 16-byte sc1 loads per lane,
// all in flight at once, until no word is poison; then put them into LDS
template <int K>
__device__ __forceinline__ void fetch_poll(const float* g, int gstride, Ctx& c) {
    constexpr int H = K / 256, NL = 2 * H;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, 1 << 30, 0x00020000);
    u4 d[NL];
    for (int it = 0;; ++it) {
#pragma unroll
        for (int i = 0; i < NL; ++i)
            d[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((i / H) * gstride + (i % H) * 256 + c.lane * 4) * 4, 0, 16 /* sc1 */);
        bool bad = false;
#pragma unroll
        for (int i = 0; i < NL; ++i) bad |= d[i].x == kPoison || d[i].y == kPoison || d[i].z == kPoison || d[i].w == kPoison;
        if (!__any(bad)) break;
        if (c.err || it >= c.bf.max_polls) { c.err = 1; break; }
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) *reinterpret_cast<u4*>(c.x_lds + (i / H) * 4096 + (i % H) * 256 + c.lane * 4) = d[i];
}

// communication wave: the workgroup's slice of a hand-off buffer is rows (b*4 + g)*RPG + r of both vectors
template <int RPG>
__device__ __forceinline__ void publish(float* g, int gstride, const Ctx& c, int set) {
    if (c.lane < 4 * RPG * 2) {
        const int gi = c.lane / (RPG * 2), r = (c.lane / 2) % RPG, v = c.lane & 1;
        const float* p = c.part + set * kCW * 8;
        const float val = p[gi * 8 + r * 2 + v] + p[(gi + 4) * 8 + r * 2 + v];
        __hip_atomic_store(g + (size_t)v * gstride + (c.b * 4 + gi) * RPG + r, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int RPG>
__device__ __forceinline__ void poison(float* g, int gstride, const Ctx& c) {
    if (c.lane < 4 * RPG * 2) {
        const int gi = c.lane / (RPG * 2), r = (c.lane / 2) % RPG, v = c.lane & 1;
        __hip_atomic_store(reinterpret_cast<unsigned*>(g) + (size_t)v * gstride + (c.b * 4 + gi) * RPG + r, kPoison, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
}

#define LOG(e, k) do { if (c.comm && c.b == kLogWG && c.lane == 0) c.bf.log[(l * 5 + (e)) * 3 + (k)] = wall_clock64(); } while (0)
#define DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

#ifdef PUBLISH_FIRST
#define BAR3() BAR()   // third barrier per edge: the publish stores are issued before the next slab's loads
#else
#define BAR3()
#endif
#define BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// The two kinds of wave run separate loops with the same sequence of barriers (two per edge), so the register
// allocation of one does not carry the other's live ranges (128 VGPRs of slabs / 128 VGPRs of fetched input).
// Compute side of one layer; on entry A holds the qkv slab, B attention + o_proj, C gate, D up (issued earlier).  On exit
// the next layer's slabs sit one ring position further: qkv in B, attention + o_proj in C, gate in D, up in A.
__device__ __forceinline__ void layer_compute(f4 (&A)[8], f4 (&B)[8], f4 (&C)[8], f4 (&D)[8], Ctx& c, int l) {
    const Bufs& bf = c.bf;
    const size_t Wl = (size_t)l * kLayerFloats, Wn = Wl + kLayerFloats;
    const bool more = l + 1 < bf.layers;
#ifdef DEFER_ISSUE
    // a freed slab is re-issued only after the NEXT edge's input has arrived, so the communication wave's polling loads
    // never queue behind a fresh slab in this CU's memory pipeline
    BAR(); if (l > 0) issue<P4, 0>(D, Wl + kO4, c); consume<P0, 0>(A, c, 0); BAR(); BAR3();
    BAR(); issue<P5, 0>(A, Wl + kO5, c); consume<P1, 0>(B, c, 0); BAR(); BAR3();
    BAR(); consume<P2, 2>(B, c, 0); BAR(); BAR3();
    BAR(); if (more) issue<P0, 0>(B, Wn, c); consume<P3, 0>(C, c, 0); consume<P4, 0>(D, c, 1); BAR(); BAR3();
    BAR(); if (more) { issue<P1, 0>(C, Wn + kO1, c); issue<P2, 2>(C, Wn + kO2, c); issue<P3, 0>(D, Wn + kO3, c); }
    consume<P5, 0>(A, c, 0); BAR(); BAR3();
#else
    BAR(); consume<P0, 0>(A, c, 0); BAR(); BAR3();
    issue<P5, 0>(A, Wl + kO5, c);
    BAR(); consume<P1, 0>(B, c, 0); BAR(); BAR3();
    BAR(); consume<P2, 2>(B, c, 0); BAR(); BAR3();
    if (more) issue<P0, 0>(B, Wn, c);
    BAR(); consume<P3, 0>(C, c, 0); consume<P4, 0>(D, c, 1); BAR(); BAR3();
    if (more) { issue<P1, 0>(C, Wn + kO1, c); issue<P2, 2>(C, Wn + kO2, c); issue<P3, 0>(D, Wn + kO3, c); }
    BAR(); consume<P5, 0>(A, c, 0); BAR(); BAR3();
    if (more) issue<P4, 0>(A, Wn + kO4, c);
#endif
}

// Communication side of one layer: per edge, fetch the input (polling), re-poison the buffer before the previous one,
// barrier (input in LDS), barrier (partial sums in LDS), publish.
__device__ __forceinline__ void layer_comm(Ctx& c, int l) {
    const Bufs& bf = c.bf;
    // ---- e0 qkv: xA -> q
    fetch_poll<1024>(bf.xA, 1024, c); poison<4>(bf.h, 8192, c); poison<4>(bf.h + 4096, 8192, c); LOG(0, 0);
    BAR(); BAR();
    LOG(0, 1); publish<3>(bf.q, 3072, c, 0); LOG(0, 2);
    BAR3();
    // ---- e1 attention: q -> a
    fetch_poll<1024>(bf.q, 3072, c); poison<1>(bf.xA, 1024, c); LOG(1, 0);
    BAR(); BAR();
    LOG(1, 1); publish<1>(bf.a, 1024, c, 0); LOG(1, 2);
    BAR3();
    // ---- e2 o_proj: a -> xB
    fetch_poll<1024>(bf.a, 1024, c); poison<3>(bf.q, 3072, c); LOG(2, 0);
    BAR(); BAR();
    LOG(2, 1); publish<1>(bf.xB, 1024, c, 0); LOG(2, 2);
    BAR3();
    // ---- e3 gate + up: xB -> h
    fetch_poll<1024>(bf.xB, 1024, c); poison<1>(bf.a, 1024, c); LOG(3, 0);
    BAR(); BAR();
    LOG(3, 1); publish<4>(bf.h, 8192, c, 0); publish<4>(bf.h + 4096, 8192, c, 1); LOG(3, 2);
    BAR3();
    // ---- e4 down: h[0:4096] -> xA
    fetch_poll<4096>(bf.h, 8192, c); poison<1>(bf.xB, 1024, c); LOG(4, 0);
    BAR(); BAR();
    LOG(4, 1); publish<1>(bf.xA, 1024, c, 0); LOG(4, 2);
    BAR3();
}

__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 3))) void persist_kernel(Bufs bf) {
    __shared__ __attribute__((aligned(16))) float x_lds[2 * 4096];
    __shared__ float part[2 * kCW * 8];
    Ctx c;
    c.bf = bf; c.x_lds = x_lds; c.part = part;
    c.b = blockIdx.x; c.lane = threadIdx.x & 63; c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    c.comm = c.wave == kCW;
    c.grp = c.b * 4 + (c.wave & 3); c.half = (c.wave >> 2) & 1;
    c.err = 0;
    c.wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bf.W), 0, 0xFFFFFFFF, 0x00020000);
    if (threadIdx.x == 0) bf.stamps[2 * c.b] = wall_clock64();
    if (c.comm) {
        for (int l = 0; l < bf.layers; ++l) layer_comm(c, l);
        if (c.err && c.lane == 0) atomicExch(bf.err, 1);
    } else {
        f4 A[8], B[8], C[8], D[8];
        issue<P0, 0>(A, 0, c);
        issue<P1, 0>(B, kO1, c); issue<P2, 2>(B, kO2, c);
        issue<P3, 0>(C, kO3, c);
        issue<P4, 0>(D, kO4, c);
        for (int l = 0; l < bf.layers; l += 4) {
            layer_compute(A, B, C, D, c, l);
            if (l + 1 < bf.layers) layer_compute(B, C, D, A, c, l + 1);
            if (l + 2 < bf.layers) layer_compute(C, D, A, B, c, l + 2);
            if (l + 3 < bf.layers) layer_compute(D, A, B, C, c, l + 3);
        }
    }
    BAR();
    if (threadIdx.x == 0) bf.stamps[2 * c.b + 1] = wall_clock64();
}

int main(int argc, char** argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 30;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    Bufs bf{};
    float* W;
    const size_t wfloats = kLayerFloats * (size_t)layers;
    CK(hipMalloc(&W, wfloats * 4));
    {   // every weight = 1 / K of its phase, so every dot product of an all-ones vector is 1
        std::vector<float> hl(kLayerFloats);
        size_t o = 0;
        auto fill = [&](size_t n, int K) { for (size_t i = 0; i < n; ++i) hl[o + i] = 1.0f / K; o += n; };
        fill(P0::floats, 1024); fill(P1::floats, 1024); fill(P2::floats, 1024); fill(P3::floats, 1024); fill(P4::floats, 1024); fill(P5::floats, 4096);
        for (int l = 0; l < layers; ++l) CK(hipMemcpy(W + (size_t)l * kLayerFloats, hl.data(), kLayerFloats * 4, hipMemcpyHostToDevice));
    }
    bf.W = W;
    CK(hipMalloc(&bf.xA, 2 * 1024 * 4)); CK(hipMalloc(&bf.q, 2 * 3072 * 4)); CK(hipMalloc(&bf.a, 2 * 1024 * 4));
    CK(hipMalloc(&bf.xB, 2 * 1024 * 4)); CK(hipMalloc(&bf.h, 2 * 8192 * 4));
    CK(hipMalloc(&bf.err, 4)); CK(hipMalloc(&bf.stamps, kWG * 2 * 8)); CK(hipMalloc(&bf.log, (size_t)layers * 5 * 3 * 8));
    bf.layers = layers;
    bf.max_polls = 100000;
    printf("layer weights %.1f MB, %d layers, %d workgroups x %d threads\n", kLayerFloats * 4 / 1e6, layers, kWG, kThreads);
    std::vector<float> ones(2 * 1024, 1.0f), xa(2 * 1024);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(bf.err, 0, 4));
        CK(hipMemset(bf.q, 0xFF, 2 * 3072 * 4)); CK(hipMemset(bf.a, 0xFF, 2 * 1024 * 4));
        CK(hipMemset(bf.xB, 0xFF, 2 * 1024 * 4)); CK(hipMemset(bf.h, 0xFF, 2 * 8192 * 4));
        CK(hipMemcpy(bf.xA, ones.data(), 2 * 1024 * 4, hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(persist_kernel, dim3(kWG), dim3(kThreads), 0, st, bf);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        int err = 0;
        CK(hipMemcpy(&err, bf.err, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(xa.data(), bf.xA, 2 * 1024 * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (float v : xa) bad += !(std::fabs(v - 1.0f) < 1e-4f);
        std::vector<unsigned long long> hs(kWG * 2);
        CK(hipMemcpy(hs.data(), bf.stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < kWG; ++i) { t0 = std::min(t0, hs[2 * i]); t1 = std::max(t1, hs[2 * i + 1]); }
        const double us = (t1 - t0) * 0.01;
        printf("rep %d: event %.1f us, in-kernel span %.1f us = %.2f us per layer (%.0f GB/s), timeout flag %d, outputs != 1.0: %d of 2048\n", rep,
               ms * 1e3, us, us / layers, kLayerFloats * 4.0 * layers / us / 1e3, err, bad);
    }
    {
        std::vector<unsigned long long> lg((size_t)layers * 5 * 3);
        CK(hipMemcpy(lg.data(), bf.log, lg.size() * 8, hipMemcpyDeviceToHost));
        const char* nm[5] = {"qkv", "attn", "o_proj", "gate+up", "down"};
        const int l = layers / 2;
        unsigned long long prev = l ? lg[((l - 1) * 5 + 4) * 3 + 2] : lg[0];
        printf("workgroup %d, layer %d (us): fetch = previous publish -> complete input in registers, compute = LDS + barriers + dot products, "
               "publish = drain + stores issued\n", kLogWG, l);
        for (int e = 0; e < 5; ++e) {
            const unsigned long long* p = &lg[(l * 5 + e) * 3];
            printf("  %-8s fetch %.2f compute %.2f publish %.2f\n", nm[e], (p[0] - prev) * 0.01, (p[1] - p[0]) * 0.01, (p[2] - p[1]) * 0.01);
            prev = p[2];
        }
    }
    return 0;
}
