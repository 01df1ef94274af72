// gfx950 kernels for the many-row regime of the T3 path (>= 8 activation rows: batched decode of
// many utterances, SURVEY §8d cfg 3/4, and the prompt prefill): the linears become skinny GEMMs
// y[M,N] = A[M,K] * W[N,K]^T on the matrix cores.
//
//   fp32 weights -> v_mfma_f32_16x16x4_f32  (exact fp32: a k-ordered fmaf chain per output)
//   bf16 weights -> v_mfma_f32_16x16x32_bf16 (bf16 activations, fp32 accumulate)
//
// The weight matrix is still the dominant stream, so the shape stays "GEMV-like": one block owns 16
// output features and ALL rows of its 128-row chunk, its 4 waves split K, operands go straight from
// global memory to the MFMA fragment registers (16-byte loads: one B fragment = 16 weight rows x 64
// contiguous bytes; activations are L2-resident and re-read per block), and the K-partials are
// combined through LDS in a fixed order.  grid = (feature tiles, row chunks).
#include "kernels.hpp"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

// In-kernel clock stamps of the GEMM kernels (scripts/micro/gemm_micro.cpp reads them through GemmArgs::dbg): diagnostic
// builds only (make stamps / -DT3K_STAMPS); the shipped kernels carry neither the test nor the stores.
#ifdef T3K_STAMPS
#define T3K_DBG_DECL(cond, blk) unsigned long long* dbg = (cond) ? a.dbg + 4 * (size_t)(blk) : nullptr
#define T3K_DBG_STAMP(i) do { if (dbg) dbg[i] = wall_clock64(); } while (0)
#else
#define T3K_DBG_DECL(cond, blk) do { } while (0)
#define T3K_DBG_STAMP(i) do { } while (0)
#endif

namespace t3k {

namespace {

constexpr int kHeadDim = 64;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // 16 bytes in flight (a native vector: arrays of HIP's uint4 struct are copied with memcpy and can stay in scratch memory)

__device__ __forceinline__ uint16_t f2bf(float f) {
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

template <typename AT> struct Act;
template <> struct Act<float> {
    __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float round(float v) { return v; }   // the value store() leaves, as a float
};
template <> struct Act<uint16_t> {
    __device__ static __forceinline__ void store(uint16_t* p, float v) { *p = f2bf(v); }
    __device__ static __forceinline__ float round(float v) { return __uint_as_float((uint32_t)f2bf(v) << 16); }
};

// fragment loaders / MFMA steps per weight type: one "k-block" = 64 bytes of a row
struct GemmF32 {
    using AT = float;
    static constexpr int KB = 16;  // k elements per k-block
    using Frag = float4;
    __device__ static __forceinline__ Frag loadA(const void* base, size_t elem) {
        return *reinterpret_cast<const float4*>(static_cast<const float*>(base) + elem);
    }
    __device__ static __forceinline__ Frag loadW(const void* base, size_t elem) { return loadA(base, elem); }
    __device__ static __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
        return c;
    }
};
// bf16 weight storage with fp32 activation rows: the weights are widened in registers and every product is an exact
// fp32 fma, i.e. the arithmetic of the <= 4-row GEMV kernels on bf16 weights.  Used for the prompt prefill of
// sessions that decode on the GEMV path, so prefill and decode see the same numerics (weights rounded, nothing else).
struct GemmBF16W {
    using AT = float;
    static constexpr int KB = 16;
    using Frag = float4;
    __device__ static __forceinline__ Frag loadA(const void* base, size_t elem) { return GemmF32::loadA(base, elem); }
    __device__ static __forceinline__ Frag loadW(const void* base, size_t elem) {
        const uint2 u = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(base) + elem);
        return float4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                      __uint_as_float(u.y & 0xFFFF0000u)};
    }
    __device__ static __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) { return GemmF32::mma(a, b, c); }
};
struct GemmBF16 {
    using AT = uint16_t;
    static constexpr int KB = 32;
    using Frag = bf16x8;
    __device__ static __forceinline__ Frag loadA(const void* base, size_t elem) {
        return *reinterpret_cast<const bf16x8*>(static_cast<const uint16_t*>(base) + elem);
    }
    __device__ static __forceinline__ Frag loadW(const void* base, size_t elem) { return loadA(base, elem); }
    __device__ static __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

// output feature of column n (0..15) of feature tile t
template <int EPI>
__device__ __forceinline__ int tile_feature(int t, int n, int N) {
    if (EPI == EPI_ROPE_KV) {  // 8 rotate_half pairs of one head matrix: {i0..i0+7} and {i0+32..i0+39}
        const int hm = t >> 2, i0 = (t & 3) << 3;
        return hm * kHeadDim + i0 + (n & 7) + ((n >> 3) << 5);
    } else if (EPI == EPI_SWIGLU) {  // 8 gate rows and their 8 up rows
        return (n >> 3) * (N >> 1) + 8 * t + (n & 7);
    }
    return 16 * t + n;
}

template <typename G, int MT, int KS, int EPI>
__global__ __launch_bounds__(KS * 64) void mfma_gemm_kernel(GemmArgs a) {
    using AT = typename G::AT;
    constexpr int KB = G::KB;
    extern __shared__ __attribute__((aligned(16))) float red[];  // [KS][MT*16][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x;
    const int m0 = blockIdx.y * (MT * 16);
    const int Mc = min(a.M - m0, MT * 16);  // rows of this chunk
    const int K = a.K;
    const int kw = K / KS;                  // k range of this wave
    const int kbeg = wave * kw;
    const int per_lane = KB / 4;            // elements per lane per k-block

    int f = tile_feature<EPI>(tile, nl, a.N);
    f = f < a.N ? f : a.N - 1;
    const size_t wbase = (size_t)f * K + kbeg + g * per_lane;
    size_t abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int row = m0 + mt * 16 + nl;
        row = row < a.M ? row : a.M - 1;
        abase[mt] = (size_t)row * K + kbeg + g * per_lane;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // two register stages of UN k-blocks each: the loads of stage s+1 (weights from HBM, activation rows from
    // L2) are in flight while the MFMAs of stage s run -- a k-block is otherwise one exposed L2 round trip
    const int nkb = kw / KB;
    constexpr int UN = (MT >= 8) ? 2 : 4;
    typename G::Frag b0[UN], a0[UN][MT], b1[UN], a1[UN][MT];
    auto load_stage = [&](int j0, typename G::Frag (&bs)[UN], typename G::Frag (&as)[UN][MT]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = (j0 + u < nkb) ? j0 + u : nkb - 1;
            bs[u] = G::loadW(a.W, wbase + (size_t)j * KB);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) as[u][mt] = G::loadA(a.A, abase[mt] + (size_t)j * KB);
        }
    };
    auto compute_stage = [&](int j0, typename G::Frag (&bs)[UN], typename G::Frag (&as)[UN][MT]) {
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (j0 + u < nkb) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = G::mma(as[u][mt], bs[u], acc[mt]);
            }
    };
    // The refills are unconditional (load_stage clamps its k-block to the last one, so the two stages past the end
    // re-request cached lines): a condition around them makes the compiler wait for vmcnt(0) before every
    // compute_stage, i.e. no overlap at all ("exact wait counts").
    load_stage(0, b0, a0);
    for (int j0 = 0; j0 < nkb; j0 += 2 * UN) {
        load_stage(j0 + UN, b1, a1);
        compute_stage(j0, b0, a0);
        load_stage(j0 + 2 * UN, b0, a0);
        compute_stage(j0 + UN, b1, a1);
    }
    // K-partials -> LDS: C layout col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * MT + mt) * 16 + g * 4 + r) * 16 + nl] = acc[mt][r];
    __syncthreads();
    auto total = [&](int m, int n) {
        const int o = m * 16 + n;
        float t = red[o];
#pragma unroll
        for (int w2 = 1; w2 < KS; ++w2) t += red[w2 * MT * 256 + o];
        return t;
    };
    if (EPI == EPI_STORE || EPI == EPI_RESID) {
        for (int it = tid; it < Mc * 16; it += KS * 64) {
            const int m = it >> 4, n = it & 15;
            const int ff = 16 * tile + n;
            if (ff < a.N) {
                const size_t o = (size_t)(m0 + m) * a.ldy + ff;
                float v = total(m, n);
                if (a.bias != nullptr) v = __fadd_rn(v, a.bias[ff]);
                a.y[o] = (EPI == EPI_RESID) ? __fadd_rn(a.resid[o], v) : v;
            }
        }
    } else if (EPI == EPI_GELU) {  // gelu_tanh(c_fc(x) + b)  (R/t3.R:1303-1304)
        for (int it = tid; it < Mc * 16; it += KS * 64) {
            const int m = it >> 4, n = it & 15;
            const int ff = 16 * tile + n;
            if (ff < a.N) {
                float v = total(m, n);
                if (a.bias != nullptr) v = __fadd_rn(v, a.bias[ff]);
                const float inner = 0.7978845608028654f * (v + 0.044715f * v * v * v);
                Act<AT>::store(static_cast<AT*>(a.yact) + (size_t)(m0 + m) * a.N + ff, 0.5f * v * (1.0f + tanhf(inner)));
            }
        }
    } else if (EPI == EPI_SWIGLU) {
        const int F = a.N >> 1;
        for (int it = tid; it < Mc * 8; it += KS * 64) {
            const int m = it >> 3, p = it & 7;
            const float gte = total(m, p), up = total(m, p + 8);
            const float sg = 1.0f / (1.0f + expf(-gte));
            Act<AT>::store(static_cast<AT*>(a.yact) + (size_t)(m0 + m) * F + 8 * tile + p,
                           __fmul_rn(__fmul_rn(gte, sg), up));
        }
    } else {  // EPI_ROPE_KV
        const int H = a.n_heads;
        const int hm = tile >> 2, i0 = (tile & 3) << 3;
        const int kind = hm / H, head = hm % H;
        const int d = H * kHeadDim;
        for (int it = tid; it < Mc * 8; it += KS * 64) {
            const int m = it >> 3, p = it & 7;
            const int mi = m0 + m;
            const int pos = a.posmap[mi];
            if (pos < 0) continue;
            const int i = i0 + p;
            float lo = total(m, p), hi = total(m, p + 8);
            if (a.bias != nullptr) {
                lo = __fadd_rn(lo, a.bias[hm * kHeadDim + i]);
                hi = __fadd_rn(hi, a.bias[hm * kHeadDim + i + 32]);
            }
            if (kind < 2 && a.rope) {
                const float cs = a.rope_cos[(size_t)pos * kHeadDim + i];
                const float sn = a.rope_sin[(size_t)pos * kHeadDim + i];
                const float nlo = __fadd_rn(__fmul_rn(lo, cs), __fmul_rn(-hi, sn));
                const float nhi = __fadd_rn(__fmul_rn(hi, cs), __fmul_rn(lo, sn));
                lo = nlo; hi = nhi;
            }
            if (kind == 0) {
                float* q = a.q + (size_t)mi * d + head * kHeadDim;
                q[i] = lo; q[i + 32] = hi;
            } else {
                const int crow = a.rowmap ? a.rowmap[mi] : mi;
                float* cb = static_cast<float*>(kind == 1 ? a.kcache : a.vcache) + (((size_t)crow * H + head) * a.max_len + pos) * kHeadDim;
                cb[i] = lo; cb[i + 32] = hi;
            }
        }
    }
}

// ---- RMSNorm over rows: out[m] = w * (x[m] * rsqrt(mean(x[m]^2) + eps))  (R/llama.R:65-72) -------------
// With ks > 0 the row is first completed from the K-split partial slabs of the residual GEMM before it
// (x[m] += bias + sum_s part[s][m], s ascending; written back), which saves the separate fold launch.
constexpr int kNormMaxPer = 8;   // d <= 2048
template <typename AT>
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(float* __restrict__ x, const float* __restrict__ w,
                                                           float eps, AT* __restrict__ out, int d, const float* __restrict__ part,
                                                           int ks, const float* __restrict__ pbias, int M) {
    // thread t owns the 4 consecutive elements 4t..4t+3 of each 1024-element slice of the row (d % 4 == 0): the row,
    // its <= 8 partial slabs, the bias and the norm weight all move as 16-byte accesses
    constexpr int kV = kNormMaxPer / 4;   // float4 per thread
    __shared__ float sred[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    float* xr = x + (size_t)m * d;
    float4 v[kV], wv[kV];   // the norm weight is requested with the row: a load after the barrier would add one memory latency
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < kV; ++j) {
        const int i = 4 * tid + j * 1024;
        v[j] = wv[j] = float4{0.f, 0.f, 0.f, 0.f};
        if (i < d) {
            float4 t = *reinterpret_cast<const float4*>(xr + i);
            wv[j] = *reinterpret_cast<const float4*>(w + i);
            if (ks > 0) {
                float4 pp[8];   // all partials requested at once, then added in order
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2)
                    pp[s2] = s2 < ks ? *reinterpret_cast<const float4*>(part + ((size_t)s2 * M + m) * d + i) : float4{0.f, 0.f, 0.f, 0.f};
                float4 p = pp[0];
#pragma unroll
                for (int s2 = 1; s2 < 8; ++s2)
                    if (s2 < ks) { p.x += pp[s2].x; p.y += pp[s2].y; p.z += pp[s2].z; p.w += pp[s2].w; }
                if (pbias != nullptr) {
                    const float4 bb = *reinterpret_cast<const float4*>(pbias + i);
                    p.x = __fadd_rn(p.x, bb.x); p.y = __fadd_rn(p.y, bb.y); p.z = __fadd_rn(p.z, bb.z); p.w = __fadd_rn(p.w, bb.w);
                }
                t.x = __fadd_rn(t.x, p.x); t.y = __fadd_rn(t.y, p.y); t.z = __fadd_rn(t.z, p.z); t.w = __fadd_rn(t.w, p.w);
                *reinterpret_cast<float4*>(xr + i) = t;
            }
            v[j] = t;
            ss = fmaf(t.x, t.x, ss); ss = fmaf(t.y, t.y, ss); ss = fmaf(t.z, t.z, ss); ss = fmaf(t.w, t.w, ss);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if ((tid & 63) == 0) sred[tid >> 6] = ss;
    __syncthreads();
    const float tot = (sred[0] + sred[1]) + (sred[2] + sred[3]);
    const float rs = 1.0f / sqrtf(tot / (float)d + eps);
#pragma unroll
    for (int j = 0; j < kV; ++j) {
        const int i = 4 * tid + j * 1024;
        if (i < d) {
            const float4 ww = wv[j];
            AT* o = out + (size_t)m * d + i;
            const float y0 = __fmul_rn(ww.x, __fmul_rn(v[j].x, rs)), y1 = __fmul_rn(ww.y, __fmul_rn(v[j].y, rs));
            const float y2 = __fmul_rn(ww.z, __fmul_rn(v[j].z, rs)), y3 = __fmul_rn(ww.w, __fmul_rn(v[j].w, rs));
            if constexpr (sizeof(AT) == 2) {
                *reinterpret_cast<uint2*>(o) = uint2{(uint32_t)f2bf(y0) | ((uint32_t)f2bf(y1) << 16),
                                                     (uint32_t)f2bf(y2) | ((uint32_t)f2bf(y3) << 16)};
            } else {
                *reinterpret_cast<float4*>(o) = float4{y0, y1, y2, y3};
            }
        }
    }
}

// ---- LayerNorm over rows (Turbo, R/t3.R:1200-1203): two-pass mean / biased variance; same optional fold ------
template <typename AT>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, float eps, AT* __restrict__ out, int d,
                                                             const float* __restrict__ part, int ks, const float* __restrict__ pbias,
                                                             int M, float* __restrict__ row_mean) {
    // same 16-byte access shape as rmsnorm_rows_kernel: thread t owns elements 4t..4t+3 of each 1024-element slice
    constexpr int kV = kNormMaxPer / 4;
    __shared__ float sred[8];
    const int m = blockIdx.x, tid = threadIdx.x;
    float* xr = x + (size_t)m * d;
    float4 v[kV], wv[kV], bv[kV];   // weight and bias requested with the row (see rmsnorm_rows_kernel)
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < kV; ++j) {
        const int i = 4 * tid + j * 1024;
        v[j] = wv[j] = bv[j] = float4{0.f, 0.f, 0.f, 0.f};
        if (i < d) {
            float4 t = *reinterpret_cast<const float4*>(xr + i);
            wv[j] = *reinterpret_cast<const float4*>(w + i);
            bv[j] = *reinterpret_cast<const float4*>(b + i);
            if (ks > 0) {
                float4 pp[8];   // all partials requested at once, then added in order
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2)
                    pp[s2] = s2 < ks ? *reinterpret_cast<const float4*>(part + ((size_t)s2 * M + m) * d + i) : float4{0.f, 0.f, 0.f, 0.f};
                float4 p = pp[0];
#pragma unroll
                for (int s2 = 1; s2 < 8; ++s2)
                    if (s2 < ks) { p.x += pp[s2].x; p.y += pp[s2].y; p.z += pp[s2].z; p.w += pp[s2].w; }
                if (pbias != nullptr) {
                    const float4 bb = *reinterpret_cast<const float4*>(pbias + i);
                    p.x = __fadd_rn(p.x, bb.x); p.y = __fadd_rn(p.y, bb.y); p.z = __fadd_rn(p.z, bb.z); p.w = __fadd_rn(p.w, bb.w);
                }
                t.x = __fadd_rn(t.x, p.x); t.y = __fadd_rn(t.y, p.y); t.z = __fadd_rn(t.z, p.z); t.w = __fadd_rn(t.w, p.w);
                *reinterpret_cast<float4*>(xr + i) = t;
            }
            v[j] = t;
            sm += (t.x + t.y) + (t.z + t.w);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
    if ((tid & 63) == 0) sred[tid >> 6] = sm;
    __syncthreads();
    const float mean = ((sred[0] + sred[1]) + (sred[2] + sred[3])) / (float)d;
    if (row_mean != nullptr && tid == 0) row_mean[m] = mean;   // for the LayerNorm split of the next residual GEMM (GemmArgs::resid_mean)
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < kV; ++j) {
        const int i = 4 * tid + j * 1024;
        if (i < d) {
            const float c0 = v[j].x - mean, c1 = v[j].y - mean, c2 = v[j].z - mean, c3 = v[j].w - mean;
            ss = fmaf(c0, c0, ss); ss = fmaf(c1, c1, ss); ss = fmaf(c2, c2, ss); ss = fmaf(c3, c3, ss);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if ((tid & 63) == 0) sred[4 + (tid >> 6)] = ss;
    __syncthreads();
    const float var = ((sred[4] + sred[5]) + (sred[6] + sred[7])) / (float)d;
    const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int j = 0; j < kV; ++j) {
        const int i = 4 * tid + j * 1024;
        if (i < d) {
            const float4 ww = wv[j], bb = bv[j];
            AT* o = out + (size_t)m * d + i;
            const float y0 = __fadd_rn(__fmul_rn(__fmul_rn(v[j].x - mean, rs), ww.x), bb.x);
            const float y1 = __fadd_rn(__fmul_rn(__fmul_rn(v[j].y - mean, rs), ww.y), bb.y);
            const float y2 = __fadd_rn(__fmul_rn(__fmul_rn(v[j].z - mean, rs), ww.z), bb.z);
            const float y3 = __fadd_rn(__fmul_rn(__fmul_rn(v[j].w - mean, rs), ww.w), bb.w);
            if constexpr (sizeof(AT) == 2) {
                *reinterpret_cast<uint2*>(o) = uint2{(uint32_t)f2bf(y0) | ((uint32_t)f2bf(y1) << 16),
                                                     (uint32_t)f2bf(y2) | ((uint32_t)f2bf(y3) << 16)};
            } else {
                *reinterpret_cast<float4*>(o) = float4{y0, y1, y2, y3};
            }
        }
    }
}

// ---- many-row attention: one block per (query item, head) ----------------------------------------------------
// (R/llama.R:257-264, src/t3_decode.cpp:151-160.)  With >= 8 rows x 16 heads there are enough (item, head) pairs
// to fill the chip without splitting the keys over blocks, so the softmax finishes inside the block and the output
// rows are written normalised, in the o_proj GEMM's activation type: no partial buffers, no merge launch.
// A wave reads 8 keys per instruction: lane = (key kg = lane >> 3, dims dc*8 .. dc*8+7), i.e. 1 KiB (bf16) or
// 2 KiB (fp32) of contiguous cache per load; the 4 waves interleave key groups.  Every (wave, kg) stream keeps its
// own running max / sum / accumulator (no cross-lane traffic inside the loop) and the 32 streams merge once through
// LDS in a fixed order.
template <typename KVT> struct KvLoad;
template <> struct KvLoad<float> {
    struct Raw { float4 a, b; };
    // the cache is read once per step and is far larger than L2 + Infinity Cache: non-temporal loads
    __device__ static __forceinline__ Raw fetch(const float* p) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v a = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
        const f4v b = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p + 4));
        return Raw{float4{a.x, a.y, a.z, a.w}, float4{b.x, b.y, b.z, b.w}};
    }
    __device__ static __forceinline__ void unpack(const Raw& r, float (&o)[8]) {
        o[0] = r.a.x; o[1] = r.a.y; o[2] = r.a.z; o[3] = r.a.w; o[4] = r.b.x; o[5] = r.b.y; o[6] = r.b.z; o[7] = r.b.w;
    }
};
template <> struct KvLoad<uint16_t> {
    struct Raw { uint4 u; };
    __device__ static __forceinline__ Raw fetch(const uint16_t* p) {
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        const u4v u = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(p));
        return Raw{uint4{u.x, u.y, u.z, u.w}};
    }
    __device__ static __forceinline__ void unpack(const Raw& r, float (&o)[8]) {
        const uint4 u = r.u;
        o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xFFFF0000u);
        o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xFFFF0000u);
        o[4] = __uint_as_float(u.z << 16); o[5] = __uint_as_float(u.z & 0xFFFF0000u);
        o[6] = __uint_as_float(u.w << 16); o[7] = __uint_as_float(u.w & 0xFFFF0000u);
    }
};

template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <typename KVT> __device__ __forceinline__ float attn_exp(float v) {
    if constexpr (sizeof(KVT) == 2) return __expf(v);
    else return expf(v);
}

// PfArgs (kernels.hpp): prefetch block p of a launch whose blocks have nthr threads
__device__ __forceinline__ void prefetch_block(const PfArgs& f, unsigned p, unsigned nthr) {
    unsigned acc = 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const char* base = static_cast<const char*>(f.ptr[r]);
        if (f.unit[r] == 0) {
            const unsigned lines = f.bytes[r] >> 7;
            for (unsigned i = p * nthr + threadIdx.x; i < lines; i += (unsigned)f.n_blocks * nthr) acc ^= *reinterpret_cast<const unsigned*>(base + ((size_t)i << 7));
        } else {
            const unsigned xcd = p & 7u, pi = p >> 3, npx = (unsigned)f.n_blocks >> 3;   // this XCD's prefetch blocks
            const unsigned lpu = f.unit[r] >> 7, units = f.bytes[r] / f.unit[r], upx = units >> 3;
            const unsigned share = upx * lpu;
            for (unsigned i = pi * nthr + threadIdx.x; i < share; i += npx * nthr) {
                const unsigned k = i / lpu, u = f.remap[r] ? xcd * upx + k : 8u * k + xcd;
                acc ^= *reinterpret_cast<const unsigned*>(base + (size_t)u * f.unit[r] + ((size_t)(i - k * lpu) << 7));
            }
        }
    }
    f.sink[(size_t)p * nthr + threadIdx.x] = acc;
}

template <typename KVT, typename OT, int NW, bool PF = false>   // NW waves per block; PF: grid rows y >= 1 are prefetch blocks (PfArgs)
__global__ __launch_bounds__(64 * NW, sizeof(KVT) == 2 ? 4 : 2) void attn_rows_kernel(AttnRowsArgs a) {   // bf16 cache: 4 waves per SIMD (128 VGPRs)
    constexpr int NS = NW * 8;   // key streams of the block: (wave, key group)

    __shared__ float sm_m[NS], sm_l[NS];
    __shared__ __attribute__((aligned(16))) float sm_acc[NS][kHeadDim + 4];
    // flat grid: block = qi * H + h (the heads of an item are adjacent, as they are in the cache); a 2-D grid with the
    // items in y is capped at 65535 items, which a prefill of a few hundred prompts exceeds
    if constexpr (PF) {
        if (blockIdx.y != 0) {   // (block-uniform)
            prefetch_block(a.pf, (blockIdx.y - 1) * gridDim.x + blockIdx.x, 64 * NW);
            return;
        }
    }
#ifdef T3K_DIAG_ATTN_SPLIT   // measurement aid (make diag; results wrong): S blocks per (item, head), each over 1/S of the keys, all writing the same row
    const unsigned dsplit = a.diag_split > 1 ? (unsigned)a.diag_split : 1u;
    const unsigned bx = blockIdx.x / dsplit, dsp = blockIdx.x % dsplit;
#else
    const unsigned bx = blockIdx.x;
#endif
    const int h = (int)(bx % (unsigned)a.n_heads), qi = (int)(bx / (unsigned)a.n_heads);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = lane >> 3, dc = lane & 7;
    const int pos = a.posmap[qi];
    const int d = a.n_heads * kHeadDim;
    OT* orow = static_cast<OT*>(a.out) + (size_t)qi * d + h * kHeadDim;
    if (pos < 0) {   // inactive item: defined (zero) activation row, as the merge kernel wrote
        if (tid < kHeadDim) Act<OT>::store(orow + tid, 0.f);
        if (a.dot_out != nullptr && tid == 0) a.dot_out[(size_t)qi * a.n_heads + h] = 0.f;
        return;
    }
    const int crow = a.rowmap ? a.rowmap[qi] : qi;
#ifdef T3K_DIAG_ATTN_SPLIT
    int L = pos + 1;
    size_t base = ((size_t)crow * a.n_heads + h) * a.max_len * kHeadDim;
    {
        const int Lq = (L + (int)dsplit - 1) / (int)dsplit;
        int k0 = (int)dsp * Lq, Ls = L - k0 < Lq ? L - k0 : Lq;
        if (Ls <= 0) { k0 = 0; Ls = 1; }
        base += (size_t)k0 * kHeadDim;
        L = Ls;
    }
#else
    const int L = pos + 1;
    const size_t base = ((size_t)crow * a.n_heads + h) * a.max_len * kHeadDim;
#endif
    const KVT* kp = static_cast<const KVT*>(a.kcache) + base + dc * 8;
    const KVT* vp = static_cast<const KVT*>(a.vcache) + base + dc * 8;
    float qv[8];
    {
        const float* qp = a.q + (size_t)qi * d + h * kHeadDim + dc * 8;
        const float4 q0 = *reinterpret_cast<const float4*>(qp), q1 = *reinterpret_cast<const float4*>(qp + 4);
        // scores are q.k / sqrt(64): the scale is applied to the dot product (R/llama.R:259)
        qv[0] = q0.x; qv[1] = q0.y; qv[2] = q0.z; qv[3] = q0.w; qv[4] = q1.x; qv[5] = q1.y; qv[6] = q1.z; qv[7] = q1.w;
    }
    // (requested here, used after the merge: at the end it would be one more memory latency in every block's tail)
    const float dv = a.dot_vec != nullptr ? a.dot_vec[h * kHeadDim + lane] : 0.f;
    float m_run = -INFINITY, l_run = 0.f, acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    // keys of this stream: wave*8 + kg, then every 32nd; a trip takes two of them (k and k + 32: four cache loads per
    // lane).  The loads run kAhead trips ahead of the arithmetic in a register ring (the cache comes from HBM:
    // a trip that waits for its own loads exposes one full memory latency per 64 keys of the block).
    constexpr int kAhead = 3;
    using Raw = typename KvLoad<KVT>::Raw;
    Raw rka[kAhead], rkb[kAhead], rva[kAhead], rvb[kAhead];
    const int kfirst = wave * 8 + kg;
    const int trips = (L + 2 * NS - 1) / (2 * NS);   // block-uniform: streams whose last trip has no key left get masked scores
    auto issue = [&](int t, auto SC) {
        constexpr int slot = decltype(SC)::value;
        const int ka = kfirst + t * (2 * NS), kb = ka + NS;
        const size_t oa = (size_t)(ka < L ? ka : 0) * kHeadDim, ob = (size_t)(kb < L ? kb : 0) * kHeadDim;
        rka[slot] = KvLoad<KVT>::fetch(kp + oa);
        rkb[slot] = KvLoad<KVT>::fetch(kp + ob);
        rva[slot] = KvLoad<KVT>::fetch(vp + oa);
        rvb[slot] = KvLoad<KVT>::fetch(vp + ob);
    };
    // arithmetic of trip t on ring slot SC; no branch (an empty stream keeps m = -inf, l = 0, acc = 0)
    auto consume = [&](int t, auto SC) {
        constexpr int slot = decltype(SC)::value;
        const int ka = kfirst + t * (2 * NS), kb = ka + NS;
        const bool oka = ka < L, okb = kb < L;
        float kva[8], kvb[8], vva[8], vvb[8];
        KvLoad<KVT>::unpack(rka[slot], kva);
        KvLoad<KVT>::unpack(rkb[slot], kvb);
        KvLoad<KVT>::unpack(rva[slot], vva);
        KvLoad<KVT>::unpack(rvb[slot], vvb);
        // packed fp32 multiply-adds (v_pk_fma_f32: two lanes of arithmetic per instruction): element pairs (e, e + 1)
        // of both keys; the per-key dot product is the sum of an even and an odd chain
        f32x2 da = {0.f, 0.f}, db = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2 q2 = {qv[e], qv[e + 1]};
            da = __builtin_elementwise_fma(q2, f32x2{kva[e], kva[e + 1]}, da);
            db = __builtin_elementwise_fma(q2, f32x2{kvb[e], kvb[e + 1]}, db);
        }
        float sa = da.x + da.y, sb = db.x + db.y;
        // sum over the 8 lanes of the key (lanes kg*8 .. kg*8+7): lane ^ 1, lane ^ 2 inside the quad, then the other
        // quad of the 8 -- as DPP moves (VALU, no LDS round trip; __shfl_xor compiles to ds_bpermute), same sums
        sa += dpp_f32<0xB1>(sa); sb += dpp_f32<0xB1>(sb);     // quad_perm [1,0,3,2]
        sa += dpp_f32<0x4E>(sa); sb += dpp_f32<0x4E>(sb);     // quad_perm [2,3,0,1]
        sa += dpp_f32<0x141>(sa); sb += dpp_f32<0x141>(sb);   // row_half_mirror: lane 7 - i of the 8, i.e. the other quad's sum
        sa = oka ? sa * 0.125f : -INFINITY;
        sb = okb ? sb * 0.125f : -INFINITY;
        const float mn = fmaxf(m_run, fmaxf(sa, sb));
        const float mref = mn > -INFINITY ? mn : 0.f;   // nothing seen yet: all three factors become exp(-inf) = 0
        // bf16 cache (values carry 8 significant bits): the hardware exponential; fp32 cache: libm's
        const float sc = attn_exp<KVT>(m_run - mref), pa = attn_exp<KVT>(sa - mref), pb = attn_exp<KVT>(sb - mref);
        l_run = fmaf(l_run, sc, pa + pb);
        const f32x2 sc2 = {sc, sc}, pa2 = {pa, pa}, pb2 = {pb, pb};
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            f32x2 a2 = {acc[e], acc[e + 1]};
            a2 = __builtin_elementwise_fma(pb2, f32x2{vvb[e], vvb[e + 1]}, __builtin_elementwise_fma(pa2, f32x2{vva[e], vva[e + 1]}, a2 * sc2));
            acc[e] = a2.x; acc[e + 1] = a2.y;
        }
        m_run = mn;
    };
    using std::integral_constant;
    static_assert(kAhead == 3, "the ring below is written out for three slots");
    const integral_constant<int, 0> S0{};
    const integral_constant<int, 1> S1{};
    const integral_constant<int, 2> S2{};
    // Consume a slot, refill it -- with NO branch around a load and a single loop exit.  A conditional refill makes
    // the compiler wait for vmcnt(0) in every trip (it must cover the path on which nothing was issued), and a loop
    // with several exits makes it copy the ring at the back edge (same wait): either way one full memory latency
    // per trip, 22 us per block at L = 264.  So the trip count is padded to a multiple of kAhead, trips past the
    // end are masked in consume() and their loads (and the last kAhead refills) are clamped to key 0 of the head:
    // requests for one already cached line, no HBM traffic.
    // (sched_barrier: the scheduler otherwise reorders the prologue's loads, slot 0 last, and the loop header then has
    // to wait for everything.)
#define T3K_SB() __builtin_amdgcn_sched_barrier(0)
    issue(0, S0); T3K_SB(); issue(1, S1); T3K_SB(); issue(2, S2); T3K_SB();
    for (int t = 0; t < trips; t += kAhead) {
        consume(t, S0); T3K_SB(); issue(t + kAhead, S0); T3K_SB();
        consume(t + 1, S1); T3K_SB(); issue(t + kAhead + 1, S1); T3K_SB();
        consume(t + 2, S2); T3K_SB(); issue(t + kAhead + 2, S2); T3K_SB();
    }
#undef T3K_SB
    // (NW = 1: merging the 8 streams of the one wave in registers -- a shuffle butterfly over lane bits 3..5 instead of LDS +
    // barrier -- measured slower: 1.828 vs 1.802 ms per step at 64 utterances; every lane then pays the exponentials.)
    // merge the NS streams (NW waves x 8 key groups): stream id = wave*8 + kg
    const int sid = wave * 8 + kg;
    if (dc == 0) { sm_m[sid] = m_run; sm_l[sid] = l_run; }
#pragma unroll
    for (int e = 0; e < 8; ++e) sm_acc[sid][dc * 8 + e] = acc[e];
    __syncthreads();
    if (tid < kHeadDim) {
        float M = -INFINITY;
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) M = fmaxf(M, sm_m[s2]);
        float den = 0.f, o = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            const float ms = sm_m[s2];
            const float ws = ms > -INFINITY ? expf(ms - M) : 0.f;
            den = fmaf(ws, sm_l[s2], den);
            o = fmaf(ws, sm_acc[s2][tid], o);
        }
        const float ov = o / den;
        Act<OT>::store(orow + tid, ov);
        if (a.dot_out != nullptr) {   // (block-uniform) the stored values of this head . dot_vec, for the LayerNorm split of o_proj
            float pv = Act<OT>::round(ov) * dv;
#pragma unroll
            for (int s2 = 32; s2 > 0; s2 >>= 1) pv += __shfl_xor(pv, s2, 64);
            if (tid == 0) a.dot_out[(size_t)qi * a.n_heads + h] = pv;
        }
    }
}

// ---- x[m] = embeds of item m (prefill: flat list of (row, position) items) --------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* const* __restrict__ src_rows, float* __restrict__ x, int d) {
    const int m = blockIdx.x;
    const float* s = src_rows[m];
    for (int i = threadIdx.x; i < d; i += 256) x[(size_t)m * d + i] = s[i];
}


// ---- bf16 many-row GEMM, second generation ------------------------------------------------------------------
// y[M,N] = A[M,K] * W[N,K]^T with bf16 operands, fp32 accumulate, for up to 128 rows per block.
//
// Block = 8 waves = 128 rows x 32 output features x one K range.  Wave w owns rows 16w..16w+15 and both 16-feature
// sub-tiles, so it holds 2 accumulator tiles.  A 128-k step's tile (128 activation rows + 32 weight rows, 256 B
// each, whole lines: 16 consecutive lanes fetch one row's run) goes global -> registers -> LDS once per block and
// is shared by the 8 waves; register rings keep kDA steps of activations (from L2) and kDW steps of weights (from
// HBM: for K <= 1024 the block's whole weight slice) in flight.  Lane (n, g) reads k = g*32..g*32+31 of its row
// from LDS per step: MFMA j of the step uses k = g*32 + 8j..8j+7 from both operands, a fixed permutation of the
// dot product's terms.  The two sub-tiles hold PARTNER features of the fused epilogues (gate / up, rotate_half
// i / i+32), so SwiGLU and RoPE finish in registers.  Measured at 128 rows (profiles/r01_micro_gemm_m128.txt):
// first tile in LDS after 1.4-3.8 us, ~0.65 us per step (LDS-bandwidth bound: 40 KB written + 96 KB read).
// Residual GEMMs with few feature tiles (N = 1024) split K over blockIdx.z into fp32 partial slabs that
// fold_partials_kernel sums in a fixed order (deterministic, no atomics).
template <int EPI>
__device__ __forceinline__ int tile32_feature(int t, int r32, int N) {
    const int s = r32 >> 4, n = r32 & 15;
    if (EPI == EPI_ROPE_KV) return (t >> 1) * kHeadDim + ((t & 1) << 4) + n + (s << 5);   // i = 16u + n, partner i + 32
    if (EPI == EPI_SWIGLU) return s * (N >> 1) + 16 * t + n;                                // gate row, partner up row
    return 32 * t + r32;
}

// Activation arithmetic by output type.  fp32 rows keep libm's expf / tanhf and IEEE division (the 1e-3 bar against the
// fp32 oracle and greedy-token identity).  bf16 rows are rounded to 8 significant bits on store, so the hardware
// exponential and reciprocal (about 1 ulp of fp32 each) are exact enough, and at 1024 rows the difference is
// 3 us per gate_up launch chip-wide (about 30 VALU instructions per output against 8).
template <typename T> __device__ __forceinline__ float act_sigmoid(float v) {
    if constexpr (sizeof(T) == 2) return __frcp_rn(1.0f + __expf(-v));
    else return 1.0f / (1.0f + expf(-v));
}
template <typename T> __device__ __forceinline__ float act_tanh(float v) {
    if constexpr (sizeof(T) == 2) return 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * v));   // exp overflow -> rcp(inf) = 0 -> 1
    else return tanhf(v);
}

// Everything the epilogues READ from global memory is fetched before their first store: the stores may alias the
// tables as far as the compiler knows, so a load inside the store loop waits for the previous store and the loop
// runs at one memory latency per row (qkv at 1024 rows: 6.5 us of epilogue for 16 rows per lane).
struct EpiRow { int pos, crow; };          // cache position and cache row of an activation row (EPI_ROPE_KV)
struct EpiCol { float b0, b1; };           // bias of the two partner features of a column (0 without bias)

template <int EPI>
__device__ __forceinline__ EpiRow rows_pos(const GemmArgs& a, int mi) {
    EpiRow e{-1, mi};
    if (EPI == EPI_ROPE_KV && mi < a.M) {
        e.pos = a.posmap[mi];
        if (a.rowmap) e.crow = a.rowmap[mi];
    }
    return e;
}
template <int EPI>
__device__ __forceinline__ EpiCol cols_bias(const GemmArgs& a, int tile, int nl) {
    EpiCol c{0.f, 0.f};
    if (a.bias != nullptr) {
        const int f0 = tile32_feature<EPI>(tile, nl, a.N), f1 = tile32_feature<EPI>(tile, 16 + nl, a.N);
        if (f0 < a.N) c.b0 = a.bias[f0];
        if (f1 < a.N) c.b1 = a.bias[f1];
    }
    return c;
}
// RoPE factors of (row position, column pair) -- {1, 0} where nothing rotates (V tiles, no RoPE, inactive row)
template <int EPI>
__device__ __forceinline__ float2 rope_cs(const GemmArgs& a, int tile, int nl, int pos) {
    float2 r{1.f, 0.f};
    if (EPI == EPI_ROPE_KV && a.rope && pos >= 0 && (tile >> 1) / a.n_heads < 2) {
        const int i = ((tile & 1) << 4) + nl;
        r.x = a.rope_cos[(size_t)pos * kHeadDim + i];
        r.y = a.rope_sin[(size_t)pos * kHeadDim + i];
    }
    return r;
}

// Epilogue of one output row mi for the PAIR of partner features that column nl of a 32-feature tile holds
// (v0: sub-tile 0, v1: sub-tile 1; tile32_feature gives their feature ids).  Shared by gemm_rows_kernel and
// gemm_wide_kernel.  Stores only (plus the residual read of the unsplit EPI_RESID form).
template <int EPI, typename T>
__device__ __forceinline__ void rows_epilogue(const GemmArgs& a, int tile, int nl, int mi, float v0, float v1, int kz, bool ksplit,
                                              EpiRow er, EpiCol ec, float2 cs) {
    if (mi >= a.M) return;
    const int f0 = tile32_feature<EPI>(tile, nl, a.N), f1 = tile32_feature<EPI>(tile, 16 + nl, a.N);
    if (EPI == EPI_STORE || EPI == EPI_RESID) {
        if (ksplit) {   // K split over blocks: fp32 partial slab of this K range (the fold adds the bias)
            float* pr = a.part + ((size_t)kz * a.M + mi) * a.N;
            if (f0 < a.N) pr[f0] = v0;
            if (f1 < a.N) pr[f1] = v1;
        } else {
            const size_t o = (size_t)mi * a.ldy;
            if (f0 < a.N) {
                if (a.bias != nullptr) v0 = __fadd_rn(v0, ec.b0);
                a.y[o + f0] = (EPI == EPI_RESID) ? __fadd_rn(a.resid[o + f0], v0) : v0;
            }
            if (f1 < a.N) {
                if (a.bias != nullptr) v1 = __fadd_rn(v1, ec.b1);
                a.y[o + f1] = (EPI == EPI_RESID) ? __fadd_rn(a.resid[o + f1], v1) : v1;
            }
        }
    } else if (EPI == EPI_GELU) {   // gelu_tanh(c_fc(x) + b)  (R/t3.R:1303-1304)
        T* yo = static_cast<T*>(a.yact) + (size_t)mi * a.N;
        if (a.bias != nullptr) { v0 = __fadd_rn(v0, ec.b0); v1 = __fadd_rn(v1, ec.b1); }
        const float i0 = 0.7978845608028654f * (v0 + 0.044715f * v0 * v0 * v0);
        const float i1 = 0.7978845608028654f * (v1 + 0.044715f * v1 * v1 * v1);
        if (f0 < a.N) Act<T>::store(yo + f0, 0.5f * v0 * (1.0f + act_tanh<T>(i0)));
        if (f1 < a.N) Act<T>::store(yo + f1, 0.5f * v1 * (1.0f + act_tanh<T>(i1)));
    } else if (EPI == EPI_SWIGLU) {   // silu(gate) * up  (R/llama.R:302, 307)
        const float sg = act_sigmoid<T>(v0);
        Act<T>::store(static_cast<T*>(a.yact) + (size_t)mi * (a.N >> 1) + 16 * tile + nl, __fmul_rn(__fmul_rn(v0, sg), v1));
    } else {  // EPI_ROPE_KV
        const int pos = er.pos;
        if (pos < 0) return;
        const int H = a.n_heads, hm = tile >> 1, i = ((tile & 1) << 4) + nl;
        const int kind = hm / H, head = hm % H;
        float lo = v0, hi = v1;
        if (a.bias != nullptr) {
            lo = __fadd_rn(lo, ec.b0);
            hi = __fadd_rn(hi, ec.b1);
        }
        if (kind < 2 && a.rope) {
            const float nlo = __fadd_rn(__fmul_rn(lo, cs.x), __fmul_rn(-hi, cs.y));
            const float nhi = __fadd_rn(__fmul_rn(hi, cs.x), __fmul_rn(lo, cs.y));
            lo = nlo; hi = nhi;
        }
        if (kind == 0) {
            float* q = a.q + (size_t)mi * (H * kHeadDim) + head * kHeadDim;
            q[i] = lo; q[i + 32] = hi;
        } else {
            const size_t co = (((size_t)er.crow * H + head) * a.max_len + pos) * kHeadDim;
            if (a.kv_bf16) {
                uint16_t* cb = static_cast<uint16_t*>(kind == 1 ? a.kcache : a.vcache) + co;
                cb[i] = f2bf(lo); cb[i + 32] = f2bf(hi);
            } else {
                float* cb = static_cast<float*>(kind == 1 ? a.kcache : a.vcache) + co;
                cb[i] = lo; cb[i + 32] = hi;
            }
        }
    }
}

template <int EPI, typename T, int SC, int AP = 4>
__global__ __launch_bounds__(512) void gemm_rows_kernel(GemmArgs a) {
    constexpr int E = 16 / (int)sizeof(T);   // elements per 16-byte piece: 8 (bf16) or 4 (fp32)
    constexpr int KC = 16 * E, LD = KC + E;  // 256-byte step rows, row stride 272 B: the 16 rows of a fragment read fall in distinct banks
    // LDS rows per step: 32 * AP activation rows + 32 weight rows.
    // AP: staging passes of 32 activation rows (4 = 128 rows; 1 = launches of <= 32 rows, which then write 12 KB to LDS
    // per step instead of 40 and keep a quarter of the activation ring).
    constexpr int NP = AP;
    constexpr int kARows = 32 * AP, kRows = kARows + 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char tl_raw[];
    T* tl = reinterpret_cast<T*>(tl_raw);   // [2][kRows][LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x, m0 = blockIdx.y * (32 * AP), kz = blockIdx.z;
    const int K = a.K;
    // steps of this block.  SC > 0: known at compile time (the decode shapes: 8 or 4 steps), so that every `s < S`
    // below folds away -- with run-time conditions around the ring refills the compiler waits for vmcnt(0) in every
    // step (it must cover the path on which nothing was issued) and the register rings hide nothing.
    const int S = SC > 0 ? SC : (K / KC) / (int)gridDim.z;
    T3K_DBG_DECL(a.dbg != nullptr && tid == 0, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
    T3K_DBG_STAMP(0);
    const int kbeg = kz * S * KC;
    const T* Wp = static_cast<const T*>(a.W);
    const T* Ap = static_cast<const T*>(a.A);

    // staging: thread t moves the 16 B piece t & 15 of tile rows (t >> 4) + 32 p, so 16 consecutive lanes fetch one
    // whole 256-byte run: p = 0..AP-1 activation rows, src[4] the weight rows.  (Reading the activation fragments
    // straight from global memory and staging only the weights was measured 7 % slower end to end: DESIGN.md 3.2.)
    const int srow = tid >> 4, spiece = tid & 15;
    const T* src[5];
    const bool live = m0 + wave * 16 < a.M && wave * 16 < 32 * AP;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        int r = m0 + p * 32 + srow;
        r = r < a.M ? r : a.M - 1;
        src[p] = Ap + (size_t)r * K + kbeg + spiece * E;
    }
    {
        int sf = tile32_feature<EPI>(tile, srow, a.N);
        sf = sf < a.N ? sf : a.N - 1;
        src[4] = Wp + (size_t)sf * K + kbeg + spiece * E;
    }
    const int soff = srow * LD + spiece * E;

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // register rings: the weights come from HBM (2-3 us under load) and are requested kDW steps ahead, which for the
    // decode shapes (K <= 1024 per block) is the block's whole weight slice at kernel start (default cache policy:
    // non-temporal weight loads measured 1-2 % slower here, gate_up 21.1 -> 21.7 us, down 11.3 -> 12.3); the activation rows come
    // from L2 and run kDA steps ahead.
    constexpr int kDW = 8, kDA = 3;
    uint4 rw[kDW], ra[kDA][NP];
    auto load_a = [&](int step, int p) -> uint4 {   // piece p of the thread's activation data of a step
        return *reinterpret_cast<const uint4*>(src[p] + (size_t)step * KC);
    };
#pragma unroll
    for (int q = 0; q < kDW; ++q) rw[q] = (q < S) ? *reinterpret_cast<const uint4*>(src[4] + (size_t)q * KC) : uint4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < kDA; ++q)
#pragma unroll
        for (int p = 0; p < NP; ++p) ra[q][p] = q < S ? load_a(q, p) : uint4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<uint4*>(&tl[soff + p * 32 * LD]) = ra[0][p];
    if (kDA < S) {
#pragma unroll
        for (int p = 0; p < NP; ++p) ra[0][p] = load_a(kDA, p);
    }
    *reinterpret_cast<uint4*>(&tl[soff + kARows * LD]) = rw[0];
    if (kDW < S) rw[0] = *reinterpret_cast<const uint4*>(src[4] + (size_t)kDW * KC);
    __syncthreads();
    T3K_DBG_STAMP(1);
    // 24 = lcm(kDW, kDA): ring slots are compile-time indices inside the unrolled body
    for (int s0 = 0; s0 < S; s0 += 24) {
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const int s = s0 + u;
            if (s < S) {   // block-uniform
                // ring slot (s+1) % kDA holds the staging pieces of step s+1
                const int ua = (u + 1) % kDA, uw = (u + 1) % kDW;
                if (live) {
                    // lane (nl, g) reads piece j of the 64 bytes g of its row for MFMA j (both operands from LDS): a fixed
                    // permutation of the dot product's terms over the MFMA k slots, the same for both operands
                    const T* tb = tl + (s & 1) * (kRows * LD);
                    const T* ab = tb + (wave * 16 + nl) * LD + g * (4 * E);
                    const T* wb = tb + (kARows + nl) * LD + g * (4 * E);
                    constexpr int jstride = E;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint4 araw = *reinterpret_cast<const uint4*>(ab + j * E);
                        if constexpr (sizeof(T) == 2) {
                            const bf16x8 av = __builtin_bit_cast(bf16x8, araw);
                            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(wb + j * jstride);
                            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(wb + 16 * LD + j * jstride);
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b0, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b1, acc1, 0, 0, 0);
                        } else {
                            // fp32 operands: exact fp32 fma chains (v_mfma_f32_16x16x4_f32, one k per lane group)
                            const float4 av = __builtin_bit_cast(float4, araw);
                            const float4 b0 = *reinterpret_cast<const float4*>(wb + j * jstride);
                            const float4 b1 = *reinterpret_cast<const float4*>(wb + 16 * LD + j * jstride);
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0.x, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b1.x, acc1, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b0.y, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1.y, acc1, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b0.z, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b1.z, acc1, 0, 0, 0);
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b0.w, acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b1.w, acc1, 0, 0, 0);
                        }
                    }
                }
                // after the step's MFMAs: tile of step s+1 -> the LDS buffer last read in step s-1, then refill the ring
                // slots.  (Order within the step is immaterial: staging before or after the MFMAs measures the same.
                // gemm_micro, gate_up at 128 rows, 8 steps: barriers + global loads alone 2.75 us; + staging
                // writes 2.77; + fragment reads and MFMAs 4.30; all three 5.43 -- the LDS serves 136 KB per step and the
                // reads and writes contend for it.)
                if (s + 1 < S) {
                    T* dst = tl + ((s + 1) & 1) * (kRows * LD) + soff;
#pragma unroll
                    for (int p = 0; p < NP; ++p) *reinterpret_cast<uint4*>(dst + p * 32 * LD) = ra[ua][p];
                    *reinterpret_cast<uint4*>(dst + kARows * LD) = rw[uw];
                }
                if (s + 1 + kDA < S) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) ra[ua][p] = load_a(s + 1 + kDA, p);
                }
                if (s + 1 + kDW < S) rw[uw] = *reinterpret_cast<const uint4*>(src[4] + (size_t)(s + 1 + kDW) * KC);
                __syncthreads();
            }
        }
    }
    T3K_DBG_STAMP(2);
    if (!live) return;
    // accumulator layout: row = g*4 + r of the wave's 16 rows, column nl of the sub-tile
    {
        const EpiCol ec = cols_bias<EPI>(a, tile, nl);
        EpiRow er[4];
        float2 cs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) er[r] = rows_pos<EPI>(a, m0 + wave * 16 + g * 4 + r);
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[r] = rope_cs<EPI>(a, tile, nl, er[r].pos);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            rows_epilogue<EPI, T>(a, tile, nl, m0 + wave * 16 + g * 4 + r, acc0[r], acc1[r], kz, gridDim.z > 1, er[r], ec, cs[r]);
    }
    T3K_DBG_STAMP(3);
}

// ---- bf16 GEMM for more than 128 rows (512 utterances per GPU, the prompt prefill of a batch) --------------------
// With many rows the linears are real GEMMs (1 TFLOP per decode step at 1024 rows) and gemm_rows_kernel's 16 x 32
// wave tile is LDS-bound at ~8 % of the MFMA peak (136 KB of LDS traffic per 0.5 M multiply-adds).  Here a block of
// 8 waves owns 64*MT rows x 32*NT features: waves sit 4 (rows) x 2 (features), a wave holds MT x NT accumulator
// tiles of 16 x 16 (MT = NT = 4: 64 rows x 64 features, 64 accumulator registers), so per 64-k step the block
// writes (64 MT + 32 NT) x 128 B to LDS and its waves read 8 x (16 MT + 16 NT) x 128 B: 176 KB per 2.1 M
// multiply-adds at MT = NT = 4, a third of the traffic per product.  A step's rows are 128 B (8 lanes fetch one
// run), row stride 144 B (16-row fragment reads hit 16 distinct bank groups); register rings keep kWA steps of
// activations (L2) and kWW steps of weights (HBM / MALL) in flight; two LDS buffers, one barrier per step.
// Feature tiles keep the partner layout of tile32_feature, so the epilogues are those of gemm_rows_kernel.
// Blocks that share a weight tile (the row chunks) are placed on the same XCD: linear block id -> XCD id % 8, and
// within an XCD the row chunk runs fastest.  K may be split over blockIdx.y for the residual GEMMs (fp32 partial
// slabs, folded in a fixed order by the next row norm or fold_partials_kernel).
template <int EPI, int MT, int NT>
__global__ __launch_bounds__(512) void gemm_wide_kernel(GemmArgs a, int tiles32, int chunks, int nblk) {
    using T = uint16_t;
    constexpr int BM = 64 * MT, BN = 32 * NT, LD = 72;   // LDS row stride in elements: 128 B of data + 16 B
    constexpr int kRows = BM + BN;
    constexpr int PW = BN / 64;                          // weight pieces per thread and step (A pieces: MT)
    static_assert(NT == 2 || NT == 4, "feature groups of 64 or 128");
    extern __shared__ __attribute__((aligned(16))) unsigned char tlw_raw[];
    T* tl = reinterpret_cast<T*>(tlw_raw);               // [2][kRows][LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware placement: consecutive linear ids go round the 8 XCDs; slot = position inside the XCD's sequence
    const int lin = blockIdx.x, xcd = lin & 7, slot = lin >> 3;
    const int blk = (slot / chunks) * 8 + xcd, chunk = slot % chunks;
    if (blk >= nblk) return;                             // block-uniform
    const int m0 = chunk * BM, kz = blockIdx.y;
    const int K = a.K;
    const int S = (K / 64) / (int)gridDim.y;             // 64-k steps of this block
    T3K_DBG_DECL(a.dbg != nullptr && tid == 0 && blockIdx.x + gridDim.x * blockIdx.y < 4096, blockIdx.x + gridDim.x * blockIdx.y);
    T3K_DBG_STAMP(0);
    const int kbeg = kz * S * 64;
    const T* Wp = static_cast<const T*>(a.W);
    const T* Ap = static_cast<const T*>(a.A);

    const int srow = tid >> 3, spiece = tid & 7;         // staging: 8 lanes per 128-byte run, rows srow + 64 p
    const T* srcA[MT];
    const T* srcW[PW];
#pragma unroll
    for (int p = 0; p < MT; ++p) {
        int r = m0 + p * 64 + srow;
        r = r < a.M ? r : a.M - 1;
        srcA[p] = Ap + (size_t)r * K + kbeg + spiece * 8;
    }
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        const int wr = p * 64 + srow;                    // weight row of the block: 32-feature tile wr / 32, row wr % 32
        int t = blk * NT + (wr >> 5);
        t = t < tiles32 ? t : tiles32 - 1;
        int f = tile32_feature<EPI>(t, wr & 31, a.N);
        f = f < a.N ? f : a.N - 1;
        srcW[p] = Wp + (size_t)f * K + kbeg + spiece * 8;
    }
    const int soff = srow * LD + spiece * 8;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int kWA = 2, kWW = 4;                      // ring depths (steps); the unrolled body covers lcm = 4 steps
    uint4 ra[kWA][MT], rw[kWW][PW];
#pragma unroll
    for (int q = 0; q < kWW; ++q)
#pragma unroll
        for (int p = 0; p < PW; ++p) rw[q][p] = q < S ? *reinterpret_cast<const uint4*>(srcW[p] + (size_t)q * 64) : uint4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < kWA; ++q)
#pragma unroll
        for (int p = 0; p < MT; ++p) ra[q][p] = q < S ? *reinterpret_cast<const uint4*>(srcA[p] + (size_t)q * 64) : uint4{0u, 0u, 0u, 0u};
    // step 0 -> LDS buffer 0, refill ring slot 0
#pragma unroll
    for (int p = 0; p < MT; ++p) *reinterpret_cast<uint4*>(&tl[soff + p * 64 * LD]) = ra[0][p];
#pragma unroll
    for (int p = 0; p < PW; ++p) *reinterpret_cast<uint4*>(&tl[soff + (BM + p * 64) * LD]) = rw[0][p];
    if (kWA < S) {
#pragma unroll
        for (int p = 0; p < MT; ++p) ra[0][p] = *reinterpret_cast<const uint4*>(srcA[p] + (size_t)kWA * 64);
    }
    if (kWW < S) {
#pragma unroll
        for (int p = 0; p < PW; ++p) rw[0][p] = *reinterpret_cast<const uint4*>(srcW[p] + (size_t)kWW * 64);
    }
    __syncthreads();
    T3K_DBG_STAMP(1);
    // One step.  U = s % 4 comes as a type so that the ring slots are literal indices (the rings stay in registers).
    // CK = false is the steady state: no branches at all -- the tile of step s + 1 is always staged (after the last
    // step it lands in a buffer nobody reads) and the refill loads clamp their step to S - 1 -- so that the compiler
    // counts the outstanding loads exactly (with the conditional form every step waited for vmcnt(0): 2.5 us per step
    // instead of 0.6).  CK = true keeps the conditions and serves the S % 4 tail.
    auto step = [&](auto UC, auto CKC, int s) {
        constexpr int u = decltype(UC)::value;
        constexpr bool CK = decltype(CKC)::value;
        if (CK && s >= S) return;   // block-uniform
        constexpr int ua = (u + 1) % kWA, uw = (u + 1) % kWW;   // ring slots holding step s + 1
        const T* tb = tl + (s & 1) * (kRows * LD);
        // lane (nl, g), MFMA j of the step: the 16 bytes (4 j + g) of its row, k = 32 j + 8 g .. + 7
        const T* ab = tb + (wm * (16 * MT) + nl) * LD + g * 8;
        const T* wb = tb + (BM + wn * (16 * NT) + nl) * LD + g * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf16x8 af[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(ab + i * 16 * LD + j * 32);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(wb + n * 16 * LD + j * 32);
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[i][n], 0, 0, 0);
            }
        }
        if (!CK || s + 1 < S) {   // step s + 1 -> the buffer last read in step s - 1
            T* dst = tl + ((s + 1) & 1) * (kRows * LD) + soff;
#pragma unroll
            for (int p = 0; p < MT; ++p) *reinterpret_cast<uint4*>(dst + p * 64 * LD) = ra[ua][p];
#pragma unroll
            for (int p = 0; p < PW; ++p) *reinterpret_cast<uint4*>(dst + (BM + p * 64) * LD) = rw[uw][p];
        }
        if (!CK || s + 1 + kWA < S) {
            const int sa = CK ? s + 1 + kWA : min(s + 1 + kWA, S - 1);
#pragma unroll
            for (int p = 0; p < MT; ++p) ra[ua][p] = *reinterpret_cast<const uint4*>(srcA[p] + (size_t)sa * 64);
        }
        if (!CK || s + 1 + kWW < S) {
            const int sw = CK ? s + 1 + kWW : min(s + 1 + kWW, S - 1);
#pragma unroll
            for (int p = 0; p < PW; ++p) rw[uw][p] = *reinterpret_cast<const uint4*>(srcW[p] + (size_t)sw * 64);
        }
        __syncthreads();
    };
    using std::integral_constant;
    int s0 = 0;
    for (; s0 + 4 <= S; s0 += 4) {
        step(integral_constant<int, 0>{}, std::false_type{}, s0);
        step(integral_constant<int, 1>{}, std::false_type{}, s0 + 1);
        step(integral_constant<int, 2>{}, std::false_type{}, s0 + 2);
        step(integral_constant<int, 3>{}, std::false_type{}, s0 + 3);
    }
    step(integral_constant<int, 0>{}, std::true_type{}, s0);
    step(integral_constant<int, 1>{}, std::true_type{}, s0 + 1);
    step(integral_constant<int, 2>{}, std::true_type{}, s0 + 2);
    T3K_DBG_STAMP(2);
    // wave (wm, wn): rows m0 + 16 (MT wm + i) + 4 g + r; sub-tiles 2 tt / 2 tt + 1 are the partner halves of 32-feature
    // tile blk NT + wn NT/2 + tt
    {
        constexpr int TT = NT / 2;
        EpiCol ec[TT];
        EpiRow er[MT][4];
        float2 cs[MT][4][TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) ec[tt] = cols_bias<EPI>(a, min(blk * NT + wn * TT + tt, tiles32 - 1), nl);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) er[i][r] = rows_pos<EPI>(a, m0 + (wm * MT + i) * 16 + g * 4 + r);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) cs[i][r][tt] = rope_cs<EPI>(a, blk * NT + wn * TT + tt, nl, er[i][r].pos);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
                    const int t = blk * NT + wn * TT + tt;
                    if (t < tiles32)
                        rows_epilogue<EPI, T>(a, t, nl, m0 + (wm * MT + i) * 16 + g * 4 + r, acc[i][2 * tt][r], acc[i][2 * tt + 1][r], kz,
                                              gridDim.y > 1, er[i][r], ec[tt], cs[i][r][tt]);
                }
    }
    T3K_DBG_STAMP(3);
}

// ---- bf16 many-row GEMM, third generation: activation-stationary, for <= 128 rows --------------------------------
// gemm_rows_kernel stages activations AND weights through LDS once per 128-k step, with a block barrier per step:
// at <= 128 rows it is bound by LDS traffic and by the barriers, not by HBM.  Here the MC activation rows of a block
// (MC = 16 or 32 rows x KB k bf16; 32 x 1024: 64 KB) go to LDS ONCE and stay; the weights never touch LDS: they are
// read from a FRAGMENT-MAJOR copy of the matrix (fragment_weights_kernel: the 64 x 16 B of one MFMA B operand -- 16
// features x 32 k -- are contiguous, so a wave's load of one fragment is one 1 KB run), straight into the registers the
// MFMA takes.  A block owns NSUB sub-tiles of 16 features (NSUB / 2 partner tiles of 32, as tile32_feature lays them
// out) and its 8 waves are NSUB sub-tiles x KW = 8 / NSUB slices of K: wave (sub, kw) holds MC / 16 accumulator tiles
// (MC rows x 16 features).  Every global load of the kernel is issued at its start -- activation pieces first, then all
// JW = KB / 32 / KW weight fragments of the wave -- and the activation rows reach LDS in phases of 256 k: phase q + 1 is
// written while phase q is multiplied (one barrier per phase, no buffer reuse).  The K slices' accumulators meet in LDS
// (fixed order kw = 0, 1, ..: deterministic), where the partner sub-tiles also meet for the fused epilogues.  K > KB
// (down_proj) is split over blockIdx.z into the fp32 partial slabs the row norms fold.
// Measured (scripts/micro/gemm_micro.cpp, 128 rows, cold weights, back to back): qkv 7.9 -> 6.3 us, gate_up 10.8 -> 8.8,
// down 10.0 -> 8.1 against gemm_rows_kernel; the 400-step session at 64 utterances 2.017 -> 1.845 ms per step, at 8
// utterances 1.147 -> 0.980.  Two things the first version got wrong, both visible only in the ISA: a run-time
// `if (tile < live tiles)` around the MFMAs made every LDS read wait on its own (loop 8.3 us instead of 2.4), and a
// staging array of HIP's uint4 STRUCT stayed in scratch memory (copied with memcpy; the native vector type does not).
// fp32 operands (T = float) take the same kernel in the same byte shapes: a "k-step" is 64 bytes of a row for both types
// (32 bf16 or 16 fp32 k; one v_mfma_f32_16x16x32_bf16, or four v_mfma_f32_16x16x4_f32 -- lane (n, g) holds k = 16 j + 4 g
// + i for MFMA i, exact fp32 products), a phase is 8 steps (512 B of every row), a fragment is 1 KB.
template <int EPI, typename T>
__global__ __launch_bounds__(64) void fragment_weights_kernel(const T* __restrict__ W, T* __restrict__ out, int N, int K) {
    constexpr int E = 16 / (int)sizeof(T), KS = 4 * E;   // elements per 16 B; k per step
    const int ksteps = K / KS;
    const int q = blockIdx.x / ksteps, j = blockIdx.x % ksteps;   // sub-tile, k-step
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    const int f = tile32_feature<EPI>(q >> 1, ((q & 1) << 4) + n, N);
    uint4 v{0u, 0u, 0u, 0u};
    if (f < N) v = *reinterpret_cast<const uint4*>(W + (size_t)f * K + KS * j + E * g);
    reinterpret_cast<uint4*>(out)[(size_t)blockIdx.x * 64 + lane] = v;
}

constexpr int kAsRedStride = 20;   // floats per (row) of the reduction buffer: the 4 lane groups of a store hit 4 x 16 distinct banks

// NRM: split norm (GemmArgs): 0 none, 1 consumer (row scales from row_ss), 2 producer (EPI_RESID: next_act / next_ss), 3 producer
// of the LayerNorm form (row means from resid_mean / row_dot / mean_shift) -- a template argument, not a run-time test: with the
// tests in the common kernel every launch measured 2 % slower.  (A first LayerNorm form computed the products activation row x
// column means inside this kernel, from the staging registers: 8 more 16-byte loads per thread, o_proj 6.1 us at 8 rows against
// 4.3 for the K-split LDS-staged kernel; the attention launch has each head's 64 values in registers anyway.)
template <int EPI, typename T, int NSUB, int KB, int MC, int NRM = 0, bool PF = false>
__global__ __launch_bounds__(512) void gemm_as_kernel(GemmArgs a, int tiles32) {
    constexpr int E = 16 / (int)sizeof(T);   // elements per 16-byte piece: 8 (bf16) or 4 (fp32)
    constexpr int KS = 4 * E;                // k per step (64 B of a row): 32 or 16
    constexpr int PHK = 8 * KS;              // k per staging phase (512 B of every row): 256 or 128
    constexpr int KW = 8 / NSUB;             // K slices: waves per sub-tile
    constexpr int NPH = KB / PHK;            // staging phases
    constexpr int JQ = 8 / KW;               // k-steps per wave and phase
    constexpr int JW = NPH * JQ;             // k-steps (weight fragments) per wave
    constexpr int NT = MC / 16;              // accumulator tiles per wave = staging pieces per thread and phase
    constexpr int LDA = KB + E;              // LDS row stride in elements (+16 B: 16-row fragment reads fall in distinct banks)
    constexpr int RS = kAsRedStride;
    static_assert(NPH >= 2 && NPH <= 16, "2 to 16 staging phases");
    constexpr int kLdsMain = (MC * LDA * (int)sizeof(T)) > (8 * MC * RS * 4) ? (MC * LDA * (int)sizeof(T)) : (8 * MC * RS * 4);
    extern __shared__ __attribute__((aligned(16))) unsigned char as_raw[];
    T* lds = reinterpret_cast<T*>(as_raw);   // [MC][LDA], later float red[KW][NSUB][MC][RS]
    float* rs_s = reinterpret_cast<float*>(as_raw + kLdsMain);   // [MC] row scales of the split norm, behind both
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = lane & 15, g = lane >> 4;
    const int sub = wave % NSUB, kw = wave / NSUB;
    const int m0 = blockIdx.y * MC, kz = blockIdx.z;
    const int K = a.K;
    if constexpr (PF) {   // grid rows y >= 1 are prefetch blocks (PfArgs); the GEMM itself has one row chunk and K unsplit
        if (blockIdx.y != 0) {   // (block-uniform)
            prefetch_block(a.pf, (blockIdx.y - 1) * gridDim.x + blockIdx.x, 512);
            return;
        }
    }
    T3K_DBG_DECL(a.dbg != nullptr && tid == 0, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
    T3K_DBG_STAMP(0);
    // Every global load of the kernel is issued here, activations first: they come from L2 and are consumed first, and
    // loads complete in order.  Phase q of the activation rows = k PHK q .. PHK q + PHK - 1 of the block's range: 32 pieces
    // of 16 B per row, piece tid + 512 i -> row (tid + 512 i) / 32 (32 consecutive threads fetch one 512-byte run).
    u32x4 va[NPH * NT];
    {
        const T* Ap = static_cast<const T*>(a.A) + (size_t)kz * KB + (tid & 31) * E;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            int row = m0 + ((tid + 512 * i) >> 5);
            row = row < a.M ? row : a.M - 1;
            const T* rp = Ap + (size_t)row * K;
#pragma unroll
            for (int q = 0; q < NPH; ++q) va[q * NT + i] = *reinterpret_cast<const u32x4*>(rp + q * PHK);
        }
    }
    // this wave's weight fragments: sub-tile qs, k-steps 8 q + JQ kw + jj of the block's range (1 KB each, contiguous per wave)
    // producer of the split norm: XCD j (blocks with id % 8 == j) owns a run of consecutive feature tiles, so that every 128-byte
    // line of the rows it writes (fp32 y and the activation-type copy) is written inside one XCD's L2, not in 32-64-byte pieces
    // from several
    const int bx = (NRM >= 2 && (gridDim.x & 7) == 0) ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    int qs = bx * NSUB + sub;
    qs = qs < 2 * tiles32 ? qs : 2 * tiles32 - 1;
    const u32x4* wp = reinterpret_cast<const u32x4*>(a.Wf) + ((size_t)qs * (K / KS) + (size_t)kz * (KB / KS) + kw * JQ) * 64 + lane;
    u32x4 wf[JW];
#pragma unroll
    for (int q = 0; q < NPH; ++q)
#pragma unroll
        for (int jj = 0; jj < JQ; ++jj) wf[q * JQ + jj] = wp[(size_t)(q * 8 + jj) * 64];
    // split norm, consumer side: thread (row = tid / 16, j = tid % 16) fetches the partial sums of squares of tiles j and j + 16
    // of its row (requested last, used after the last phase: they are never waited for early)
    float ssa = 0.f, ssb = 0.f;
    if (NRM == 1 && (tid >> 4) < MC) {   // wave-uniform
        int row = m0 + (tid >> 4);
        row = row < a.M ? row : a.M - 1;
        const int j = tid & 15;
        if (j < a.row_ss_tiles) ssa = a.row_ss[(size_t)j * a.M + row];
        if (j + 16 < a.row_ss_tiles) ssb = a.row_ss[(size_t)(j + 16) * a.M + row];
    }
    // LayerNorm producer: thread (row = tid / 16, j = tid % 16) fetches the attention launch's products (attention row of head j
    // and j + 16) . mean_vec of its row, and the old row mean (requested last as well)
    float rda = 0.f, rdb = 0.f, mu_old = 0.f;
    if (NRM == 3 && (tid >> 4) < MC) {   // wave-uniform
        int row = m0 + (tid >> 4);
        row = row < a.M ? row : a.M - 1;
        const int j = tid & 15;
        if (j < a.row_dot_n) rda = a.row_dot[(size_t)row * a.row_dot_n + j];
        if (j + 16 < a.row_dot_n) rdb = a.row_dot[(size_t)row * a.row_dot_n + j + 16];
        mu_old = a.resid_mean[row];
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the loads above the stores below (the compiler otherwise sinks the weight loads)
    T* sdst = lds + (tid >> 5) * LDA + (tid & 31) * E;
#pragma unroll
    for (int i = 0; i < NT; ++i) *reinterpret_cast<u32x4*>(sdst + i * 16 * LDA) = va[i];
    __syncthreads();
    T3K_DBG_STAMP(1);
    // straight-line code (rows past M are copies of row M - 1 and are dropped by the epilogue): no branch may sit between
    // the fragment reads and the MFMAs, or every read is waited for on its own
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const T* ab = lds + nl * LDA + (kw * JQ) * KS + g * E;   // lane (nl, g), phase q, step jj: k = PHK q + KS (JQ kw + jj) + E g .. + E - 1
    // (the phase index comes as a type: with a loop variable the staging registers' indices are not constants when the
    // compiler decides where `va` lives, and it keeps the array in scratch memory)
    auto phase = [&](auto QC) {
        constexpr int q = decltype(QC)::value;
        if constexpr (q < NPH) {
            if constexpr (q + 1 < NPH) {   // phase q + 1 -> LDS while phase q is multiplied
#pragma unroll
                for (int i = 0; i < NT; ++i) *reinterpret_cast<u32x4*>(sdst + i * 16 * LDA + (q + 1) * PHK) = va[(q + 1) * NT + i];
            }
#pragma unroll
            for (int jj = 0; jj < JQ; ++jj) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const T* ap = ab + t * 16 * LDA + q * PHK + jj * KS;
                    if constexpr (sizeof(T) == 2) {
                        const bf16x8 bfr = __builtin_bit_cast(bf16x8, wf[q * JQ + jj]);
                        const bf16x8 af = *reinterpret_cast<const bf16x8*>(ap);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[t], 0, 0, 0);
                    } else {
                        const f32x4 bfr = __builtin_bit_cast(f32x4, wf[q * JQ + jj]);
                        const f32x4 af = *reinterpret_cast<const f32x4*>(ap);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bfr.x, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bfr.y, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bfr.z, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bfr.w, acc[t], 0, 0, 0);
                    }
                }
            }
            __syncthreads();   // phase q + 1 is visible; after the last phase: every wave is done with the activation rows
        }
    };
    phase(std::integral_constant<int, 0>{});
    phase(std::integral_constant<int, 1>{});
    phase(std::integral_constant<int, 2>{});
    phase(std::integral_constant<int, 3>{});
    phase(std::integral_constant<int, 4>{});
    phase(std::integral_constant<int, 5>{});
    phase(std::integral_constant<int, 6>{});
    phase(std::integral_constant<int, 7>{});
    if constexpr (NPH > 8) {   // K = 4096 unsplit (down_proj as the producer of the next layer's split norm)
        phase(std::integral_constant<int, 8>{});
        phase(std::integral_constant<int, 9>{});
        phase(std::integral_constant<int, 10>{});
        phase(std::integral_constant<int, 11>{});
        phase(std::integral_constant<int, 12>{});
        phase(std::integral_constant<int, 13>{});
        phase(std::integral_constant<int, 14>{});
        phase(std::integral_constant<int, 15>{});
    }
    T3K_DBG_STAMP(2);
    if constexpr (NRM == 1) {   // 16 lanes per row add their partial sums in a fixed butterfly order
        float sv = ssa + ssb;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) sv += __shfl_xor(sv, o, 64);
        if ((tid & 15) == 0 && (tid >> 4) < MC) rs_s[tid >> 4] = 1.0f / sqrtf(sv / (float)K + a.norm_eps);
    }
    if constexpr (NRM == 3) {   // rs_s[row] = the new row mean; 16 lanes per row add the heads' products in a fixed butterfly order
        float sv = rda + rdb;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) sv += __shfl_xor(sv, o, 64);
        if ((tid & 15) == 0 && (tid >> 4) < MC) rs_s[tid >> 4] = mu_old + (sv + a.mean_shift);
    }
    float* red = reinterpret_cast<float*>(as_raw);   // the buffer becomes the reduction area
    {
        float* rb = red + (size_t)((kw * NSUB + sub) * MC) * RS + nl;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) rb[(t * 16 + g * 4 + r) * RS] = acc[t][r];   // accumulator row g*4 + r, column nl
    }
    __syncthreads();
    // epilogue items: (partner tile p of the block, row, column) -- NSUB / 2 * MC * 16 of them.  All the loads of a
    // thread's items come before its first store (see rows_epilogue).
    constexpr int ITEMS = NSUB / 2 * MC * 16;
    constexpr int NIT = (ITEMS + 511) / 512;
    float v0[NIT], v1[NIT];
    EpiRow er[NIT];
    EpiCol ec[NIT];
    float2 cs[NIT];
    int tile_[NIT], mi_[NIT];
    bool ok_[NIT];
    const int c = tid & 15;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int it = tid + 512 * i;
        const int row = (it >> 4) % MC, p = ((it >> 4) / MC) % (NSUB / 2);
        const int tile = bx * (NSUB / 2) + p;
        mi_[i] = m0 + row;
        ok_[i] = it < ITEMS && tile < tiles32 && mi_[i] < a.M;
        tile_[i] = tile < tiles32 ? tile : tiles32 - 1;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int w = 0; w < KW; ++w) {
            s0 += red[(size_t)(((w * NSUB + 2 * p) * MC) + row) * RS + c];
            s1 += red[(size_t)(((w * NSUB + 2 * p + 1) * MC) + row) * RS + c];
        }
        if constexpr (NRM == 1) {
            const float rsr = rs_s[row];
            s0 *= rsr; s1 *= rsr;
        }
        v0[i] = s0; v1[i] = s1;
        er[i] = rows_pos<EPI>(a, ok_[i] ? mi_[i] : a.M);
        ec[i] = cols_bias<EPI>(a, tile_[i], c);
    }
    if constexpr (EPI == EPI_RESID && NRM >= 2) {
        {   // split norm, producer side (unsplit K, N % 32 == 0: every column of a tile is a feature)
            float rv0[NIT], rv1[NIT], w0[NIT], w1[NIT];
#pragma unroll
            for (int i = 0; i < NIT; ++i) {   // all loads before the first store (resid may alias y)
                const size_t o = (size_t)(ok_[i] ? mi_[i] : 0) * a.ldy + 32 * tile_[i] + c;
                rv0[i] = a.resid[o]; rv1[i] = a.resid[o + 16];
                w0[i] = a.next_norm_w[32 * tile_[i] + c]; w1[i] = a.next_norm_w[32 * tile_[i] + 16 + c];
            }
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                float x0 = v0[i], x1 = v1[i];
                if (a.bias != nullptr) { x0 = __fadd_rn(x0, ec[i].b0); x1 = __fadd_rn(x1, ec[i].b1); }
                x0 = __fadd_rn(rv0[i], x0); x1 = __fadd_rn(rv1[i], x1);
                float z0 = x0, z1 = x1;   // LayerNorm form: centred on the row mean
                if constexpr (NRM == 3) {
                    const float mu = rs_s[((tid + 512 * i) >> 4) % MC];
                    z0 = x0 - mu; z1 = x1 - mu;
                }
                float sq = ok_[i] ? fmaf(z0, z0, z1 * z1) : 0.f;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) sq += __shfl_xor(sq, o, 64);   // the 16 columns of (tile, row): lanes c = 0..15
                if (ok_[i]) {
                    const size_t o = (size_t)mi_[i] * a.ldy + 32 * tile_[i] + c;
                    a.y[o] = x0; a.y[o + 16] = x1;
                    T* na = static_cast<T*>(a.next_act) + (size_t)mi_[i] * a.N + 32 * tile_[i] + c;
                    Act<T>::store(na, __fmul_rn(w0[i], z0));
                    Act<T>::store(na + 16, __fmul_rn(w1[i], z1));
                    if (c == 0) a.next_ss[(size_t)tile_[i] * a.M + mi_[i]] = sq;
                }
            }
            T3K_DBG_STAMP(3);
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) cs[i] = rope_cs<EPI>(a, tile_[i], c, er[i].pos);
#pragma unroll
    for (int i = 0; i < NIT; ++i)
        if (ok_[i]) rows_epilogue<EPI, T>(a, tile_[i], c, mi_[i], v0[i], v1[i], kz, gridDim.z > 1, er[i], ec[i], cs[i]);
    T3K_DBG_STAMP(3);
}

// y[m][i] = resid[m][i] + bias[i] + sum_s part[s][m][i], s ascending (the K-split of gemm_rows_kernel).
// grid (M, N / 256): every thread fetches its <= 8 partials at once, then adds them in order.
__global__ __launch_bounds__(256) void fold_partials_kernel(float* __restrict__ y, const float* __restrict__ resid,
                                                            const float* __restrict__ part, int ks, const float* __restrict__ bias,
                                                            int M, int N, int ldy) {
    const int m = blockIdx.x, i = blockIdx.y * 256 + threadIdx.x;
    if (i >= N) return;
    float p[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) p[s] = s < ks ? part[((size_t)s * M + m) * N + i] : 0.f;
    const size_t o = (size_t)m * ldy + i;
    const float r = resid != nullptr ? resid[o] : 0.f;
    float v = p[0];
#pragma unroll
    for (int s = 1; s < 8; ++s)
        if (s < ks) v += p[s];
    if (bias != nullptr) v = __fadd_rn(v, bias[i]);
    y[o] = resid != nullptr ? __fadd_rn(r, v) : v;
}

// rows per block of gemm_rows_kernel (32 * AP) for a launch of M rows (bf16 operands).  A block's time is mostly fixed
// cost plus LDS traffic per step (32 * AP activation rows + 32 weight rows written, the fragments read back), so with
// <= 128 rows smaller row chunks beat one 128-row chunk where the launch has few feature tiles: twice the blocks, 88
// instead of 136 KB of LDS traffic per step each.  Measured at 128 rows over a 400-step session (step 2.153 ms with
// 128-row chunks everywhere): qkv and gate_up in 64-row chunks 2.051 ms, o_proj (K-split, 2 steps per block) in 32-row
// chunks as well 2.020 ms; 32-row chunks for qkv (2.059) or gate_up (2.082) and smaller chunks for down (2.023-2.099)
// do not pay.  The results are bit-identical (a row's sums do not depend on its chunk).
template <int EPI>
inline int rows_chunk_ap(int M, int K) {
    if (M <= 32) return 1;
    if (M > 128) return 4;
    if (EPI == EPI_RESID) return K <= 1024 ? 1 : (M <= 64 ? 2 : 4);
    return 2;
}

template <int EPI, typename T, int SC>
void launch_rows_ap(const GemmArgs& a, int ap, int tiles, int ks, hipStream_t s) {
    constexpr int row_b = (128 + 8) * 2;   // LDS row stride in bytes (bf16 and fp32 alike: 256 B of data + 16 B)
    const dim3 block(512);
    if constexpr (sizeof(T) == 2) {
        if (ap == 1) {
            hipLaunchKernelGGL((gemm_rows_kernel<EPI, T, SC, 1>), dim3(tiles, (a.M + 31) / 32, ks), block, 2 * 64 * row_b, s, a);
            return;
        }
        if (ap == 2) {
            hipLaunchKernelGGL((gemm_rows_kernel<EPI, T, SC, 2>), dim3(tiles, (a.M + 63) / 64, ks), block, 2 * 96 * row_b, s, a);
            return;
        }
    }
    hipLaunchKernelGGL((gemm_rows_kernel<EPI, T, SC, 4>), dim3(tiles, (a.M + 127) / 128, ks), block, 2 * 160 * row_b, s, a);
}

template <int EPI, typename T>
hipError_t launch_gemm_rows(const GemmArgs& a, hipStream_t s) {
    constexpr int KC = 16 * (16 / (int)sizeof(T));   // the kernel's step: 128 (bf16) or 64 (fp32) k
    int tiles;
    if (EPI == EPI_ROPE_KV) tiles = (a.N / kHeadDim) * 2;
    else if (EPI == EPI_SWIGLU) tiles = a.N / 32;
    else tiles = (a.N + 31) / 32;
    const int chunks = (a.M + 127) / 128;
    int ks = 1;
    if ((EPI == EPI_RESID || EPI == EPI_STORE) && a.part != nullptr) {
        // few feature tiles: split K over blocks until the launch has ~192 blocks (steps per block stay even)
        while (ks < 8 && (long long)tiles * chunks * ks < 192 && a.K % (2 * KC * ks * 2) == 0 &&
               (size_t)(2 * ks) * a.M * a.N <= a.part_floats)
            ks *= 2;
    }
    constexpr int lds = 2 * 160 * (128 + 8) * 2;   // two step tiles of the 128-row form: 87,040 B
    constexpr int SC16 = sizeof(T) == 4 ? 16 : 0;  // fp32 operands: 64-k steps, K = 1024 is 16 of them
    // the attribute belongs to the (kernel, device) pair: one flag per EPI instantiation and device
    static bool attr_done[64] = {};   // (static in a function template: one array per <EPI, T>)
    int dev = 0;
    hipError_t ed = hipGetDevice(&dev);
    if (ed != hipSuccess) return ed;
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
        const void* fns[5] = {reinterpret_cast<const void*>(gemm_rows_kernel<EPI, T, 0, 4>),
                              reinterpret_cast<const void*>(gemm_rows_kernel<EPI, T, 8, 4>),
                              reinterpret_cast<const void*>(gemm_rows_kernel<EPI, T, 4, 4>),
                              reinterpret_cast<const void*>(gemm_rows_kernel<EPI, T, 2, 4>),
                              reinterpret_cast<const void*>(gemm_rows_kernel<EPI, T, SC16, 4>)};
        for (const void* fn : fns) {
            hipError_t e0 = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e0 != hipSuccess) return e0;
        }
        if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    const int steps = a.K / KC / ks;
    const int ap = sizeof(T) == 2 ? rows_chunk_ap<EPI>(a.M, a.K) : 4;
    if (steps == 8) launch_rows_ap<EPI, T, 8>(a, ap, tiles, ks, s);
    else if (steps == 4) launch_rows_ap<EPI, T, 4>(a, ap, tiles, ks, s);
    else if (steps == 2) launch_rows_ap<EPI, T, 2>(a, ap, tiles, ks, s);
    else if (sizeof(T) == 4 && steps == 16) launch_rows_ap<EPI, T, SC16>(a, 4, tiles, ks, s);
    else launch_rows_ap<EPI, T, 0>(a, 4, tiles, ks, s);
    hipError_t e = hipGetLastError();
    if (a.ks_out != nullptr) *a.ks_out = ks;
    if (e != hipSuccess || ks == 1) return e;
    if (a.ks_out != nullptr && EPI == EPI_RESID && a.resid == a.y) return e;   // the caller folds (fused into its norm launch)
    hipLaunchKernelGGL(fold_partials_kernel, dim3(a.M, (a.N + 255) / 256), dim3(256), 0, s, a.y, EPI == EPI_RESID ? a.resid : nullptr, a.part, ks,
                       a.bias, a.M, a.N, a.ldy);
    return hipGetLastError();
}

template <int EPI>
inline int tiles32_of(int N) {
    if (EPI == EPI_ROPE_KV) return (N / kHeadDim) * 2;
    if (EPI == EPI_SWIGLU) return N / 32;
    return (N + 31) / 32;
}

template <int EPI, typename T, int NSUB, int KB, int MC, int NRM = 0>
hipError_t launch_gemm_as_v(const GemmArgs& a, int tiles32, int gz, hipStream_t s) {
    constexpr int lds_a = MC * (KB * (int)sizeof(T) + 16), lds_r = 8 * MC * kAsRedStride * 4;
    constexpr int lds = (lds_a > lds_r ? lds_a : lds_r) + MC * 4;   // + the split norm's row scales; bf16: 66,176 B at KB = 1024, MC = 32
    static bool attr_done[64] = {};   // one flag per instantiation and device
    int dev = 0;
    hipError_t ed = hipGetDevice(&dev);
    if (ed != hipSuccess) return ed;
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_as_kernel<EPI, T, NSUB, KB, MC, NRM>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e0 != hipSuccess) return e0;
        if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    dim3 grid((2 * tiles32 + NSUB - 1) / NSUB, (a.M + MC - 1) / MC, gz);
    if constexpr (sizeof(T) == 2 && MC == 16) {   // (bf16, one chunk of up to 16 rows: the regime the prefetch is for)
        // Prefetch blocks ride as further grid ROWS of a launch with one row chunk and K unsplit whose width is a multiple of 8 (block
        // -> XCD stays x % 8).  Rows, not layers: with the same blocks placed at z >= 1 the consumers found nothing in their L2s (Llama,
        // 4 utterances: 0.842 ms per step against 0.825) -- the dispatcher does not deal the blocks of further z layers over the XCDs
        // the way it deals those of further rows.
        if (a.pf.n_blocks > 0 && a.pf.sink != nullptr && grid.z == 1 && grid.y == 1 && (grid.x & 7) == 0) {
            static bool attr_pf[64] = {};
            if (dev < 0 || dev >= 64 || !attr_pf[dev]) {
                hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_as_kernel<EPI, T, NSUB, KB, MC, NRM, true>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                if (e0 != hipSuccess) return e0;
                if (dev >= 0 && dev < 64) attr_pf[dev] = true;
            }
            GemmArgs b = a;
            const unsigned rows = ((unsigned)a.pf.n_blocks + grid.x - 1) / grid.x;
            b.pf.n_blocks = (int)(rows * grid.x);   // whole rows (the sink holds kPfMaxBlocks blocks)
            if (b.pf.n_blocks <= kPfMaxBlocks) {
                grid.y = 1 + rows;
                hipLaunchKernelGGL((gemm_as_kernel<EPI, T, NSUB, KB, MC, NRM, true>), grid, dim3(512), lds, s, b, tiles32);
                return hipGetLastError();
            }
        }
    }
    hipLaunchKernelGGL((gemm_as_kernel<EPI, T, NSUB, KB, MC, NRM>), grid, dim3(512), lds, s, a, tiles32);
    return hipGetLastError();
}

constexpr int kAsMinBlocks = 192;   // sub-tiles per block and K range per block are chosen to leave at least this many blocks

// Rows per block of the activation-stationary kernel: 16 up to 32 rows, else 32.  Measured over 400-step sessions (bf16,
// ms per step, chunks of 64 / 32 / 16 rows): 128 rows 1.895 / 1.852 / 1.930, 96 rows 1.710 / 1.656 / 1.707, 64 rows
// 1.456 / 1.354 / 1.386, 48 rows 1.353 / 1.263 / 1.298, 32 rows 1.236 / 1.136 / 1.112, 16 rows 1.118 / 1.017 / 0.981:
// smaller chunks put more blocks in flight (two 66 KB blocks fit a CU) and shorten the staging each block waits for;
// the weights are re-requested once per chunk, but from L2.  (kAsMinBlocks 128 / 192 / 320 / 512: 192 is best or tied at
// every batch size.)  fp32 operands, the same rule (64 utterances, 200 steps: 32-row chunks 3.454 ms per step, 16-row chunks
// 3.807, the LDS-staged kernel 3.603).
template <typename T>
inline int as_chunk_rows(int M) { return M <= 32 ? 16 : 32; }
// 4 sub-tiles per block when that still leaves kAsMinBlocks blocks, else 2
inline bool as_four(int tiles32, int chunks, int gz) { return (long long)((2 * tiles32 + 3) / 4) * chunks * gz >= kAsMinBlocks; }

template <int EPI, typename T, int NSUB, int KB>
hipError_t launch_gemm_as_m(const GemmArgs& a, int tiles32, int gz, hipStream_t s) {
    if constexpr (EPI == EPI_SWIGLU || EPI == EPI_GELU || EPI == EPI_ROPE_KV) {
        if (a.row_ss != nullptr) {
            if (as_chunk_rows<T>(a.M) == 16) return launch_gemm_as_v<EPI, T, NSUB, KB, 16, 1>(a, tiles32, gz, s);
            return launch_gemm_as_v<EPI, T, NSUB, KB, 32, 1>(a, tiles32, gz, s);
        }
    }
    if constexpr (EPI == EPI_RESID) {
        if (a.next_act != nullptr && a.row_dot != nullptr) {
            if (as_chunk_rows<T>(a.M) == 16) return launch_gemm_as_v<EPI, T, NSUB, KB, 16, 3>(a, tiles32, gz, s);
            return launch_gemm_as_v<EPI, T, NSUB, KB, 32, 3>(a, tiles32, gz, s);
        }
        if (a.next_act != nullptr) {
            if (as_chunk_rows<T>(a.M) == 16) return launch_gemm_as_v<EPI, T, NSUB, KB, 16, 2>(a, tiles32, gz, s);
            return launch_gemm_as_v<EPI, T, NSUB, KB, 32, 2>(a, tiles32, gz, s);
        }
    }
    if (as_chunk_rows<T>(a.M) == 16) return launch_gemm_as_v<EPI, T, NSUB, KB, 16>(a, tiles32, gz, s);
    return launch_gemm_as_v<EPI, T, NSUB, KB, 32>(a, tiles32, gz, s);
}

// Activation-stationary launch, or `handled = false` when the shape is not its own (the caller falls through).
template <int EPI, typename T>
hipError_t launch_gemm_as(const GemmArgs& a, hipStream_t s, bool& handled) {
    handled = false;
    const bool splittable = (EPI == EPI_RESID || EPI == EPI_STORE) && a.part != nullptr;
    if (EPI == EPI_SWIGLU && a.N % 32) return hipSuccess;
    if constexpr (EPI == EPI_RESID && sizeof(T) == 2) {
        // down_proj as the producer of the NEXT layer's pre-attention RMS norm: K = 4096 unsplit in 16-row chunks (128 KB of LDS), so
        // that the rows leave the block complete -- no partial slabs and no norm launch to fold them (the caller passes no `part`)
        if (a.next_act != nullptr && a.row_dot == nullptr && a.part == nullptr && a.K == 4096 && (a.N % 32) == 0 && a.next_norm_w && a.next_ss) {
            handled = true;
            if (a.ks_out != nullptr) *a.ks_out = 1;
            return launch_gemm_as_v<EPI_RESID, T, 2, 4096, 16, 2>(a, tiles32_of<EPI>(a.N), 1, s);
        }
    }
    const int mc = as_chunk_rows<T>(a.M);
    const int tiles32 = tiles32_of<EPI>(a.N), chunks = (a.M + mc - 1) / mc;
    // the block's K range: all of K (1024 or 512), or for the residual GEMMs a split into fp32 partial slabs -- 1024 k
    // each, 512 when that leaves fewer than kAsMinBlocks blocks
    int KB = 0;
    if (a.K == 1024 || a.K == 512) KB = a.K;
    else if (splittable && a.K % 1024 == 0) {
        KB = 1024;
        if ((long long)tiles32 * chunks * (a.K / 1024) < kAsMinBlocks && a.K / 512 <= 8) KB = 512;
    }
    if (KB == 0) return hipSuccess;
    const int gz = a.K / KB;
    if (gz > 8 || (gz > 1 && (size_t)gz * a.M * a.N > a.part_floats)) return hipSuccess;
    if (a.next_act != nullptr && (EPI != EPI_RESID || gz != 1 || (a.N % 32) || !a.next_norm_w || !a.next_ss)) return hipSuccess;
    if ((a.row_dot != nullptr) != (a.resid_mean != nullptr) || (a.row_dot != nullptr && (a.next_act == nullptr || a.row_dot_n < 1 || a.row_dot_n > 32)))
        return hipSuccess;
    if (a.row_ss != nullptr && ((EPI != EPI_SWIGLU && EPI != EPI_GELU && EPI != EPI_ROPE_KV) || a.row_ss_tiles < 1 || a.row_ss_tiles > 32)) return hipSuccess;
    const bool four = as_four(tiles32, chunks, gz);
    handled = true;
    hipError_t e;
    if (KB == 1024) e = four ? launch_gemm_as_m<EPI, T, 4, 1024>(a, tiles32, gz, s) : launch_gemm_as_m<EPI, T, 2, 1024>(a, tiles32, gz, s);
    else e = four ? launch_gemm_as_m<EPI, T, 4, 512>(a, tiles32, gz, s) : launch_gemm_as_m<EPI, T, 2, 512>(a, tiles32, gz, s);
    if (a.ks_out != nullptr) *a.ks_out = gz;
    if (e != hipSuccess || gz == 1) return e;
    if (a.ks_out != nullptr && EPI == EPI_RESID && a.resid == a.y) return e;   // the caller folds (fused into its norm launch)
    hipLaunchKernelGGL(fold_partials_kernel, dim3(a.M, (a.N + 255) / 256), dim3(256), 0, s, a.y, EPI == EPI_RESID ? a.resid : nullptr, a.part, gz,
                       a.bias, a.M, a.N, a.ldy);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_gemm_as_e(const GemmArgs& a, int epilogue, hipStream_t s, bool& handled) {
    switch (epilogue) {
        case EPI_STORE: return launch_gemm_as<EPI_STORE, T>(a, s, handled);
        case EPI_RESID: return launch_gemm_as<EPI_RESID, T>(a, s, handled);
        case EPI_GELU: return launch_gemm_as<EPI_GELU, T>(a, s, handled);
        case EPI_SWIGLU: return launch_gemm_as<EPI_SWIGLU, T>(a, s, handled);
        case EPI_ROPE_KV: return launch_gemm_as<EPI_ROPE_KV, T>(a, s, handled);
    }
    handled = false;
    return hipErrorInvalidValue;
}

template <int EPI, int MT, int NT>
hipError_t launch_gemm_wide_v(const GemmArgs& a, int tiles32, hipStream_t s) {
    const int nblk = (tiles32 + NT - 1) / NT, chunks = (a.M + 64 * MT - 1) / (64 * MT);
    int ks = 1;
    if ((EPI == EPI_RESID || EPI == EPI_STORE) && a.part != nullptr) {
        // few blocks: split K until the launch has ~192 (at least two 64-k steps per block)
        while (ks < 8 && (long long)nblk * chunks * ks < 192 && (a.K / 64) % (2 * ks) == 0 && a.K / 64 / (2 * ks) >= 2 &&
               (size_t)(2 * ks) * a.M * a.N <= a.part_floats)
            ks *= 2;
    }
    constexpr int lds = 2 * (64 * MT + 32 * NT) * 144;   // two step tiles: 110,592 B at MT = NT = 4
    static bool attr_done[64] = {};   // one flag per instantiation and device
    int dev = 0;
    hipError_t ed = hipGetDevice(&dev);
    if (ed != hipSuccess) return ed;
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<EPI, MT, NT>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e0 != hipSuccess) return e0;
        if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    const int grid_x = 8 * ((nblk + 7) / 8) * chunks;
    hipLaunchKernelGGL((gemm_wide_kernel<EPI, MT, NT>), dim3(grid_x, ks), dim3(512), lds, s, a, tiles32, chunks, nblk);
    hipError_t e = hipGetLastError();
    if (a.ks_out != nullptr) *a.ks_out = ks;
    if (e != hipSuccess || ks == 1) return e;
    if (a.ks_out != nullptr && EPI == EPI_RESID && a.resid == a.y) return e;   // the caller folds (fused into its norm launch)
    hipLaunchKernelGGL(fold_partials_kernel, dim3(a.M, (a.N + 255) / 256), dim3(256), 0, s, a.y, EPI == EPI_RESID ? a.resid : nullptr, a.part, ks,
                       a.bias, a.M, a.N, a.ldy);
    return hipGetLastError();
}

// block shape by the number of blocks it leaves: 256 x 128, 256 x 64, 128 x 64 (rows x features)
template <int EPI>
hipError_t launch_gemm_wide(const GemmArgs& a, hipStream_t s) {
    int tiles32;
    if (EPI == EPI_ROPE_KV) tiles32 = (a.N / kHeadDim) * 2;
    else if (EPI == EPI_SWIGLU) tiles32 = a.N / 32;
    else tiles32 = (a.N + 31) / 32;
    const long long c4 = (a.M + 255) / 256;
    if (((tiles32 + 3) / 4) * c4 >= 128) return launch_gemm_wide_v<EPI, 4, 4>(a, tiles32, s);
    if (((tiles32 + 1) / 2) * c4 >= 128) return launch_gemm_wide_v<EPI, 4, 2>(a, tiles32, s);
    return launch_gemm_wide_v<EPI, 2, 2>(a, tiles32, s);
}

// Does the register-blocked kernel leave enough blocks for M rows?  (>= 128 before any K split with one of its
// shapes, or M >= 512.)  Measured at 256 rows (gemm_micro): gate_up 17.9 vs 20.2 us and head 18.3 vs 29.9 us in its
// favour, qkv 14.2 vs 11.6 and o_proj 10.5 vs 8.9 against (96 and 32 blocks).
inline bool gemm_wide_pays(const GemmArgs& a, int epilogue) {
    if (a.M >= 512) return true;
    const long long tiles32 = epilogue == EPI_ROPE_KV ? (a.N / kHeadDim) * 2 : (a.N + 31) / 32;
    return ((tiles32 + 1) / 2) * ((a.M + 127) / 128) >= 128;
}

template <typename G, int MT, int KS, int EPI>
hipError_t launch_gemm_k(const GemmArgs& a, int tiles, hipStream_t s) {
    const int chunks = (a.M + MT * 16 - 1) / (MT * 16);
    const size_t lds = (size_t)KS * MT * 256 * sizeof(float);
    auto kfn = mfma_gemm_kernel<G, MT, KS, EPI>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kfn, dim3(tiles, chunks), dim3(KS * 64), lds, s, a);
    return hipGetLastError();
}

template <typename G, int EPI>
hipError_t launch_gemm_mt(const GemmArgs& a, int tiles, hipStream_t s) {
    // rows per block (MT*16): as many as possible (each block streams its 16 weight rows once) while the
    // launch still has >= ~256 blocks; K is split over 8 waves when it is long enough
    int mt = 8;
    while (mt > 1 && (a.M <= (mt / 2) * 16 || (long long)tiles * ((a.M + mt * 16 - 1) / (mt * 16)) < 256)) mt >>= 1;
    const bool ks8 = (a.K % (8 * G::KB) == 0) && a.K >= 2048;
#define T3K_GEMM_CASE(MTV)                                                          \
    case MTV:                                                                       \
        return ks8 ? launch_gemm_k<G, MTV, 8, EPI>(a, tiles, s) : launch_gemm_k<G, MTV, 4, EPI>(a, tiles, s);
    switch (mt) {
        T3K_GEMM_CASE(1)
        T3K_GEMM_CASE(2)
        T3K_GEMM_CASE(4)
        default: return ks8 ? launch_gemm_k<G, 8, 8, EPI>(a, tiles, s) : launch_gemm_k<G, 8, 4, EPI>(a, tiles, s);
    }
#undef T3K_GEMM_CASE
}

template <typename G>
hipError_t launch_gemm_g(const GemmArgs& a, int epi, hipStream_t s) {
    if (a.K % (4 * G::KB)) return hipErrorInvalidValue;
    switch (epi) {
        case EPI_STORE: return launch_gemm_mt<G, EPI_STORE>(a, (a.N + 15) / 16, s);
        case EPI_RESID: return launch_gemm_mt<G, EPI_RESID>(a, (a.N + 15) / 16, s);
        case EPI_GELU: return launch_gemm_mt<G, EPI_GELU>(a, (a.N + 15) / 16, s);
        case EPI_SWIGLU: return launch_gemm_mt<G, EPI_SWIGLU>(a, a.N / 16, s);
        case EPI_ROPE_KV: return launch_gemm_mt<G, EPI_ROPE_KV>(a, a.N / 16, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace

// Sub-tiles (16 features) per block column of the activation-stationary launch of this shape (bf16 operands), and its K layers:
// what a prefetch for that launch needs to know (PfArgs::unit).  Mirrors launch_gemm_as.
int gemm_as_plan(int M, int N, int K, int epilogue, bool splittable, int* gz_out) {
    const int mc = as_chunk_rows<uint16_t>(M), chunks = (M + mc - 1) / mc;
    int tiles32;
    if (epilogue == EPI_ROPE_KV) tiles32 = (N / kHeadDim) * 2;
    else if (epilogue == EPI_SWIGLU) tiles32 = N / 32;
    else tiles32 = (N + 31) / 32;
    int KB = 0;
    if (K == 1024 || K == 512) KB = K;
    else if (splittable && K % 1024 == 0) {
        KB = 1024;
        if ((long long)tiles32 * chunks * (K / 1024) < kAsMinBlocks && K / 512 <= 8) KB = 512;
    }
    if (KB == 0) return 0;
    const int gz = K / KB;
    if (gz_out) *gz_out = gz;
    return as_four(tiles32, chunks, gz) ? 4 : 2;
}

hipError_t launch_gemm(const GemmArgs& a, int epilogue, int wdtype, hipStream_t s) {
    if (!a.W || !a.A || a.M <= 0 || a.N <= 0 || a.K <= 0) return hipErrorInvalidValue;
    if ((epilogue == EPI_STORE || epilogue == EPI_RESID) && (!a.y || a.ldy <= 0)) return hipErrorInvalidValue;
    if (epilogue == EPI_RESID && !a.resid) return hipErrorInvalidValue;
    if (epilogue == EPI_SWIGLU && (!a.yact || (a.N % 16))) return hipErrorInvalidValue;
    if (epilogue == EPI_GELU && !a.yact) return hipErrorInvalidValue;
    if (epilogue == EPI_ROPE_KV &&
        (a.N != 3 * a.n_heads * kHeadDim || !a.posmap || !a.q || !a.kcache || !a.vcache || a.max_len <= 0 ||
         (a.rope && (!a.rope_cos || !a.rope_sin))))
        return hipErrorInvalidValue;
    // 128-row tile kernel (gemm_rows_kernel): bf16 operands, or fp32 operands from 64 rows up: its block is 128 rows x 32 features whatever M is, and the exact-fp32 MFMA is
    // 16x slower than the bf16 one, so with few rows (one live wave per block) the first generation's 16-feature tiles
    // with K split over the waves win (8 rows, Turbo batched: 5156 vs 3516 tok/s)
    // 256 rows or more of bf16 operands: the register-blocked kernel (gemm_wide_kernel) where it leaves enough blocks
    if ((wdtype == 1 || wdtype == 0) && a.Wf != nullptr && a.M <= kAsMaxRows) {
        bool handled = false;
        const hipError_t e = wdtype == 1 ? launch_gemm_as_e<uint16_t>(a, epilogue, s, handled) : launch_gemm_as_e<float>(a, epilogue, s, handled);
        if (handled || e != hipSuccess) return e;
    }
    if (a.next_act != nullptr || a.row_ss != nullptr) return hipErrorInvalidValue;   // only the activation-stationary kernel has the split norm
    constexpr int wide_min = 256;
    if (wdtype == 1 && a.M >= wide_min && a.K % 128 == 0 && (epilogue != EPI_SWIGLU || a.N % 32 == 0) &&
        gemm_wide_pays(a, epilogue)) {
        switch (epilogue) {
            case EPI_STORE: return launch_gemm_wide<EPI_STORE>(a, s);
            case EPI_RESID: return launch_gemm_wide<EPI_RESID>(a, s);
            case EPI_GELU: return launch_gemm_wide<EPI_GELU>(a, s);
            case EPI_SWIGLU: return launch_gemm_wide<EPI_SWIGLU>(a, s);
            case EPI_ROPE_KV: return launch_gemm_wide<EPI_ROPE_KV>(a, s);
        }
    }
    if ((wdtype == 1 || (wdtype == 0 && a.M >= 64)) && a.K % 256 == 0 && (epilogue != EPI_SWIGLU || a.N % 32 == 0)) {
        if (wdtype == 1) {
            switch (epilogue) {
                case EPI_STORE: return launch_gemm_rows<EPI_STORE, uint16_t>(a, s);
                case EPI_RESID: return launch_gemm_rows<EPI_RESID, uint16_t>(a, s);
                case EPI_GELU: return launch_gemm_rows<EPI_GELU, uint16_t>(a, s);
                case EPI_SWIGLU: return launch_gemm_rows<EPI_SWIGLU, uint16_t>(a, s);
                case EPI_ROPE_KV: return launch_gemm_rows<EPI_ROPE_KV, uint16_t>(a, s);
            }
        } else {
            switch (epilogue) {
                case EPI_STORE: return launch_gemm_rows<EPI_STORE, float>(a, s);
                case EPI_RESID: return launch_gemm_rows<EPI_RESID, float>(a, s);
                case EPI_GELU: return launch_gemm_rows<EPI_GELU, float>(a, s);
                case EPI_SWIGLU: return launch_gemm_rows<EPI_SWIGLU, float>(a, s);
                case EPI_ROPE_KV: return launch_gemm_rows<EPI_ROPE_KV, float>(a, s);
            }
        }
    }
    if (wdtype == 2) return launch_gemm_g<GemmBF16W>(a, epilogue, s);   // bf16 weights, fp32 activation rows
    return wdtype == 1 ? launch_gemm_g<GemmBF16>(a, epilogue, s) : launch_gemm_g<GemmF32>(a, epilogue, s);
}

bool gemm_as_takes(int M, int K) { return M > 0 && M <= kAsMaxRows && (K == 1024 || K == 512); }

size_t weight_fragment_elems(int N, int K, int epilogue) {
    const int t = epilogue == EPI_ROPE_KV ? (N / kHeadDim) * 2 : (N + 31) / 32;
    return (size_t)t * 32 * K;
}

namespace {
template <typename T>
hipError_t launch_weight_fragments_t(const T* w, T* o, int N, int K, int epilogue, hipStream_t s) {
    constexpr int frag_elems = 16 * 4 * (16 / (int)sizeof(T));   // 16 features x one 64-byte step
    const size_t blocks = weight_fragment_elems(N, K, epilogue) / frag_elems;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks), block(64);
    switch (epilogue) {
        case EPI_ROPE_KV: hipLaunchKernelGGL((fragment_weights_kernel<EPI_ROPE_KV, T>), grid, block, 0, s, w, o, N, K); break;
        case EPI_SWIGLU: hipLaunchKernelGGL((fragment_weights_kernel<EPI_SWIGLU, T>), grid, block, 0, s, w, o, N, K); break;
        default: hipLaunchKernelGGL((fragment_weights_kernel<EPI_STORE, T>), grid, block, 0, s, w, o, N, K); break;
    }
    return hipGetLastError();
}
}  // namespace

hipError_t launch_weight_fragments(const void* W, void* out, int N, int K, int epilogue, int wdtype, hipStream_t s) {
    if (!W || !out || N <= 0 || K <= 0 || (K & 31) || (wdtype != 0 && wdtype != 1)) return hipErrorInvalidValue;
    if (epilogue == EPI_ROPE_KV && N % kHeadDim) return hipErrorInvalidValue;
    if (epilogue == EPI_SWIGLU && (N % 32)) return hipErrorInvalidValue;
    if (wdtype == 1) return launch_weight_fragments_t<uint16_t>(static_cast<const uint16_t*>(W), static_cast<uint16_t*>(out), N, K, epilogue, s);
    return launch_weight_fragments_t<float>(static_cast<const float*>(W), static_cast<float*>(out), N, K, epilogue, s);
}

hipError_t launch_rmsnorm_rows(float* x, const float* w, float eps, void* out, int M, int d, int act_bf16, hipStream_t s,
                               const float* part, int ks, const float* pbias) {
    if (!x || !w || !out || M <= 0 || d <= 0 || d > 256 * kNormMaxPer || (d & 3) || (ks > 0 && !part)) return hipErrorInvalidValue;
    if (act_bf16)
        hipLaunchKernelGGL(rmsnorm_rows_kernel<uint16_t>, dim3(M), dim3(256), 0, s, x, w, eps, static_cast<uint16_t*>(out), d, part,
                           ks, pbias, M);
    else
        hipLaunchKernelGGL(rmsnorm_rows_kernel<float>, dim3(M), dim3(256), 0, s, x, w, eps, static_cast<float*>(out), d, part, ks,
                           pbias, M);
    return hipGetLastError();
}

hipError_t launch_layernorm_rows(float* x, const float* w, const float* b, float eps, void* out, int M, int d, int act_bf16,
                                 hipStream_t s, const float* part, int ks, const float* pbias, float* row_mean) {
    if (!x || !w || !b || !out || M <= 0 || d <= 0 || d > 256 * kNormMaxPer || (d & 3) || (ks > 0 && !part)) return hipErrorInvalidValue;
    if (act_bf16)
        hipLaunchKernelGGL(layernorm_rows_kernel<uint16_t>, dim3(M), dim3(256), 0, s, x, w, b, eps, static_cast<uint16_t*>(out), d,
                           part, ks, pbias, M, row_mean);
    else
        hipLaunchKernelGGL(layernorm_rows_kernel<float>, dim3(M), dim3(256), 0, s, x, w, b, eps, static_cast<float*>(out), d, part,
                           ks, pbias, M, row_mean);
    return hipGetLastError();
}

hipError_t launch_fold_partials(float* x, const float* part, int ks, const float* pbias, int M, int d, hipStream_t s) {
    if (!x || !part || ks < 1 || M <= 0 || d <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fold_partials_kernel, dim3(M, (d + 255) / 256), dim3(256), 0, s, x, x, part, ks, pbias, M, d, d);
    return hipGetLastError();
}

hipError_t launch_attn_rows(const AttnRowsArgs& a, hipStream_t s) {
    if (!a.q || !a.kcache || !a.vcache || !a.posmap || !a.out || a.M <= 0 || a.n_heads <= 0 || a.max_len <= 0)
        return hipErrorInvalidValue;
    if ((long long)a.M * a.n_heads > 0x7FFFFFFFLL || (a.dot_vec != nullptr) != (a.dot_out != nullptr)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)a.M * (unsigned)a.n_heads);
    // Waves per (item, head) block by the number of such blocks: 1 from 512 blocks, 2 from 256, else 4.  With one wave per
    // block all 2048 blocks of a 128-row step are resident at once (8 per CU) and start and end at different times; with
    // 4-wave blocks they ran as two rounds in lockstep, and HBM idled through both rounds' prologues and merges.
    // Measured, 400-step bf16 sessions, ms per step with 4 / 2 / 1 waves: 64 utterances (2048 blocks) 1.852 / 1.807 /
    // 1.802; 16 utterances (512) 1.114 / 1.090 / 1.070; 512 utterances 7.909 / 7.851 / 7.827; 4 utterances (128 blocks)
    // 0.943 / 0.940 / 0.979; Turbo 16 x L 407..707 (256 blocks) 0.877 / 0.864 / 0.933; Turbo 8 (128) 0.823 / 0.840 / 0.919;
    // fp32 cache, 64 utterances 4.279 / 4.374 / 4.110.
    const long long blocks = (long long)a.M * a.n_heads;
    static const int nw8_below = [] { const char* e = std::getenv("T3MI_ATTN_NW8"); return e ? std::atoi(e) : 0; }();   // experiment: 8 waves below this block count
    const int nw = blocks >= 512 ? 1 : (blocks >= 256 ? 2 : (blocks < nw8_below ? 8 : 4));
    if (nw == 8) {
        const dim3 block(512);
        if (a.kv_bf16 && a.out_bf16) hipLaunchKernelGGL((attn_rows_kernel<uint16_t, uint16_t, 8>), grid, block, 0, s, a);
        else if (a.kv_bf16) hipLaunchKernelGGL((attn_rows_kernel<uint16_t, float, 8>), grid, block, 0, s, a);
        else if (a.out_bf16) hipLaunchKernelGGL((attn_rows_kernel<float, uint16_t, 8>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((attn_rows_kernel<float, float, 8>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    if (nw == 1) {
        const dim3 block(64);
        if (a.kv_bf16 && a.out_bf16) hipLaunchKernelGGL((attn_rows_kernel<uint16_t, uint16_t, 1>), grid, block, 0, s, a);
        else if (a.kv_bf16) hipLaunchKernelGGL((attn_rows_kernel<uint16_t, float, 1>), grid, block, 0, s, a);
        else if (a.out_bf16) hipLaunchKernelGGL((attn_rows_kernel<float, uint16_t, 1>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((attn_rows_kernel<float, float, 1>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    if (nw == 2) {
        const dim3 block(128);
        if (a.kv_bf16 && a.out_bf16) hipLaunchKernelGGL((attn_rows_kernel<uint16_t, uint16_t, 2>), grid, block, 0, s, a);
        else if (a.kv_bf16) hipLaunchKernelGGL((attn_rows_kernel<uint16_t, float, 2>), grid, block, 0, s, a);
        else if (a.out_bf16) hipLaunchKernelGGL((attn_rows_kernel<float, uint16_t, 2>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((attn_rows_kernel<float, float, 2>), grid, block, 0, s, a);
        return hipGetLastError();
    }
    const dim3 block(256);
#ifdef T3K_DIAG_ATTN_SPLIT
    {
        static const int ds = [] { const char* e = std::getenv("T3MI_DIAG_ATTN_SPLIT"); return e ? std::atoi(e) : 0; }();
        if (ds > 1 && a.kv_bf16 && a.out_bf16) {
            AttnRowsArgs b = a;
            b.diag_split = ds;
            hipLaunchKernelGGL((attn_rows_kernel<uint16_t, uint16_t, 4>), dim3(grid.x * ds), block, 0, s, b);
            return hipGetLastError();
        }
    }
#endif
    if (a.pf.n_blocks > 0 && a.pf.sink != nullptr && (grid.x & 7) == 0 && a.kv_bf16 && a.out_bf16) {
        // few (item, head) blocks (this branch: fewer than 256): further grid rows prefetch the next launches' weights (PfArgs)
        AttnRowsArgs b = a;
        const unsigned rows = ((unsigned)a.pf.n_blocks + grid.x - 1) / grid.x;
        b.pf.n_blocks = (int)(rows * grid.x);
        if (b.pf.n_blocks <= kPfMaxBlocks) {
            hipLaunchKernelGGL((attn_rows_kernel<uint16_t, uint16_t, 4, true>), dim3(grid.x, 1 + rows), block, 0, s, b);
            return hipGetLastError();
        }
    }
    if (a.kv_bf16 && a.out_bf16) hipLaunchKernelGGL((attn_rows_kernel<uint16_t, uint16_t, 4>), grid, block, 0, s, a);
    else if (a.kv_bf16) hipLaunchKernelGGL((attn_rows_kernel<uint16_t, float, 4>), grid, block, 0, s, a);
    else if (a.out_bf16) hipLaunchKernelGGL((attn_rows_kernel<float, uint16_t, 4>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((attn_rows_kernel<float, float, 4>), grid, block, 0, s, a);
    return hipGetLastError();
}

hipError_t launch_gather_rows(const float* const* src_rows, float* x, int M, int d, hipStream_t s) {
    if (!src_rows || !x || M <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(M), dim3(256), 0, s, src_rows, x, d);
    return hipGetLastError();
}

}  // namespace t3k
