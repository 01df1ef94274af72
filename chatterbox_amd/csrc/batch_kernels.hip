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

namespace t3k {

namespace {

constexpr int kHeadDim = 64;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint16_t f2bf(float f) {
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

template <typename AT> struct Act;
template <> struct Act<float> {
    __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct Act<uint16_t> {
    __device__ static __forceinline__ void store(uint16_t* p, float v) { *p = f2bf(v); }
};

// fragment loaders / MFMA steps per weight type: one "k-block" = 64 bytes of a row
struct GemmF32 {
    using AT = float;
    static constexpr int KB = 16;  // k elements per k-block
    using Frag = float4;
    __device__ static __forceinline__ Frag load(const void* base, size_t elem) {
        return *reinterpret_cast<const float4*>(static_cast<const float*>(base) + elem);
    }
    __device__ static __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
        return c;
    }
};
struct GemmBF16 {
    using AT = uint16_t;
    static constexpr int KB = 32;
    using Frag = bf16x8;
    __device__ static __forceinline__ Frag load(const void* base, size_t elem) {
        return *reinterpret_cast<const bf16x8*>(static_cast<const uint16_t*>(base) + elem);
    }
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
            bs[u] = G::load(a.W, wbase + (size_t)j * KB);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) as[u][mt] = G::load(a.A, abase[mt] + (size_t)j * KB);
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
    load_stage(0, b0, a0);
    for (int j0 = 0; j0 < nkb; j0 += 2 * UN) {
        if (j0 + UN < nkb) load_stage(j0 + UN, b1, a1);
        compute_stage(j0, b0, a0);
        if (j0 + 2 * UN < nkb) load_stage(j0 + 2 * UN, b0, a0);
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
                float* cb = (kind == 1 ? a.kcache : a.vcache) + (((size_t)crow * H + head) * a.max_len + pos) * kHeadDim;
                cb[i] = lo; cb[i + 32] = hi;
            }
        }
    }
}

// ---- RMSNorm over rows: out[m] = w * (x[m] * rsqrt(mean(x[m]^2) + eps))  (R/llama.R:65-72) -------------
template <typename AT>
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float eps, AT* __restrict__ out, int d) {
    __shared__ float sred[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const float* xr = x + (size_t)m * d;
    float ss = 0.f;
    for (int i = tid; i < d; i += 256) ss = fmaf(xr[i], xr[i], ss);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if ((tid & 63) == 0) sred[tid >> 6] = ss;
    __syncthreads();
    const float tot = (sred[0] + sred[1]) + (sred[2] + sred[3]);
    const float rs = 1.0f / sqrtf(tot / (float)d + eps);
    for (int i = tid; i < d; i += 256) Act<AT>::store(out + (size_t)m * d + i, __fmul_rn(w[i], __fmul_rn(xr[i], rs)));
}

// ---- LayerNorm over rows (Turbo, R/t3.R:1200-1203): two-pass mean / biased variance -------------------------
template <typename AT>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, float eps, AT* __restrict__ out, int d) {
    __shared__ float sred[8];
    const int m = blockIdx.x, tid = threadIdx.x;
    const float* xr = x + (size_t)m * d;
    float sm = 0.f;
    for (int i = tid; i < d; i += 256) sm += xr[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
    if ((tid & 63) == 0) sred[tid >> 6] = sm;
    __syncthreads();
    const float mean = ((sred[0] + sred[1]) + (sred[2] + sred[3])) / (float)d;
    float ss = 0.f;
    for (int i = tid; i < d; i += 256) {
        const float c = xr[i] - mean;
        ss = fmaf(c, c, ss);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if ((tid & 63) == 0) sred[4 + (tid >> 6)] = ss;
    __syncthreads();
    const float var = ((sred[4] + sred[5]) + (sred[6] + sred[7])) / (float)d;
    const float rs = 1.0f / sqrtf(var + eps);
    for (int i = tid; i < d; i += 256)
        Act<AT>::store(out + (size_t)m * d + i, __fadd_rn(__fmul_rn(__fmul_rn(xr[i] - mean, rs), w[i]), b[i]));
}

// ---- merge of the split-softmax attention partials into the o_proj activation rows ----------------------
template <typename AT>
__global__ __launch_bounds__(256) void attn_merge_kernel(const float* __restrict__ pm, const float* __restrict__ pl,
                                                         const float* __restrict__ pacc, const int* __restrict__ posmap,
                                                         AT* __restrict__ out, int H, int n_split_max, int kps) {
    const int qi = blockIdx.x;
    const int pos = posmap[qi];
    const int d = H * kHeadDim;
    for (int i = threadIdx.x; i < d; i += 256) {
        float o = 0.f;
        if (pos >= 0) {
            const int head = i >> 6, dd = i & 63;
            const int ns = pos / kps + 1;
            const size_t b = ((size_t)qi * H + head) * n_split_max;
            float M = -INFINITY;
            for (int s = 0; s < ns; ++s) M = fmaxf(M, pm[b + s]);
            float den = 0.f;
            for (int s = 0; s < ns; ++s) {
                const float ws = expf(pm[b + s] - M);
                den = fmaf(ws, pl[b + s], den);
                o = fmaf(ws, pacc[(b + s) * kHeadDim + dd], o);
            }
            o /= den;
        }
        Act<AT>::store(out + (size_t)qi * d + i, o);
    }
}

// ---- x[m] = embeds of item m (prefill: flat list of (row, position) items) --------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* const* __restrict__ src_rows, float* __restrict__ x, int d) {
    const int m = blockIdx.x;
    const float* s = src_rows[m];
    for (int i = threadIdx.x; i < d; i += 256) x[(size_t)m * d + i] = s[i];
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
    return wdtype == 1 ? launch_gemm_g<GemmBF16>(a, epilogue, s) : launch_gemm_g<GemmF32>(a, epilogue, s);
}

hipError_t launch_rmsnorm_rows(const float* x, const float* w, float eps, void* out, int M, int d, int act_bf16,
                               hipStream_t s) {
    if (!x || !w || !out || M <= 0 || d <= 0) return hipErrorInvalidValue;
    if (act_bf16)
        hipLaunchKernelGGL(rmsnorm_rows_kernel<uint16_t>, dim3(M), dim3(256), 0, s, x, w, eps, static_cast<uint16_t*>(out), d);
    else
        hipLaunchKernelGGL(rmsnorm_rows_kernel<float>, dim3(M), dim3(256), 0, s, x, w, eps, static_cast<float*>(out), d);
    return hipGetLastError();
}

hipError_t launch_layernorm_rows(const float* x, const float* w, const float* b, float eps, void* out, int M, int d,
                                 int act_bf16, hipStream_t s) {
    if (!x || !w || !b || !out || M <= 0 || d <= 0) return hipErrorInvalidValue;
    if (act_bf16)
        hipLaunchKernelGGL(layernorm_rows_kernel<uint16_t>, dim3(M), dim3(256), 0, s, x, w, b, eps, static_cast<uint16_t*>(out), d);
    else
        hipLaunchKernelGGL(layernorm_rows_kernel<float>, dim3(M), dim3(256), 0, s, x, w, b, eps, static_cast<float*>(out), d);
    return hipGetLastError();
}

hipError_t launch_attn_merge(const float* pm, const float* pl, const float* pacc, const int* posmap, void* out, int M,
                             int H, int n_split_max, int kps, int act_bf16, hipStream_t s) {
    if (!pm || !pl || !pacc || !posmap || !out || M <= 0) return hipErrorInvalidValue;
    if (act_bf16)
        hipLaunchKernelGGL(attn_merge_kernel<uint16_t>, dim3(M), dim3(256), 0, s, pm, pl, pacc, posmap,
                           static_cast<uint16_t*>(out), H, n_split_max, kps);
    else
        hipLaunchKernelGGL(attn_merge_kernel<float>, dim3(M), dim3(256), 0, s, pm, pl, pacc, posmap,
                           static_cast<float*>(out), H, n_split_max, kps);
    return hipGetLastError();
}

hipError_t launch_gather_rows(const float* const* src_rows, float* x, int M, int d, hipStream_t s) {
    if (!src_rows || !x || M <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(M), dim3(256), 0, s, src_rows, x, d);
    return hipGetLastError();
}

}  // namespace t3k
