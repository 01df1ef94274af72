// Dense fp32 GEMM for S3Gen: the linears of basic_transformer_block (R/s3gen.R:246-372), the causal
// convolutions as GEMMs over window rows (R/s3gen.R:96-215), the conformer's linears (R/conformer.R:200-403) and the
// vocoder's convolutions (R/hifigan.R:82-150, 529-599):  y[M][N] = epi(LN?(A[M][K]) . W[N][K]^T + b).
//
// The decoder's many-row kernels (batch_kernels.hip) are built for a weight STREAM from HBM shared by <= 128 rows; here the
// operands are a megabyte each and live in L2, M is 10^2 .. 10^4 rows and a GEMM is 0.3 - 0.8 GFLOP, i.e. 2 - 5 us of the
// exact-fp32 matrix-core time (v_mfma_f32_16x16x4_f32, 256 FLOP / clk / CU).  What decides is how many CUs work and how
// little each block waits, so: small blocks (4 waves, a BM x BN tile of 32 x 32 or 32 x 64: 256 - 1536 blocks at
// 1000 rows), 35 - 52 KB of LDS so that three to four blocks share a CU and cover each other's staging, 64-k chunks of both
// operands double-buffered in LDS with ONE barrier per chunk, the next chunks' global loads in flight under the current
// chunk's MFMAs in a register ring (no load under a condition: exact wait counts).  Rows of both tiles sit at a stride of 68 floats (272 B), the
// 16 rows of a fragment read fall into distinct banks.
//
// MFMA operand roles: the WEIGHT rows are the MFMA's row operand and the activation rows its column operand, so a lane ends
// up with 4 consecutive FEATURES of one activation row: bias / residual / store are 16-byte accesses.  Lane (n, g) feeds
// element k = 16 j + 4 g + i of its row into MFMA i of k-group j -- a fixed permutation of the dot product's terms,
// the same for both operands.
//
// LN: a LayerNorm over the K columns of every A row in front of the linear layer (K = the row width; basic_transformer_block's
// norm1 / norm3, R/s3gen.R:359-372; the conformer's norm_mha / norm_ff) has no launch of its own and costs the multiply loop
// nothing: with W' = W diag(w_ln), s = W' 1 and b' = b + W b_ln (all three made once, when the checkpoint is loaded),
//     LN(x) W^T + b = rstd (W' x - mean s) + b',
// so the GEMM runs on the raw rows and the epilogue applies the row's mean and rstd.  The statistics (two-pass: mean, then the
// centred squares) come from the staging registers when the whole K range is in the ring (K <= 64 D), else from a second read
// of the rows.  A first form normalised the rows while staging them: 32 more VALU instructions and two LDS reads per chunk
// and thread made every such GEMM 25 - 30 % slower (scripts/micro/dense_micro.hip: ff1 8.3 -> 10.7 us at 1000 rows).
#include "cfm_kernels.hpp"
#include "kernels.hpp"

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

namespace t3k {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kKC = 64;          // k per chunk
constexpr int kLdT = kKC + 4;    // LDS row stride in floats

// sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), every lane gets it: VALU moves, no LDS round trip
template <int CTRL> __device__ __forceinline__ float dg_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dg_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dg_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dg_dpp<0x141>(v);   // row_half_mirror: the other quad of the 8
    v += dg_dpp<0x140>(v);   // row_mirror: the other half of the 16
    return v;
}

template <class F, int... S>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, S...>) { (f(std::integral_constant<int, S>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int BM, int BN, int EPI, bool LN, int D>
__global__ __launch_bounds__(256) void dense_gemm_kernel(DenseGemmArgs a) {
    constexpr int MT = BM / 32, NT = BN / 32;        // 16-row / 16-feature tiles per wave (waves sit 2 x 2)
    constexpr int NLD = (BM + BN) / 16;              // staging pieces (16 B) per thread and chunk
    constexpr int NLA = BM / 16;                     // of which activation rows
    extern __shared__ __attribute__((aligned(16))) float dg_lds[];   // [2][(BM + BN)][kLdT], then (LN) mean | rstd of the BM rows
    float* stat = dg_lds + 2 * (BM + BN) * kLdT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = lane & 15, g = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int K = a.K, nc = K / kKC;
    const int sr = tid >> 4, sc = (tid & 15) * 4;    // staging: thread -> (row sr + 16 i, columns sc .. sc + 3)

    // global source of staging piece i: activation rows first, then weight rows (rows past M repeat row M - 1: never stored)
    const float* src[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        if (i < NLA) {
            int row = m0 + sr + 16 * i;
            row = row < a.M ? row : a.M - 1;
            src[i] = a.A + (size_t)row * a.lda + sc;
        } else {
            src[i] = a.W + (size_t)(n0 + sr + 16 * (i - NLA)) * K + sc;
        }
    }
    // Register ring of D chunks: slot s holds a chunk = s (mod D).  A block has 1 - 2 MFLOP per 64-k chunk (0.2 - 0.4 us of
    // matrix-core time) against 0.7 - 2 us of load latency (the activations were written by another XCD: they come over the
    // fabric), so a prefetch distance of one chunk leaves the block waiting on every chunk; D chunks in flight cover it.
    // No load sits under a condition (the loop is split into a steady part that refills the ring and a tail of D chunks that
    // does not): the wait counts stay exact.
    f32x4 st[D][NLD];
    auto fetch_part = [&](int c, auto SC, auto LO, auto HI) {
        constexpr int slot = decltype(SC)::value;
#pragma unroll
        for (int i = decltype(LO)::value; i < decltype(HI)::value; ++i) st[slot][i] = *reinterpret_cast<const f32x4*>(src[i] + c * kKC);
    };
    using std::integral_constant;
    auto fetch = [&](int c, auto SC) { fetch_part(c, SC, integral_constant<int, 0>{}, integral_constant<int, NLD>{}); };
    auto stash = [&](int buf, auto SC) {
        constexpr int slot = decltype(SC)::value;
        float* d = dg_lds + (size_t)buf * (BM + BN) * kLdT + sr * kLdT + sc;
#pragma unroll
        for (int i = 0; i < NLD; ++i) *reinterpret_cast<f32x4*>(d + i * 16 * kLdT) = st[slot][i];
    };
    // prologue: chunks 0 .. D - 1 into the ring
    static_for<D>([&](auto SC) { fetch(decltype(SC)::value, SC); });
    if constexpr (LN) {
        // row statistics for the epilogue: the 16 threads that stage a row (tid & 15, one DPP row) hold all of it when the whole
        // K range is in the ring; otherwise they read it again (L2-resident), twice
        const float invK = 1.0f / (float)K;
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            float sum = 0.f, sq = 0.f, mean;
            if (nc <= D) {   // (block-uniform; nc % D == 0: every slot holds a distinct chunk)
#pragma unroll
                for (int s = 0; s < D; ++s) sum += (st[s][i].x + st[s][i].y) + (st[s][i].z + st[s][i].w);
                mean = row16_sum(sum) * invK;
#pragma unroll
                for (int s = 0; s < D; ++s) {
                    const f32x4 dl = st[s][i] - mean;
                    sq += (dl.x * dl.x + dl.y * dl.y) + (dl.z * dl.z + dl.w * dl.w);
                }
            } else {
                for (int c = 0; c < nc; ++c) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(src[i] + c * kKC);
                    sum += (v.x + v.y) + (v.z + v.w);
                }
                mean = row16_sum(sum) * invK;
                for (int c = 0; c < nc; ++c) {
                    const f32x4 dl = *reinterpret_cast<const f32x4*>(src[i] + c * kKC) - mean;
                    sq += (dl.x * dl.x + dl.y * dl.y) + (dl.z * dl.z + dl.w * dl.w);
                }
            }
            sq = row16_sum(sq);
            if ((tid & 15) == 0) {   // read after the loop's barriers
                stat[sr + 16 * i] = mean;
                stat[BM + sr + 16 * i] = 1.0f / sqrtf(sq * invK + a.ln_eps);
            }
        }
    }
    stash(0, integral_constant<int, 0>{});
    __syncthreads();

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fragment bases inside a buffer: activation tile rows wm * BM/2 + 16 t + nl, weight tile rows BM + wn * BN/2 + 16 t + nl
    const int fa = (wm * (BM / 2) + nl) * kLdT + 4 * g;
    const int fw = (BM + wn * (BN / 2) + nl) * kLdT + 4 * g;
    // chunk cc (ring slot d = cc % D, already in LDS buffer cc & 1): refill slot d with chunk cc + D, multiply, move chunk cc + 1
    // (slot (d + 1) % D) into the other buffer; one barrier per chunk
    auto step = [&](int cc, auto DC, auto REFILL, auto LAST) {
        constexpr int d = decltype(DC)::value;
        constexpr int nslot = (d + 1) % D;
        if constexpr (decltype(REFILL)::value) fetch(cc + D, integral_constant<int, d>{});
        const float* b = dg_lds + (size_t)(cc & 1) * (BM + BN) * kLdT;
#pragma unroll
        for (int j = 0; j < kKC / 16; ++j) {
            f32x4 av[MT], wv[NT];
#pragma unroll
            for (int t = 0; t < MT; ++t) av[t] = *reinterpret_cast<const f32x4*>(b + fa + t * 16 * kLdT + 16 * j);
#pragma unroll
            for (int t = 0; t < NT; ++t) wv[t] = *reinterpret_cast<const f32x4*>(b + fw + t * 16 * kLdT + 16 * j);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int tm = 0; tm < MT; ++tm)
#pragma unroll
                    for (int tn = 0; tn < NT; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[tn][i], av[tm][i], acc[tm][tn], 0, 0, 0);
        }
        if constexpr (!decltype(LAST)::value) {
            stash((cc + 1) & 1, integral_constant<int, nslot>{});
            __syncthreads();
        }
    };
    using std::true_type;
    using std::false_type;
    int c = 0;
    for (; c + D < nc; c += D)   // nc % D == 0 (the launcher picks D)
        static_for<D>([&](auto DC) { step(c + decltype(DC)::value, DC, true_type{}, false_type{}); });
    static_for<D>([&](auto DC) {
        step(c + decltype(DC)::value, DC, false_type{}, integral_constant<bool, decltype(DC)::value == D - 1>{});
    });
    // epilogue: lane (nl, g) holds features f .. f + 3 (f = 16 tile + 4 g) of activation row nl of its tiles
#pragma unroll
    for (int tm = 0; tm < MT; ++tm) {
        const int row = m0 + wm * (BM / 2) + 16 * tm + nl;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) {
            const int f = n0 + wn * (BN / 2) + 16 * tn + 4 * g;
            f32x4 v = acc[tm][tn];
            if constexpr (LN) {   // rstd (W' x - mean s); the bias below is b' = b + W b_ln
                const int rl = wm * (BM / 2) + 16 * tm + nl;
                const float mean = stat[rl], rstd = stat[BM + rl];
                const f32x4 cs = *reinterpret_cast<const f32x4*>(a.ln_colsum + f);
                v.x = __fmul_rn(rstd, __fsub_rn(v.x, __fmul_rn(mean, cs.x))); v.y = __fmul_rn(rstd, __fsub_rn(v.y, __fmul_rn(mean, cs.y)));
                v.z = __fmul_rn(rstd, __fsub_rn(v.z, __fmul_rn(mean, cs.z))); v.w = __fmul_rn(rstd, __fsub_rn(v.w, __fmul_rn(mean, cs.w)));
            }
            if (a.bias != nullptr) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + f);
                v.x = __fadd_rn(v.x, bv.x); v.y = __fadd_rn(v.y, bv.y); v.z = __fadd_rn(v.z, bv.z); v.w = __fadd_rn(v.w, bv.w);
            }
            if (row < a.M) {
                float* yp = a.y + (size_t)row * a.ldy + f;
                if constexpr (EPI == EPI_RESID) {
                    const f32x4 r = *reinterpret_cast<const f32x4*>(a.resid + (size_t)row * a.ldy + f);
                    v.x = __fadd_rn(r.x, v.x); v.y = __fadd_rn(r.y, v.y); v.z = __fadd_rn(r.z, v.z); v.w = __fadd_rn(r.w, v.w);
                }
                if constexpr (EPI == EPI_GELU) {   // gelu_tanh (R/s3gen.R:308), the form of batch_kernels.hip's epilogue
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = v[e];
                        const float inner = 0.7978845608028654f * (x + 0.044715f * x * x * x);
                        v[e] = 0.5f * x * (1.0f + tanhf(inner));
                    }
                }
                *reinterpret_cast<f32x4*>(yp) = v;
            }
        }
    }
}

template <int BM, int BN, int EPI, bool LN, int D>
hipError_t launch_dense_d(const DenseGemmArgs& a, hipStream_t s) {
    constexpr int lds = (2 * (BM + BN) * kLdT + (LN ? 2 * BM : 0)) * 4;
    static bool attr_done[64] = {};   // one flag per instantiation and device
    int dev = 0;
    hipError_t ed = hipGetDevice(&dev);
    if (ed != hipSuccess) return ed;
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(dense_gemm_kernel<BM, BN, EPI, LN, D>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e0 != hipSuccess) return e0;
        if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
    hipLaunchKernelGGL((dense_gemm_kernel<BM, BN, EPI, LN, D>), dim3((a.M + BM - 1) / BM, a.N / BN), dim3(256), lds, s, a);
    return hipGetLastError();
}

// Ring depth: a divisor of the chunk count.  Measured (dense_micro, 200 - 4000 rows): 2 chunks in flight are as good as 4 or 8
// and a depth of 1 costs 10 - 25 %; with a LayerNorm the depth that holds the whole row (4 for K = 256, 8 for K = 512) saves the
// second read of the rows.
template <int BM, int BN, int EPI, bool LN>
hipError_t launch_dense_v(const DenseGemmArgs& a, hipStream_t s) {
    const int nc = a.K / kKC;
    if constexpr (LN) {
        if constexpr (BM + BN <= 64) {
            if (nc == 8) return launch_dense_d<BM, BN, EPI, LN, 8>(a, s);
        }
        if (nc == 4) return launch_dense_d<BM, BN, EPI, LN, 4>(a, s);
    }
    if (nc % 2 == 0) return launch_dense_d<BM, BN, EPI, LN, 2>(a, s);
    if (nc % 3 == 0) return launch_dense_d<BM, BN, EPI, LN, 3>(a, s);
    if (nc % 5 == 0) return launch_dense_d<BM, BN, EPI, LN, 5>(a, s);
    return launch_dense_d<BM, BN, EPI, LN, 1>(a, s);
}

template <int EPI, bool LN>
hipError_t launch_dense_e(const DenseGemmArgs& a, hipStream_t s) {
    // Tile: 32 x 32 (one accumulator tile per wave) is the fastest or tied on every estimator shape from 200 to 4000 rows except
    // narrow outputs with many rows (N = 256 from 2000 rows: 32 x 64 saves 1 - 1.5 us); 64 x 64 was never the fastest
    // (profiles/r03_dense_gemm_micro.txt).
    if (a.N % 64 == 0 && a.N <= 512 && a.M >= 1536) return launch_dense_v<32, 64, EPI, LN>(a, s);
    return launch_dense_v<32, 32, EPI, LN>(a, s);
}

}  // namespace

bool dense_gemm_supported(int M, int N, int K, int epi) {
    return M > 0 && N > 0 && K > 0 && N % 32 == 0 && K % kKC == 0 && (epi == EPI_STORE || epi == EPI_RESID || epi == EPI_GELU);
}

void dense_fold_layernorm(float* W, const float* bias, const float* ln_w, const float* ln_b, int N, int K, float* bias_out, float* colsum_out) {
    for (int n = 0; n < N; ++n) {
        float* w = W + (size_t)n * K;
        double sb = 0.0, sw = 0.0;
        for (int k = 0; k < K; ++k) {
            sb += (double)w[k] * (double)ln_b[k];
            w[k] = w[k] * ln_w[k];
            sw += (double)w[k];
        }
        bias_out[n] = (float)((bias ? (double)bias[n] : 0.0) + sb);
        colsum_out[n] = (float)sw;
    }
}

hipError_t launch_dense_gemm(const DenseGemmArgs& a, hipStream_t s) {
    if (!dense_gemm_supported(a.M, a.N, a.K, a.epi) || !a.A || !a.W || !a.y || a.lda < a.K || a.ldy < a.N) return hipErrorInvalidValue;
    if ((a.lda & 3) || (a.ldy & 3)) return hipErrorInvalidValue;           // 16-byte accesses
    if (a.epi == EPI_RESID && !a.resid) return hipErrorInvalidValue;
    const bool ln = a.ln_colsum != nullptr;
    switch (a.epi) {
        case EPI_STORE: return ln ? launch_dense_e<EPI_STORE, true>(a, s) : launch_dense_e<EPI_STORE, false>(a, s);
        case EPI_RESID: return ln ? launch_dense_e<EPI_RESID, true>(a, s) : launch_dense_e<EPI_RESID, false>(a, s);
        default: return ln ? launch_dense_e<EPI_GELU, true>(a, s) : launch_dense_e<EPI_GELU, false>(a, s);
    }
}

}  // namespace t3k
