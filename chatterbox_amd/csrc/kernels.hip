// gfx950 (MI355X, CDNA4) kernels of the T3 decode hot path.  wave = 64 lanes.
//
// Decode at 1-4 activation rows is a pure weight stream (2.05 GB fp32 per step):
// every kernel here is HBM-bound and is shaped for memory-level parallelism, not FLOPs --
// weights go straight from HBM to VGPRs as 16-byte loads, one 1-KiB contiguous piece per
// wave-instruction, all of a wave's loads issued before the first use; no LDS staging for
// operands that are read once (cdna guide, "GEMV / M <= 16" row).  The activation rows share
// one pass over the weights (the CFG cond/uncond pair costs one weight read, not two).
//
// Summation orders are fixed (no atomics), so results are bit-reproducible run to run.
#include "kernels.hpp"

#include <hip/hip_runtime.h>

#include <cstdint>

namespace t3k {

namespace {

constexpr int kWave = 64;
constexpr int kHeadDim = 64;

// Wave-wide reductions on the DPP / readlane path: 4 row-local DPP steps (~8 cycles each) + 4
// v_readlane, instead of 6 dependent ds_bpermute round trips through the LDS crossbar (~100 cycles
// each) -- the reductions sit on the critical path of every latency-bound decode kernel.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);   // row_half_mirror
    v += dpp_f<0x140>(v);   // row_mirror: every lane of a 16-lane row now holds the row sum
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// sum within each aligned group of 16 lanes (all 16 lanes get the result)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    return v;
}

#ifdef T3K_STAMPS
// diagnostic build: 100 MHz constant clock, comparable across CUs
#define T3K_STAMP(args, gw, slot)                                                                  \
    do {                                                                                           \
        if ((args).stamps != nullptr && (threadIdx.x & 63) == 0 && (gw) < T3K_STAMP_WAVES)         \
            (args).stamps[(size_t)(gw) * T3K_STAMP_SLOTS + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define T3K_STAMP(args, gw, slot) do { } while (0)
#endif

// ---- weight element types --------------------------------------------------------------
struct F32W {
    static constexpr int E = 4;  // elements per 16-byte load
    __device__ static __forceinline__ void load(const void* base, size_t elem_off, float (&o)[4]) {
        // read-once weight stream: non-temporal (does not displace the activations / KV in L2)
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(static_cast<const float*>(base) + elem_off));
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
};
struct BF16W {
    static constexpr int E = 8;
    __device__ static __forceinline__ void load(const void* base, size_t elem_off, float (&o)[8]) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(static_cast<const uint16_t*>(base) + elem_off));
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xFFFF0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xFFFF0000u);
        o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xFFFF0000u);
        o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xFFFF0000u);
    }
};

// ---- GEMV: y[r][n] = sum_k W[n][k] * xs[r][k] ---------------------------------------------------
//
// Block = NW waves.  The block first issues the weight loads of its first chunk (longest latency
// first), then builds the activation rows xs[R][K] ONCE in LDS through one of three prologues:
//   PRO_PLAIN : copy x                                   (down_proj: x = SwiGLU output)
//   PRO_NORM  : RMSNorm, w * (x * rsqrt(mean(x^2)+eps))  (R/llama.R:65-72)
//   PRO_ATTN  : combine the split-softmax partials of the decode attention kernel
// A wave owns a chunk of ROWS output features over the whole K (KSPLIT == 1) or one quarter of K
// (KSPLIT == 4: the 4 waves of a group combine through LDS in a fixed order).  Lane l holds
// k = q*64*E + l*E .. +E-1 of every row, so one wave-instruction reads 1 KiB contiguous.
// Which features form a chunk depends on the epilogue:
//   STORE/RESID : n = ROWS*c + j
//   ROPE_KV     : both halves (i, i+32) of rotate_half pairs of one head of q, k or v
//   SWIGLU      : gate rows and the matching up rows (F = N/2 apart)
template <int EPI, int ROWS>
__device__ __forceinline__ int chunk_row(int c, int j, int N) {
    constexpr int HP = ROWS / 2;  // pairs per chunk (ROPE / SWIGLU)
    if (EPI == EPI_ROPE_KV) {
        const int per_head = 32 / HP;                 // chunks per 64-row head matrix
        const int hm = c / per_head, i0 = (c % per_head) * HP;
        return hm * kHeadDim + i0 + (j % HP) + (j / HP) * 32;
    } else if (EPI == EPI_SWIGLU) {
        return (j / HP) * (N >> 1) + HP * c + (j % HP);
    } else {
        return ROWS * c + j;
    }
}

template <typename WT, int R, int NK, int KSPLIT, int ROWS, int EPI, int PRO>
// The operands every wave needs before it can issue its first load come first in the signature as plain scalars, so
// that a build with -mllvm -amdgpu-kernarg-preload-count (Makefile: PRELOAD) receives them in SGPRs at wave launch
// instead of through a kernarg round trip (0.4-0.5 us in the stamp timeline); on the replayed-graph bench the two
// builds measure the same.
__global__ __launch_bounds__(512) void gemv_kernel(const void* __restrict__ pW, const float* __restrict__ px, const float* __restrict__ pln_w,
                                                   const float* presid, const int* __restrict__ prowpos,
                                                   const float* __restrict__ prope_cur, int pN, int n_chunks, GemvArgs a) {
    constexpr int E = WT::E;
    constexpr int KW = NK * kWave * E;  // K elements handled by one wave
    constexpr int K = KW * KSPLIT;
    static_assert(EPI == EPI_STORE || EPI == EPI_RESID || EPI == EPI_GELU || (ROWS % 2 == 0), "paired epilogues need even ROWS");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                                          // [R][K], PRO_NORM / PRO_ATTN
    float* part = smem + (PRO != PRO_PLAIN ? R * K : 0);       // [NW][max(ROWS,2)][R] scratch
    // block size from the kernel arguments (scalar loads), not blockDim: the compiler fetched the byte of blockDim it needed
    // with a VECTOR load from the implicit arguments and waited for it before issuing anything else (KSPLIT = 4)
    const int tid = threadIdx.x, nthr = a.waves * 64;
    const int lane = tid & 63, wave = tid >> 6, nw = a.waves;
    const int grp = wave / KSPLIT, kpart = wave % KSPLIT, ngrp = nw / KSPLIT;
    const int kbase = kpart * KW;

    // block-uniform trip count: every wave runs the same number of iterations (barriers inside)
    int cb = (int)blockIdx.x * ngrp;
    const int cstride = (int)gridDim.x * ngrp;
    int c = cb + grp;

    [[maybe_unused]] const int gwave = (int)blockIdx.x * nw + wave;
    T3K_STAMP(a, gwave, 0);

    // ---- 1. small L2-resident operands FIRST.  Vector-memory results return in issue order, so anything
    // requested after the weight stream only becomes usable once the weights have landed; requested
    // before it, the norm / softmax-merge math overlaps the HBM flight of the weights.
    // 1a. epilogue operands that do not depend on the GEMV (cache slot, RoPE row, residual)
    int e_pos = -1;
    float e_c = 0.f, e_s = 0.f, e_res = 0.f;
    {
        constexpr int HP = ROWS / 2;
        if (EPI == EPI_ROPE_KV) {
            if (lane < HP * R) {
                const int r = lane % R;
                e_pos = prowpos[r];
                if (c < n_chunks) {
                    const int per_head = 32 / HP;
                    const int i = (c % per_head) * HP + lane / R;   // rotate_half pair index 0..31
                    // rope_cur[r] = cos/sin rows of row r's CURRENT position, staged by the sampler / prefill
                    // kernel so that this load does not wait for rowpos (cos[i] == cos[i+32]: emb = cat(freqs, freqs))
                    e_c = prope_cur[(size_t)(2 * r) * kHeadDim + i];
                    e_s = prope_cur[(size_t)(2 * r + 1) * kHeadDim + i];
                }
            }
        } else if (EPI == EPI_RESID) {
            if (lane < ROWS * R && c < n_chunks && (KSPLIT == 1 || kpart == 0)) {
                const int n = ROWS * c + lane / R;
                if (n < pN) e_res = presid[(size_t)(lane % R) * a.ldy + n];
            }
        }
    }
    // 1b. activation rows: this lane's k-slice (PLAIN / NORM, one L2 round trip per wave, no barrier on the
    // critical path) or the split-softmax partials of the attention kernel (ATTN, merged once per block)
    float xv[R][NK][E];
    constexpr int NXJ = (K / 4 + 127) / 128;   // float4 slices per thread per row at >= 128 threads
    constexpr bool kNormPro = (PRO == PRO_NORM || PRO == PRO_LNORM);
    float4 gx[kNormPro ? NXJ : 1], vx[kNormPro ? R : 1][kNormPro ? NXJ : 1];
    float rs_row[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rs_row[r] = 1.0f;
    constexpr int SPC = (R <= 2) ? 8 : 4;                        // attention splits merged per pass
    constexpr int ITEMS = (R * K / 4 + 255) / 256;               // (row, float4) items per thread, 256-thread block
    float pmv[PRO == PRO_ATTN ? ITEMS : 1][SPC], plv[PRO == PRO_ATTN ? ITEMS : 1][SPC];
    float4 pav[PRO == PRO_ATTN ? ITEMS : 1][SPC];
    int a_pos[PRO == PRO_ATTN ? ITEMS : 1];
    if (PRO == PRO_ATTN) {
        constexpr int H = K / kHeadDim;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int idx = tid + it * 256;
            const bool ok = idx < R * (K / 4);
            const int r = ok ? idx / (K / 4) : 0, k4 = ok ? idx % (K / 4) : 0;
            const int head = k4 >> 4, dd = (k4 & 15) << 2;
            a_pos[it] = ok ? prowpos[r] : -1;
            const size_t b = ((size_t)r * H + head) * a.n_split_max;
#pragma unroll
            for (int sp = 0; sp < SPC; ++sp) {
                const int sc = sp < a.n_split_max ? sp : a.n_split_max - 1;   // clamp, masked below
                pmv[it][sp] = a.pm[b + sc];
                plv[it][sp] = a.pl[b + sc];
                pav[it][sp] = *reinterpret_cast<const float4*>(a.pacc + (b + sc) * kHeadDim + dd);
            }
        }
    } else if (PRO == PRO_NORM || PRO == PRO_LNORM) {
        // the block loads x and the norm weight ONCE (each thread a slice) instead of once per wave:
        // per-wave copies cost more L1 bandwidth than the weight stream itself (12 KiB per wave)
#pragma unroll
        for (int j = 0; j < NXJ; ++j) {
            const int i = (tid + j * nthr) * 4;
            if (i < K) {
#pragma unroll
                for (int r = 0; r < R; ++r) vx[r][j] = *reinterpret_cast<const float4*>(px + (size_t)r * K + i);
                gx[j] = *reinterpret_cast<const float4*>(pln_w + i);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < NK; ++q)
#pragma unroll
                for (int e4 = 0; e4 < E; e4 += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(px + (size_t)r * K + kbase + q * kWave * E + lane * E + e4);
                    xv[r][q][e4 + 0] = v.x; xv[r][q][e4 + 1] = v.y; xv[r][q][e4 + 2] = v.z; xv[r][q][e4 + 3] = v.w;
                }
    }

    // ---- 2. the weight stream: every load of the wave's first chunk, 1 KiB contiguous per instruction
    float w[ROWS][NK][E];
    auto load_chunk = [&](int cc) {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            int row = chunk_row<EPI, ROWS>(cc, j, pN);
            row = row < pN ? row : pN - 1;
            const size_t off = (size_t)row * K + kbase + lane * E;
#pragma unroll
            for (int q = 0; q < NK; ++q) WT::load(pW, off + (size_t)q * kWave * E, w[j][q]);
        }
    };
    load_chunk(c);   // unconditional: load_chunk clamps its rows (a wave without a chunk re-requests the last row), and a
                     // condition around these loads makes the compiler wait for vmcnt(0) before the prologue math
    T3K_STAMP(a, gwave, 7);   // every load of the first chunk issued (nothing waited for yet)

    // ---- 3. prologue math while the weights are in flight
#ifdef T3K_STAMPS
    if (PRO == PRO_NORM) {
        asm volatile("" ::"v"(vx[0][0].x));
        T3K_STAMP(a, gwave, 5);   // this wave's x slices landed

    } else if (PRO == PRO_PLAIN) {
        asm volatile("" ::"v"(xv[0][0][0]));
        T3K_STAMP(a, gwave, 5);
    } else if (PRO == PRO_ATTN) {
        asm volatile("" ::"v"(pmv[0][0]), "v"(pav[0][0].x));
        T3K_STAMP(a, gwave, 5);
    }
#endif
    if (PRO == PRO_NORM) {
        static_assert(PRO != PRO_NORM || KSPLIT == 1, "norm prologue needs the whole row in one wave");
        // xs[r][k] = w_ln[k] * x[r][k]; the per-row scale rsqrt(mean(x^2)+eps) is a scalar and is applied
        // to the dot products in the epilogue (R/llama.R:65-72 up to fp32 rounding order), so one barrier
        // suffices and the reduction is off the FMA critical path.
        float ssq[R];
#pragma unroll
        for (int r = 0; r < R; ++r) ssq[r] = 0.f;
#pragma unroll
        for (int j = 0; j < NXJ; ++j) {
            const int i = (tid + j * nthr) * 4;
            if (i < K) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float4 v = vx[r][j];
                    ssq[r] = fmaf(v.x, v.x, ssq[r]); ssq[r] = fmaf(v.y, v.y, ssq[r]);
                    ssq[r] = fmaf(v.z, v.z, ssq[r]); ssq[r] = fmaf(v.w, v.w, ssq[r]);
                    float4 o;
                    o.x = __fmul_rn(gx[j].x, v.x); o.y = __fmul_rn(gx[j].y, v.y);
                    o.z = __fmul_rn(gx[j].z, v.z); o.w = __fmul_rn(gx[j].w, v.w);
                    *reinterpret_cast<float4*>(xs + r * K + i) = o;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float t = wave_sum(ssq[r]);
            if (lane == 0) part[wave * R + r] = t;
        }
        __syncthreads();
        T3K_STAMP(a, gwave, 6);   // past the block barrier of the norm prologue
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float t = 0.f;
            for (int ww = 0; ww < nw; ++ww) t += part[ww * R + r];
            rs_row[r] = 1.0f / sqrtf(t / (float)K + a.eps);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < NK; ++q)
#pragma unroll
                for (int e4 = 0; e4 < E; e4 += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(xs + r * K + kbase + q * kWave * E + lane * E + e4);
                    xv[r][q][e4 + 0] = v.x; xv[r][q][e4 + 1] = v.y; xv[r][q][e4 + 2] = v.z; xv[r][q][e4 + 3] = v.w;
                }
    } else if (PRO == PRO_LNORM) {
        // LayerNorm (R/t3.R:1200-1203): xs = w_ln * (x - mean); the scalar rsqrt(var + eps) is applied to the
        // dot products in the epilogue and the LayerNorm bias is folded into `bias` at load time
        // (bias_eff[n] = bias[n] + sum_k W[n][k] * b_ln[k]).  Two-pass mean / variance.
        float sm[R];
#pragma unroll
        for (int r = 0; r < R; ++r) sm[r] = 0.f;
#pragma unroll
        for (int j = 0; j < NXJ; ++j) {
            const int i = (tid + j * nthr) * 4;
            if (i < K) {
#pragma unroll
                for (int r = 0; r < R; ++r) sm[r] += (vx[r][j].x + vx[r][j].y) + (vx[r][j].z + vx[r][j].w);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float t = wave_sum(sm[r]);
            if (lane == 0) part[wave * R + r] = t;
        }
        __syncthreads();
        float ssq[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float t = 0.f;
            for (int ww = 0; ww < nw; ++ww) t += part[ww * R + r];
            sm[r] = t / (float)K;   // mean
            ssq[r] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < NXJ; ++j) {
            const int i = (tid + j * nthr) * 4;
            if (i < K) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float4 v = vx[r][j];
                    v.x -= sm[r]; v.y -= sm[r]; v.z -= sm[r]; v.w -= sm[r];
                    ssq[r] = fmaf(v.x, v.x, ssq[r]); ssq[r] = fmaf(v.y, v.y, ssq[r]);
                    ssq[r] = fmaf(v.z, v.z, ssq[r]); ssq[r] = fmaf(v.w, v.w, ssq[r]);
                    float4 o;
                    o.x = __fmul_rn(gx[j].x, v.x); o.y = __fmul_rn(gx[j].y, v.y);
                    o.z = __fmul_rn(gx[j].z, v.z); o.w = __fmul_rn(gx[j].w, v.w);
                    *reinterpret_cast<float4*>(xs + r * K + i) = o;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float t = wave_sum(ssq[r]);
            if (lane == 0) part[(nw + wave) * R + r] = t;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float t = 0.f;
            for (int ww = 0; ww < nw; ++ww) t += part[(nw + ww) * R + r];
            rs_row[r] = 1.0f / sqrtf(t / (float)K + a.eps);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < NK; ++q)
#pragma unroll
                for (int e4 = 0; e4 < E; e4 += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(xs + r * K + kbase + q * kWave * E + lane * E + e4);
                    xv[r][q][e4 + 0] = v.x; xv[r][q][e4 + 1] = v.y; xv[r][q][e4 + 2] = v.z; xv[r][q][e4 + 3] = v.w;
                }
    } else if (PRO == PRO_ATTN) {
        constexpr int H = K / kHeadDim;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int idx = tid + it * 256;
            const int pos = a_pos[it];
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pos >= 0) {
                const int ns = pos / a.keys_per_split + 1;
                float M = -INFINITY, den = 0.f;
                for (int s0 = 0; s0 < ns; s0 += SPC) {
                    if (s0 > 0) {  // long contexts: further passes (the first pass was requested up front)
                        const int r = idx / (K / 4), k4 = idx % (K / 4);
                        const size_t b = ((size_t)r * H + (k4 >> 4)) * a.n_split_max;
#pragma unroll
                        for (int sp = 0; sp < SPC; ++sp) {
                            const int sc = s0 + sp < a.n_split_max ? s0 + sp : a.n_split_max - 1;
                            pmv[it][sp] = a.pm[b + sc];
                            plv[it][sp] = a.pl[b + sc];
                            pav[it][sp] = *reinterpret_cast<const float4*>(a.pacc + (b + sc) * kHeadDim + ((k4 & 15) << 2));
                        }
                    }
                    float Mc = M;
#pragma unroll
                    for (int sp = 0; sp < SPC; ++sp)
                        if (s0 + sp < ns) Mc = fmaxf(Mc, pmv[it][sp]);
                    const float scale = expf(M - Mc);   // 0 on the first pass
                    o.x *= scale; o.y *= scale; o.z *= scale; o.w *= scale;
                    den *= scale;
#pragma unroll
                    for (int sp = 0; sp < SPC; ++sp)
                        if (s0 + sp < ns) {
                            const float ws = expf(pmv[it][sp] - Mc);
                            den = fmaf(ws, plv[it][sp], den);
                            o.x = fmaf(ws, pav[it][sp].x, o.x); o.y = fmaf(ws, pav[it][sp].y, o.y);
                            o.z = fmaf(ws, pav[it][sp].z, o.z); o.w = fmaf(ws, pav[it][sp].w, o.w);
                        }
                    M = Mc;
                }
                o.x /= den; o.y /= den; o.z /= den; o.w /= den;
            }
            if (idx < R * (K / 4)) *reinterpret_cast<float4*>(xs + (size_t)idx * 4) = o;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < NK; ++q)
#pragma unroll
                for (int e4 = 0; e4 < E; e4 += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(xs + r * K + kbase + q * kWave * E + lane * E + e4);
                    xv[r][q][e4 + 0] = v.x; xv[r][q][e4 + 1] = v.y; xv[r][q][e4 + 2] = v.z; xv[r][q][e4 + 3] = v.w;
                }
    }

#ifdef T3K_STAMPS
    asm volatile("" :: "v"(xv[0][0][0]));
    T3K_STAMP(a, gwave, 1);   // activation rows ready
#endif
    for (; cb < n_chunks; cb += cstride, c += cstride) {
        const bool live = c < n_chunks;
        float acc[ROWS][R];
#pragma unroll
        for (int j = 0; j < ROWS; ++j)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < NK; ++q)
#pragma unroll
                    for (int e = 0; e < E; ++e) s = fmaf(w[j][q][e], xv[r][q][e], s);
                acc[j][r] = s;
            }
#ifdef T3K_STAMPS
        asm volatile("" :: "v"(acc[0][0]));
        T3K_STAMP(a, gwave, 2);   // weights arrived, FMAs done
#endif
        const int cn = c + cstride;
        if (cn < n_chunks) load_chunk(cn);  // next chunk's loads fly during the reduction
#pragma unroll
        for (int j = 0; j < ROWS; ++j)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[j][r] = wave_sum(acc[j][r]);
                if (PRO == PRO_NORM || PRO == PRO_LNORM) acc[j][r] = __fmul_rn(acc[j][r], rs_row[r]);
            }

        if (KSPLIT > 1) {
            __syncthreads();  // previous iteration's readers are done with part[]
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < ROWS; ++j)
#pragma unroll
                    for (int r = 0; r < R; ++r) part[(wave * ROWS + j) * R + r] = acc[j][r];
            }
            __syncthreads();
            // every wave of the group re-reads the K-partials in a fixed order
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float s = part[((grp * KSPLIT) * ROWS + j) * R + r];
#pragma unroll
                    for (int ww = 1; ww < KSPLIT; ++ww) s += part[((grp * KSPLIT + ww) * ROWS + j) * R + r];
                    acc[j][r] = s;
                }
            if (kpart != 0) continue;  // one wave of the group writes
        }
#ifdef T3K_STAMPS
        asm volatile("" :: "v"(acc[0][0]));
        T3K_STAMP(a, gwave, 3);   // reductions done
#endif
        if (!live) continue;

        if (EPI == EPI_GELU) {
            // gelu_tanh(c_fc(x) + b)  (R/t3.R:1303-1304)
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (lane == j * R + r) {
                        const int n = ROWS * c + j;
                        if (n < pN) {
                            float v = acc[j][r];
                            if (a.bias != nullptr) v = __fadd_rn(v, a.bias[n]);
                            const float inner = 0.7978845608028654f * (v + 0.044715f * v * v * v);
                            a.y[(size_t)r * a.ldy + n] = 0.5f * v * (1.0f + tanhf(inner));
                        }
                    }
        } else if (EPI == EPI_STORE || EPI == EPI_RESID) {
            const bool first = (cb == (int)blockIdx.x * ngrp);
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (lane == j * R + r) {
                        const int n = ROWS * c + j;
                        if (n < pN) {
                            const size_t o = (size_t)r * a.ldy + n;
                            float v = acc[j][r];
                            if (a.bias != nullptr) v = __fadd_rn(v, a.bias[n]);
                            if (EPI == EPI_RESID) {
                                const float res = first ? e_res : presid[o];
                                a.y[o] = __fadd_rn(res, v);
                            } else {
                                a.y[o] = v;
                            }
                        }
                    }
        } else if (EPI == EPI_SWIGLU) {
            // silu(g) * u with silu(g) = g * sigmoid(g)  (R/llama.R:302, 307)
            constexpr int HP = ROWS / 2;
#pragma unroll
            for (int j = 0; j < HP; ++j)
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (lane == j * R + r) {
                        const float g = acc[j][r], u = acc[j + HP][r];
                        const float sg = 1.0f / (1.0f + expf(-g));
                        a.y[(size_t)r * a.ldy + HP * c + j] = __fmul_rn(__fmul_rn(g, sg), u);
                    }
        } else {  // EPI_ROPE_KV
            constexpr int HP = ROWS / 2;
            const int H = a.n_heads;
            const int per_head = 32 / HP;
            const int hm = c / per_head, i0 = (c % per_head) * HP;
            const int kind = hm / H, head = hm % H;  // 0 q, 1 k, 2 v
            const bool first = (cb == (int)blockIdx.x * ngrp);
#pragma unroll
            for (int jj = 0; jj < HP; ++jj)
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (lane == jj * R + r) {
                        const int pos = e_pos;
                        if (pos >= 0) {
                            const int i = i0 + jj;
                            float lo = acc[jj][r], hi = acc[jj + HP][r];
                            if (a.bias != nullptr) {
                                lo = __fadd_rn(lo, a.bias[hm * kHeadDim + i]);
                                hi = __fadd_rn(hi, a.bias[hm * kHeadDim + i + 32]);
                            }
                            if (kind < 2 && a.rope) {
                                // q*cos + rotate_half(q)*sin, each product rounded (R/llama.R:184-185);
                                // cos/sin rows repeat with period 32, so one (cos, sin) serves both halves
                                const float cs = first ? e_c : prope_cur[(size_t)(2 * r) * kHeadDim + i];
                                const float sn = first ? e_s : prope_cur[(size_t)(2 * r + 1) * kHeadDim + i];
                                const float nlo = __fadd_rn(__fmul_rn(lo, cs), __fmul_rn(-hi, sn));
                                const float nhi = __fadd_rn(__fmul_rn(hi, cs), __fmul_rn(lo, sn));
                                lo = nlo; hi = nhi;
                            }
                            if (kind == 0) {
                                float* q = a.y + (size_t)r * a.ldy + head * kHeadDim;
                                q[i] = lo; q[i + 32] = hi;
                            } else {
                                float* cbase = (kind == 1 ? a.kcache : a.vcache) +
                                               (((size_t)r * H + head) * a.max_len + pos) * kHeadDim;
                                cbase[i] = lo; cbase[i + 32] = hi;
                            }
                        }
                    }
        }
    }
    T3K_STAMP(a, gwave, 4);   // stores issued
}

// ---- single-query attention over the KV cache, split along the keys ---------------------------------
// (R/llama.R:257-264, src/t3_decode.cpp:151-160.)  grid (H, rows, n_split_max); block 256 = 16 groups
// of 16 lanes; a group owns one key at a time, each lane 4 of its 64 dims.  Block (h, r, s) covers keys
// [s*kps, (s+1)*kps) of row r / head h and writes an UNNORMALISED partial (m, l, acc[64]); the o_proj
// kernel's PRO_ATTN prologue merges the partials, so no atomics and no inter-block hand-off are needed.
// All K and V loads of a 64-key pass are issued before the first use (one HBM round trip per pass).
__global__ __launch_bounds__(256) void attn_split_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float red[16 * kHeadDim + 32];
    const int h = blockIdx.x, r = blockIdx.y, sp = blockIdx.z;
#ifdef T3K_STAMPS
    const int gwave = (((int)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6);
#endif
    T3K_STAMP(a, gwave, 0);
    const int pos = a.rowpos[r];
    if (pos < 0) return;
    const int L = pos + 1;
    const int k0 = sp * a.keys_per_split;
    if (k0 >= L) return;
    const int k1 = min(L, k0 + a.keys_per_split);
    const int tid = threadIdx.x, g = tid >> 4, sub = tid & 15, wave = tid >> 6;
    const int d = a.n_heads * kHeadDim;
    const int crow = a.rowmap ? a.rowmap[r] : r;
    const size_t base = ((size_t)crow * a.n_heads + h) * a.max_len * kHeadDim;
    const float* Kp = a.kcache + base;
    const float* Vp = a.vcache + base;
    const float4 q4 = *reinterpret_cast<const float4*>(a.q + (size_t)r * d + h * kHeadDim + sub * 4);

    float m_run = -INFINITY, l_run = 0.f;           // l_run: this group's share
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);   // this group's share, dims sub*4..+3
    for (int p0 = k0; p0 < k1; p0 += 64) {
        float4 kv[4], vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = p0 + g + 16 * u;
            const int jc = j < k1 ? j : k1 - 1;
            kv[u] = *reinterpret_cast<const float4*>(Kp + (size_t)jc * kHeadDim + sub * 4);
            vv[u] = *reinterpret_cast<const float4*>(Vp + (size_t)jc * kHeadDim + sub * 4);
        }
        float sc[4];
        float mloc = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float dot = fmaf(q4.w, kv[u].w, fmaf(q4.z, kv[u].z, fmaf(q4.y, kv[u].y, q4.x * kv[u].x)));
            dot = row16_sum(dot);
            sc[u] = (p0 + g + 16 * u < k1) ? dot * 0.125f : -INFINITY;  // 1/sqrt(64)
            mloc = fmaxf(mloc, sc[u]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, kWave));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, kWave));
        __syncthreads();
        if ((tid & 63) == 0) red[16 * kHeadDim + wave] = mloc;
        __syncthreads();
        const float mp = fmaxf(fmaxf(red[16 * kHeadDim], red[16 * kHeadDim + 1]),
                               fmaxf(red[16 * kHeadDim + 2], red[16 * kHeadDim + 3]));
        const float m_new = fmaxf(m_run, mp);
        const float scale = expf(m_run - m_new);  // 0 on the first pass (m_run = -inf)
        acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale;
        l_run *= scale;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float e = expf(sc[u] - m_new);  // exp(-inf) = 0 for padded keys
            l_run += e;
            acc.x = fmaf(e, vv[u].x, acc.x); acc.y = fmaf(e, vv[u].y, acc.y);
            acc.z = fmaf(e, vv[u].z, acc.z); acc.w = fmaf(e, vv[u].w, acc.w);
        }
        m_run = m_new;
    }
    __syncthreads();
    *reinterpret_cast<float4*>(red + g * kHeadDim + sub * 4) = acc;
    if (sub == 0) red[16 * kHeadDim + 8 + g] = l_run;
    __syncthreads();
    const size_t pb = ((size_t)r * a.n_heads + h) * a.n_split_max + sp;
    if (tid < kHeadDim) {
        float o = 0.f;
#pragma unroll
        for (int gg = 0; gg < 16; ++gg) o += red[gg * kHeadDim + tid];
        a.pacc[pb * kHeadDim + tid] = o;
    } else if (tid == 64) {
        float l = 0.f;
#pragma unroll
        for (int gg = 0; gg < 16; ++gg) l += red[16 * kHeadDim + 8 + gg];
        a.pl[pb] = l;
        a.pm[pb] = m_run;
    }
    T3K_STAMP(a, gwave, 4);
}

// ---- token-by-token prefill input: x[r] = embeds[r][t - start[r]] ---------------------------
__global__ __launch_bounds__(256) void prefill_embed_kernel(EmbedArgs a) {
    const int r = blockIdx.x;
    const int p = a.t - a.start[r];
    if (threadIdx.x == 0) a.rowpos[r] = p >= 0 ? p : -1;
    if (p < 0) return;
    if (threadIdx.x < 2 * kHeadDim) {   // cos / sin rows of this position for the QKV kernel's RoPE epilogue
        const int t = threadIdx.x;
        a.rope_cur[(size_t)r * 2 * kHeadDim + t] =
            (t < kHeadDim ? a.rope_cos : a.rope_sin)[(size_t)p * kHeadDim + (t & (kHeadDim - 1))];
    }
    const float* src = a.emb_rows[r] + (size_t)p * a.d;
    for (int i = threadIdx.x; i < a.d; i += blockDim.x) a.x[(size_t)r * a.d + i] = src[i];
}

// ---- drop_invalid_tokens (R/s3tokenizer.R:678-687): stable compaction of the ids below `limit` ---------------------
// One 256-thread block per utterance walks the stream in chunks of 256: keep flags -> wave ballots -> exclusive offsets
// (popcount below the lane + the earlier waves' counts through LDS) -> scatter.  Integer work, bit-exact by construction.
__global__ __launch_bounds__(256) void drop_invalid_kernel(const int* __restrict__ tokens, const int* __restrict__ n_tokens, int cap,
                                                           int limit, int* __restrict__ kept, int* __restrict__ n_kept) {
    __shared__ int wave_cnt[4];
    __shared__ int base;
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(n_tokens[u], cap);
    const int* src = tokens + (size_t)u * cap;
    int* dst = kept + (size_t)u * cap;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        const int tok = i < n ? src[i] : limit;
        const bool keep = i < n && tok >= 0 && tok < limit;
        const unsigned long long mask = __ballot(keep);
        if (lane == 0) wave_cnt[wave] = __popcll(mask);
        __syncthreads();
        int off = base + __popcll(mask & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (keep) dst[off] = tok;
        __syncthreads();
        if (tid == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (tid == 0) n_kept[u] = base;
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint32_t u = __float_as_uint(src[i]);
        uint32_t rb;
        if ((u & 0x7FFFFFFFu) > 0x7F800000u) rb = (u >> 16) | 0x40u;  // keep NaN a NaN
        else rb = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
        dst[i] = (uint16_t)rb;
    }
}

// ---- sampling tail: one workgroup (16 waves) per utterance ------------------------------------
// CFG combine -> repetition penalty -> /T -> softmax -> min-p (prob space) -> renorm -> sort desc
// -> cumsum -> top-p mask -> renorm -> draw -> EOS flag -> next-token embedding.
// R/t3.R:518-587 ; src/t3_decode.cpp:307-388.
constexpr int kSampThreads = 1024;

__device__ __forceinline__ float block_reduce_sum(float v, float* scratch) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < kSampThreads / 64; ++i) t += scratch[i];
    return t;
}
__device__ __forceinline__ float block_reduce_max(float v, float* scratch) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = scratch[0];
#pragma unroll
    for (int i = 1; i < kSampThreads / 64; ++i) t = fmaxf(t, scratch[i]);
    return t;
}

__device__ __forceinline__ float counter_uniform(unsigned long long seed, unsigned long long utt,
                                                 unsigned long long step) {
    // splitmix64 finaliser over (seed, utterance, step); identical to oracle/t3_oracle.py
    unsigned long long z = seed * 0x9E3779B97F4A7C15ULL + utt * 0xBF58476D1CE4E5B9ULL +
                           step * 0x94D049BB133111EBULL + 0x2545F4914F6CDD1DULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// order-preserving map float -> u32 (larger float = larger integer; -inf maps above 0, the sort padding)
__device__ __forceinline__ uint32_t mono_key(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float mono_value(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__device__ __forceinline__ int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

__global__ __launch_bounds__(kSampThreads) void sampler_kernel(SamplerArgs a, int npad_max) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    const int u = blockIdx.x;
    const int tid = threadIdx.x;
    const int V = a.V;
    if (a.done[u]) {
        if (tid < a.rpu) a.rowpos[u * a.rpu + tid] = -1;
        return;
    }
    // LDS carve: lg[Vpad] float | kp[npad_max] u32 | tot[1024] double | ki[npad_max] u16 | scratch[32] float | misc int[8]
    const int Vpad = (V + 3) & ~3;
    float* lg = reinterpret_cast<float*>(smraw);
    uint32_t* kp = reinterpret_cast<uint32_t*>(lg + Vpad);
    double* tot = reinterpret_cast<double*>(kp + npad_max);
    uint16_t* ki = reinterpret_cast<uint16_t*>(tot + kSampThreads);
    float* scratch = reinterpret_cast<float*>(ki + npad_max);
    int* misc = reinterpret_cast<int*>(scratch + 32);

    const int step = a.n_tokens[u];
    const float* lc = a.logits + (size_t)u * a.rpu * V;
    const uint8_t* seen = a.seen + (size_t)u * V;
    const bool use_cfg = a.cfg_w > 0.f && a.rpu > 1;
    const bool turbo = a.variant == 2;   // .turbo_sample_token (R/t3.R:1688-1742)

    // 1. CFG, penalty, temperature  (R/t3.R:518-537)
    float lmax = -INFINITY;
#pragma unroll 4
    for (int v = tid; v < V; v += kSampThreads) {
        float l = lc[v];
        if (use_cfg) l = __fadd_rn(l, __fmul_rn(a.cfg_w, __fsub_rn(l, lc[V + v])));
        if (a.mask_stop && v == a.stop_token) l = -INFINITY;
        if (a.penalty != 1.0f) {
            if (turbo) {
                // R/t3.R:1699-1708: sign-aware; step 0 penalises column BOS-1 only (Q17), later the generated ids
                const bool hit = step == 0 ? (v == a.start_token - 1) : (seen[v] != 0);
                if (hit) l = (l > 0.f) ? __fdiv_rn(l, a.penalty) : __fmul_rn(l, a.penalty);
            } else if (seen[v]) {
                l = __fdiv_rn(l, a.penalty);                                                        // R/t3.R:531 (Q2)
            }
        }
        if (a.temperature != 1.0f && !(turbo && !(a.temperature > 0.f))) l = __fdiv_rn(l, a.temperature);
        lg[v] = l;
        if (a.filt != nullptr) a.filt[(size_t)u * V + v] = l;   // parity hook (t3mi_session_filtered_logits)
        lmax = fmaxf(lmax, l);
    }
    lmax = block_reduce_max(lmax, scratch);
    float s2 = 1.f;
    if (!turbo) {
    // 2. softmax (R/t3.R:540)
    float se = 0.f;
    for (int v = tid; v < V; v += kSampThreads) {
        const float e = expf(lg[v] - lmax);
        lg[v] = e;
        se += e;
    }
    se = block_reduce_sum(se, scratch);
    float pmax = 0.f;
    for (int v = tid; v < V; v += kSampThreads) {
        const float p = __fdiv_rn(lg[v], se);
        lg[v] = p;
        pmax = fmaxf(pmax, p);
    }
    pmax = block_reduce_max(pmax, scratch);
    // 3. min-p in probability space + renorm (R/t3.R:543-548, Q5)
    const float thr = __fmul_rn(a.min_p, pmax);
    s2 = 0.f;
    for (int v = tid; v < V; v += kSampThreads) {
        float p = lg[v];
        if (p < thr) p = 0.f;
        lg[v] = p;
        s2 += p;
    }
    s2 = block_reduce_sum(s2, scratch);
    }  // !turbo
    int tok;
    int kkeep;
    float s3;
    if (a.top_p <= 0.f) {
        // top_p = 0: cumsum > 0 everywhere, only sorted index 0 survives in both variants (Q6), i.e. the
        // arg-max of the filtered distribution with ties to the lowest id -- no sort needed.
        unsigned long long best = 0ull;
        for (int v = tid; v < V; v += kSampThreads) {
            unsigned bits;
            bool valid;
            if (turbo) {   // arg-max of the filtered logits: top-k / top-p keep the first sorted entry (R/t3.R:1722-1727)
                const float l = lg[v];
                bits = mono_key(l);
                valid = l > -INFINITY;
            } else {
                const float p = __fdiv_rn(lg[v], s2);
                bits = __float_as_uint(p);
                valid = p > 0.f;
            }
            const unsigned long long key = ((unsigned long long)bits << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)v);
            if (valid && key > best) best = key;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long t = __shfl_xor(best, o, kWave);
            best = t > best ? t : best;
        }
        unsigned long long* sk = reinterpret_cast<unsigned long long*>(tot);
        __syncthreads();
        if ((tid & 63) == 0) sk[tid >> 6] = best;
        __syncthreads();
        best = sk[0];
#pragma unroll
        for (int i = 1; i < kSampThreads / 64; ++i) best = sk[i] > best ? sk[i] : best;
        kkeep = best ? 1 : 0;
        tok = best ? (int)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull)) : -1;
        s3 = turbo ? 1.0f : __uint_as_float((unsigned)(best >> 32));
        if (tid == 0) {
            kp[0] = turbo ? __float_as_uint(1.0f) : (unsigned)(best >> 32);
            ki[0] = (uint16_t)(tok < 0 ? 0xFFFF : tok);
        }
        __syncthreads();
    } else {
    // 4. compact the non-zero entries into (prob bits, id) keys
    uint32_t kth = 0u;
    if (turbo && a.top_k > 0 && a.top_k < V) {
        // top-k threshold = the k-th largest value (ties kept, R/t3.R:1714-1719): radix select over the
        // order-preserving integer image of the logits, 8 bits per pass, histogram in LDS
        int* hist = reinterpret_cast<int*>(tot);
        uint32_t prefix = 0u;
        int remaining = a.top_k;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const uint32_t himask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
            for (int v = tid; v < V; v += kSampThreads) {
                const uint32_t key = mono_key(lg[v]);
                if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int acc = 0, dsel = 0;
                for (int b = 255; b >= 0; --b) {
                    const int c = hist[b];
                    if (acc + c >= remaining) { dsel = b; break; }
                    acc += c;
                }
                misc[3] = dsel;
                misc[4] = remaining - acc;
            }
            __syncthreads();
            prefix |= (uint32_t)misc[3] << shift;
            remaining = misc[4];
            __syncthreads();
        }
        kth = prefix;
    }
    // 4. compact the non-zero entries (>= tau) into (prob bits, id) keys
    auto compact = [&](float tau) {
        if (tid == 0) misc[0] = 0;
        __syncthreads();
        for (int v = tid; v < V; v += kSampThreads) {
            if (turbo) {
                const float l = lg[v];
                const uint32_t key = mono_key(l);
                if (l > -INFINITY && key >= kth) {
                    const int slot = atomicAdd(&misc[0], 1);
                    kp[slot] = key;
                    ki[slot] = (uint16_t)v;
                }
            } else {
                const float p = __fdiv_rn(lg[v], s2);
                if (p > 0.f && p >= tau) {
                    const int slot = atomicAdd(&misc[0], 1);
                    kp[slot] = __float_as_uint(p);
                    ki[slot] = (uint16_t)v;
                }
            }
        }
        __syncthreads();
        return misc[0];
    };
    int n = compact(0.f);
    // top-p prefilter (Llama rules) when the support is large (min_p = 0 or a flat distribution): the kept set is a
    // prefix of the sorted order whose cumulative sum stays within top_p, and a prefix sum only involves the larger
    // entries -- so sorting just the entries >= tau gives the same prefix, cumsums and draw as sorting everything,
    // provided their total exceeds top_p.  tau = 2^e from a histogram of probability mass per binary exponent, with
    // a margin far above the rounding of the unordered LDS adds; if no e qualifies (flat distribution, top_p close
    // to 1) the full support is sorted as before.  At min_p = 0 this turns the 8194-element sort (326 us) into a
    // sort of the few hundred entries that matter; supports of <= 1024 entries skip it (the sort is cheap there).
    if (!turbo && a.top_p < 1.0f && n > 1024) {
        float* hs = reinterpret_cast<float*>(tot);
        if (tid < 256) hs[tid] = 0.f;
        __syncthreads();
        for (int i = tid; i < n; i += kSampThreads) atomicAdd(&hs[kp[i] >> 23], __uint_as_float(kp[i]));
        __syncthreads();
        if (tid == 0) {
            float acc = 0.f;
            int sel = 0;
            for (int b2 = 255; b2 > 0; --b2) {
                acc += hs[b2];
                if (acc >= a.top_p + 1e-3f) { sel = b2; break; }
            }
            misc[3] = sel;
        }
        __syncthreads();
        const int sel = misc[3];
        __syncthreads();
        if (sel > 0) n = compact(__uint_as_float((unsigned)sel << 23));   // tau = 2^(sel - 127)
    }
    // A support just above a power of two (the full vocabulary: 8194 entries, 8193 with the stop token masked) would sort 16384
    // slots: instead its n - 2^k last entries of the sorted order (smallest probability, largest id among equals) are selected
    // one by one and swapped to the tail, and the sort runs over the first 2^k.
    int nsort = n;
    {
        const int p2 = next_pow2(n < 2 ? 2 : n) >> 1;
        if (n > 1024 && n - p2 <= 4) {
            unsigned long long* red = reinterpret_cast<unsigned long long*>(tot);
            for (; nsort > p2; --nsort) {
                unsigned long long best = ~0ull;   // (key, 0xFFFF - id, slot): the minimum is the last entry of the descending order
                for (int i = tid; i < nsort; i += kSampThreads) {
                    const unsigned long long w = ((((unsigned long long)kp[i] << 16) | (unsigned long long)(0xFFFFu - ki[i])) << 14) | (unsigned long long)i;
                    best = w < best ? w : best;
                }
                red[tid] = best;
                __syncthreads();
                for (int o = kSampThreads >> 1; o > 0; o >>= 1) {
                    if (tid < o && red[tid + o] < red[tid]) red[tid] = red[tid + o];
                    __syncthreads();
                }
                if (tid == 0) {
                    const int pos = (int)(red[0] & 0x3FFFull), last = nsort - 1;
                    const uint32_t tp = kp[pos]; kp[pos] = kp[last]; kp[last] = tp;
                    const uint16_t ti = ki[pos]; ki[pos] = ki[last]; ki[last] = ti;
                }
                __syncthreads();
            }
        }
    }
    const int npad = next_pow2(nsort < 2 ? 2 : nsort);   // (== nsort when entries were moved to the tail: nothing to pad)
    for (int i = nsort + tid; i < npad; i += kSampThreads) {
        kp[i] = 0u;
        ki[i] = 0xFFFFu;
    }
    __syncthreads();
    // 5. bitonic sort: descending probability, ties by ascending id (= stable descending sort)
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            // one compare-exchange per thread and trip: pair q of this stage = (i, i | j) with bit j of i clear
            for (int q = tid; q < (npad >> 1); q += kSampThreads) {
                const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1)), ixj = i | j;
                const uint32_t pa = kp[i], pb = kp[ixj];
                const uint16_t ia = ki[i], ib = ki[ixj];
                const bool a_first = (pa > pb) || (pa == pb && ia < ib);
                const bool up = ((i & k) == 0);
                if (up ? !a_first : a_first) {
                    kp[i] = pb; kp[ixj] = pa;
                    ki[i] = ib; ki[ixj] = ia;
                }
            }
            __syncthreads();
        }
    }
    if (turbo) {
        // softmax of the sorted, filtered logits (R/t3.R:1723): exp(l - l_max) / sum, as probabilities in kp[]
        const float l0 = n > 0 ? mono_value(kp[0]) : 0.f;
        float es = 0.f;
        for (int i = tid; i < n; i += kSampThreads) {
            const float e = expf(mono_value(kp[i]) - l0);
            kp[i] = __float_as_uint(e);
            es += e;
        }
        es = block_reduce_sum(es, scratch);
        for (int i = tid; i < n; i += kSampThreads) kp[i] = __float_as_uint(__fdiv_rn(__uint_as_float(kp[i]), es));
        __syncthreads();
    }
    // 6. inclusive cumsum in double, rounded to float per prefix (libtorch CPU cumsum semantics)
    const int CH = ((npad > n ? npad : n) + kSampThreads - 1) / kSampThreads;
    const int i0 = tid * CH;
    double csum = 0.0;
    for (int i = i0; i < i0 + CH && i < n; ++i) csum += (double)__uint_as_float(kp[i]);
    tot[tid] = csum;
    __syncthreads();
    if (tid < 64) {
        // wave 0: exclusive scan of the 1024 chunk totals, 16 per lane
        double loc[16];
        double run = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            loc[q] = run;
            run += tot[tid * 16 + q];
        }
        double incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double t = __shfl_up(incl, o, kWave);
            if (tid >= o) incl += t;
        }
        const double excl = incl - run;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot[tid * 16 + q] = excl + loc[q];
    }
    __syncthreads();
    // cum[] aliases lg[] (the dense probabilities are no longer needed)
    float* cum = lg;
    {
        double run = tot[tid];
        for (int i = i0; i < i0 + CH && i < n; ++i) {
            run += (double)__uint_as_float(kp[i]);
            cum[i] = (float)run;
        }
    }
    __syncthreads();
    // 7. top-p mask: the kept set is a prefix of the sorted order
    //    r  : keep i == 0 or cum[i]   <= top_p   (R/t3.R:557-560, drops the crossing token)
    //    cpp: keep i == 0 or cum[i-1] <= top_p   (src/t3_decode.cpp:350-356, keeps it)
    if (tid == 0) misc[1] = 0;
    __syncthreads();
    {
        int cnt = 0;
        for (int i = tid; i < n; i += kSampThreads) {
            bool keep;
            if (i == 0) keep = true;
            else if (turbo && a.top_p >= 1.0f) keep = true;                   // R/t3.R:1722: filter only if top_p < 1
            else if (a.variant == 1) keep = !(cum[i - 1] > a.top_p);
            else keep = !(cum[i] > a.top_p);
            cnt += keep ? 1 : 0;
        }
        if (cnt) atomicAdd(&misc[1], cnt);
    }
    __syncthreads();
    kkeep = misc[1];
    s3 = 0.f;
    for (int i = tid; i < kkeep; i += kSampThreads) s3 += __uint_as_float(kp[i]);
    s3 = block_reduce_sum(s3, scratch);
    // 8. draw: inverse CDF over the kept prefix with the build's counter RNG
    const float uu = counter_uniform(a.seed, a.utt_ids[u], (unsigned long long)step);
    const float target = __fmul_rn(uu, kkeep > 0 ? cum[kkeep - 1] : 0.f);
    if (tid == 0) misc[2] = 0;
    __syncthreads();
    {
        int cnt = 0;
        for (int i = tid; i < kkeep; i += kSampThreads) cnt += (cum[i] <= target) ? 1 : 0;
        if (cnt) atomicAdd(&misc[2], cnt);
    }
    __syncthreads();
    int jsel = misc[2];
    if (jsel > kkeep - 1) jsel = kkeep - 1;
    tok = (kkeep > 0) ? (int)ki[jsel < 0 ? 0 : jsel] : -1;
    }  // general (sorted) path
    if (a.drawn != nullptr && tid == 0) a.drawn[u] = tok;
    if (a.forced != nullptr && step < a.n_forced) tok = a.forced[(size_t)u * a.n_forced + step];
    if (tok < 0 || tok >= V) tok = a.stop_token;  // degenerate logits (all NaN): end the utterance

    if (a.dist_ids != nullptr) {
        const int m = kkeep < a.dist_cap ? kkeep : a.dist_cap;
        for (int i = tid; i < m; i += kSampThreads) {
            a.dist_ids[(size_t)u * a.dist_cap + i] = (int)ki[i];
            a.dist_probs[(size_t)u * a.dist_cap + i] = __fdiv_rn(__uint_as_float(kp[i]), s3);
        }
        if (tid == 0) a.dist_n[u] = kkeep;
    }
    // 9. bookkeeping + next-token embedding (R/t3.R:569-587; src/t3_decode.cpp:365-388)
    const int pos_next = a.s0[u] + step;  // cache slot of the token just sampled
    int dn = 0;
    int max_new = a.max_new_u[u];   // this slot's own limit (<= the session's; t3mi_session_refill may lower it)
    if (a.cache_cap > 0) max_new = min(max_new, a.cache_cap - a.s0[u] - 1 + a.cap_extra);   // per utterance, R/t3.R:996
    if (tok == a.stop_token) dn = 1;
    else if (step + 1 >= max_new) dn = 2;
    else if (pos_next >= a.max_len || step + 1 >= a.max_pos) dn = 3;
    if (tid == 0) {
        if (step < a.cap) a.tokens[(size_t)u * a.cap + step] = tok;
        a.n_tokens[u] = step + 1;
        a.seen[(size_t)u * V + tok] = 1;
        a.done[u] = dn;
    }
    if (tid < a.rpu) a.rowpos[u * a.rpu + tid] = dn ? -1 : pos_next;
    if (!dn && a.rope_cos != nullptr && tid < 2 * kHeadDim) {
        const float v = (tid < kHeadDim ? a.rope_cos : a.rope_sin)[(size_t)pos_next * kHeadDim + (tid & (kHeadDim - 1))];
        for (int rr = 0; rr < a.rpu; ++rr) a.rope_cur[((size_t)u * a.rpu + rr) * 2 * kHeadDim + tid] = v;
    }
    if (!dn) {
        const float* e = a.emb + (size_t)tok * a.d;
        // Q11: BOS holds row 0 of the position table; Turbo adds no position term (R/t3.R:1628)
        const float* pe = a.pos_emb != nullptr ? a.pos_emb + (size_t)(step + 1) * a.d : nullptr;
        for (int i = tid; i < a.d; i += kSampThreads) {
            const float v = pe != nullptr ? __fadd_rn(e[i], pe[i]) : e[i];
            for (int rr = 0; rr < a.rpu; ++rr) a.x[((size_t)u * a.rpu + rr) * a.d + i] = v;
        }
    }
}

template <typename WT, int R, int NK, int KSPLIT, int ROWS, int EPI, int PRO>
hipError_t launch_gemv_t(const GemvArgs& a, hipStream_t s) {
    constexpr int K = NK * kWave * WT::E * KSPLIT;
    const int n_chunks = (a.N + ROWS - 1) / ROWS;
    // waves per block: aim at one block per CU (256 CUs) with every load of the launch in flight at once
    int nw = a.waves > 0 ? a.waves : 4;
    if (PRO == PRO_ATTN) nw = 4;   // the merge prologue is written for a 256-thread block
    if ((PRO == PRO_NORM || PRO == PRO_LNORM) && nw < 2) nw = 2;   // the norm prologue slices x over >= 128 threads
    if (nw % KSPLIT) nw = ((nw + KSPLIT - 1) / KSPLIT) * KSPLIT;
    if (nw > 8) nw = 8;
    const int per_block = nw / KSPLIT;
    int grid = (n_chunks + per_block - 1) / per_block;
    if (grid > 256 * 8) grid = 256 * 8;
    const size_t lds = ((PRO != PRO_PLAIN ? (size_t)R * K : 0) + (size_t)nw * (ROWS < 2 ? 2 : ROWS) * R + 32) * sizeof(float);
    auto kfn = gemv_kernel<WT, R, NK, KSPLIT, ROWS, EPI, PRO>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    GemvArgs al = a;
    al.waves = nw;   // the kernel reads its block size from here
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(nw * 64), lds, s, a.W, a.x, a.ln_w, a.resid, a.rowpos, a.rope_cur, a.N, n_chunks, al);
    return hipGetLastError();
}

template <typename WT, int R, int NK>
hipError_t launch_gemv_rn(const GemvArgs& a, int epi, int ksplit, hipStream_t s) {
    const int rpc = a.rows_per_chunk;
    if (a.layernorm && ksplit == 1) {   // Turbo (GPT-2) linears behind a LayerNorm
        switch (epi) {
            case EPI_STORE: return launch_gemv_t<WT, R, NK, 1, 2, EPI_STORE, PRO_LNORM>(a, s);
            case EPI_ROPE_KV: return launch_gemv_t<WT, R, NK, 1, 2, EPI_ROPE_KV, PRO_LNORM>(a, s);
            case EPI_GELU: return launch_gemv_t<WT, R, NK, 1, 2, EPI_GELU, PRO_LNORM>(a, s);
            default: return hipErrorInvalidValue;
        }
    }
    if (ksplit == 1) {
        switch (epi) {
            case EPI_STORE:
                if (rpc == 1) return launch_gemv_t<WT, R, NK, 1, 1, EPI_STORE, PRO_NORM>(a, s);
                if (rpc == 2) return launch_gemv_t<WT, R, NK, 1, 2, EPI_STORE, PRO_NORM>(a, s);
                return launch_gemv_t<WT, R, NK, 1, 4, EPI_STORE, PRO_NORM>(a, s);
            case EPI_ROPE_KV:
                if (rpc == 4) return launch_gemv_t<WT, R, NK, 1, 4, EPI_ROPE_KV, PRO_NORM>(a, s);
                return launch_gemv_t<WT, R, NK, 1, 2, EPI_ROPE_KV, PRO_NORM>(a, s);
            case EPI_SWIGLU:
                if (rpc == 4) return launch_gemv_t<WT, R, NK, 1, 4, EPI_SWIGLU, PRO_NORM>(a, s);
                return launch_gemv_t<WT, R, NK, 1, 2, EPI_SWIGLU, PRO_NORM>(a, s);
            case EPI_RESID:
                if (a.pacc != nullptr) {
                    if (rpc == 2) return launch_gemv_t<WT, R, NK, 1, 2, EPI_RESID, PRO_ATTN>(a, s);
                    return launch_gemv_t<WT, R, NK, 1, 1, EPI_RESID, PRO_ATTN>(a, s);
                }
                return launch_gemv_t<WT, R, NK, 1, 1, EPI_RESID, PRO_PLAIN>(a, s);
        }
    } else if (ksplit == 4 && epi == EPI_RESID && a.pacc == nullptr) {
        if (rpc == 1) return launch_gemv_t<WT, R, NK, 4, 1, EPI_RESID, PRO_PLAIN>(a, s);
        if (rpc == 2) return launch_gemv_t<WT, R, NK, 4, 2, EPI_RESID, PRO_PLAIN>(a, s);
        return launch_gemv_t<WT, R, NK, 4, 4, EPI_RESID, PRO_PLAIN>(a, s);
    }
    return hipErrorInvalidValue;
}

template <typename WT>
hipError_t launch_gemv_w(const GemvArgs& a, int epi, hipStream_t s) {
    const int unit = kWave * WT::E;
    if (a.K % unit) return hipErrorInvalidValue;
    int units = a.K / unit, ksplit = 1;
    if (units > 4) {
        if (units % 4) return hipErrorInvalidValue;
        units /= 4;
        ksplit = 4;
    }
    if (units != 1 && units != 2 && units != 4) return hipErrorInvalidValue;
#define T3K_DISPATCH_R(NKV)                                                      \
    switch (a.rows) {                                                            \
        case 1: return launch_gemv_rn<WT, 1, NKV>(a, epi, ksplit, s);            \
        case 2: return launch_gemv_rn<WT, 2, NKV>(a, epi, ksplit, s);            \
        case 4: return launch_gemv_rn<WT, 4, NKV>(a, epi, ksplit, s);            \
        default: return hipErrorInvalidValue;                                    \
    }
    if (units == 1) { T3K_DISPATCH_R(1) }
    if (units == 2) { T3K_DISPATCH_R(2) }
    T3K_DISPATCH_R(4)
#undef T3K_DISPATCH_R
}

}  // namespace

size_t sampler_lds_bytes(int V) {
    int npad = 2;
    while (npad < V) npad <<= 1;
    const size_t Vpad = (size_t)((V + 3) & ~3);
    return Vpad * 4 + (size_t)npad * 4 + (size_t)kSampThreads * 8 + (size_t)npad * 2 + 32 * 4 + 8 * 4;
}

hipError_t launch_gemv(const GemvArgs& a, int epilogue, int wdtype, hipStream_t s) {
    if (a.W == nullptr || a.y == nullptr || a.N <= 0 || a.K <= 0) return hipErrorInvalidValue;
    const bool norm = (epilogue != EPI_RESID);
    if (norm && (a.ln_w == nullptr || a.x == nullptr)) return hipErrorInvalidValue;
    if (epilogue == EPI_GELU && !a.layernorm) return hipErrorInvalidValue;
    if (epilogue == EPI_RESID) {
        if (a.resid == nullptr) return hipErrorInvalidValue;
        if (a.pacc != nullptr) {  // attention-merge prologue
            if (a.pm == nullptr || a.pl == nullptr || a.rowpos == nullptr || a.keys_per_split <= 0 ||
                a.n_split_max <= 0 || a.K % kHeadDim)
                return hipErrorInvalidValue;
        } else if (a.x == nullptr) {
            return hipErrorInvalidValue;
        }
    }
    if (epilogue == EPI_SWIGLU && (a.N % 8)) return hipErrorInvalidValue;
    if (epilogue == EPI_ROPE_KV) {
        if (a.N != 3 * a.n_heads * kHeadDim || a.rowpos == nullptr || a.kcache == nullptr ||
            a.vcache == nullptr || a.rope_cur == nullptr || a.max_len <= 0)
            return hipErrorInvalidValue;
    }
    return wdtype == 1 ? launch_gemv_w<BF16W>(a, epilogue, s) : launch_gemv_w<F32W>(a, epilogue, s);
}

hipError_t launch_attn_decode(const AttnArgs& a, hipStream_t s) {
    if (a.rows <= 0 || a.n_heads <= 0 || a.max_len <= 0 || a.n_split_max <= 0 || a.keys_per_split <= 0 ||
        (a.keys_per_split % 64) || a.pm == nullptr || a.pl == nullptr || a.pacc == nullptr)
        return hipErrorInvalidValue;
    if ((long long)a.n_split_max * a.keys_per_split < a.max_len) return hipErrorInvalidValue;
    hipLaunchKernelGGL(attn_split_kernel, dim3(a.n_heads, a.rows, a.n_split_max), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_prefill_embed(const EmbedArgs& a, hipStream_t s) {
    if (a.rows <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(prefill_embed_kernel, dim3(a.rows), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_sampler(const SamplerArgs& a, hipStream_t s) {
    if (a.n_utt <= 0 || a.V <= 0 || a.V > 16384 || a.V > 65535) return hipErrorInvalidValue;
    int npad = 2;
    while (npad < a.V) npad <<= 1;
    const size_t lds = sampler_lds_bytes(a.V);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sampler_kernel, dim3(a.n_utt), dim3(kSampThreads), lds, s, a, npad);
    return hipGetLastError();
}

hipError_t init_device() {
    // the sampler carves > 64 KiB of dynamic LDS; call once per device before any launch / capture
    return hipFuncSetAttribute(reinterpret_cast<const void*>(sampler_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_drop_invalid(const int* tokens, const int* n_tokens, int cap, int limit, int* kept, int* n_kept, int n_utt,
                               hipStream_t s) {
    if (!tokens || !n_tokens || !kept || !n_kept || cap <= 0 || n_utt <= 0 || limit <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(drop_invalid_kernel, dim3(n_utt), dim3(256), 0, s, tokens, n_tokens, cap, limit, kept, n_kept);
    return hipGetLastError();
}

hipError_t launch_f32_to_bf16(const float* src, void* dst, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(2048), dim3(256), 0, s, src, static_cast<uint16_t*>(dst), n);
    return hipGetLastError();
}

}  // namespace t3k
