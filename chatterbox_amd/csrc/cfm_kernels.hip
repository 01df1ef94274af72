// gfx950 kernels of the S3Gen flow-matching decoder (CFM estimator; R/s3gen.R:64-528, 639-751) that the
// T3 kernels do not already provide.  Activations live as rows [B*T, C] fp32 (channels contiguous): the reference's
// (B, C, T) <-> (B, T, C) transposes (R/s3gen.R:482, 487, ...) are views here.  The linears and the causal convolutions
// (as GEMMs over 3-tap row windows) run on launch_gemm's exact-fp32 MFMA kernels (batch_kernels.hip); this file holds
//   cfm_attn_kernel      unmasked multi-head attention over a whole sequence, flash form, fp32 MFMA (R/s3gen.R:260-290)
//   cfm_norm_mish_kernel LayerNorm over channels -> Mish -> * mask (+ time-embedding row)    (R/s3gen.R:186-196, 227)
//   cfm_window3_kernel   the 3-tap causal window of conv(x * mask) as GEMM rows               (R/s3gen.R:138-142, 189)
//   small row kernels    concat, mask, activation, sinusoidal embedding, CFG + Euler update  (R/s3gen.R:72-91, 471-478, 734-737)
#include "cfm_kernels.hpp"

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

namespace t3k {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kDh = 64;        // head_dim of cfm_attention (R/s3gen.R:241)
constexpr int kLd = kDh + 4;   // LDS row stride of a K / V tile: 272 B, the 16 rows of a fragment read fall in distinct banks

__device__ __forceinline__ float mish_f(float x) {
    // R/s3gen.R:167: x * tanh(softplus(x)); nnf_softplus returns x above its threshold of 20
    const float sp = x > 20.f ? x : log1pf(expf(x));
    return x * tanhf(sp);
}

// ---- attention --------------------------------------------------------------------------------------------------
// grid (query blocks, heads, sequences); a block = NW waves = 16 NW queries of one (sequence, head) against all T keys.
// Per 64-key tile (staged once per block in LDS, next tile's global loads in flight under the arithmetic):
//   S^T = K Q^T   : A = K[key][dim] from LDS, B = Q[query][dim] from registers  -> C[row = key][col = query]
//   online softmax: a lane holds 16 scores of ONE query (col = lane & 15); the 4 lanes of a query meet through 2 shuffles
//   O^T += V^T P^T: A = V[key][d] from LDS, B = P^T straight from the score registers -- the k index of an MFMA step is
//                   a free permutation of the keys, so step r of key tile kt takes keys {16 kt + 4 g + r}: exactly the
//                   C-layout register r of lane group g.  No transposition through LDS.
// v_mfma_f32_16x16x4_f32 throughout: exact fp32 products, fp32 accumulate.
// Keys split over blocks (n_split > 1) when the (query group, head, sequence) count alone would leave SIMDs idle -- a
// 16-query wave is the unit of work and a 500-frame CFG pair has only 512 of them for 1024 SIMDs: split `sp` takes key
// tiles [sp * tps, (sp + 1) * tps) and leaves an unnormalised partial (o, running max, sum) that cfm_attn_merge_kernel
// combines in a fixed order; n_split == 1 writes the normalised rows directly.
// REL (the conformer's rel_position_attention, R/conformer.R:226-279): q gets pos_bias_u added on load and every score gets
// the position term bd[head][query][T - 1 - query + key] (conf_bd_kernel's product, read through rel_shift's index map,
// R/conformer.R:281-307) before the 1 / sqrt(d_k) scale.
template <int NW, bool REL>
__global__ __launch_bounds__(64 * NW) void cfm_attn_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T, int ld,
                                                           int inner, float scale, int n_split, int tps, float* __restrict__ po,
                                                           float* __restrict__ pml, int M_total, const float* __restrict__ qbias,
                                                           const float* __restrict__ bd) {
    __shared__ __attribute__((aligned(16))) float kt_s[64 * kLd];
    __shared__ __attribute__((aligned(16))) float vt_s[64 * kLd];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nl = lane & 15, g = lane >> 4;
    const int head = blockIdx.y, seq = blockIdx.z;
    const size_t row0 = (size_t)seq * T;
    const int qblk = blockIdx.x / n_split, sp = blockIdx.x % n_split;
    const int q_lo = (qblk * NW + wave) * 16;
    const int qi = min(q_lo + nl, T - 1);
    const float* qp = qkv + (row0 + qi) * ld + head * kDh + 4 * g;
    f32x4 qv[4];   // native vectors: arrays of HIP's float4 struct can end up in scratch memory
#pragma unroll
    for (int i = 0; i < 4; ++i) qv[i] = *reinterpret_cast<const f32x4*>(qp + 16 * i);
    if (REL) {
#pragma unroll
        for (int i = 0; i < 4; ++i) qv[i] += *reinterpret_cast<const f32x4*>(qbias + head * kDh + 4 * g + 16 * i);   // q + pos_bias_u
    }
    // row of this lane's query in the position product: entry (T - 1 - query) + key
    const float* bdq = REL ? bd + ((size_t)head * T + qi) * (size_t)(2 * T - 1) + (T - 1 - qi) : nullptr;
    const float* kbase = qkv + row0 * ld + inner + head * kDh;
    const float* vbase = kbase + inner;

    constexpr int PF = 1024 / (64 * NW);   // float4 per thread per tile (64 keys x 16 float4)
    f32x4 pk[PF], pv[PF];
    const int j_lo = sp * tps, n_tiles = min((T + 63) / 64, j_lo + tps);   // this split's tiles [j_lo, n_tiles)
    auto prefetch = [&](int j) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int e = tid + u * 64 * NW, key = e >> 4, c4 = (e & 15) << 2;
            const int kk = min(j * 64 + key, T - 1);   // rows past the end are masked below; the load stays in bounds
            pk[u] = *reinterpret_cast<const f32x4*>(kbase + (size_t)kk * ld + c4);
            pv[u] = *reinterpret_cast<const f32x4*>(vbase + (size_t)kk * ld + c4);
        }
    };
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    prefetch(j_lo);
    for (int j = j_lo; j < n_tiles; ++j) {
        __syncthreads();   // the previous tile's reads are done
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int e = tid + u * 64 * NW, key = e >> 4, c4 = (e & 15) << 2;
            *reinterpret_cast<f32x4*>(&kt_s[key * kLd + c4]) = pk[u];
            *reinterpret_cast<f32x4*>(&vt_s[key * kLd + c4]) = pv[u];
        }
        __syncthreads();
        prefetch(j + 1 < n_tiles ? j + 1 : j);   // unconditional (clamped): the compiler can count the loads in flight
        f32x4 s[4];
        float pb[4][4];
        if (REL) {   // requested before the MFMAs that hide them; keys past the end are clamped (masked below)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) pb[kt][r] = bdq[min(j * 64 + 16 * kt + 4 * g + r, T - 1)];
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(&kt_s[(16 * kt + nl) * kLd + 16 * i + 4 * g]);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[0], qv[i][0], s[kt], 0, 0, 0);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[1], qv[i][1], s[kt], 0, 0, 0);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[2], qv[i][2], s[kt], 0, 0, 0);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[3], qv[i][3], s[kt], 0, 0, 0);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = j * 64 + 16 * kt + 4 * g + r;
                const float raw = REL ? s[kt][r] + pb[kt][r] : s[kt][r];       // matrix_ac + matrix_bd, R/conformer.R:257
                const float v = key < T ? raw * scale : -INFINITY;             // R/s3gen.R:276
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);      // finite from the first tile on (a split's first tile holds a valid key)
        // the hardware exponential (v_exp_f32 on x log2 e, about 1 ulp): libm's expf is ~20 VALU instructions, 16 of them per lane
        // and tile sat between the two MFMA phases (15.9 -> 14.9 us per launch at T = 500); probabilities are summed and
        // normalised, the bar is 1e-3 of the output's magnitude
        const float alpha = __expf(m_run - m_new);
        float ls = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[kt][r] - m_new);
                s[kt][r] = p;
                ls += p;
            }
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vr = &vt_s[(16 * kt + 4 * g + r) * kLd + nl];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[16 * dt], s[kt][r], o[dt], 0, 0, 0);
            }
    }
    if (q_lo + nl < T) {
        const size_t row = row0 + q_lo + nl;
        if (n_split == 1) {
            const float inv = 1.0f / l_run;
            float* op = out + row * inner + head * kDh + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<float4*>(op + 16 * dt) = float4{o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv};
        } else {
            float* op = po + ((size_t)sp * M_total + row) * inner + head * kDh + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<float4*>(op + 16 * dt) = float4{o[dt][0], o[dt][1], o[dt][2], o[dt][3]};
            if (g == 0) {
                float* ml = pml + (((size_t)sp * M_total + row) * (inner / kDh) + head) * 2;
                ml[0] = m_run;
                ml[1] = l_run;
            }
        }
    }
}

// out[row][head][d] = sum_s po[s] * exp(m_s - m) / sum_s l_s * exp(m_s - m), s ascending: one thread per 4 output values
__global__ __launch_bounds__(256) void cfm_attn_merge_kernel(const float* __restrict__ po, const float* __restrict__ pml,
                                                             float* __restrict__ out, int M, int inner, int n_split) {
    const int H = inner / kDh, q4 = inner >> 2;
    const size_t total = (size_t)M * q4;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t row = e / q4;
        const int c4 = (int)(e % q4) << 2, head = c4 / kDh;
        float mx = -INFINITY;
        for (int s = 0; s < n_split; ++s) mx = fmaxf(mx, pml[(((size_t)s * M + row) * H + head) * 2]);
        float l = 0.f;
        float4 acc{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < n_split; ++s) {
            const float* ml = pml + (((size_t)s * M + row) * H + head) * 2;
            const float w = expf(ml[0] - mx);
            l += ml[1] * w;
            const float4 v = *reinterpret_cast<const float4*>(po + ((size_t)s * M + row) * inner + c4);
            acc.x += v.x * w; acc.y += v.y * w; acc.z += v.z * w; acc.w += v.w * w;
        }
        const float inv = 1.0f / l;
        *reinterpret_cast<float4*>(out + row * inner + c4) = float4{acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv};
    }
}

// ---- LayerNorm over channels -> Mish -> * mask (+ add[b][c]) : one wave per row, C <= 1024, C % 4 == 0 ---------------
__global__ __launch_bounds__(256) void cfm_norm_mish_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                            const float* __restrict__ b, const float* __restrict__ mask,
                                                            const float* __restrict__ add, int add_ld, float* __restrict__ out,
                                                            int M, int C, int T, float eps) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int n4 = C >> 2;
    const float* xr = in + (size_t)m * C;
    float4 v[4];
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = j * 64 + lane;
        v[j] = i < n4 ? *reinterpret_cast<const float4*>(xr + 4 * i) : float4{0.f, 0.f, 0.f, 0.f};
        sm += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
    const float mean = sm / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j * 64 + lane < n4) {
            const float c0 = v[j].x - mean, c1 = v[j].y - mean, c2 = v[j].z - mean, c3 = v[j].w - mean;
            ss = fmaf(c0, c0, ss); ss = fmaf(c1, c1, ss); ss = fmaf(c2, c2, ss); ss = fmaf(c3, c3, ss);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rs = 1.0f / sqrtf(ss / (float)C + eps);
    const float mk = mask[m];
    const float* ar = add != nullptr ? add + (size_t)(m / T) * add_ld : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = j * 64 + lane;
        if (i < n4) {
            const float4 ww = *reinterpret_cast<const float4*>(w + 4 * i), bb = *reinterpret_cast<const float4*>(b + 4 * i);
            float4 y;
            y.x = mish_f((v[j].x - mean) * rs * ww.x + bb.x) * mk;
            y.y = mish_f((v[j].y - mean) * rs * ww.y + bb.y) * mk;
            y.z = mish_f((v[j].z - mean) * rs * ww.z + bb.z) * mk;
            y.w = mish_f((v[j].w - mean) * rs * ww.w + bb.w) * mk;
            if (ar != nullptr) {   // h + mlp(time_emb), R/s3gen.R:227: added AFTER the mask
                const float4 aa = *reinterpret_cast<const float4*>(ar + 4 * i);
                y.x += aa.x; y.y += aa.y; y.z += aa.z; y.w += aa.w;
            }
            *reinterpret_cast<float4*>(out + (size_t)m * C + 4 * i) = y;
        }
    }
}

// ---- causal 3-tap window: col[m][k C + c] = (x * mask)[m - 2 + k][c] inside the row's own sequence, else 0 -----------
// (causal_conv1d pads (k - 1) zeros on the left, R/s3gen.R:135-141; the conv weight is stored [out][k][in] to match).
// masked (optional) receives x * mask, the input of the 1 x 1 res_conv of the same block (R/s3gen.R:229).
__global__ __launch_bounds__(256) void cfm_window3_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                          float* __restrict__ col, float* __restrict__ masked, int M, int C, int T) {
    const int n4 = C >> 2;
    const size_t total = (size_t)M * 3 * n4;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c4 = (int)(e % n4);
        const size_t mk = e / n4;
        const int k = (int)(mk % 3);
        const int m = (int)(mk / 3);
        const int t = m % T, ts = t - 2 + k;
        float4 v{0.f, 0.f, 0.f, 0.f};
        if (ts >= 0) {
            const int ms = m - 2 + k;
            const float mv = mask[ms];
            const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)ms * C + 4 * c4);
            v = float4{xv.x * mv, xv.y * mv, xv.z * mv, xv.w * mv};
        }
        *reinterpret_cast<float4*>(col + ((size_t)m * 3 + k) * C + 4 * c4) = v;
        if (k == 2 && masked != nullptr) *reinterpret_cast<float4*>(masked + (size_t)m * C + 4 * c4) = v;
    }
}

__global__ __launch_bounds__(256) void cfm_concat_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b, int Cb,
                                                         float* __restrict__ out, int M) {
    const int C = Ca + Cb;
    const size_t total = (size_t)M * C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const size_t m = e / C;
        out[e] = c < Ca ? a[m * Ca + c] : b[m * Cb + (c - Ca)];
    }
}

__global__ __launch_bounds__(256) void cfm_mask_rows_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                                            float* __restrict__ out, int M, int C) {
    const size_t total = (size_t)M * C;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) out[e] = x[e] * mask[e / C];
}

// mode 0: silu (timestep_embedding, R/s3gen.R:111; conformer feed-forward, R/conformer.R:336), 1: mish (resnet mlp,
// R/s3gen.R:210-213), 2: leaky_relu with slope 0.01 (pre_lookahead_layer, R/conformer.R:172), 3: * arg (xscale, R/conformer.R:90)
__global__ __launch_bounds__(256) void cfm_act_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n, int mode, float arg) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const float v = x[e];
        float y;
        if (mode == 0) y = v / (1.0f + expf(-v));
        else if (mode == 1) y = mish_f(v);
        else if (mode == 2) y = v >= 0.f ? v : v * 0.01f;
        else y = v * arg;
        out[e] = y;
    }
}

// ---- conformer pieces ---------------------------------------------------------------------------------------------
// k-tap window rows of one sequence for a convolution as a GEMM: col[t][j * C + c] = xs[(t + j + off) / up][c] where
// 0 <= t + j + off < T_in * up, else 0.  off = 0: right-padded look-ahead conv (R/conformer.R:168-172); off = -(k - 1):
// left-padded causal conv (:175-178); up = 2, off = -4, k = 5: nearest-neighbour x 2 then the left-padded conv (:432-438).
__global__ __launch_bounds__(256) void rows_window_kernel(const float* __restrict__ x, float* __restrict__ col, int T_out, int T_in, int C,
                                                          int k, int off, int up) {
    const int n4 = C >> 2;
    const size_t total = (size_t)T_out * k * n4;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c4 = (int)(e % n4);
        const size_t tj = e / n4;
        const int j = (int)(tj % k), t = (int)(tj / k);
        const int sidx = t + j + off;
        float4 v{0.f, 0.f, 0.f, 0.f};
        if (sidx >= 0 && sidx < T_in * up) v = *reinterpret_cast<const float4*>(x + (size_t)(sidx / up) * C + 4 * c4);
        *reinterpret_cast<float4*>(col + ((size_t)t * k + j) * C + 4 * c4) = v;
    }
}

// espnet_rel_positional_encoding (R/conformer.R:34-95): rows for the relative positions +(T - 1) .. -(T - 1):
// pe[n][2 i] = sin(rel * div_i), pe[n][2 i + 1] = cos(rel * div_i), rel = T - 1 - n, div_i = exp(2 i * -(log(1e4) / d))
__global__ __launch_bounds__(256) void rel_pos_emb_kernel(float* __restrict__ pe, int T, int d) {
    const int half = d >> 1, P = 2 * T - 1;
    const float c = (float)(-(9.210340371976184 / (double)d));
    for (int e = blockIdx.x * 256 + threadIdx.x; e < P * half; e += gridDim.x * 256) {
        const int n = e / half, i = e % half;
        const float ang = (float)(T - 1 - n) * expf((float)(2 * i) * c);
        pe[(size_t)n * d + 2 * i] = sinf(ang);
        pe[(size_t)n * d + 2 * i + 1] = cosf(ang);
    }
}

// matrix_bd before rel_shift (R/conformer.R:253): bd[h][q][n] = (q[q] + pos_bias_v)[h] . p[n][h], n < 2T - 1.
// One wave per (16 queries, 64 positions, head): K = 64 in 16 exact-fp32 MFMA steps, operands straight from L2.
__global__ __launch_bounds__(64) void conf_bd_kernel(const float* __restrict__ qkv, int ld, const float* __restrict__ bias_v,
                                                     const float* __restrict__ p, int d, float* __restrict__ bd, int T) {
    const int lane = threadIdx.x, nl = lane & 15, g = lane >> 4;
    const int P = 2 * T - 1, n0 = blockIdx.x * 64, q0 = blockIdx.y * 16, head = blockIdx.z;
    const float* qp = qkv + (size_t)min(q0 + nl, T - 1) * ld + head * kDh + 4 * g;
    const float* bp = bias_v + head * kDh + 4 * g;
    f32x4 av[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const f32x4*>(qp + 16 * i) + *reinterpret_cast<const f32x4*>(bp + 16 * i);
    f32x4 acc[4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        const float* pr = p + (size_t)min(n0 + 16 * pt + nl, P - 1) * d + head * kDh + 4 * g;
        f32x4 bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const f32x4*>(pr + 16 * i);
        acc[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][c], bv[i][c], acc[pt], 0, 0, 0);
    }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        const int n = n0 + 16 * pt + nl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = q0 + 4 * g + r;
            if (q < T && n < P) bd[((size_t)head * T + q) * P + n] = acc[pt][r];
        }
    }
}

// sinusoidal_pos_emb (R/s3gen.R:72-91): out[b] = cat(sin(e), cos(e)), e = (scale * t[b]) * exp(j * -log(1e4) / (half - 1))
__global__ __launch_bounds__(256) void cfm_time_sin_kernel(const float* __restrict__ t, float* __restrict__ out, int B, int dim,
                                                           float scale) {
    const int half = dim >> 1;
    const float step = (float)(-9.210340371976184 / (double)(half - 1));   // -log(10000) / (half - 1), rounded to float as R torch does
    for (int e = blockIdx.x * 256 + threadIdx.x; e < B * half; e += gridDim.x * 256) {
        const int b = e / half, j = e % half;
        const float arg = (scale * t[b]) * expf((float)j * step);
        out[(size_t)b * dim + j] = sinf(arg);
        out[(size_t)b * dim + half + j] = cosf(arg);
    }
}

// Euler step with classifier-free guidance (R/s3gen.R:732-737): x += dt * ((1 + rate) * d_cond - rate * d_uncond);
// without guidance (basic_euler, R/s3gen.R:644-646; d_uncond == nullptr): x += dt * d
__global__ __launch_bounds__(256) void cfm_euler_kernel(float* __restrict__ x, const float* __restrict__ dc, const float* __restrict__ du,
                                                        size_t n, float dt, float one_plus_rate, float rate) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const float d = du != nullptr ? __fsub_rn(__fmul_rn(one_plus_rate, dc[e]), __fmul_rn(rate, du[e])) : dc[e];
        x[e] = __fadd_rn(x[e], __fmul_rn(dt, d));
    }
}

inline int grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

int cfm_attn_splits(int n_seq, int T, int heads) {
    // 16-query waves of the launch without a key split; aim at two waves per SIMD (2048), at most kCfmAttnMaxSplit splits,
    // every split at least one 64-key tile
    const long long waves = (long long)((T + 15) / 16) * heads * n_seq;
    const int n_tiles = (T + 63) / 64;
    static const int target = getenv("T3MI_CFM_ATTN_WAVES") ? atoi(getenv("T3MI_CFM_ATTN_WAVES")) : 2048;   // tuning runs only
    int want = (int)((target + waves - 1) / waves);
    want = want < 1 ? 1 : (want > kCfmAttnMaxSplit ? kCfmAttnMaxSplit : want);
    want = want > n_tiles ? n_tiles : want;
    const int tps = (n_tiles + want - 1) / want;
    return (n_tiles + tps - 1) / tps;
}

hipError_t launch_cfm_attn(const float* qkv, float* out, int n_seq, int T, int heads, int head_dim, float* part_o, float* part_ml,
                           hipStream_t s, const float* qbias, const float* bd) {
    if (!qkv || !out || n_seq <= 0 || T <= 0 || heads <= 0 || head_dim != kDh) return hipErrorInvalidValue;
    if ((qbias == nullptr) != (bd == nullptr) || (bd != nullptr && n_seq != 1)) return hipErrorInvalidValue;
    const int inner = heads * kDh, ld = 3 * inner, M = n_seq * T;
    const float scale = 1.0f / sqrtf((float)head_dim);
    const int n_tiles = (T + 63) / 64;
    int ns = (part_o != nullptr && part_ml != nullptr) ? cfm_attn_splits(n_seq, T, heads) : 1;
    const int tps = (n_tiles + ns - 1) / ns;
    // 16 NW queries per block: the narrower block when the wider one would leave CUs idle
    const long long blocks4 = (long long)((T + 63) / 64) * heads * n_seq * ns;
    const dim3 g4(((T + 63) / 64) * ns, heads, n_seq), g2(((T + 31) / 32) * ns, heads, n_seq);
    if (bd != nullptr) {
        if (blocks4 >= 512)
            hipLaunchKernelGGL((cfm_attn_kernel<4, true>), g4, dim3(256), 0, s, qkv, out, T, ld, inner, scale, ns, tps, part_o, part_ml, M, qbias, bd);
        else
            hipLaunchKernelGGL((cfm_attn_kernel<2, true>), g2, dim3(128), 0, s, qkv, out, T, ld, inner, scale, ns, tps, part_o, part_ml, M, qbias, bd);
    } else if (blocks4 >= 512) {
        hipLaunchKernelGGL((cfm_attn_kernel<4, false>), g4, dim3(256), 0, s, qkv, out, T, ld, inner, scale, ns, tps, part_o, part_ml, M, qbias, bd);
    } else {
        hipLaunchKernelGGL((cfm_attn_kernel<2, false>), g2, dim3(128), 0, s, qkv, out, T, ld, inner, scale, ns, tps, part_o, part_ml, M, qbias, bd);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || ns == 1) return e;
    hipLaunchKernelGGL(cfm_attn_merge_kernel, dim3(grid_for((size_t)M * (inner >> 2))), dim3(256), 0, s, part_o, part_ml, out, M, inner, ns);
    return hipGetLastError();
}

hipError_t launch_cfm_norm_mish(const float* in, const float* w, const float* b, const float* mask, const float* add, int add_ld,
                                float* out, int M, int C, int T, float eps, hipStream_t s) {
    if (!in || !w || !b || !mask || !out || M <= 0 || C <= 0 || C > 1024 || (C & 3) || T <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cfm_norm_mish_kernel, dim3((M + 3) / 4), dim3(256), 0, s, in, w, b, mask, add, add_ld, out, M, C, T, eps);
    return hipGetLastError();
}

hipError_t launch_cfm_window3(const float* x, const float* mask, float* col, float* masked, int M, int C, int T, hipStream_t s) {
    if (!x || !mask || !col || M <= 0 || C <= 0 || (C & 3) || T <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cfm_window3_kernel, dim3(grid_for((size_t)M * 3 * (C >> 2))), dim3(256), 0, s, x, mask, col, masked, M, C, T);
    return hipGetLastError();
}

hipError_t launch_cfm_concat(const float* a, int Ca, const float* b, int Cb, float* out, int M, hipStream_t s) {
    if (!a || !b || !out || M <= 0 || Ca <= 0 || Cb <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cfm_concat_kernel, dim3(grid_for((size_t)M * (Ca + Cb))), dim3(256), 0, s, a, Ca, b, Cb, out, M);
    return hipGetLastError();
}

hipError_t launch_cfm_mask_rows(const float* x, const float* mask, float* out, int M, int C, hipStream_t s) {
    if (!x || !mask || !out || M <= 0 || C <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cfm_mask_rows_kernel, dim3(grid_for((size_t)M * C)), dim3(256), 0, s, x, mask, out, M, C);
    return hipGetLastError();
}

hipError_t launch_cfm_act(const float* x, float* out, size_t n, int mode, hipStream_t s, float arg) {
    if (!x || !out || n == 0 || mode < 0 || mode > 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cfm_act_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, out, n, mode, arg);
    return hipGetLastError();
}

hipError_t launch_rows_window(const float* x, float* col, int T_out, int T_in, int C, int k, int off, int up, hipStream_t s) {
    if (!x || !col || T_out <= 0 || T_in <= 0 || C <= 0 || (C & 3) || k <= 0 || up <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rows_window_kernel, dim3(grid_for((size_t)T_out * k * (C >> 2))), dim3(256), 0, s, x, col, T_out, T_in, C, k, off, up);
    return hipGetLastError();
}

hipError_t launch_rel_pos_emb(float* pe, int T, int d, hipStream_t s) {
    if (!pe || T <= 0 || d < 2 || (d & 1)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rel_pos_emb_kernel, dim3(grid_for((size_t)(2 * T - 1) * (d >> 1))), dim3(256), 0, s, pe, T, d);
    return hipGetLastError();
}

hipError_t launch_conf_bd(const float* qkv, int ld, const float* bias_v, const float* p, int d, float* bd, int T, int heads, hipStream_t s) {
    if (!qkv || !bias_v || !p || !bd || T <= 0 || heads <= 0 || d != heads * kDh) return hipErrorInvalidValue;
    hipLaunchKernelGGL(conf_bd_kernel, dim3((2 * T - 1 + 63) / 64, (T + 15) / 16, heads), dim3(64), 0, s, qkv, ld, bias_v, p, d, bd, T);
    return hipGetLastError();
}

hipError_t launch_cfm_time_sin(const float* t, float* out, int B, int dim, float scale, hipStream_t s) {
    if (!t || !out || B <= 0 || dim < 4 || (dim & 1)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cfm_time_sin_kernel, dim3(grid_for((size_t)B * (dim >> 1))), dim3(256), 0, s, t, out, B, dim, scale);
    return hipGetLastError();
}

hipError_t launch_cfm_euler(float* x, const float* d_cond, const float* d_uncond, size_t n, float dt, float one_plus_rate, float rate,
                            hipStream_t s) {
    if (!x || !d_cond || n == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cfm_euler_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, d_cond, d_uncond, n, dt, one_plus_rate, rate);
    return hipGetLastError();
}

}  // namespace t3k
