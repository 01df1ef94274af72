// Kernels of the prompt builder: embedding gathers with the learned position term, the emotion
// row, and the perceiver's plain multi-head attention.  They run once per utterance on a few hundred rows, so
// they are written for clarity and exact fp32 op order, not for bandwidth; the linears of the conditioning
// encoder go through launch_gemm (fp32 MFMA) and its LayerNorm through launch_layernorm_rows.
//   reference: R/t3.R:48-90 (learned_position_embeddings), :161-195 (attention_block), :278-315 (t3_cond_enc),
//              :372-403 (prepare_input_embeds); Turbo :1447-1477.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace t3k {

namespace {

// out[r] = table[ids[r]] (+ pos[r])        nn_embedding(tokens + 1) + emb(arange) (R/t3.R:60-68, 366, 379-380)
__global__ __launch_bounds__(256) void embed_rows_kernel(const float* __restrict__ table, const int* __restrict__ ids,
                                                         const float* __restrict__ pos, float* __restrict__ out, int d) {
    const int r = blockIdx.x;
    const float* src = table + (size_t)ids[r] * d;
    const float* pr = pos != nullptr ? pos + (size_t)r * d : nullptr;
    for (int i = threadIdx.x; i < d; i += 256) out[(size_t)r * d + i] = pr != nullptr ? __fadd_rn(src[i], pr[i]) : src[i];
}

// out[i] = w[i] * s                        Linear(1, d, bias = FALSE) on the emotion scalar (R/t3.R:268, 305-310)
__global__ __launch_bounds__(256) void scale_row_kernel(const float* __restrict__ w, float s, float* __restrict__ out, int d) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < d) out[i] = __fmul_rn(w[i], s);
}

// Plain multi-head attention, no mask (R/t3.R:178-190): block (head h, query m); scores over all Mk keys in LDS,
// softmax normalised BEFORE the value product (as nnf_softmax then torch_matmul), output column per thread.
__global__ __launch_bounds__(256) void mha_rows_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, float* __restrict__ out, int Mk, int H,
                                                       int Dh, float scale) {
    extern __shared__ float sm[];   // [Dh] query | [Mk] scores
    __shared__ float red[8];
    float* qs = sm;
    float* sc = sm + Dh;
    const int h = blockIdx.x, m = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = H * Dh;
    for (int i = tid; i < Dh; i += 256) qs[i] = q[(size_t)m * d + h * Dh + i];
    __syncthreads();
    for (int j = wave; j < Mk; j += 4) {
        const float* kr = k + (size_t)j * d + h * Dh;
        float s = 0.f;
        for (int i = lane; i < Dh; i += 64) s = fmaf(qs[i], kr[i], s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) sc[j] = s * scale;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int j = tid; j < Mk; j += 256) mx = fmaxf(mx, sc[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < Mk; j += 256) {
        const float e = expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    for (int j = tid; j < Mk; j += 256) sc[j] = sc[j] / sum;
    __syncthreads();
    for (int i = tid; i < Dh; i += 256) {
        const float* vc = v + h * Dh + i;
        float acc = 0.f;
        for (int j = 0; j < Mk; ++j) acc = fmaf(sc[j], vc[(size_t)j * d], acc);
        out[(size_t)m * d + h * Dh + i] = acc;
    }
}

}  // namespace

hipError_t launch_embed_rows(const float* table, const int* ids, const float* pos, float* out, int n, int d, hipStream_t s) {
    if (!table || !ids || !out || n <= 0 || d <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(embed_rows_kernel, dim3(n), dim3(256), 0, s, table, ids, pos, out, d);
    return hipGetLastError();
}

hipError_t launch_scale_row(const float* w, float sc, float* out, int d, hipStream_t s) {
    if (!w || !out || d <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(scale_row_kernel, dim3((d + 255) / 256), dim3(256), 0, s, w, sc, out, d);
    return hipGetLastError();
}

hipError_t launch_mha_rows(const float* q, const float* k, const float* v, float* out, int Mq, int Mk, int H, int Dh,
                           hipStream_t s) {
    if (!q || !k || !v || !out || Mq <= 0 || Mk <= 0 || H <= 0 || Dh <= 0) return hipErrorInvalidValue;
    const size_t lds = (size_t)(Dh + Mk) * sizeof(float);
    if (lds > 60 * 1024) return hipErrorInvalidValue;   // <= ~15k keys: far above any conditioning prompt (150 / 375 tokens)
    hipLaunchKernelGGL(mha_rows_kernel, dim3(H, Mq), dim3(256), lds, s, q, k, v, out, Mk, H, Dh, 1.0f / sqrtf((float)Dh));
    return hipGetLastError();
}

}  // namespace t3k
