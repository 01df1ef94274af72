// Launch wrappers of the CFM-estimator kernels (cfm_kernels.hip); the GEMMs are launch_gemm's (kernels.hpp).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>

namespace t3k {

// cfm_attention (R/s3gen.R:260-290) without mask: qkv [n_seq * T][3 * heads * 64] (q | k | v), out [n_seq * T][heads * 64].
// part_o [kCfmAttnMaxSplit][n_seq * T][heads * 64] and part_ml [kCfmAttnMaxSplit][n_seq * T][heads][2] (both or neither):
// scratch for the key split the launch uses when it has too few 16-query waves to fill the chip (merged in a second launch).
constexpr int kCfmAttnMaxSplit = 8;
// qbias [heads * 64] and bd [heads][T][2T - 1] (both or neither, one sequence): the conformer's relative-position form
// (R/conformer.R:226-279) -- q + pos_bias_u, and bd[h][i][T - 1 - i + j] added to score (i, j) before the scale.
hipError_t launch_cfm_attn(const float* qkv, float* out, int n_seq, int T, int heads, int head_dim, float* part_o, float* part_ml,
                           hipStream_t s, const float* qbias = nullptr, const float* bd = nullptr);
// out[m] = mish(LayerNorm_C(in[m])) * mask[m] (+ add[m / T][0 .. C))     (causal_block1d, R/s3gen.R:186-196, :227)
hipError_t launch_cfm_norm_mish(const float* in, const float* w, const float* b, const float* mask, const float* add, int add_ld,
                                float* out, int M, int C, int T, float eps, hipStream_t s);
// col[m][k * C + c] = (x * mask)[m - 2 + k][c] within the row's sequence of T rows, zero before its start; masked
// (optional) = x * mask                                                     (causal_conv1d, R/s3gen.R:135-141)
hipError_t launch_cfm_window3(const float* x, const float* mask, float* col, float* masked, int M, int C, int T, hipStream_t s);
hipError_t launch_cfm_concat(const float* a, int Ca, const float* b, int Cb, float* out, int M, hipStream_t s);
hipError_t launch_cfm_mask_rows(const float* x, const float* mask, float* out, int M, int C, hipStream_t s);
// mode 0 silu, 1 mish, 2 leaky_relu(0.01), 3 multiply by arg
hipError_t launch_cfm_act(const float* x, float* out, size_t n, int mode, hipStream_t s, float arg = 0.f);
// col[t][j * C + c] = x[(t + j + off) / up][c] inside [0, T_in * up), else 0: the k-tap windows of a padded (and nearest-
// upsampled) conv1d as GEMM rows (R/conformer.R:168-178, 432-438); weights stored [out][tap][in]
hipError_t launch_rows_window(const float* x, float* col, int T_out, int T_in, int C, int k, int off, int up, hipStream_t s);
// espnet_rel_positional_encoding's slice for a sequence of T (R/conformer.R:34-95): pe [2T - 1][d]
hipError_t launch_rel_pos_emb(float* pe, int T, int d, hipStream_t s);
// bd[h][q][n] = (q + pos_bias_v)[q][h] . p[n][h]   (matrix_bd before rel_shift, R/conformer.R:247-253); q inside qkv rows
hipError_t launch_conf_bd(const float* qkv, int ld, const float* bias_v, const float* p, int d, float* bd, int T, int heads, hipStream_t s);
// sinusoidal_pos_emb (R/s3gen.R:72-91): t [B] on device -> out [B][dim]
hipError_t launch_cfm_time_sin(const float* t, float* out, int B, int dim, float scale, hipStream_t s);
// x += dt * (one_plus_rate * d_cond - rate * d_uncond)   (R/s3gen.R:734-737);  d_uncond == nullptr: x += dt * d_cond (:644-646)
hipError_t launch_cfm_euler(float* x, const float* d_cond, const float* d_uncond, size_t n, float dt, float one_plus_rate, float rate,
                            hipStream_t s);

// Dense fp32 GEMM for the S3Gen shapes (a few hundred to a few thousand rows, K and N of 256 .. 1536): y = epi(LN?(A) W^T + b).
// Exact-fp32 MFMA products.  ln_colsum != nullptr: a LayerNorm over the K columns of every A row (K = the row width) sits in
// front of the layer and has been FOLDED into it by the loader -- W holds W diag(w_ln), bias holds b + W b_ln, ln_colsum[n] =
// sum_k W'[n][k] -- so that LN(x) W^T + b = rstd (W' x - mean ln_colsum) + bias: the kernel multiplies the raw rows and its
// epilogue applies the row statistics it computed on the way (dense_fold_layernorm does the folding).  resid may alias y.
struct DenseGemmArgs {
    const float* A = nullptr;      // [M][lda]
    int lda = 0;
    const float* W = nullptr;      // [N][K] row-major, as the checkpoint stores nn_linear weights
    const float* bias = nullptr;   // [N] or nullptr
    float* y = nullptr;            // [M][ldy]
    int ldy = 0;
    const float* resid = nullptr;  // epi 1: [M][ldy]
    const float* ln_colsum = nullptr;   // [N]
    float ln_eps = 1e-5f;
    int M = 0, N = 0, K = 0;
    int epi = 0;                   // EPI_STORE (0), EPI_RESID (1: resid + acc + bias), EPI_GELU (4: gelu_tanh(acc + bias))
};
// true when launch_dense_gemm takes the shape (N % 32 == 0, K % 64 == 0, epi one of the three)
bool dense_gemm_supported(int M, int N, int K, int epi);
// Host side of the fold, on fp32 host copies: W [N][K] in place -> W diag(w_ln); bias_out [N] = (bias or 0) + W b_ln (with the
// ORIGINAL W); colsum_out [N] = row sums of the folded W.  Sums in double.
void dense_fold_layernorm(float* W, const float* bias, const float* ln_w, const float* ln_b, int N, int K, float* bias_out, float* colsum_out);
hipError_t launch_dense_gemm(const DenseGemmArgs& a, hipStream_t s);

}  // namespace t3k
