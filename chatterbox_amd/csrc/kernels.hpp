// Launch wrappers of the gfx950 kernels behind the t3mi C ABI (see kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

namespace t3k {

enum Epilogue : int {
    EPI_STORE = 0,    // y[r][n] = acc                                  (speech_head, R/t3.R:515)
    EPI_RESID = 1,    // y[r][n] = resid[r][n] + acc                     (o_proj / down_proj, R/llama.R:355,360)
    EPI_ROPE_KV = 2,  // RoPE on q,k; q -> qbuf, k,v -> cache at rowpos  (R/llama.R:227-245; src/t3_decode.cpp:134-148)
    EPI_SWIGLU = 3,   // y[r][n] = silu(gate_n) * up_n                   (R/llama.R:305-308)
    EPI_GELU = 4,     // y[r][n] = gelu_tanh(acc + bias)                 (Turbo c_fc, R/t3.R:1303-1304)
};

enum Prologue : int {
    PRO_PLAIN = 0,  // activation rows used as they are
    PRO_NORM = 1,   // fused RMSNorm (R/llama.R:65-72)
    PRO_ATTN = 2,   // merge of the split-softmax attention partials
    PRO_LNORM = 3,  // fused LayerNorm (Turbo, R/t3.R:1200-1203); its bias term is folded into `bias`
};

#ifdef T3K_STAMPS
#define T3K_STAMP_SLOTS 8
#define T3K_STAMP_WAVES 8192
#endif

struct GemvArgs {
    const void* W = nullptr;      // [N, K] row-major (row = output feature), fp32 or bf16
    const float* x = nullptr;     // [rows, K]
    const float* ln_w = nullptr;  // RMSNorm / LayerNorm weight [K] (norm prologue) or nullptr
    int layernorm = 0;            // 1: LayerNorm prologue (mean removed), else RMSNorm
    const float* bias = nullptr;  // [N] added to the dot products (Turbo linears), or nullptr
    int rope = 1;                 // EPI_ROPE_KV: 0 = store q/k/v without rotation (GPT-2)
    float eps = 1e-5f;
    int N = 0, K = 0;
    int rows = 0;                 // activation rows in this launch (<= 4)
    float* y = nullptr;           // [rows, ldy]
    const float* resid = nullptr; // [rows, ldy] (may alias y)
    int ldy = 0;
    // EPI_ROPE_KV
    const int* rowpos = nullptr;  // [rows] cache position of the token in x[r]; < 0 = inactive row
    const float* rope_cur = nullptr;  // [rows][2][64] cos / sin rows of each row's current position
    float* kcache = nullptr;      // this layer, these rows: [rows][H][max_len][64]
    float* vcache = nullptr;
    int n_heads = 0, max_len = 0;
    // PRO_ATTN (o_proj): partials of attn_split_kernel; x is then unused
    const float* pm = nullptr;    // [rows][H][n_split_max] running max
    const float* pl = nullptr;    // [rows][H][n_split_max] sum of exp
    const float* pacc = nullptr;  // [rows][H][n_split_max][64] unnormalised sum of exp * v
    int n_split_max = 0, keys_per_split = 0;
    int waves = 0;                // waves per block (0 = default 4)
    int rows_per_chunk = 0;       // output features per wave-chunk (0 = kernel default)
#ifdef T3K_STAMPS
    unsigned long long* stamps = nullptr;  // diagnostic build only: [T3K_STAMP_WAVES][T3K_STAMP_SLOTS] of this launch
#endif
};

struct AttnArgs {
    const float* q = nullptr;     // [rows, H*64] (RoPE applied)
    const float* kcache = nullptr;  // [rows][H][max_len][64]
    const float* vcache = nullptr;
    const int* rowpos = nullptr;  // [rows]; query r attends to positions 0..rowpos[r] (< 0 = inactive)
    const int* rowmap = nullptr;  // [rows] cache row of query r (nullptr = identity; prefill: many queries per row)
    float* pm = nullptr;          // partial outputs, see GemvArgs
    float* pl = nullptr;
    float* pacc = nullptr;
    int rows = 0, n_heads = 0, max_len = 0;
    int n_split_max = 0, keys_per_split = 64;
#ifdef T3K_STAMPS
    unsigned long long* stamps = nullptr;
#endif
};

// many-row regime (batched decode, prefill): skinny GEMM on the matrix cores, see batch_kernels.hip
// Weight prefetch from a launch's idle CUs.  A decode step at few rows is a chain of launches that each leave most of the chip idle
// and each wait for their weights' first bytes from HBM; n_blocks EXTRA blocks appended to a launch request one dword per 128-byte
// line of up to two matrices that LATER launches stream, so that those find them in L2.  XCD-aware (what makes it work: hits
// in the Infinity Cache alone gain nothing): region i is a sequence of units of unit[i] bytes -- the weight range one block
// column of the consuming activation-stationary GEMM reads -- whose block column u runs on XCD u % 8 (remap[i] = 0) or on XCD
// u / (units / 8) (1: the split norm's producer); the prefetch blocks are further ROWS of the launch's grid (blockIdx.y >= 1: known
// without reading a kernel argument -- a test on an argument in front of the kernel's first loads cost every launch 0.2 us), each
// row as wide as the launch's own (a multiple of 8), so prefetch block p sits on XCD p % 8 and touches only the units its XCD's
// L2 will be asked for.  unit[i] = 0: no mapping, lines dealt over all blocks.  Every
// prefetch thread leaves the xor of the words it touched in sink [n_blocks * block size] (a buffer of its own, never read: the
// store is what keeps the loads in the program).
constexpr int kPfMaxBlocks = 1024;   // prefetch blocks of one launch (the sink buffer holds 512 words for each)
struct PfArgs {
    const void* ptr[2] = {nullptr, nullptr};
    unsigned bytes[2] = {0, 0};
    unsigned unit[2] = {0, 0};
    int remap[2] = {0, 0};
    int n_blocks = 0;
    unsigned* sink = nullptr;
};

struct GemmArgs {
    const void* W = nullptr;      // [N, K] row-major, fp32 or bf16
    const void* Wf = nullptr;     // the same matrix fragment-major (launch_weight_fragments with this launch's epilogue and type), or nullptr
    const void* A = nullptr;      // [M, K] activation rows: float (fp32 weights) or bf16 (bf16 weights)
    int M = 0, N = 0, K = 0;
    // Split RMS norm (activation-stationary kernel only; launch_gemm fails if it would run another kernel).
    // Producer -- an unsplit EPI_RESID launch: besides y it writes next_act[m][f] = next_norm_w[f] * y[m][f] in the activation
    // type and next_ss[t][m] = the sum of y[m][f]^2 over the 32 features of tile t (N / 32 tiles).
    void* next_act = nullptr;
    const float* next_norm_w = nullptr;
    float* next_ss = nullptr;
    // LayerNorm form of the producer (Turbo; resid_mean and row_dot both set, or neither): the row mean is known before the product
    // is -- mean(y[m]) = resid_mean[m] + A[m] . colmean(W) + mean(bias) -- and the attention launch before o_proj has every head's 64
    // values of A[m] in registers: it leaves row_dot[m][h] = A[m][64 h .. 64 h + 63] . colmean (AttnRowsArgs::dot_vec / dot_out).
    // next_act = next_norm_w * (y - mean), next_ss sums (y - mean)^2.  The consumer is the same (its bias must carry W . b_ln, the
    // LayerNorm bias folded in).
    const float* resid_mean = nullptr;   // [M] mean of resid[m] (launch_layernorm_rows' row_mean)
    const float* row_dot = nullptr;      // [M][row_dot_n]
    int row_dot_n = 0;                   // <= 32
    float mean_shift = 0.f;              // mean(bias)
    // Consumer -- A holds such rows: output row m is scaled by 1 / sqrt(sum_t row_ss[t][m] / K + norm_eps) before the epilogue
    // (the GEMM is linear in its rows), which completes w * x * rsqrt(mean(x^2) + eps) without a norm launch.
    const float* row_ss = nullptr;
    int row_ss_tiles = 0;
    float norm_eps = 0.f;
    float* y = nullptr;           // EPI_STORE / EPI_RESID output [M, ldy] fp32
    const float* resid = nullptr; // may alias y
    int ldy = 0;
    void* yact = nullptr;         // EPI_SWIGLU output [M, N/2] / EPI_GELU output [M, N], activation type
    const float* bias = nullptr;  // [N] or nullptr
    int rope = 1;
    // EPI_ROPE_KV
    const int* posmap = nullptr;  // [M] cache slot of item m (< 0 = inactive)
    const int* rowmap = nullptr;  // [M] cache row of item m (nullptr = identity)
    const float* rope_cos = nullptr;  // [max_len, 64]
    const float* rope_sin = nullptr;
    float* q = nullptr;           // [M, H*64] fp32
    void* kcache = nullptr;       // this layer: [cache rows][H][max_len][64], fp32 or bf16 (kv_bf16)
    void* vcache = nullptr;
    int kv_bf16 = 0;
    int n_heads = 0, max_len = 0;
    float* part = nullptr;        // scratch for K-split partial sums (EPI_STORE / EPI_RESID), or nullptr: no split
    size_t part_floats = 0;
    int* ks_out = nullptr;        // host: receives the K-split factor; with it set a residual GEMM in place (y == resid)
                                  // leaves the fold of its partial slabs to the caller (launch_*norm_rows / launch_fold_partials)
    unsigned long long* dbg = nullptr;  // diagnostic builds (-DT3K_STAMPS; scripts/micro/gemm_micro.cpp): [block][4] clock stamps; ignored otherwise
    PfArgs pf;                    // activation-stationary kernel, one row chunk, K unsplit: see PfArgs (last: the other offsets stay)
};

struct EmbedArgs {                // token-by-token prefill input (R/t3.R:494, :505)
    const float* const* emb_rows = nullptr;  // [rows] device pointers to [s0[r], d]
    const int* start = nullptr;   // [rows] first prefill step at which row r is live (right-aligned prompts)
    int t = 0;                    // prefill step
    int d = 0, rows = 0;
    float* x = nullptr;           // [rows, d]
    int* rowpos = nullptr;        // [rows] out
    const float* rope_cos = nullptr;  // [max_len, 64] (R/llama.R:87-112)
    const float* rope_sin = nullptr;
    float* rope_cur = nullptr;    // [rows][2][64] out
};

struct SamplerArgs {
    const float* logits = nullptr;  // [n_utt*rpu, V]
    int V = 0, rpu = 1, n_utt = 0;
    float cfg_w = 0.f, temperature = 1.f, top_p = 1.f, min_p = 0.f, penalty = 1.f;
    int variant = 0, top_k = 0, stop_token = -1, start_token = -1, mask_stop = 0, max_new_tokens = 0;
    unsigned long long seed = 0;
    const unsigned long long* utt_ids = nullptr;  // [n_utt] global index of the utterance in each slot (RNG stream = seed, utt_ids[u])
    const int* max_new_u = nullptr;  // [n_utt] token limit of each slot (max_new_tokens, or what t3mi_session_refill gave the slot)
    int cache_cap = 0;            // > 0: utterance u stops after min(max_new_tokens, cache_cap - s0[u] - 1 + cap_extra) tokens
    int cap_extra = 0;            // Turbo: +1 (first token comes from the prefill logits)
    uint8_t* seen = nullptr;      // [n_utt, V]
    int* tokens = nullptr;        // [n_utt, cap]
    int cap = 0;
    int* n_tokens = nullptr;      // [n_utt]
    int* done = nullptr;          // [n_utt] 0 live, 1 stop token, 2 max_new_tokens, 3 cache full
    int* rowpos = nullptr;        // [n_utt*rpu]
    const int* s0 = nullptr;      // [n_utt] prompt length
    const int* forced = nullptr;  // [n_utt, n_forced] or nullptr
    int n_forced = 0;
    const float* emb = nullptr;      // speech_emb [V, d]
    const float* pos_emb = nullptr;  // speech_pos_emb [max_pos, d]
    int d = 0, max_pos = 0, max_len = 0;
    float* x = nullptr;           // [n_utt*rpu, d] next-token embedding out
    const float* rope_cos = nullptr;  // [max_len, 64]
    const float* rope_sin = nullptr;
    float* rope_cur = nullptr;    // [n_utt*rpu][2][64] out: RoPE rows of the next token's position
    int* dist_ids = nullptr;      // [n_utt, dist_cap] debug (may be nullptr)
    float* dist_probs = nullptr;
    int* dist_n = nullptr;
    int* drawn = nullptr;         // [n_utt] token drawn before teacher forcing (debug)
    int dist_cap = 0;
    float* filt = nullptr;        // [n_utt, V] debug (may be nullptr): the logits row after CFG, penalty and temperature
};

size_t sampler_lds_bytes(int V);

// once per device (current device), before the first launch
hipError_t init_device();
// wdtype: 0 fp32, 1 bf16 (launch_gemm: with bf16 activation rows; 2 = bf16 weights with fp32 activation rows).
// All launches are asynchronous on `s`.
hipError_t launch_gemv(const GemvArgs& a, int epilogue, int wdtype, hipStream_t s);
hipError_t launch_attn_decode(const AttnArgs& a, hipStream_t s);
hipError_t launch_prefill_embed(const EmbedArgs& a, hipStream_t s);
hipError_t launch_sampler(const SamplerArgs& a, hipStream_t s);
hipError_t launch_gemm(const GemmArgs& a, int epilogue, int wdtype, hipStream_t s);
// 16-feature sub-tiles per block column (2 or 4; 0: not an activation-stationary shape) and K layers of the activation-stationary
// launch of this shape with bf16 operands: what a prefetch for that launch has to know (PfArgs::unit)
int gemm_as_plan(int M, int N, int K, int epilogue, bool splittable, int* gz_out);
// Fragment-major copy of a matrix W [N, K] (wdtype 0 fp32 / 1 bf16) for the activation-stationary GEMM (<= kAsMaxRows
// rows; GemmArgs::Wf): per 16-feature sub-tile (in the partner order of `epilogue`) and 64-byte k-step, the 64 x 16 B of
// one MFMA operand.  weight_fragment_elems = elements of the copy (N rounded up to whole 32-feature tiles).
constexpr int kAsMaxRows = 1024;
// true when a launch with Wf set, M rows and this K runs on the activation-stationary kernel with K unsplit (the shape the split
// norm needs on both sides)
bool gemm_as_takes(int M, int K);
size_t weight_fragment_elems(int N, int K, int epilogue);
hipError_t launch_weight_fragments(const void* W, void* out, int N, int K, int epilogue, int wdtype, hipStream_t s);
// row norms of the many-row path; with ks > 0 they first complete x from the K-split partial slabs of the residual
// GEMM before them: x[m] += pbias + sum_s part[s][m] (written back)
// (row_mean: [M] receives the mean of every completed row, or nullptr)
hipError_t launch_layernorm_rows(float* x, const float* w, const float* b, float eps, void* out, int M, int d, int act_bf16,
                                 hipStream_t s, const float* part = nullptr, int ks = 0, const float* pbias = nullptr,
                                 float* row_mean = nullptr);
hipError_t launch_rmsnorm_rows(float* x, const float* w, float eps, void* out, int M, int d, int act_bf16, hipStream_t s,
                               const float* part = nullptr, int ks = 0, const float* pbias = nullptr);
hipError_t launch_fold_partials(float* x, const float* part, int ks, const float* pbias, int M, int d, hipStream_t s);
// many-row attention (batched decode, prefill): one block per (query item, head) over keys 0..posmap[item]; writes
// the normalised output rows in the activation type of the o_proj GEMM.  kv_bf16 / out_bf16: element types.
struct AttnRowsArgs {
    const float* q = nullptr;       // [M, H*64] fp32 (RoPE applied)
    const void* kcache = nullptr;   // this layer: [cache rows][H][max_len][64]
    const void* vcache = nullptr;
    const int* posmap = nullptr;    // [M] last key of item m (< 0 = inactive)
    const int* rowmap = nullptr;    // [M] cache row of item m (nullptr = identity)
    void* out = nullptr;            // [M, H*64]
    int M = 0, n_heads = 0, max_len = 0, kv_bf16 = 0, out_bf16 = 0;
    const float* dot_vec = nullptr; // [H*64] and dot_out [M][H], both or neither: dot_out[m][h] = the stored values of head h . dot_vec's
    float* dot_out = nullptr;       // slice of head h (GemmArgs::row_dot of the o_proj launch that follows)
    PfArgs pf;                      // weight prefetch from the launch's idle CUs (few (item, head) blocks)
#ifdef T3K_DIAG_ATTN_SPLIT
    int diag_split = 0;             // make diag only: blocks per (item, head), each over a share of the keys (results wrong)
#endif
};
hipError_t launch_attn_rows(const AttnRowsArgs& a, hipStream_t s);
hipError_t launch_gather_rows(const float* const* src_rows, float* x, int M, int d, hipStream_t s);
// ---- prompt builder (prompt_kernels.hip; R/t3.R:48-90, 161-195, 278-315, 372-403) ----
// out[r] = table[ids[r]] (+ pos[r]), r < n; ids on device, checked against the table size by the caller
hipError_t launch_embed_rows(const float* table, const int* ids, const float* pos, float* out, int n, int d, hipStream_t s);
// out[i] = w[i] * sc
hipError_t launch_scale_row(const float* w, float sc, float* out, int d, hipStream_t s);
// unmasked multi-head attention: q [Mq, H*Dh], k / v [Mk, H*Dh] -> out [Mq, H*Dh], all fp32
hipError_t launch_mha_rows(const float* q, const float* k, const float* v, float* out, int Mq, int Mk, int H, int Dh,
                           hipStream_t s);
// drop_invalid_tokens (R/s3tokenizer.R:678-687) on device: kept[u][0 .. n_kept[u]) = the ids of tokens[u][0 .. min(n_tokens[u], cap))
// that are < limit, in order.  One block per utterance.
hipError_t launch_drop_invalid(const int* tokens, const int* n_tokens, int cap, int limit, int* kept, int* n_kept, int n_utt,
                               hipStream_t s);
// fp32 -> bf16 (round to nearest even) on device, n elements
hipError_t launch_f32_to_bf16(const float* src, void* dst, size_t n, hipStream_t s);

}  // namespace t3k
