/* t3mi -- MI355X-native T3 speech-token decoder: C ABI (plain pointers and sizes).
 *
 * Drop-in boundary for chatterbox's T3 decode path.  The reference crosses from R to
 * native code exactly once per utterance:
 *     .Call("cpp_t3_decode", <11 SEXP>, PACKAGE = "chatterbox")      R/t3.R:1010-1036
 *     SEXP cpp_t3_decode(SEXP x 11)                                  src/t3_decode.cpp:207-219
 * Its tensor arguments are R-torch external pointers (src/t3_decode.cpp:26-32) that are
 * unusable without libtorch, so this ABI takes (i) a model handle opened from the
 * UNCHANGED safetensors file, (ii) the prompt embeddings as a plain float array and
 * (iii) the same decode parameters as the reference's named config list
 * (R/t3.R:1022-1034).  The R glue that binds these entry points is r-package/src/r_glue.c;
 * INTEGRATION.md shows the binding.
 *
 * All functions return T3MI_OK (0) or a negative status; t3mi_last_error() gives the
 * text.  Nothing throws across this boundary.  One model handle = one GPU = one HIP
 * stream; calls on one handle are not re-entrant (the reference is single-threaded too,
 * SURVEY §8b "Threading").
 */
#ifndef T3MI_H
#define T3MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T3MI_OK 0
#define T3MI_ERR_ARG (-1)     /* bad argument (Rf_error in the reference glue)            */
#define T3MI_ERR_IO (-2)      /* checkpoint missing / malformed / tensor shape mismatch   */
#define T3MI_ERR_HIP (-3)     /* HIP runtime failure (no device, OOM, launch error)       */
#define T3MI_ERR_STATE (-4)   /* call out of order (no active session, ...)               */
#define T3MI_WARN_CACHE_FULL 1 /* generation stopped: KV cache full (src/t3_decode.cpp:394-397) */

/* model variants: t3_model (Llama-520M, R/t3.R:326-442) / t3_model_turbo (GPT-2, R/t3.R:1420-1478) */
#define T3MI_VARIANT_LLAMA 0
#define T3MI_VARIANT_TURBO 1

/* storage type of the streamed weight matrices (arithmetic is always fp32 accumulate) */
#define T3MI_W_F32 0
#define T3MI_W_BF16 1

/* sampling-tail rule (the two reference backends differ) */
#define T3MI_SAMPLER_R 0      /* R/t3.R:529-567   : BOS-quirk penalty column, drops crossing token */
#define T3MI_SAMPLER_CPP 1    /* src/t3_decode.cpp:318-362 : empty seen set, keeps crossing token  */
#define T3MI_SAMPLER_TURBO 2  /* R/t3.R:1688-1742 : sign-aware penalty, top-k, top-p on logits     */

typedef struct t3mi_model t3mi_model;

/* Decode parameters == arguments of t3_inference() (R/t3.R:461-463) plus the fields of
 * the config list handed to cpp_t3_decode (R/t3.R:1022-1034). */
typedef struct t3mi_params {
    int32_t max_new_tokens;      /* t3_inference default 1000                              */
    float temperature;           /* 0.8                                                     */
    float cfg_weight;            /* 0.5 ; > 0 requires 2 rows per utterance (cond, uncond)  */
    float top_p;                 /* 0.95 ; 0 => deterministic arg-max           */
    float min_p;                 /* 0.05                                                    */
    float repetition_penalty;    /* 1.2                                                     */
    int32_t start_token;         /* start_speech_token 6561 (R/t3.R:19)                     */
    int32_t stop_token;          /* stop_speech_token 6562 (R/t3.R:20)                      */
    int32_t sampler;             /* T3MI_SAMPLER_*                                          */
    int32_t top_k;               /* turbo only (R/t3.R:1579), 0 = off                       */
    int32_t max_cache_len;       /* 0 = prompt + max_new_tokens (R path, Q13); else cap (cpp: 350): utterance u then
                                    generates at most min(max_new_tokens, max_cache_len - s0[u] - 1) (R/t3.R:996) */
    int32_t mask_stop;           /* 1 = never emit stop_token (fixed-length bench workloads) */
    uint64_t seed;               /* stream of the build's counter RNG (libtorch's is not reproducible) */
    int64_t first_utterance_id;  /* global index of utterance 0 of this call: utterance u draws from the RNG stream
                                    (seed, first_utterance_id + u), so a batch split over calls / ranks samples the
                                    same tokens as the unsplit batch; 0 for a single call            */
    int32_t drop_invalid_from;   /* > 0: the generate entry points return the stream with every id >= this value removed,
                                    order kept: drop_invalid_tokens (R/s3tokenizer.R:678-687, SPEECH_VOCAB_SIZE 6561), which
                                    generate() applies to the T3 output before S3Gen (R/tts.R:366); filtered on the
                                    device.  0 = return the raw stream (t3_inference's own contract, Q14)               */
    int32_t reserved_;
} t3mi_params;

/* what t3mi_open_cached did with the packed bf16 weight file */
#define T3MI_CACHE_NONE 0      /* no cache path given, fp32 storage, or the file could not be written       */
#define T3MI_CACHE_WRITTEN 1   /* no file existed: weights converted from the checkpoint, cache written     */
#define T3MI_CACHE_USED 2      /* weights read from the cache (header and checksum matched)                  */
#define T3MI_CACHE_REPLACED 3  /* a stale / foreign / damaged file was found: converted again and rewritten  */

typedef struct t3mi_info {
    int32_t variant, n_layers, d_model, n_heads, head_dim, d_ff, vocab, max_pos, wdtype, device;
    int64_t weight_bytes;        /* bytes streamed per decode step (P * b_w of SURVEY §8d)  */
    int32_t cache_state;         /* T3MI_CACHE_*                                             */
    int32_t has_prompt_tensors;  /* 1: text_emb / cond_enc.* present (the *_tokens entry points work) */
} t3mi_info;

/* per-kernel timing of one decode step (HIP events on the library's stream) */
typedef struct t3mi_kstat {
    char name[32];
    int32_t launches;            /* launches of this kernel per decode step                 */
    float total_us;              /* summed device time of those launches in one step        */
    double bytes;                /* algorithmic bytes those launches move                   */
} t3mi_kstat;

/* Fill `p` with the defaults of t3_inference (R/t3.R:461-463), sampler = T3MI_SAMPLER_R. */
void t3mi_default_params(t3mi_params* p);

/* Text of the last error on this thread (never NULL). */
const char* t3mi_last_error(void);

/* Number of visible HIP devices (0 on a box without a GPU; never fails). */
int t3mi_device_count(void);

/* Open a model from the reference's safetensors layout (keys of R/t3.R:1065-1097,
 * R/llama.R:566-607; turbo keys R/t3.R:1508-1549), upload to `device`, build RoPE tables
 * (R/llama.R:87-152).  Replaces load_t3_weights + .get_cpp_weights (R/t3.R:884-912). */
int t3mi_open(const char* safetensors_path, int variant, int device, int wdtype, t3mi_model** out);
/* The same with a packed bf16 weight cache (the reference loads CPU-first and converts on every start,
 * R/tts.R:72-79): with wdtype = T3MI_W_BF16 and a non-NULL `cache_path`, the bf16 image of the streamed matrices is read
 * from that file when its header matches the checkpoint (size, mtime in ns, variant, dims, layout version) and its
 * checksum holds; otherwise the weights are converted from the checkpoint as t3mi_open does and the file is (re)written
 * atomically (best effort: an unwritable path is not an error).  t3mi_get_info().cache_state tells which happened.
 * The usual place is next to the checkpoint in the HF cache: "<safetensors_path>.t3mi-bf16". */
int t3mi_open_cached(const char* safetensors_path, int variant, int device, int wdtype, const char* cache_path,
                     t3mi_model** out);
/* Validate a checkpoint WITHOUT a GPU: header, every tensor name / shape / dtype the loaders read (R/t3.R:1065-1145,
 * R/llama.R:566-607; Turbo R/t3.R:1508-1549), and fill the dimensions.  T3MI_ERR_IO with the tensor's name for a
 * truncated, mis-shaped or foreign file -- the same checks t3mi_open makes before its first allocation. */
int t3mi_inspect(const char* safetensors_path, int variant, t3mi_info* info);
int t3mi_close(t3mi_model* m);
int t3mi_get_info(const t3mi_model* m, t3mi_info* info);

/* One utterance, whole loop: prefill + decode until stop_token / max_new_tokens / cache full.
 * == t3_inference (R/t3.R:461-605) after prepare_input_embeds, and == t3_inference_cpp's
 * prefill + .Call("cpp_t3_decode") (R/t3.R:987-1036).
 *   embeds : host float32 [rows, s0, d_model], row-major (rows = 2 with CFG: cond, uncond)
 *   out    : receives 0-indexed token ids INCLUDING the stop token if hit (Q14); cap >= max_new_tokens, and
 *            cap >= max_new_tokens + 1 on the Turbo variant, which samples one token from the prefill logits and
 *            then up to max_new_tokens more (R/t3.R:1614-1658); a smaller cap is rejected with T3MI_ERR_ARG
 * Returns T3MI_OK, T3MI_WARN_CACHE_FULL, or an error. */
int t3mi_generate(t3mi_model* m, const float* embeds, int rows, int s0, const t3mi_params* p,
                  int32_t* out, int32_t cap, int32_t* n_out);

/* Many independent utterances decoded together (no reference counterpart
 * beyond the serial loop of tts_chunked, R/tts.R:462-474).  embeds[u] is [rows_per_utt, s0[u], d];
 * out is [n_utt, cap]; n_out[u] tokens valid per utterance. */
int t3mi_generate_batch(t3mi_model* m, int n_utt, int rows_per_utt, const float* const* embeds,
                        const int32_t* s0, const t3mi_params* p, int32_t* out, int32_t cap,
                        int32_t* n_out);

/* ---- prompt construction on device ------------------------------------
 * Replaces t3_cond (R/t3.R:103-113) + prepare_conditioning / prepare_input_embeds (R/t3.R:361-403;
 * Turbo R/t3.R:1447-1477), i.e. learned_position_embeddings (R/t3.R:48-90), t3_cond_enc (:252-316) and the
 * perceiver resampler (:140-242), so the caller passes what t3_inference() itself receives: the
 * conditioning and the text token ids.  Needs the checkpoint's text_emb.* / cond_enc.* tensors
 * (R/t3.R:1076-1141); on a decode-only checkpoint these calls return T3MI_ERR_STATE. */
typedef struct t3mi_cond {
    const float* speaker_emb;                  /* host [n_speaker], voice-encoder embedding (256)          */
    int32_t n_speaker;
    const int32_t* cond_prompt_speech_tokens;  /* host [n_cond_tokens] 0-indexed ids (150 / Turbo 375), or NULL */
    int32_t n_cond_tokens;
    float emotion_adv;                         /* exaggeration scalar, 0.5 (R/t3.R:104); ignored by Turbo   */
    int32_t has_emotion_adv;                   /* 0 = cond$emotion_adv is NULL (R/t3.R:304)                 */
} t3mi_cond;

/* rows (2 with cfg_weight > 0 on the Llama variant), prompt length S0 and conditioning length of the
 * prompt t3_inference would build for n_text text tokens (start / stop text tokens NOT counted in n_text;
 * they are added here as R/t3.R:473-477 does; Turbo adds none, R/t3.R:1590). */
int t3mi_prompt_shape(t3mi_model* m, const t3mi_cond* cond, int n_text, float cfg_weight, int32_t* rows,
                      int32_t* s0, int32_t* len_cond);
/* prepare_input_embeds (R/t3.R:372-403 / :1457-1477) as t3_inference calls it (R/t3.R:473-494): writes the
 * host array out[rows, s0, d_model] (parity hook; the generate calls below keep the prompt in HBM). */
int t3mi_prepare_input_embeds(t3mi_model* m, const t3mi_cond* cond, const int32_t* text_tokens, int n_text,
                              float cfg_weight, int32_t start_token, float* out, size_t cap_floats);
/* t3mi_prepare_input_embeds for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc, code
 * that holds a HIP graph capture open on ANOTHER stream in thread-local or relaxed mode: the call enqueues on the model's
 * own stream and waits for it, so it can neither be captured into the caller's graph nor run while a global-mode capture
 * is open).  NO counterpart in the reference.  Same inputs and same values as
 * t3mi_prepare_input_embeds, but (i) the result [rows, s0, d_model] is written to the CALLER'S DEVICE buffer
 * `out_device` (device memory on the model's GPU; checked with hipPointerGetAttributes, a host pointer is T3MI_ERR_ARG) and (ii) all working memory (uploaded ids, speaker vector, the perceiver's
 * intermediate rows) lives in a fixed 8.02 MiB array inside the library's code object, one copy per GPU, held until the
 * process ends: the call makes no hipMalloc / hipFree.  Limits (T3MI_ERR_ARG beyond them, with the numbers in the
 * message): n_cond + n_text + 3 <= 4096 ids, speaker embedding <= 1024 floats, (4 * max(n_cond, n_query) + 4 * n_query)
 * * d_model <= 2,097,152 floats (e.g. d_model 1024, 32 queries: up to 480 conditioning tokens).  Nothing in the array
 * needs to start at zero (every row is written before it is read).  All calls on one GPU share that GPU's array: calls
 * that overlap on one GPU race; the call returns after the model's stream has drained. */
int t3mi_prepare_input_embeds_static(t3mi_model* m, const t3mi_cond* cond, const int32_t* text_tokens, int n_text,
                                     float cfg_weight, int32_t start_token, float* out_device, size_t cap_floats);
/* t3_inference(model, cond, text_tokens, ...) (R/t3.R:461-605) / t3_inference_turbo (R/t3.R:1578-1677):
 * prompt built on device, then the same loop as t3mi_generate. */
int t3mi_generate_tokens(t3mi_model* m, const t3mi_cond* cond, const int32_t* text_tokens, int n_text,
                         const t3mi_params* p, int32_t* out, int32_t cap, int32_t* n_out);
int t3mi_generate_batch_tokens(t3mi_model* m, int n_utt, const t3mi_cond* conds,
                               const int32_t* const* text_tokens, const int32_t* n_text, const t3mi_params* p,
                               int32_t* out, int32_t cap, int32_t* n_out);
/* t3mi_session_begin with the prompts built on device from tokens. */
int t3mi_session_begin_tokens(t3mi_model* m, int n_utt, const t3mi_cond* conds, const int32_t* const* text_tokens,
                              const int32_t* n_text, const t3mi_params* p);

/* ---- session API: the same loop in pieces (bench timing, parity hooks) ---------------- */

/* Allocate KV slabs (create_kv_cache, R/llama_traced.R:274-297), upload prompts, run the
 * prefill (R/t3.R:505 / :987-992) and leave the first logits pending.  `embeds_on_device`
 * != 0 means every embeds[u] is already a device pointer (inputs resident in HBM). */
int t3mi_session_begin(t3mi_model* m, int n_utt, int rows_per_utt, const float* const* embeds,
                       const int32_t* s0, const t3mi_params* p, int embeds_on_device);
/* Teacher forcing (parity hook): utterance u emits forced[u*n_forced + i] at step i instead of
 * its own sample.  Call between session_begin and the first session_decode. */
int t3mi_session_force(t3mi_model* m, const int32_t* forced, int n_forced);
/* Continuous batching: slots utts[0..n) of the running session take new utterances while the other slots keep their state (cache rows,
 * tokens, pending logits).  The reference has no counterpart: its tts_chunked runs the sentences one after the other (R/tts.R:466-474),
 * and a fixed batch idles every slot whose sentence has ended ("refill from a host queue").  embeds[i] is the
 * [rows_per_utt, s0[i], d] prompt of the utterance for slot utts[i] (as in t3mi_session_begin); max_new_tokens[i] its own token limit
 * (NULL: the session's; never above it); utterance_ids[i] the global index that names its random stream (NULL: first_utterance_id +
 * slot) -- with the id it would have had in one big batch, a refilled utterance draws the same tokens.  The prompt plus one token must
 * fit the cache rows the session was begun with.  Waits for the enqueued steps, so read a finished slot's tokens BEFORE refilling it;
 * the new prompts are prefilled together and the call returns with every slot's first logits pending.  Not available on a
 * teacher-forced session or with T3MI_PREFILL=token.  Arguments are checked before anything is touched (a refused call leaves the
 * session as it was); after a HIP error inside the call the session must be ended. */
int t3mi_session_refill(t3mi_model* m, int n, const int32_t* utts, const float* const* embeds, const int32_t* s0,
                        const int32_t* max_new_tokens, const int64_t* utterance_ids, int embeds_on_device);
/* The same with the prompts built on device from conditioning and text ids (prepare_input_embeds, R/t3.R:372-403). */
int t3mi_session_refill_tokens(t3mi_model* m, int n, const int32_t* utts, const t3mi_cond* conds,
                               const int32_t* const* text_tokens, const int32_t* n_text, const int32_t* max_new_tokens,
                               const int64_t* utterance_ids);
/* The whole job on top of t3mi_session_refill: n_utt utterances through `slots` batch slots; every `poll_steps` decode steps the slots
 * whose utterance has ended (stop token, its own limit max_new_each[i] <= p->max_new_tokens, cache full) deliver their tokens and take the
 * next waiting utterances.  Per-utterance results, status codes and the out / n_out layout are those of t3mi_generate_batch /
 * t3mi_generate_batch_tokens (utterance i draws from stream (seed, first_utterance_id + i) wherever it runs); max_new_each may be
 * NULL.  With p->max_cache_len = 0 the cache rows are sized for the longest prompt + limit of the job.  stats (may be NULL) receives
 * {decode steps, refill calls, sum over steps of occupied slots, slots}.  flags: 0, or T3MI_CONTINUOUS_LONGEST_FIRST -- the queue is
 * started in the order of the utterances' limits (max_new_each) or, without limits, of their text / prompt length, longest first, so
 * that the job does not end on one straggler; the results do not depend on the order.  Replaces the serial loop of tts_chunked
 * (R/tts.R:466-474). */
#define T3MI_CONTINUOUS_LONGEST_FIRST 1
int t3mi_generate_continuous(t3mi_model* m, int n_utt, int rows_per_utt, const float* const* embeds, const int32_t* s0,
                             const int32_t* max_new_each, int slots, int poll_steps, int flags, const t3mi_params* p, int32_t* out,
                             int32_t cap, int32_t* n_out, int32_t* stats);
int t3mi_generate_continuous_tokens(t3mi_model* m, int n_utt, const t3mi_cond* conds, const int32_t* const* text_tokens,
                                    const int32_t* n_text, const int32_t* max_new_each, int slots, int poll_steps, int flags,
                                    const t3mi_params* p, int32_t* out, int32_t cap, int32_t* n_out, int32_t* stats);
/* Enqueue `n_steps` decode steps (one token per live utterance per step) on the model's
 * stream; returns without waiting. */
int t3mi_session_decode(t3mi_model* m, int n_steps);
/* Same, bracketed by HIP events on that stream; waits and returns elapsed milliseconds. */
int t3mi_session_decode_timed(t3mi_model* m, int n_steps, float* ms);
int t3mi_session_sync(t3mi_model* m);
/* n_tokens[u], done[u] (1 = hit stop / max_new_tokens / cache full) after a sync. */
int t3mi_session_status(t3mi_model* m, int32_t* n_tokens, int32_t* done);
int t3mi_session_tokens(t3mi_model* m, int utt, int32_t* out, int32_t cap, int32_t* n_out);
/* Raw speech-head logits [rows_total, vocab] of the most recent step (parity hook). */
int t3mi_session_logits(t3mi_model* m, float* out);
/* Overwrite the pending logits [rows_total, vocab] (the ones the NEXT decode step's sampling tail will read) with a host
 * array.  Parity hook with no counterpart in the reference's API: it is how a test hands the reference's own recorded
 * logits (outputs/t3_debug.safetensors `raw_logits_step0`, replayed by scripts/debug_t3_step.R:108-214) to the device
 * sampling tail without the weights that produced them. */
int t3mi_session_set_logits(t3mi_model* m, const float* logits);
/* Keep, from the next decode step on, every utterance's logits row after CFG combine, repetition penalty and temperature
 * (R/t3.R:518-537; the values scripts/debug_t3_step.R:119-151 compares with `logits_after_cfg` / `logits_after_temp`).
 * Call between session_begin and the first session_decode; costs one [n_utt, vocab] fp32 buffer and a store per logit. */
int t3mi_session_keep_filtered(t3mi_model* m);
/* out[vocab] = that row for utterance `utt` at the most recent step (T3MI_ERR_STATE without t3mi_session_keep_filtered). */
int t3mi_session_filtered_logits(t3mi_model* m, int utt, float* out);
/* Final filtered distribution of utterance `utt` at the most recent step: ids in descending
 * probability, their renormalised probabilities; *n = support size (parity hook). */
int t3mi_session_dist(t3mi_model* m, int utt, int32_t* ids, float* probs, int32_t cap, int32_t* n);
/* drawn[u] = the token utterance u's sampler drew at the most recent step BEFORE teacher forcing
 * replaced it (parity hook for the draw itself). */
int t3mi_session_drawn(t3mi_model* m, int32_t* drawn);
/* Time every kernel of ONE decode step with HIP events (advances the session by one step). */
int t3mi_session_profile_step(t3mi_model* m, t3mi_kstat* stats, int cap, int* n_stats);
int t3mi_session_end(t3mi_model* m);

/* Raw HIP stream of the handle (hipStream_t as void*), for callers that time with their own events. */
void* t3mi_stream(t3mi_model* m);

/* ---- S3Gen flow-matching decoder: the CFM estimator and its Euler solvers (first piece) -------------------
 * What generate() runs on T3's speech tokens after the conformer encoder (R/tts.R:392-398 -> s3gen$inference R/s3gen.R:987-1056
 * -> causal_masked_diff_xvec$forward :805-890 -> causal_cfm$forward :594-636).  Sequences use the reference's layouts:
 * (B, 80, T) float32 row-major, mask (B, 1, T), spks (B, 80), t / r [B].  fp32 arithmetic (exact-fp32 MFMA GEMMs). */
typedef struct t3mi_cfm t3mi_cfm;
typedef struct t3mi_cfm_info {
    int32_t mel, hidden, n_mid, n_tfm, heads, head_dim, d_ff, time_in, time_dim, meanflow, device, reserved_;
    int64_t weight_bytes;
} t3mi_cfm_info;
/* cfm_estimator(...) + load_cfm_estimator_weights (R/s3gen.R:389-444, 1070-1178): reads the estimator's tensors from the
 * UNCHANGED s3gen safetensors file under `prefix` (NULL = "flow.decoder.estimator.", R/s3gen.R:1231); shapes come from the
 * file; a `time_embed_mixer.weight` makes it a meanflow estimator (R/s3gen.R:404-413).  n_heads: 0 = inner / 64. */
int t3mi_cfm_open(const char* safetensors_path, const char* prefix, int device, int n_heads, t3mi_cfm** out);
int t3mi_cfm_close(t3mi_cfm* m);
int t3mi_cfm_get_info(const t3mi_cfm* m, t3mi_cfm_info* info);
/* cfm_estimator$forward(x, mask, mu, t, spks, cond, r) (R/s3gen.R:446-520): host arrays in, out (B, 80, T).  r = NULL unless
 * the model is a meanflow one.  Attention is unmasked (compute_attn_mask returns NULL, :522-527); the mask multiplies. */
int t3mi_cfm_estimator(t3mi_cfm* m, int B, int T, const float* x, const float* mask, const float* mu, const float* t,
                       const float* spks, const float* cond, const float* r, float* out);
/* causal_cfm$forward (R/s3gen.R:594-636) for one utterance, untraced: z (1, 80, T) is the start noise the reference takes
 * from its own torch_randn buffer (`rand_noise[,, 1:T] * temperature`, :616-619; meanflow: randn with `noised_mels` laid over
 * the generated part, :610-615) -- libtorch's stream cannot be reproduced, so the caller owns it.  meanflow = 0: cosine
 * schedule and solve_euler (:651-751) with classifier-free guidance as a batch of two (row 2: mu / spks / cond zero),
 * d = (1 + cfg_rate) d_cond - cfg_rate d_uncond (0.7 in the reference, :554), t accumulated in float32.  meanflow = 1:
 * linear schedule and basic_euler (:639-649), no guidance, the estimator sees (t, r).  out (1, 80, T).  device_ms (may be
 * NULL): HIP-event time of the n_timesteps estimator calls and Euler updates on the handle's stream. */
int t3mi_cfm_solve(t3mi_cfm* m, int T, const float* z, const float* mu, const float* mask, const float* spks, const float* cond,
                   int n_timesteps, double cfg_rate, int meanflow, float* out, float* device_ms);


/* ---- the flow module in front of the estimator: speech tokens -> mel (second piece) -----------------------
 * causal_masked_diff_xvec (R/s3gen.R:768-891): input_embedding, spk_embed_affine_layer, the upsampling conformer encoder
 * (upsample_conformer_encoder_full, R/conformer.R:464-593: linear_no_subsampling, pre_lookahead_layer, 6 + 4
 * conformer_encoder_layers with rel_position_attention, upsample_1d), encoder_proj, then causal_cfm.  One utterance per
 * call, every position valid -- what s3gen$inference passes (R/s3gen.R:1006-1011). */
typedef struct t3mi_flow_info {
    int32_t vocab, d_model, heads, d_ff, n_blocks, n_up_blocks, spk_embed_dim, mel;
    int64_t weight_bytes;
} t3mi_flow_info;
/* load_s3gen_weights' flow part (R/s3gen.R:1215-1231; load_conformer_encoder_weights R/conformer.R:604-716) from the unchanged
 * s3gen safetensors file; the handle also serves t3mi_cfm_estimator / t3mi_cfm_solve. */
int t3mi_flow_open(const char* safetensors_path, int device, t3mi_cfm** out);
int t3mi_flow_get_info(const t3mi_cfm* m, t3mi_flow_info* info);
/* upsample_conformer_encoder_full$forward (R/conformer.R:517-578), parity hook: x [T, d_model] (embedded tokens), out [2T, d_model]. */
int t3mi_flow_encoder(t3mi_cfm* m, int T, const float* x, float* out);
/* causal_masked_diff_xvec$forward (R/s3gen.R:805-890): token / prompt_token are 0-indexed ids; prompt_feat [n_feat, mel] (the
 * reference mel, truncated or zero-padded to 2 * n_prompt frames, :857-865); embedding [spk_embed_dim] (the x-vector);
 * noise (1, mel, noise_T), noise_T >= 2 * (n_prompt + n_token) (see t3mi_cfm_solve); finalize = 0 drops the 6 look-ahead
 * frames (:844-846).  out (1, mel, *n_frames) = the generated part, *n_frames = 2 * n_token (- 6).  device_ms (may be
 * NULL): HIP-event time of the encoder and the solve. */
int t3mi_flow_inference(t3mi_cfm* m, const int32_t* token, int n_token, const int32_t* prompt_token, int n_prompt,
                        const float* prompt_feat, int n_feat, const float* embedding, const float* noise, int noise_T,
                        int n_timesteps, double cfg_rate, int meanflow, int finalize, float* out, int cap_frames, int* n_frames,
                        float* device_ms);

/* ---- HiFT vocoder: mel -> waveform (third piece) ----------------------------------------------------------
 * hift_generator as create_s3gen_vocoder configures it (R/hifigan.R:337-627, 788-817): F0 predictor, harmonic-plus-noise
 * source (9 harmonics at 24 kHz), ISTFTNet decoder (upsampling 8 x 5 x 3, residual blocks with Snake activations, 16-point
 * inverse STFT, hop 4): 480 samples per mel frame.  One utterance per call. */
typedef struct t3mi_hift t3mi_hift;
typedef struct t3mi_hift_info {
    int32_t mel, base_channels, f0_channels, n_harmonics, samples_per_frame, sampling_rate, device, reserved_;
    int64_t weight_bytes;
} t3mi_hift_info;
/* create_s3gen_vocoder + load_hifigan_weights (R/hifigan.R:640-817) from the unchanged s3gen safetensors file, keys under
 * `prefix` (NULL = "mel2wav."): weight-normalised convolutions arrive as parametrizations.weight.original0 / original1 pairs and
 * are folded at load (w = g v / ||v||, :649-667). */
int t3mi_hift_open(const char* safetensors_path, const char* prefix, int device, t3mi_hift** out);
int t3mi_hift_close(t3mi_hift* m);
int t3mi_hift_get_info(const t3mi_hift* m, t3mi_hift_info* info);
/* hift_generator$inference(speech_feat, cache_source) (R/hifigan.R:601-626): mel (80, T) -> audio [480 T], clamped
 * to +-0.99 (:577).  The reference draws the harmonics' start phases (uniform(-pi, pi), entry 0 forced to 0) and the source
 * noise (standard normal, [n_harmonics][480 T]) from torch's generator (:237-249): here the caller hands them in (phase NULL:
 * zeros; noise NULL: no noise term).  source_out [480 T] and f0_out [T] (either may be NULL) return the source signal and the
 * predicted F0 (the list elements `source` / `f0` of the reference's result).  cache_source [cache_len] (NULL / 0: none): the
 * source of the previous chunk's tail, written over the first cache_len samples of this call's source before decoding
 * (s[,, 1:cache_len] <- cache_source, :617-620; cache_len <= 480 T).  s3gen$inference then applies its 20 ms fade-in
 * (R/s3gen.R:1047-1049), which stays with the caller.  device_ms (may be NULL): HIP-event time of the pass. */
int t3mi_hift_inference(t3mi_hift* m, int T, const float* mel, const float* phase, const float* noise, const float* cache_source,
                        int cache_len, float* audio, float* source_out, float* f0_out, float* device_ms);

#ifdef __cplusplus
}
#endif
#endif /* T3MI_H */
