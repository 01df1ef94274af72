// Prompt builder of the T3 path on device: replaces prepare_conditioning / prepare_input_embeds
// (R/t3.R:361-403; Turbo R/t3.R:1447-1477) so the caller hands over token ids, a speaker embedding and the
// emotion scalar instead of a [rows, S0, d] float array.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>

#include "safetensors.hpp"

namespace t3mi {

struct PromptModel;   // conditioning-encoder / text-embedding tensors in HBM + scratch rows

// what the builder borrows from the decoder's model
struct PromptEnv {
    int variant = 0;                   // T3MI_VARIANT_*
    int d = 0, vocab = 0, max_pos = 0; // speech table sizes
    const float* speech_emb = nullptr; // [vocab, d]
    const float* speech_pos = nullptr; // [max_pos, d] or nullptr (Turbo)
    int start_token = 0;
    hipStream_t stream = nullptr;
};

struct PromptCond {
    const float* speaker_emb = nullptr;  // host [n_speaker]
    int n_speaker = 0;
    const int32_t* cond_tokens = nullptr; // host [n_cond] 0-indexed speech ids, or nullptr
    int n_cond = 0;
    float emotion = 0.5f;
    int has_emotion = 1;
};

// 0 = loaded, 1 = the checkpoint has no prompt tensors (*out = nullptr, not an error), < 0 = malformed (err filled)
// validate_only: check names, shapes and dtypes, touch no device (*out stays nullptr)
int prompt_load(const SafeTensorsFile& st, int variant, int d, PromptModel** out, std::string& err, bool validate_only = false);
void prompt_free(PromptModel* pm);
// start of a batch of prompt_build calls: sizes the staging arenas for `total_ids` token ids (conditioning + text +
// start / stop + BOS, summed over the batch) and n_utt speaker vectors.  The stream must be drained (the previous
// batch's uploads and kernels done) before this is called.
int prompt_batch_begin(PromptModel* pm, size_t total_ids, int n_utt, std::string& err);
// shape of the prompt these inputs give: rows (2 with CFG on the Llama variant), S0, conditioning length
int prompt_shape(const PromptModel* pm, const PromptEnv& env, const PromptCond& c, int n_text, float cfg_weight, int* rows,
                 int* s0, int* len_cond, std::string& err);
// Caller-owned working memory for one prompt_build call (the entry point that allocates nothing): `rows` holds the
// perceiver's intermediate rows, `ids` the uploaded token ids, `spk` the uploaded speaker vector.  Capacities in elements.
struct PromptScratch {
    float* rows = nullptr;
    size_t rows_floats = 0;
    int* ids = nullptr;
    size_t ids_cap = 0;
    float* spk = nullptr;
    size_t spk_cap = 0;
};
// scratch floats / ids prompt_build needs for these inputs (what a PromptScratch must hold)
void prompt_scratch_need(const PromptModel* pm, const PromptEnv& env, const PromptCond& c, int n_text, size_t* rows_floats,
                         size_t* n_ids);
// writes [rows, s0, d] fp32 at out_dev (asynchronous on env.stream; host inputs are copied before returning).
// ext == nullptr: working memory comes from the model's own (growing, hipMalloc'ed) buffers, sized by prompt_batch_begin.
// ext != nullptr: working memory is *ext, nothing is allocated or kept; the call returns T3MI_ERR_ARG if *ext is too
// small, and the host inputs must stay valid until the stream has drained (the caller synchronises).
int prompt_build(PromptModel* pm, const PromptEnv& env, const PromptCond& c, const int32_t* text_tokens, int n_text,
                 float cfg_weight, float* out_dev, std::string& err, const PromptScratch* ext = nullptr);

}  // namespace t3mi
