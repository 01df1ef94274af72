# MI355X-native T3 decode behind the reference's R API.
# Drop this file into the package's R/ directory (it replaces the bodies of t3_inference /
# t3_inference_cpp, R/t3.R:461-605, 932-1050; signatures and return values are unchanged) and add
#     useDynLib(chatterbox, .registration = TRUE)
# to NAMESPACE (missing in the reference, SURVEY Q18).

.t3mi <- new.env(parent = emptyenv())

#' Path of the T3 safetensors file the reference's loaders read: the same hfhub cache lookup as load_chatterbox
#' (R/tts.R:51-52, 72-79 -> get_model_paths()$t3_cfg, R/download.R:129-150) and load_chatterbox_turbo
#' (R/tts.R:104-106, 127 -> get_turbo_model_paths()$t3_turbo_v1, R/download.R:241-263).  options(chatterbox.t3_path = ,
#' chatterbox.t3_turbo_path = ) override it (a checkpoint outside the hub cache).
t3mi_default_path <- function (variant = c("llama", "turbo"))
{
    variant <- match.arg(variant)
    opt <- getOption(if (variant == "turbo") "chatterbox.t3_turbo_path" else "chatterbox.t3_path")
    if (!is.null(opt)) return(opt)
    path <- if (variant == "turbo") get_turbo_model_paths()$t3_turbo_v1 else get_model_paths()$t3_cfg
    if (is.null(path) || !file.exists(path))
        stop("T3 checkpoint not found: run download_chatterbox_models() (or download_chatterbox_turbo_models()), ",
             "or set options(chatterbox.t3_path = '<file>.safetensors')", call. = FALSE)
    path
}

#' Open (once per file, device and dtype) the native model.  `path = NULL` resolves the file the R model itself was
#' loaded from (t3mi_default_path); the handle is cached, so every t3_inference() call after the first reuses the
#' weights already resident in HBM.
t3mi_model <- function (path = NULL, device = 0L, dtype = c("f32", "bf16"), variant = c("llama", "turbo"),
                        cache = FALSE)
{
    dtype <- match.arg(dtype)
    variant <- match.arg(variant)
    if (is.null(path)) path <- t3mi_default_path(variant)
    if (!is.character(path) || length(path) != 1L || !file.exists(path))
        stop("t3mi_model: '", format(path), "' is not a readable safetensors file", call. = FALSE)
    key <- paste(normalizePath(path), device, dtype, variant)
    if (is.null(.t3mi[[key]])) {
        # cache = TRUE keeps the packed bf16 weights next to the checkpoint in the HF cache ("<file>.t3mi-bf16"): the next
        # session reads half the bytes and converts nothing; a stale or damaged cache is detected and rewritten
        cache_path <- if (isTRUE(cache)) paste0(path, ".t3mi-bf16") else if (is.character(cache)) cache else NULL
        .t3mi[[key]] <- .Call("t3mi_r_open", path, if (variant == "turbo") 1L else 0L, as.integer(device),
                              if (dtype == "bf16") 1L else 0L, cache_path, PACKAGE = "chatterbox")
    }
    .t3mi[[key]]
}

# Seed of the library's counter RNG.  The reference draws from torch's global generator (torch_multinomial, R/t3.R:566), so
# repeated calls differ and torch_manual_seed() governs; the native sampler has its own stream (libtorch's cannot be
# reproduced), so a call without an explicit `seed` takes one from R's RNG: set.seed() governs, repeated calls differ.
.t3mi_seed <- function (seed) if (is.null(seed)) as.double(sample.int(.Machine$integer.max, 1L)) else as.double(seed)

# t3_cond (R/t3.R:103-113) as the plain vectors the native prompt builder takes
.t3mi_cond <- function (cond)
{
    emo <- cond$emotion_adv
    if (inherits(emo, "torch_tensor")) emo <- as.numeric(emo$cpu())
    list(speaker_emb = as.numeric(cond$speaker_emb$cpu()),
         tokens = if (is.null(cond$cond_prompt_speech_tokens)) NULL
                  else as.integer(cond$cond_prompt_speech_tokens$cpu()),
         emotion_adv = if (is.null(emo)) NULL else as.numeric(emo)[1])
}

.t3mi_decode <- function (model, cond, text_tokens, cfg, native, sampler, max_cache_len = 0L)
{
    config <- model$config
    device <- model$text_emb$weight$device
    cfg$stop_token <- as.integer(config$stop_speech_token)
    cfg$start_token <- as.integer(config$start_speech_token)
    cfg$sampler <- as.integer(sampler)
    cfg$max_cache_len <- as.integer(max_cache_len)
    if (is.null(cond$cond_prompt_speech_emb)) {
        # the whole of t3_inference natively: conditioning encoder, perceiver, embeddings, prefill, decode
        # (prepare_input_embeds R/t3.R:372-403 runs on the GPU; the ids stay 0-indexed as tokenized)
        ids <- .Call("t3mi_r_generate_tokens", native, list(.t3mi_cond(cond)),
                     list(as.integer(text_tokens$cpu())), cfg, PACKAGE = "chatterbox")[[1]]
        return(list(ids = ids, device = device, stop = config$stop_speech_token))
    }
    # pre-computed prompt-speech embeddings (not produced by generate()): prompt built in R torch
    if (text_tokens$dim() == 1) text_tokens <- text_tokens$unsqueeze(1)
    text_tokens <- torch::nnf_pad(text_tokens, c(1, 0), value = config$start_text_token)
    text_tokens <- torch::nnf_pad(text_tokens, c(0, 1), value = config$stop_text_token)
    bos <- torch::torch_tensor(matrix(config$start_speech_token, nrow = 1), device = device,
                               dtype = torch::torch_long())
    if (cfg$cfg_weight > 0) {
        text_tokens <- torch::torch_cat(list(text_tokens, text_tokens), dim = 1)
        bos <- torch::torch_cat(list(bos, bos), dim = 1)
    }
    embeds <- torch::with_no_grad(model$prepare_input_embeds(cond, text_tokens, bos, cfg$cfg_weight)$embeds)
    e <- as.array(embeds$to(device = "cpu", dtype = torch::torch_float32()))   # [rows, S0, d], column-major
    ids <- .Call("t3mi_r_generate", native, e, as.integer(dim(e)), cfg, PACKAGE = "chatterbox")
    list(ids = ids, device = device, stop = config$stop_speech_token)
}

#' Same signature and return value as the reference's t3_inference (R/t3.R:461-463, 598-604).
t3_inference <- function (model, cond, text_tokens, max_new_tokens = 1000,
                          temperature = 0.8, cfg_weight = 0.5, top_p = 0.95,
                          min_p = 0.05, repetition_penalty = 1.2,
                          native = t3mi_model(), seed = NULL)
{
    cfg <- list(max_tokens = as.integer(max_new_tokens), temperature = as.double(temperature),
                cfg_weight = as.double(cfg_weight), top_p = as.double(top_p), min_p = as.double(min_p),
                rep_penalty = as.double(repetition_penalty), seed = .t3mi_seed(seed))
    r <- .t3mi_decode(model, cond, text_tokens, cfg, native, sampler = 0L)
    if (length(r$ids) && r$ids[length(r$ids)] == r$stop) message("EOS detected at step ", length(r$ids))
    if (length(r$ids) > 0) torch::torch_tensor(r$ids, dtype = torch::torch_long(), device = r$device)
    else torch::torch_tensor(integer(0), device = r$device)
}

#' Same signature and return value as t3_inference_cpp (R/t3.R:932-935, 1039-1049).
t3_inference_cpp <- function (model, cond, text_tokens, max_new_tokens = 1000,
                              temperature = 0.8, cfg_weight = 0.5, top_p = 0.95,
                              min_p = 0.05, repetition_penalty = 1.2, max_cache_len = 350L,
                              native = t3mi_model(), seed = NULL)
{
    cfg <- list(max_tokens = as.integer(max_new_tokens), temperature = as.double(temperature),
                cfg_weight = as.double(cfg_weight), top_p = as.double(top_p), min_p = as.double(min_p),
                rep_penalty = as.double(repetition_penalty), seed = .t3mi_seed(seed))
    r <- .t3mi_decode(model, cond, text_tokens, cfg, native, sampler = 1L, max_cache_len = max_cache_len)
    ids <- r$ids
    eos <- match(r$stop, ids)
    if (!is.na(eos)) {
        message("EOS detected at step ", eos)
        ids <- ids[seq_len(eos - 1L)]
    }
    if (length(ids) > 0L) torch::torch_tensor(ids, dtype = torch::torch_long(), device = r$device)
    else torch::torch_tensor(integer(0), device = r$device)
}

#' Same signature and return value as t3_inference_traced (R/t3.R:685-688, 863-870): the R sampling rule on a
#' cache of max_cache_len slots, at most max_cache_len - cond_len - 1 tokens (R/t3.R:753), stop token kept.
#' Nothing is traced: the decode step is a replayed HIP graph on either path.
t3_inference_traced <- function (model, cond, text_tokens, max_new_tokens = 1000,
                                 temperature = 0.8, cfg_weight = 0.5, top_p = 0.95,
                                 min_p = 0.05, repetition_penalty = 1.2, max_cache_len = 350L,
                                 native = t3mi_model(), seed = NULL)
{
    cfg <- list(max_tokens = as.integer(max_new_tokens), temperature = as.double(temperature),
                cfg_weight = as.double(cfg_weight), top_p = as.double(top_p), min_p = as.double(min_p),
                rep_penalty = as.double(repetition_penalty), seed = .t3mi_seed(seed))
    r <- suppressWarnings(.t3mi_decode(model, cond, text_tokens, cfg, native, sampler = 0L,
                                       max_cache_len = max_cache_len))
    if (length(r$ids) && r$ids[length(r$ids)] == r$stop) message("EOS detected at step ", length(r$ids))
    if (length(r$ids) > 0L) torch::torch_tensor(r$ids, dtype = torch::torch_long(), device = r$device)
    else torch::torch_tensor(integer(0), device = r$device)
}

#' Same signature and return value as t3_inference_turbo (R/t3.R:1578-1580, 1662-1676): GPT-2 backbone, no CFG,
#' top-k / top-p logit processors; the trailing stop token is removed.
t3_inference_turbo <- function (model, cond, text_tokens, max_new_tokens = 1000,
                                temperature = 0.8, top_k = 1000L, top_p = 0.95,
                                repetition_penalty = 1.2,
                                native = t3mi_model(variant = "turbo"), seed = NULL)
{
    config <- model$config
    device <- model$speech_emb$weight$device
    cfg <- list(max_tokens = as.integer(max_new_tokens), temperature = as.double(temperature),
                cfg_weight = 0, top_k = as.integer(top_k), top_p = as.double(top_p), min_p = 0,
                rep_penalty = as.double(repetition_penalty), sampler = 2L, seed = .t3mi_seed(seed),
                start_token = as.integer(config$start_speech_token),
                stop_token = as.integer(config$stop_speech_token))
    # prompt (speaker row + prompt-token embeddings + text embeddings + BOS, R/t3.R:1447-1477) built on the GPU
    ids <- .Call("t3mi_r_generate_tokens", native, list(.t3mi_cond(cond)), list(as.integer(text_tokens$cpu())),
                 cfg, PACKAGE = "chatterbox")[[1]]
    n <- length(ids)
    if (n > 0L && ids[n] == config$stop_speech_token) {
        message("EOS at step ", n)
        ids <- ids[seq_len(n - 1L)]
    }
    if (length(ids) > 0L) torch::torch_tensor(ids, dtype = torch::torch_long(), device = device)
    else torch::torch_tensor(integer(0), device = device)
}

# the continuous-batching entries of the config list (read by t3mi_r_generate_tokens / t3mi_r_generate_batch)
.t3mi_continuous <- function (cfg, slots, each, poll_steps, n)
{
    if (is.null(slots) && is.null(each)) return(cfg)
    if (!is.null(each) && length(each) != n) stop("max_new_tokens_each needs one entry per utterance")
    cfg$slots <- as.integer(if (is.null(slots)) n else slots)
    cfg$poll_steps <- as.integer(poll_steps)
    cfg$longest_first <- 1L      # start the longest texts first (the results do not depend on the order)
    if (!is.null(each)) cfg$max_tokens_each <- as.integer(each)
    cfg
}

#' Many utterances at once (the batch axis tts_chunked leaves serial, R/tts.R:462-474): `conds` and
#' `text_tokens_list` are lists of the per-sentence arguments of t3_inference; returns a list of token tensors.
#' `slots`: continuous batching (t3mi_generate_continuous) -- the utterances go through that many batch slots and a slot takes the
#' next waiting utterance as soon as its own has ended, instead of one batch that runs until its longest utterance ends;
#' `max_new_tokens_each`: one limit per utterance (at most max_new_tokens).  Utterance i draws from the stream (seed, i) either way.
t3_inference_batch <- function (model, conds, text_tokens_list, max_new_tokens = 1000,
                                temperature = 0.8, cfg_weight = 0.5, top_p = 0.95,
                                min_p = 0.05, repetition_penalty = 1.2,
                                native = t3mi_model(), seed = NULL, slots = NULL, max_new_tokens_each = NULL, poll_steps = 16L)
{
    config <- model$config
    device <- model$text_emb$weight$device
    cfg <- list(max_tokens = as.integer(max_new_tokens), temperature = as.double(temperature),
                cfg_weight = as.double(cfg_weight), top_p = as.double(top_p), min_p = as.double(min_p),
                rep_penalty = as.double(repetition_penalty), sampler = 0L, seed = .t3mi_seed(seed),
                start_token = as.integer(config$start_speech_token),
                stop_token = as.integer(config$stop_speech_token))
    cfg <- .t3mi_continuous(cfg, slots, max_new_tokens_each, poll_steps, length(conds))
    ids <- .Call("t3mi_r_generate_tokens", native, lapply(conds, .t3mi_cond),
                 lapply(text_tokens_list, function (t) as.integer(t$cpu())), cfg, PACKAGE = "chatterbox")
    lapply(ids, function (v) torch::torch_tensor(v, dtype = torch::torch_long(), device = device))
}

#' Turbo counterpart of t3_inference_batch: per-sentence semantics of t3_inference_turbo (trailing stop token removed).
t3_inference_turbo_batch <- function (model, conds, text_tokens_list, max_new_tokens = 1000,
                                      temperature = 0.8, top_k = 1000L, top_p = 0.95, repetition_penalty = 1.2,
                                      native = t3mi_model(variant = "turbo"), seed = NULL, slots = NULL,
                                      max_new_tokens_each = NULL, poll_steps = 16L)
{
    config <- model$config
    device <- model$speech_emb$weight$device
    cfg <- list(max_tokens = as.integer(max_new_tokens), temperature = as.double(temperature),
                cfg_weight = 0, top_k = as.integer(top_k), top_p = as.double(top_p), min_p = 0,
                rep_penalty = as.double(repetition_penalty), sampler = 2L, seed = .t3mi_seed(seed),
                start_token = as.integer(config$start_speech_token),
                stop_token = as.integer(config$stop_speech_token))
    cfg <- .t3mi_continuous(cfg, slots, max_new_tokens_each, poll_steps, length(conds))
    ids <- .Call("t3mi_r_generate_tokens", native, lapply(conds, .t3mi_cond),
                 lapply(text_tokens_list, function (t) as.integer(t$cpu())), cfg, PACKAGE = "chatterbox")
    lapply(ids, function (v) {
        n <- length(v)
        if (n > 0L && v[n] == config$stop_speech_token) v <- v[seq_len(n - 1L)]
        torch::torch_tensor(v, dtype = torch::torch_long(), device = device)
    })
}

#' Warn when the native handle reads another file than the one the R model was loaded from (a fine-tuned checkpoint given
#' to load_chatterbox() by path while t3mi_model() resolved the hub-cache file): the tokens would come from other weights.
.t3mi_check_source <- function (model, native_path)
{
    src <- model$t3_path
    if (!is.null(src) && !is.null(native_path) && file.exists(src) && normalizePath(src) != normalizePath(native_path))
        warning("native T3 weights come from '", native_path, "' but the R model was loaded from '", src,
                "'; pass native = t3mi_model('", src, "')", call. = FALSE)
    invisible(NULL)
}

#' tts_chunked (R/tts.R:455-480) with the sentence loop turned into ONE batched T3 call: same arguments, same
#' sentence split (strsplit(text, "(?<=[.!?])\\s+", perl = TRUE), R/tts.R:462), same per-sentence tokenisation,
#' conditioning and defaults as generate() (R/tts.R:278-293, 341-348), audio concatenated in sentence order.
#' Everything after the speech tokens (drop_invalid_tokens, S3Gen under autocast on CUDA, vocoder: R/tts.R:366-412) is the
#' reference's, including its warning for a sentence that yields no valid token (R/tts.R:368-371).  `devices`: more than
#' one GPU index spreads the sentences over that many worker processes (t3_inference_sharded below).
tts_chunked <- function (model, text, voice, chunk_size = 200, exaggeration = 0.5, cfg_weight = 0.5,
                         temperature = 0.8, top_p = 0.9, autocast = NULL, traced = FALSE, top_k = 1000L,
                         repetition_penalty = 1.2, seed = NULL, devices = 0L,
                         native = t3mi_model(variant = if (isTRUE(model$turbo)) "turbo" else "llama",
                                             device = devices[1L]),
                         s3gen_native = NULL,      # t3mi_s3gen() (R/s3gen_native.R): the waveform side on the MI355X as well
                         slots = NULL)             # batch slots of t3_inference_batch (continuous batching): long texts with sentences of
                                                   # very different length keep every slot busy; NULL = all sentences in one batch
{
    if (!is_loaded(model)) stop("Model not loaded. Call load_chatterbox() first.")
    sentences <- strsplit(text, "(?<=[.!?])\\s+", perl = TRUE)[[1]]
    if (length(sentences) == 0L) return(list(audio = numeric(0), sample_rate = S3GEN_SR))
    device <- model$device
    is_turbo <- isTRUE(model$turbo)
    use_autocast <- if (is.null(autocast)) grepl("^cuda", device) else autocast
    if (is.character(voice)) voice <- create_voice_embedding(model, voice, autocast = use_autocast)
    else if (!inherits(voice, "voice_embedding")) stop("voice must be a voice_embedding object or path to reference audio")
    toks <- lapply(sentences, function (s) {
        ids <- if (is_turbo) tokenize_text_gpt2(model$tokenizer, punc_norm(s)) else tokenize_text(model$tokenizer, s)
        torch::torch_tensor(ids, dtype = torch::torch_long())
    })
    cond <- t3_cond(speaker_emb = voice$ve_embedding, cond_prompt_speech_tokens = voice$cond_prompt_speech_tokens,
                    emotion_adv = if (is_turbo) NULL else exaggeration)
    conds <- rep(list(cond), length(sentences))
    seed <- .t3mi_seed(seed)      # ONE seed for the call: sentence i draws from the stream (seed, i) on whichever GPU decodes it
    message(sprintf("Generating speech tokens for %d sentences in one batch...", length(sentences)))
    speech <- if (length(devices) > 1L)
        t3_inference_sharded(model$t3, conds, toks, devices = devices, turbo = is_turbo, seed = seed,
                             cfg_weight = cfg_weight, temperature = temperature, top_p = top_p, top_k = top_k,
                             repetition_penalty = repetition_penalty)
    else if (is_turbo)
        t3_inference_turbo_batch(model$t3, conds, toks, temperature = temperature, top_k = top_k, top_p = top_p,
                                 repetition_penalty = repetition_penalty, native = native, seed = seed, slots = slots)
    else
        t3_inference_batch(model$t3, conds, toks, cfg_weight = cfg_weight, temperature = temperature, top_p = top_p,
                           native = native, seed = seed, slots = slots)
    all_audio <- numeric(0)
    for (i in seq_along(sentences)) {
        tk <- drop_invalid_tokens(speech[[i]])
        if (length(tk) == 0L) {
            warning("No valid speech tokens generated")              # R/tts.R:368-371
            next
        }
        vals <- as.integer(tk)
        if (is_turbo) vals <- c(vals, rep(S3GEN_SIL, 3L))
        st <- torch::torch_tensor(vals, dtype = torch::torch_long())$unsqueeze(1L)$to(device = device)
        if (!is.null(s3gen_native)) {     # no R-torch op between the ids and the samples; draws from R's RNG (set.seed governs)
            audio <- s3gen_inference_native(st, voice$ref_dict, finalize = TRUE, meanflow = is_turbo, native = s3gen_native)[[1]]
            all_audio <- c(all_audio, as.numeric(audio$squeeze()$cpu()))
            next
        }
        synth <- function ()
            model$s3gen$inference(speech_tokens = st, ref_dict = voice$ref_dict, finalize = TRUE, traced = traced,
                                  n_cfm_timesteps = if (is_turbo) 2L else NULL)[[1]]
        audio <- torch::with_no_grad({
            if (use_autocast) torch::with_autocast(device_type = "cuda", synth())    # R/tts.R:389-401
            else synth()
        })
        all_audio <- c(all_audio, as.numeric(audio$squeeze()$cpu()))
    }
    list(audio = all_audio, sample_rate = S3GEN_SR)
}

#' The utterance sharder of SURVEY §8e on the R side: the loop tts_chunked runs serially (R/tts.R:466-474) spread over the
#' GPUs of one node.  One worker PROCESS per device (parallel::makeCluster; each opens its own t3mi_model(device = i) --
#' weights replicated, 2 GB fp32 / 1 GB bf16 of 288 GB), utterance u goes to worker floor(u * G / U) (contiguous blocks, the
#' partition of chatterbox_amd/shard.py), every worker decodes its block as ONE batched call with first_utterance_id = the
#' block's first global index, so a sampled utterance draws from the stream (seed, u) whatever the number of GPUs; the only
#' exchange is the integer token vectors coming back, in utterance order.  No collective, nothing crosses xGMI.  Workers and
#' their handles are cached for the session (weights stay resident); `t3mi_cluster_stop()` ends them.
t3_inference_sharded <- function (model, conds, text_tokens_list, devices = 0L, turbo = FALSE, seed = NULL,
                                  max_new_tokens = 1000, temperature = 0.8, cfg_weight = 0.5, top_p = 0.95, min_p = 0.05,
                                  top_k = 1000L, repetition_penalty = 1.2, dtype = "f32", path = NULL)
{
    U <- length(conds)
    G <- length(devices)
    stopifnot(U > 0L, G > 0L, length(text_tokens_list) == U)
    config <- model$config
    out_device <- model$speech_emb$weight$device
    variant <- if (turbo) "turbo" else "llama"
    if (is.null(path)) path <- t3mi_default_path(variant)
    cfg <- list(max_tokens = as.integer(max_new_tokens), temperature = as.double(temperature),
                cfg_weight = if (turbo) 0 else as.double(cfg_weight), top_p = as.double(top_p),
                min_p = if (turbo) 0 else as.double(min_p), top_k = as.integer(top_k),
                rep_penalty = as.double(repetition_penalty), sampler = if (turbo) 2L else 0L,
                start_token = as.integer(config$start_speech_token), stop_token = as.integer(config$stop_speech_token),
                seed = .t3mi_seed(seed))
    plain_conds <- lapply(conds, .t3mi_cond)                       # plain vectors: they travel to the workers
    plain_text <- lapply(text_tokens_list, function (t) as.integer(t$cpu()))
    owner <- (seq_len(U) - 1L) * G %/% U                            # utterance u (0-based) -> worker floor(u * G / U)
    cl <- .t3mi_cluster(devices)
    jobs <- lapply(seq_len(G), function (g) {
        mine <- which(owner == g - 1L)
        list(device = as.integer(devices[g]), idx = mine, conds = plain_conds[mine], text = plain_text[mine],
             cfg = c(cfg, list(first_utterance_id = as.double(if (length(mine)) mine[1L] - 1L else 0))),
             path = path, variant = variant, dtype = dtype)
    })
    parts <- parallel::clusterApply(cl, jobs, function (job) {
        if (!length(job$idx)) return(list())
        h <- chatterbox:::t3mi_model(job$path, device = job$device, dtype = job$dtype, variant = job$variant)
        .Call("t3mi_r_generate_tokens", h, job$conds, job$text, job$cfg, PACKAGE = "chatterbox")
    })
    ids <- vector("list", U)
    for (g in seq_len(G)) ids[jobs[[g]]$idx] <- parts[[g]]
    lapply(ids, function (v) {
        n <- length(v)
        if (turbo && n > 0L && v[n] == config$stop_speech_token) v <- v[seq_len(n - 1L)]      # R/t3.R:1667-1671
        torch::torch_tensor(v, dtype = torch::torch_long(), device = out_device)
    })
}

.t3mi_cluster <- function (devices)
{
    key <- paste("cluster", paste(devices, collapse = ","))
    if (is.null(.t3mi[[key]])) {
        cl <- parallel::makeCluster(length(devices))               # PSOCK workers: fresh processes, no GPU state inherited
        parallel::clusterEvalQ(cl, library(chatterbox))
        .t3mi[[key]] <- cl
    }
    .t3mi[[key]]
}

t3mi_cluster_stop <- function ()
{
    for (k in grep("^cluster ", ls(.t3mi), value = TRUE)) {
        parallel::stopCluster(.t3mi[[k]])
        rm(list = k, envir = .t3mi)
    }
    invisible(NULL)
}
