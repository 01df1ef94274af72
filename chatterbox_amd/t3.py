"""Host-side mirror of the reference's T3 interface over the t3mi C ABI.

The reference's host language is R (`t3_inference`, `t3_inference_cpp`,
R/t3.R:461-605, 932-1050); R is not installed in the build image, so this module
is the Python stand-in that keeps the same function names, argument names,
defaults, return convention (0-indexed speech-token ids) and error behaviour.  The
R glue that binds the same C entry points lives in r-package/ (INTEGRATION.md).

Two call forms, told apart by the second argument:

* `t3_inference(model, cond, text_tokens, ...)` with `cond = t3_cond(...)` -- the
  reference's own signature; the prompt (conditioning encoder, perceiver, text and
  speech embeddings: R/t3.R:361-403) is built on the GPU by the library.
* `t3_inference(model, embeds, ...)` with the output of `prepare_input_embeds`
  (float32 `[rows, S0, 1024]`, rows = 2 with CFG) for callers that build the
  prompt themselves, and for checkpoints without the `cond_enc.*` tensors.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

__all__ = ["T3Model", "load_t3", "load_t3_turbo", "t3_inference", "t3_inference_cpp", "t3_inference_traced",
           "t3_inference_batch", "t3_inference_turbo", "t3_inference_turbo_batch", "Session", "T3Cond", "t3_cond",
           "prepare_input_embeds", "inspect_checkpoint", "split_sentences", "drop_invalid_tokens", "tts_chunked", "SPEECH_VOCAB_SIZE"]

_DT = {"f32": _lib.W_F32, "fp32": _lib.W_F32, "float32": _lib.W_F32, "bf16": _lib.W_BF16,
       "bfloat16": _lib.W_BF16}


class T3Model:
    """Handle to weights resident on one MI355X (replaces the R `t3_model` object for
    the decode path: load_t3_weights R/t3.R:1061-1145 + .get_cpp_weights :884-912)."""

    def __init__(self, path: str, device: int = 0, dtype: str = "f32", variant: int = _lib.VARIANT_LLAMA,
                 start_speech_token: int = 6561, stop_speech_token: int = 6562, cache=None):
        """cache: None / False = none; True = "<path>.t3mi-bf16" next to the checkpoint; a string = that file.  Only
        bf16 storage uses it (t3mi_open_cached): the packed bf16 matrices are read back instead of converted."""
        self._lib = _lib.load()
        h = C.c_void_p()
        cpath = None
        if cache:
            cpath = (path + ".t3mi-bf16") if cache is True else str(cache)
        _lib.check(self._lib.t3mi_open_cached(path.encode(), variant, int(device), _DT[dtype],
                                              cpath.encode() if cpath else None, C.byref(h)))
        self._h = h
        info = _lib.Info()
        _lib.check(self._lib.t3mi_get_info(self._h, C.byref(info)))
        self.info = info
        self.d_model, self.vocab, self.n_layers = info.d_model, info.vocab, info.n_layers
        self.variant = int(variant)
        # R/t3.R:19-20 (overridable for reduced-vocabulary test checkpoints)
        self.start_speech_token, self.stop_speech_token = int(start_speech_token), int(stop_speech_token)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.t3mi_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def load_t3(path: str, device: int = 0, dtype: str = "f32", **kw) -> T3Model:
    return T3Model(path, device, dtype, **kw)


def inspect_checkpoint(path: str, variant: int = _lib.VARIANT_LLAMA) -> dict:
    """t3mi_inspect: validate a checkpoint without a GPU (header, tensor names / shapes / dtypes) and return its
    dimensions; raises T3miError(ERR_IO) naming the offending tensor for a truncated, mis-shaped or foreign file."""
    info = _lib.Info()
    _lib.check(_lib.load().t3mi_inspect(path.encode(), int(variant), C.byref(info)))
    return {k: getattr(info, k) for k, _ in _lib.Info._fields_}


def load_t3_turbo(path: str, device: int = 0, dtype: str = "f32", **kw) -> T3Model:
    """Turbo T3 (GPT-2 backbone): t3_model_turbo + load_t3_turbo_weights, R/t3.R:1347-1557."""
    return T3Model(path, device, dtype, variant=_lib.VARIANT_TURBO, **kw)


def _seed(seed) -> int:
    """Seed of the library's counter RNG.  None = drawn from numpy's global generator, so `np.random.seed()` governs
    sampled output and repeated calls differ -- the reference draws from torch's global generator (R/t3.R:566) and
    behaves the same way under torch_manual_seed(); an int = that reproducible stream."""
    return int(np.random.randint(0, 2**31 - 1)) if seed is None else int(seed)


def _params(model: T3Model, max_new_tokens, temperature, cfg_weight, top_p, min_p, repetition_penalty,
            sampler, max_cache_len=0, seed=0, mask_stop=False, top_k=0, first_utterance_id=0, drop_invalid=None) -> _lib.Params:
    p = _lib.Params()
    model._lib.t3mi_default_params(C.byref(p))
    p.max_new_tokens = int(max_new_tokens)
    p.temperature = float(temperature)
    p.cfg_weight = float(cfg_weight)
    p.top_p = float(top_p)
    p.min_p = float(min_p)
    p.repetition_penalty = float(repetition_penalty)
    p.start_token = int(model.start_speech_token)
    p.stop_token = int(model.stop_speech_token)
    p.sampler = int(sampler)
    p.top_k = int(top_k)
    p.max_cache_len = int(max_cache_len)
    p.mask_stop = 1 if mask_stop else 0
    p.seed = _seed(seed)
    p.first_utterance_id = int(first_utterance_id)
    # drop_invalid: True = SPEECH_VOCAB_SIZE (6561, R/s3tokenizer.R:7), an int = that limit, None / False = raw stream
    p.drop_invalid_from = 0 if not drop_invalid else (SPEECH_VOCAB_SIZE if drop_invalid is True else int(drop_invalid))
    return p


class T3Cond:
    """t3_cond (R/t3.R:103-113): speaker embedding [256], optional conditioning speech tokens (0-indexed),
    emotion / exaggeration scalar (None = NULL).  Keeps the numpy arrays alive for the C struct."""

    def __init__(self, speaker_emb, cond_prompt_speech_tokens=None, emotion_adv: Optional[float] = 0.5):
        self.speaker_emb = np.ascontiguousarray(speaker_emb, dtype=np.float32).reshape(-1)
        self.cond_prompt_speech_tokens = (None if cond_prompt_speech_tokens is None else
                                          np.ascontiguousarray(cond_prompt_speech_tokens, dtype=np.int32).reshape(-1))
        self.emotion_adv = emotion_adv

    def c_struct(self) -> _lib.Cond:
        c = _lib.Cond()
        c.speaker_emb = self.speaker_emb.ctypes.data_as(C.POINTER(C.c_float))
        c.n_speaker = int(self.speaker_emb.size)
        if self.cond_prompt_speech_tokens is not None and self.cond_prompt_speech_tokens.size:
            c.cond_prompt_speech_tokens = self.cond_prompt_speech_tokens.ctypes.data_as(C.POINTER(C.c_int32))
            c.n_cond_tokens = int(self.cond_prompt_speech_tokens.size)
        c.emotion_adv = float(self.emotion_adv) if self.emotion_adv is not None else 0.0
        c.has_emotion_adv = 0 if self.emotion_adv is None else 1
        return c


def t3_cond(speaker_emb, cond_prompt_speech_tokens=None, cond_prompt_speech_emb=None, emotion_adv=0.5) -> T3Cond:
    """R/t3.R:103-113.  Pre-computed prompt-speech embeddings are not part of this path (generate() never
    passes them: R/tts.R:296-302)."""
    if cond_prompt_speech_emb is not None:
        raise NotImplementedError("cond_prompt_speech_emb: pass cond_prompt_speech_tokens (as generate() does)")
    return T3Cond(speaker_emb, cond_prompt_speech_tokens, emotion_adv)


def _text(text_tokens) -> np.ndarray:
    return np.ascontiguousarray(text_tokens, dtype=np.int32).reshape(-1)     # R/t3.R:469-471: 1-D or [1, T]


def prepare_input_embeds(model: T3Model, cond: T3Cond, text_tokens, cfg_weight: float = 0.0):
    """t3_model$prepare_input_embeds as t3_inference calls it (R/t3.R:473-494; Turbo :1593-1600): text start /
    stop tokens added (Llama variant), BOS speech token appended, CFG row pair with the text span of row 2
    zeroed.  Returns (embeds [rows, S0, d] float32, len_cond).  Parity hook; t3_inference builds the same
    prompt without the round trip to the host."""
    tt = _text(text_tokens)
    cs = cond.c_struct()
    rows, s0, lc = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    _lib.check(model._lib.t3mi_prompt_shape(model._h, C.byref(cs), tt.size, float(cfg_weight), C.byref(rows),
                                            C.byref(s0), C.byref(lc)))
    out = np.zeros((rows.value, s0.value, model.d_model), dtype=np.float32)
    _lib.check(model._lib.t3mi_prepare_input_embeds(
        model._h, C.byref(cs), tt.ctypes.data_as(C.POINTER(C.c_int32)), tt.size, float(cfg_weight),
        model.start_speech_token, out.ctypes.data_as(C.POINTER(C.c_float)), out.size))
    return out, int(lc.value)


def _generate_tokens(model: T3Model, conds: Sequence[T3Cond], texts: Sequence, p: _lib.Params, cap: int) -> List[np.ndarray]:
    n_utt = len(conds)
    cs = (_lib.Cond * n_utt)(*[c.c_struct() for c in conds])
    tts = [_text(t) for t in texts]
    ptrs = (C.POINTER(C.c_int32) * n_utt)(*[t.ctypes.data_as(C.POINTER(C.c_int32)) for t in tts])
    nt = (C.c_int32 * n_utt)(*[t.size for t in tts])
    out = np.zeros((n_utt, cap), dtype=np.int32)
    n_out = (C.c_int32 * n_utt)()
    rc = _lib.check(model._lib.t3mi_generate_batch_tokens(
        model._h, n_utt, cs, ptrs, nt, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_int32)), cap, n_out))
    if rc == _lib.WARN_CACHE_FULL:
        import warnings
        warnings.warn("KV cache full, stopping generation")       # src/t3_decode.cpp:395
    return [out[u, : n_out[u]].astype(np.int64) for u in range(n_utt)]


def _as_embeds(model: T3Model, embeds, cfg_weight: float) -> np.ndarray:
    e = np.ascontiguousarray(embeds, dtype=np.float32)
    if e.ndim == 2:
        e = e[None]
    if e.ndim != 3 or e.shape[2] != model.d_model:
        raise ValueError(f"embeds must be [rows, S0, {model.d_model}], got {e.shape}")
    if cfg_weight > 0.0 and e.shape[0] != 2:
        raise ValueError("cfg_weight > 0 needs the cond/uncond pair: embeds[2, S0, d] (R/t3.R:480-491)")
    if cfg_weight <= 0.0 and e.shape[0] != 1:
        e = e[:1].copy()       # R/t3.R:523: only the first row is used without CFG
    return e


def t3_inference(model: T3Model, cond, text_tokens=None, max_new_tokens: int = 1000, temperature: float = 0.8,
                 cfg_weight: float = 0.5, top_p: float = 0.95, min_p: float = 0.05,
                 repetition_penalty: float = 1.2, seed: Optional[int] = None) -> np.ndarray:
    """R/t3.R:461-605.  Returns 0-indexed speech tokens, INCLUDING the stop token if it
    was sampled (Q14).  `top_p = 0` is the deterministic (greedy) mode (Q6).
    `cond` is a T3Cond (then `text_tokens` are the text ids, as in the reference) or a prompt
    embedding array [rows, S0, d] (then `text_tokens` stays None)."""
    p = _params(model, max_new_tokens, temperature, cfg_weight, top_p, min_p, repetition_penalty,
                _lib.SAMPLER_R, 0, seed)
    if isinstance(cond, T3Cond):
        if text_tokens is None:
            raise ValueError("t3_inference(model, cond, text_tokens): text_tokens missing")
        return _generate_tokens(model, [cond], [text_tokens], p, int(max_new_tokens))[0]
    if text_tokens is not None:
        raise ValueError("t3_inference(model, embeds): text_tokens only goes with a T3Cond")
    e = _as_embeds(model, cond, cfg_weight)
    out = np.zeros(int(max_new_tokens), dtype=np.int32)
    n = C.c_int32(0)
    _lib.check(model._lib.t3mi_generate(
        model._h, e.ctypes.data_as(C.POINTER(C.c_float)), e.shape[0], e.shape[1], C.byref(p),
        out.ctypes.data_as(C.POINTER(C.c_int32)), out.shape[0], C.byref(n)))
    return out[: n.value].astype(np.int64)


def t3_inference_cpp(model: T3Model, cond, text_tokens=None, max_new_tokens: int = 1000, temperature: float = 0.8,
                     cfg_weight: float = 0.5, top_p: float = 0.95, min_p: float = 0.05,
                     repetition_penalty: float = 1.2, max_cache_len: int = 350, seed: Optional[int] = None) -> np.ndarray:
    """R/t3.R:932-1050 (backend = "cpp"): 350-slot cache, cpp sampling rule (Q3/Q4),
    result stripped at the first stop token (R/t3.R:1039-1045).  Same two call forms as t3_inference."""
    p = _params(model, max_new_tokens, temperature, cfg_weight, top_p, min_p, repetition_penalty,
                _lib.SAMPLER_CPP, max_cache_len, seed)
    if isinstance(cond, T3Cond):
        toks = _generate_tokens(model, [cond], [text_tokens], p, int(max_new_tokens))[0]
    else:
        e = _as_embeds(model, cond, cfg_weight)
        out = np.zeros(int(max_new_tokens), dtype=np.int32)
        n = C.c_int32(0)
        rc = _lib.check(model._lib.t3mi_generate(
            model._h, e.ctypes.data_as(C.POINTER(C.c_float)), e.shape[0], e.shape[1], C.byref(p),
            out.ctypes.data_as(C.POINTER(C.c_int32)), out.shape[0], C.byref(n)))
        if rc == _lib.WARN_CACHE_FULL:
            import warnings
            warnings.warn("KV cache full, stopping generation")       # src/t3_decode.cpp:395
        toks = out[: n.value].astype(np.int64)
    stop = np.nonzero(toks == model.stop_speech_token)[0]
    if stop.size:
        toks = toks[: stop[0]]
    return toks


def t3_inference_traced(model: T3Model, cond, text_tokens=None, max_new_tokens: int = 1000, temperature: float = 0.8,
                        cfg_weight: float = 0.5, top_p: float = 0.95, min_p: float = 0.05,
                        repetition_penalty: float = 1.2, max_cache_len: int = 350, seed: Optional[int] = None) -> np.ndarray:
    """R/t3.R:685-871 (`traced = TRUE`): the arithmetic and sampling rule of t3_inference on a fixed
    cache of `max_cache_len` slots; generation is limited to `max_cache_len - cond_len - 1` tokens
    (R/t3.R:753) without a warning, and the stop token stays in the result.  There is nothing to
    trace here (the decode step is a replayed HIP graph either way); the function exists so that
    `generate(traced = TRUE)` callers keep their semantics."""
    import warnings
    p = _params(model, max_new_tokens, temperature, cfg_weight, top_p, min_p, repetition_penalty,
                _lib.SAMPLER_R, max_cache_len, seed)
    if isinstance(cond, T3Cond):
        if text_tokens is None:
            raise ValueError("t3_inference_traced(model, cond, text_tokens): text_tokens missing")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return _generate_tokens(model, [cond], [text_tokens], p, int(max_new_tokens))[0]
    if text_tokens is not None:
        raise ValueError("t3_inference_traced(model, embeds): text_tokens only goes with a T3Cond")
    e = _as_embeds(model, cond, cfg_weight)
    out = np.zeros(int(max_new_tokens), dtype=np.int32)
    n = C.c_int32(0)
    _lib.check(model._lib.t3mi_generate(
        model._h, e.ctypes.data_as(C.POINTER(C.c_float)), e.shape[0], e.shape[1], C.byref(p),
        out.ctypes.data_as(C.POINTER(C.c_int32)), out.shape[0], C.byref(n)))
    return out[: n.value].astype(np.int64)


def t3_inference_batch(model: T3Model, embeds_list: Sequence, text_tokens_list: Optional[Sequence] = None,
                       max_new_tokens: int = 1000,
                       temperature: float = 0.8, cfg_weight: float = 0.5, top_p: float = 0.95,
                       min_p: float = 0.05, repetition_penalty: float = 1.2, seed: Optional[int] = None,
                       mask_stop: bool = False, first_utterance_id: int = 0, drop_invalid=None) -> List[np.ndarray]:
    """Independent utterances decoded together (the batch axis tts_chunked leaves serial,
    R/tts.R:462-474).  Same per-utterance semantics as t3_inference.  Utterance u of the call draws from the
    RNG stream (seed, first_utterance_id + u): a batch cut into several calls (or sharded over GPUs,
    chatterbox_amd.shard) samples the same tokens as the one-call batch."""
    p = _params(model, max_new_tokens, temperature, cfg_weight, top_p, min_p, repetition_penalty,
                _lib.SAMPLER_R, 0, seed, mask_stop, first_utterance_id=first_utterance_id, drop_invalid=drop_invalid)
    if text_tokens_list is not None:        # embeds_list holds T3Cond objects, one per utterance
        return _generate_tokens(model, list(embeds_list), list(text_tokens_list), p, int(max_new_tokens))
    es = [_as_embeds(model, e, cfg_weight) for e in embeds_list]
    rpu = es[0].shape[0]
    n_utt = len(es)
    ptrs = (C.c_void_p * n_utt)(*[e.ctypes.data for e in es])
    s0 = (C.c_int32 * n_utt)(*[e.shape[1] for e in es])
    out = np.zeros((n_utt, int(max_new_tokens)), dtype=np.int32)
    n_out = (C.c_int32 * n_utt)()
    _lib.check(model._lib.t3mi_generate_batch(
        model._h, n_utt, rpu, ptrs, s0, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_int32)),
        out.shape[1], n_out))
    return [out[u, : n_out[u]].astype(np.int64) for u in range(n_utt)]


def t3_inference_turbo_batch(model: T3Model, embeds_list: Sequence, text_tokens_list: Optional[Sequence] = None,
                             max_new_tokens: int = 1000,
                             temperature: float = 0.8, top_k: int = 1000, top_p: float = 0.95,
                             repetition_penalty: float = 1.2, seed: Optional[int] = None, mask_stop: bool = False,
                             keep_stop: bool = False, first_utterance_id: int = 0, drop_invalid=None) -> List[np.ndarray]:
    """Independent Turbo utterances decoded together; per-utterance semantics of t3_inference_turbo."""
    if model.variant != _lib.VARIANT_TURBO:
        raise ValueError("t3_inference_turbo needs a model opened with load_t3_turbo")
    p = _params(model, max_new_tokens, temperature, 0.0, top_p, 0.0, repetition_penalty,
                _lib.SAMPLER_TURBO, 0, seed, mask_stop, top_k, first_utterance_id=first_utterance_id, drop_invalid=drop_invalid)
    cap = int(max_new_tokens) + 1            # first token from the prefill + max_new_tokens more (R/t3.R:1614-1658)
    if text_tokens_list is not None:         # embeds_list holds T3Cond objects
        toks = _generate_tokens(model, list(embeds_list), list(text_tokens_list), p, cap)
        n_utt = len(toks)
    else:
        es = [_as_embeds(model, e, 0.0) for e in embeds_list]
        n_utt = len(es)
        ptrs = (C.c_void_p * n_utt)(*[e.ctypes.data for e in es])
        s0 = (C.c_int32 * n_utt)(*[e.shape[1] for e in es])
        out = np.zeros((n_utt, cap), dtype=np.int32)
        n_out = (C.c_int32 * n_utt)()
        _lib.check(model._lib.t3mi_generate_batch(
            model._h, n_utt, 1, ptrs, s0, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_int32)), cap, n_out))
        toks = [out[u, : n_out[u]].astype(np.int64) for u in range(n_utt)]
    res = []
    for u in range(n_utt):
        t = toks[u]
        if not keep_stop and t.size and t[-1] == model.stop_speech_token:
            t = t[:-1]                       # R/t3.R:1667-1671
        res.append(t)
    return res


def t3_inference_turbo(model: T3Model, cond, text_tokens=None, max_new_tokens: int = 1000, temperature: float = 0.8,
                       top_k: int = 1000, top_p: float = 0.95, repetition_penalty: float = 1.2,
                       seed: Optional[int] = None) -> np.ndarray:
    """R/t3.R:1578-1677: no CFG, no position term on the generated tokens, top-k / top-p logit processors; a
    trailing stop token is removed (:1667-1671).  `cond` is a T3Cond with `text_tokens` (the reference's
    signature) or a [1, S0, d] prompt array."""
    if isinstance(cond, T3Cond):
        return t3_inference_turbo_batch(model, [cond], [text_tokens], max_new_tokens, temperature, top_k, top_p,
                                        repetition_penalty, seed)[0]
    return t3_inference_turbo_batch(model, [cond], None, max_new_tokens, temperature, top_k, top_p,
                                    repetition_penalty, seed)[0]


SPEECH_VOCAB_SIZE = 6561     # R/s3tokenizer.R:7 (ids >= this are the start / stop / padding specials)


def drop_invalid_tokens(tokens, speech_vocab_size: int = SPEECH_VOCAB_SIZE) -> np.ndarray:
    """R/s3tokenizer.R:678-687: keep the ids below SPEECH_VOCAB_SIZE, order preserved (generate() applies it to
    the T3 output before S3Gen, R/tts.R:366)."""
    t = np.asarray(tokens, dtype=np.int64).reshape(-1)
    return t[t < int(speech_vocab_size)]


def split_sentences(text: str) -> List[str]:
    """The sentence split of tts_chunked, R/tts.R:462: strsplit(text, "(?<=[.!?])\\s+", perl = TRUE)[[1]].
    R's strsplit returns character(0) for "" and drops the empty piece after a match at the end of the string."""
    import re
    if text == "":
        return []
    # re.ASCII: R's strsplit(perl = TRUE) runs PCRE without UCP, where \s is the ASCII white space only -- a no-break space
    # (U+00A0) or an em space after a full stop does NOT split there
    parts = re.split(r"(?<=[.!?])\s+", text, flags=re.ASCII)
    if parts and parts[-1] == "":
        parts = parts[:-1]
    return parts


def tts_chunked(model: T3Model, text: str, cond: T3Cond, tokenize, chunk_size: int = 200, batched: bool = True,
                drop_invalid: bool = True, s3gen=None, ref_dict: Optional[dict] = None, slots: Optional[int] = None, **kw) -> dict:
    """tts_chunked (R/tts.R:455-480) up to the speech tokens: the text is split into sentences exactly as the
    reference does, every sentence is tokenised by `tokenize(sentence) -> text token ids` (the reference's BPE
    tokenizer, or punc_norm + the GPT-2 tokenizer on Turbo: R/tts.R:278-285 -- host string work, out of scope
    here), and the sentences go through T3 either as ONE batch (`batched`, the default: the sentences are
    independent) or one after the other like the reference's loop (R/tts.R:466-474).  In both orders
    every sentence draws from its own RNG stream (seed, sentence index); the tokens are the same whenever both orders
    run the same kernel regime (greedy fp32: identical to the oracle in both, test_tts_chunked_batch_equals_serial),
    otherwise a draw within rounding of a CDF step may differ (8 rows run MFMA GEMMs, 2 rows GEMVs: other summation
    order).  `chunk_size` is accepted
    and unused, as in the reference.  `slots`: the batched order through that many slots with continuous batching
    (t3_inference_continuous), for long texts whose sentences differ much in length.  Keyword arguments are those of t3_inference / t3_inference_turbo
    (generate() passes cfg_weight, temperature, top_p, and top_k / repetition_penalty on Turbo: R/tts.R:303-360).
    Returns {"sentences": [...], "speech_tokens": [array per sentence]} with drop_invalid_tokens applied
    (R/tts.R:366).  With `s3gen` (one (flow, vocoder) pair of chatterbox_amd.s3gen, or a list of pairs) and the voice's
    `ref_dict` the waveform side runs too (R/tts.R:365-412, s3gen.synthesize): the result then also carries "audio" -- the
    sentences' samples concatenated, as R/tts.R:474 does -- and "sample_rate" (24000)."""
    sentences = split_sentences(text)
    if not sentences:
        return {"sentences": [], "speech_tokens": []}
    kw["seed"] = _seed(kw.get("seed"))      # ONE seed per call: sentence i draws from the stream (seed, i) in both orders
    turbo = model.variant == _lib.VARIANT_TURBO
    toks = [np.asarray(tokenize(s), dtype=np.int32).reshape(-1) for s in sentences]
    fn = t3_inference_turbo_batch if turbo else t3_inference_batch
    # drop_invalid_tokens runs on the device at the end of the generate call (t3mi_params.drop_invalid_from)
    di = (drop_invalid if drop_invalid is not True else SPEECH_VOCAB_SIZE) if drop_invalid else None
    if batched and slots:
        # continuous batching: the sentences go through `slots` batch slots, a slot takes the next sentence when its own has ended
        out = t3_inference_continuous(model, [cond] * len(sentences), toks, slots=int(slots), drop_invalid=di, **kw)
    elif batched:
        out = fn(model, [cond] * len(sentences), toks, drop_invalid=di, **kw)
    else:
        out = [fn(model, [cond], [t], first_utterance_id=i, drop_invalid=di, **kw)[0] for i, t in enumerate(toks)]
    res = {"sentences": sentences, "speech_tokens": out}
    if s3gen is not None:
        from . import s3gen as _s3
        if ref_dict is None:
            raise ValueError("tts_chunked: s3gen needs the voice's ref_dict (prompt_token, prompt_feat, embedding)")
        wavs = _s3.synthesize(s3gen, out, ref_dict, seed=kw["seed"], turbo=turbo)
        res["audio"] = np.concatenate(wavs) if wavs else np.zeros(0, np.float32)
        res["sample_rate"] = _s3.S3GEN_SR
    return res


class Session:
    """The decode loop in pieces (t3mi_session_*): prefill once, then enqueue steps.
    Used by bench.py (timed region = decode steps only) and by the parity tests
    (raw logits / filtered distribution of each step, teacher forcing)."""

    def __init__(self, model: T3Model, embeds_list: Sequence, max_new_tokens: int = 1000,
                 temperature: float = 0.8, cfg_weight: float = 0.5, top_p: float = 0.95,
                 min_p: float = 0.05, repetition_penalty: float = 1.2, sampler: int = _lib.SAMPLER_R,
                 max_cache_len: int = 0, seed: int = 0, mask_stop: bool = False,
                 forced: Optional[np.ndarray] = None, device_ptrs: Optional[Sequence[Tuple[int, int]]] = None,
                 top_k: int = 0, first_utterance_id: int = 0, text_tokens_list: Optional[Sequence] = None):
        self.model = model
        L = model._lib
        if model.variant == _lib.VARIANT_TURBO:
            sampler, cfg_weight, min_p = _lib.SAMPLER_TURBO, 0.0, 0.0
        p = _params(model, max_new_tokens, temperature, cfg_weight, top_p, min_p, repetition_penalty,
                    sampler, max_cache_len, seed, mask_stop, top_k, first_utterance_id)
        self.cfg_weight = float(cfg_weight)
        if text_tokens_list is not None:
            # embeds_list holds T3Cond objects: the prompts are built on the device (t3mi_session_begin_tokens)
            conds, tts = list(embeds_list), [_text(t) for t in text_tokens_list]
            self.n_utt = len(conds)
            self.rpu = 2 if (cfg_weight > 0 and model.variant != _lib.VARIANT_TURBO) else 1
            cs = (_lib.Cond * self.n_utt)(*[c.c_struct() for c in conds])
            tp = (C.POINTER(C.c_int32) * self.n_utt)(*[t.ctypes.data_as(C.POINTER(C.c_int32)) for t in tts])
            nt = (C.c_int32 * self.n_utt)(*[t.size for t in tts])
            self._keep = (conds, tts)
            _lib.check(L.t3mi_session_begin_tokens(model._h, self.n_utt, cs, tp, nt, C.byref(p)))
            self.max_new_tokens = int(max_new_tokens) + (1 if model.variant == _lib.VARIANT_TURBO else 0)
            return
        if device_ptrs is not None:
            # [(device_address, s0)] of prompts already resident in HBM, each [rpu, s0, d] float32
            self.n_utt = len(device_ptrs)
            self.rpu = 2 if cfg_weight > 0 else 1
            ptrs = (C.c_void_p * self.n_utt)(*[int(a) for a, _ in device_ptrs])
            s0 = (C.c_int32 * self.n_utt)(*[int(s) for _, s in device_ptrs])
            on_dev = 1
            self._keep = None
        else:
            es = [_as_embeds(model, e, cfg_weight) for e in embeds_list]
            self._keep = es
            self.n_utt, self.rpu = len(es), es[0].shape[0]
            ptrs = (C.c_void_p * self.n_utt)(*[e.ctypes.data for e in es])
            s0 = (C.c_int32 * self.n_utt)(*[e.shape[1] for e in es])
            on_dev = 0
        _lib.check(L.t3mi_session_begin(model._h, self.n_utt, self.rpu, ptrs, s0, C.byref(p), on_dev))
        self.max_new_tokens = int(max_new_tokens) + (1 if model.variant == _lib.VARIANT_TURBO else 0)
        if forced is not None:
            f = np.ascontiguousarray(forced, dtype=np.int32).reshape(self.n_utt, -1)
            _lib.check(L.t3mi_session_force(model._h, f.ctypes.data_as(C.POINTER(C.c_int32)), f.shape[1]))

    def refill(self, utts: Sequence[int], embeds_list: Sequence, text_tokens_list: Optional[Sequence] = None,
               max_new_tokens: Optional[Sequence[int]] = None, utterance_ids: Optional[Sequence[int]] = None) -> None:
        """Continuous batching (t3mi_session_refill): slots `utts` take new utterances -- prompt embeddings [rows, S0, d], or T3Cond
        objects with `text_tokens_list` -- while the other slots keep decoding from where they are.  Read a finished slot's
        tokens() first.  `max_new_tokens[i]`: the utterance's own limit (at most the session's); `utterance_ids[i]`: the global
        index that names its random stream (default: first_utterance_id + slot)."""
        L, n = self.model._lib, len(utts)
        ua = (C.c_int32 * n)(*[int(u) for u in utts])
        mn = (C.c_int32 * n)(*[int(v) for v in max_new_tokens]) if max_new_tokens is not None else None
        ids = (C.c_int64 * n)(*[int(v) for v in utterance_ids]) if utterance_ids is not None else None
        if text_tokens_list is not None:
            conds, tts = list(embeds_list), [_text(t) for t in text_tokens_list]
            cs = (_lib.Cond * n)(*[c.c_struct() for c in conds])
            tp = (C.POINTER(C.c_int32) * n)(*[t.ctypes.data_as(C.POINTER(C.c_int32)) for t in tts])
            nt = (C.c_int32 * n)(*[t.size for t in tts])
            _lib.check(L.t3mi_session_refill_tokens(self.model._h, n, ua, cs, tp, nt, mn, ids))
            return
        es = [_as_embeds(self.model, e, self.cfg_weight) for e in embeds_list]
        if any(e.shape[0] != self.rpu for e in es):
            raise ValueError(f"refill: every prompt needs {self.rpu} row(s) like the session's utterances")
        ptrs = (C.c_void_p * n)(*[e.ctypes.data for e in es])
        s0 = (C.c_int32 * n)(*[e.shape[1] for e in es])
        _lib.check(L.t3mi_session_refill(self.model._h, n, ua, ptrs, s0, mn, ids, 0))

    def decode(self, n_steps: int = 1):
        _lib.check(self.model._lib.t3mi_session_decode(self.model._h, int(n_steps)))

    def decode_timed(self, n_steps: int) -> float:
        ms = C.c_float(0)
        _lib.check(self.model._lib.t3mi_session_decode_timed(self.model._h, int(n_steps), C.byref(ms)))
        return float(ms.value)

    def sync(self):
        _lib.check(self.model._lib.t3mi_session_sync(self.model._h))

    def status(self):
        n = (C.c_int32 * self.n_utt)()
        d = (C.c_int32 * self.n_utt)()
        _lib.check(self.model._lib.t3mi_session_status(self.model._h, n, d))
        return np.array(n[:], dtype=np.int64), np.array(d[:], dtype=np.int64)

    def tokens(self, utt: int = 0) -> np.ndarray:
        out = np.zeros(self.max_new_tokens, dtype=np.int32)
        n = C.c_int32(0)
        _lib.check(self.model._lib.t3mi_session_tokens(
            self.model._h, utt, out.ctypes.data_as(C.POINTER(C.c_int32)), out.shape[0], C.byref(n)))
        return out[: n.value].astype(np.int64)

    def logits(self) -> np.ndarray:
        out = np.zeros((self.n_utt * self.rpu, self.model.vocab), dtype=np.float32)
        _lib.check(self.model._lib.t3mi_session_logits(self.model._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def set_logits(self, logits) -> None:
        """Replace the pending logits [rows, vocab] (parity hook: the reference's recorded logits go through the
        device sampling tail, scripts/debug_t3_step.R:108-214)."""
        lg = np.ascontiguousarray(logits, dtype=np.float32)
        if lg.shape != (self.n_utt * self.rpu, self.model.vocab):
            raise ValueError(f"logits must be [{self.n_utt * self.rpu}, {self.model.vocab}], got {lg.shape}")
        _lib.check(self.model._lib.t3mi_session_set_logits(self.model._h, lg.ctypes.data_as(C.POINTER(C.c_float))))

    def keep_filtered(self) -> None:
        _lib.check(self.model._lib.t3mi_session_keep_filtered(self.model._h))

    def filtered_logits(self, utt: int = 0) -> np.ndarray:
        """The logits row after CFG, repetition penalty and temperature at the most recent step (R/t3.R:518-537)."""
        out = np.zeros(self.model.vocab, dtype=np.float32)
        _lib.check(self.model._lib.t3mi_session_filtered_logits(self.model._h, int(utt), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def dist(self, utt: int = 0):
        V = self.model.vocab
        ids = np.zeros(V, dtype=np.int32)
        pr = np.zeros(V, dtype=np.float32)
        n = C.c_int32(0)
        _lib.check(self.model._lib.t3mi_session_dist(
            self.model._h, utt, ids.ctypes.data_as(C.POINTER(C.c_int32)),
            pr.ctypes.data_as(C.POINTER(C.c_float)), V, C.byref(n)))
        return ids[: n.value].astype(np.int64), pr[: n.value].copy()

    def drawn(self) -> np.ndarray:
        out = (C.c_int32 * self.n_utt)()
        _lib.check(self.model._lib.t3mi_session_drawn(self.model._h, out))
        return np.array(out[:], dtype=np.int64)

    def profile_step(self):
        st = (_lib.KStat * 16)()
        n = C.c_int(0)
        _lib.check(self.model._lib.t3mi_session_profile_step(self.model._h, st, 16, C.byref(n)))
        return [dict(name=st[i].name.decode(), launches=int(st[i].launches), total_us=float(st[i].total_us),
                     bytes=float(st[i].bytes)) for i in range(n.value)]

    def end(self):
        if self.model is not None and self.model._h:
            self.model._lib.t3mi_session_end(self.model._h)
        self.model = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.end()


def prompt_length(model: T3Model, cond: T3Cond, n_text: int, cfg_weight: float = 0.5) -> int:
    """Positions of the prompt prepare_input_embeds builds for this conditioning and text length (t3mi_prompt_shape)."""
    rows, s0, lc = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    c = cond.c_struct()
    _lib.check(model._lib.t3mi_prompt_shape(model._h, C.byref(c), int(n_text), float(cfg_weight), C.byref(rows), C.byref(s0), C.byref(lc)))
    return int(s0.value)


def t3_inference_continuous(model: T3Model, embeds_list: Sequence, text_tokens_list: Optional[Sequence] = None,
                            slots: int = 64, poll_steps: int = 16, max_new_tokens=1000,
                            temperature: float = 0.8, cfg_weight: float = 0.5, top_p: float = 0.95, min_p: float = 0.05,
                            repetition_penalty: float = 1.2, top_k: int = 1000, seed: Optional[int] = None, mask_stop: bool = False,
                            first_utterance_id: int = 0, drop_invalid=None, keep_stop: bool = False,
                            longest_first: bool = False, stats: Optional[dict] = None) -> List[np.ndarray]:
    """Any number of utterances through `slots` batch slots that never idle: every `poll_steps` decode steps the slots whose
    utterance has ended (stop token, its own `max_new_tokens[i]`, cache full) hand their tokens back and take the next waiting
    utterances, prefilled together (t3mi_generate_continuous over t3mi_session_refill) while the others keep their caches.  The reference decodes sentences one after
    the other (tts_chunked, R/tts.R:466-474); a fixed batch (t3_inference_batch) runs until its LONGEST utterance ends.
    Per-utterance semantics, argument forms and random streams are those of t3_inference_batch / t3_inference_turbo_batch
    (utterance i draws from stream (seed, first_utterance_id + i) wherever it runs), so with fp32 weights the tokens are the
    batch call's (Turbo: the trailing stop token is removed unless `keep_stop`).  `max_new_tokens`: one int or one per utterance.  `longest_first`: the queue starts with the utterances
    expected to run longest (their limits, else their text / prompt length), so the job does not end on one straggler; the results
    do not depend on the order.  `stats` (optional dict) receives steps, refill calls and
    the slot occupancy."""
    n = len(embeds_list)
    if n == 0:
        return []
    turbo = model.variant == _lib.VARIANT_TURBO
    if turbo:
        cfg_weight, min_p, sampler = 0.0, 0.0, _lib.SAMPLER_TURBO
    else:
        top_k, sampler = 0, _lib.SAMPLER_R
    mx = [int(max_new_tokens)] * n if np.isscalar(max_new_tokens) else [int(v) for v in max_new_tokens]
    if len(mx) != n or min(mx) <= 0:
        raise ValueError("max_new_tokens: one positive int, or one per utterance")
    p = _params(model, max(mx), temperature, cfg_weight, top_p, min_p, repetition_penalty, sampler, 0, seed, mask_stop, top_k,
                first_utterance_id=first_utterance_id, drop_invalid=drop_invalid)
    cap = max(mx) + (1 if turbo else 0)       # Turbo's first token comes from the prefill logits (R/t3.R:1614-1658)
    out = np.zeros((n, cap), dtype=np.int32)
    n_out = (C.c_int32 * n)()
    lim = (C.c_int32 * n)(*mx)
    st = (C.c_int32 * 4)()
    L = model._lib
    flags = 1 if longest_first else 0         # T3MI_CONTINUOUS_LONGEST_FIRST
    if text_tokens_list is not None:          # embeds_list holds T3Cond objects, one per utterance
        conds, tts = list(embeds_list), [_text(t) for t in text_tokens_list]
        cs = (_lib.Cond * n)(*[c.c_struct() for c in conds])
        tp = (C.POINTER(C.c_int32) * n)(*[t.ctypes.data_as(C.POINTER(C.c_int32)) for t in tts])
        nt = (C.c_int32 * n)(*[t.size for t in tts])
        rc = _lib.check(L.t3mi_generate_continuous_tokens(model._h, n, cs, tp, nt, lim, int(slots), int(poll_steps), flags, C.byref(p),
                                                          out.ctypes.data_as(C.POINTER(C.c_int32)), cap, n_out, st))
    else:
        es = [_as_embeds(model, e, cfg_weight) for e in embeds_list]
        ptrs = (C.c_void_p * n)(*[e.ctypes.data for e in es])
        s0 = (C.c_int32 * n)(*[e.shape[1] for e in es])
        rc = _lib.check(L.t3mi_generate_continuous(model._h, n, es[0].shape[0], ptrs, s0, lim, int(slots), int(poll_steps), flags, C.byref(p),
                                                   out.ctypes.data_as(C.POINTER(C.c_int32)), cap, n_out, st))
    if rc == _lib.WARN_CACHE_FULL:
        import warnings
        warnings.warn("KV cache full, stopping generation")       # src/t3_decode.cpp:395
    if stats is not None:
        stats.update(steps=int(st[0]), refill_calls=int(st[1]), slots=int(st[3]), occupancy=st[2] / max(1, st[0] * st[3]))
    res = [out[u, : n_out[u]].astype(np.int64) for u in range(n)]
    if turbo and not keep_stop:               # as t3_inference_turbo_batch: the trailing stop token is removed (R/t3.R:1667-1671)
        res = [t[:-1] if t.size and t[-1] == model.stop_speech_token else t for t in res]
    return res
