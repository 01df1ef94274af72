"""Text -> audio with the two halves overlapped: while T3 (continuous batching, Session.refill) still decodes later sentences, the
sentences that have ended are already in S3Gen on other host threads and HIP streams.  The T3 decode at a handful of rows is a
chain of short launches that leaves most of the chip idle; S3Gen's dense launches fill it.

The reference runs sentence after sentence and, inside a sentence, T3 then S3Gen (tts_chunked, R/tts.R:455-480 over generate(),
R/tts.R:300-412); tts_chunked(..., s3gen=...) of this package batches T3 and then runs S3Gen.  Same tokens and same samples here
(sentence i draws from the streams (seed, i) in both halves), only earlier: the first sentence's audio is ready when ITS tokens
and ITS waveform are, not when the last sentence's are."""
import queue
from concurrent.futures import ThreadPoolExecutor
from typing import Iterator, Optional, Tuple

import numpy as np

from . import _lib
from . import s3gen as _s3
from .t3 import SPEECH_VOCAB_SIZE, Session, T3Cond, T3Model, _seed, prompt_length, split_sentences


def tts_stream(model: T3Model, text: str, cond: T3Cond, tokenize, s3gen, ref_dict: dict, slots: Optional[int] = None,
               poll_steps: int = 8, drop_invalid=True, max_new_tokens: int = 1000, temperature: float = 0.8,
               cfg_weight: float = 0.5, top_p: float = 0.95, min_p: float = 0.05, repetition_penalty: float = 1.2,
               top_k: int = 1000, seed: Optional[int] = None) -> Iterator[Tuple[int, str, np.ndarray, np.ndarray]]:
    """Generator over the sentences of `text` IN ORDER: (index, sentence, speech tokens, audio samples at 24 kHz), each as soon as
    its audio exists.  Arguments as tts_chunked (`tokenize`, `cond`, sampling parameters, `drop_invalid`, `slots` = T3 batch slots,
    default one per sentence) plus `s3gen` -- one (flow, vocoder) pair or a list of pairs (one host thread each) -- and the voice's
    `ref_dict`.  Tokens and samples equal tts_chunked(..., slots=slots, s3gen=..., ref_dict=...) with the same seed."""
    sentences = split_sentences(text)
    n = len(sentences)
    if n == 0:
        return
    turbo = model.variant == _lib.VARIANT_TURBO
    if turbo:
        cfg_weight, min_p = 0.0, 0.0
    else:
        top_k = 0
    seed = _seed(seed)
    limit = (SPEECH_VOCAB_SIZE if drop_invalid is True else int(drop_invalid)) if drop_invalid else 0
    toks = [np.asarray(tokenize(s), dtype=np.int32).reshape(-1) for s in sentences]
    pairs = [s3gen] if isinstance(s3gen[0], _s3.S3GenFlow) else list(s3gen)
    idle: "queue.Queue" = queue.Queue()
    for p in pairs:
        idle.put(p)

    def wave(i: int, t: np.ndarray) -> np.ndarray:
        flow, voc = idle.get()
        try:
            return _s3.synthesize_one(flow, voc, t, i, ref_dict, seed, turbo=turbo, n_cfm_timesteps=2 if turbo else None)
        finally:
            idle.put((flow, voc))

    slots = max(1, min(int(slots) if slots else n, n))
    extra = 1 if turbo else 0
    cache = max(prompt_length(model, cond, t.size, cfg_weight) for t in toks) + int(max_new_tokens) + 2 + extra
    sess = Session(model, [cond] * slots, text_tokens_list=toks[:slots], max_new_tokens=max_new_tokens, temperature=temperature,
                   cfg_weight=cfg_weight, top_p=top_p, min_p=min_p, repetition_penalty=repetition_penalty, max_cache_len=cache,
                   seed=seed, top_k=top_k)
    pool = ThreadPoolExecutor(max_workers=len(pairs))
    fut: list = [None] * n
    speech: list = [None] * n
    sent = 0
    try:
        owner = {u: u for u in range(slots)}
        nxt = slots
        while owner:
            sess.decode(poll_steps)
            _, done = sess.status()
            free = [u for u in owner if done[u] != 0]
            for u in free:
                i = owner.pop(u)
                t = sess.tokens(u)
                if turbo and t.size and t[-1] == model.stop_speech_token:
                    t = t[:-1]                                   # R/t3.R:1667-1671
                if limit:
                    t = t[t < limit]                             # drop_invalid_tokens (R/s3tokenizer.R:678-687)
                speech[i] = t
                fut[i] = pool.submit(wave, i, t)
            take = free[: max(0, n - nxt)]
            if take:
                new = list(range(nxt, nxt + len(take)))
                nxt += len(take)
                sess.refill(take, [cond] * len(new), [toks[i] for i in new], utterance_ids=new)
                owner.update(dict(zip(take, new)))
            while sent < n and fut[sent] is not None and fut[sent].done():
                yield sent, sentences[sent], speech[sent], fut[sent].result()
                sent += 1
        sess.end()
        sess = None
        while sent < n:
            yield sent, sentences[sent], speech[sent], fut[sent].result()
            sent += 1
    finally:
        if sess is not None:
            sess.end()
        pool.shutdown(wait=True)


def tts_pipelined(model: T3Model, text: str, cond: T3Cond, tokenize, s3gen, ref_dict: dict, **kw) -> dict:
    """tts_stream collected into tts_chunked's result: {"sentences", "speech_tokens", "audio", "sample_rate"}."""
    out = list(tts_stream(model, text, cond, tokenize, s3gen, ref_dict, **kw))
    wavs = [a for _, _, _, a in out]
    return {"sentences": [s for _, s, _, _ in out], "speech_tokens": [t for _, _, t, _ in out],
            "audio": np.concatenate(wavs) if wavs else np.zeros(0, np.float32), "sample_rate": _s3.S3GEN_SR}
