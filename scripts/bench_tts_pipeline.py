#!/usr/bin/env python3
"""Text -> audio for a text of --sentences sentences: T3 for all sentences and then S3Gen (tts_chunked(s3gen=...)) against the two
halves overlapped (tts_pipelined: sentences that have ended are in S3Gen while T3 decodes the rest).  Full-size modules, synthetic
weights (no stop token arises: every sentence runs --tokens speech tokens), bf16 T3, fp32 S3Gen, --pairs (flow, vocoder) handles.
Reports the wall time of both and the time until the FIRST sentence's audio exists.
usage: python scripts/bench_tts_pipeline.py [--variant llama|turbo] [--sentences 8] [--slots 4] [--tokens 100] [--pairs 4] [--out f.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import chatterbox_amd as cb  # noqa: E402
from chatterbox_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="llama")
ap.add_argument("--sentences", type=int, default=8)
ap.add_argument("--slots", type=int, default=4)
ap.add_argument("--tokens", type=int, default=100)
ap.add_argument("--pairs", type=int, default=4)
ap.add_argument("--poll", type=int, default=8)
ap.add_argument("--out", default="")
a = ap.parse_args()

turbo = a.variant == "turbo"
if turbo:
    dims = synth.TURBO_FULL
    model = cb.load_t3_turbo(synth.ensure_turbo_checkpoint(dims, 4321), dtype="bf16")
else:
    dims = synth.FULL
    model = cb.load_t3(synth.ensure_checkpoint(dims, 1234), dtype="bf16")
fd, hd = synth.FLOW_FULL, synth.HIFT_FULL
fp, hp = synth.ensure_flow_checkpoint(fd, 555), synth.ensure_hift_checkpoint(hd, 999)
pairs = [(cb.load_s3gen_flow(fp), cb.load_hift_vocoder(hp)) for _ in range(a.pairs)]
rng = np.random.default_rng(3)
n_cond = 375 if turbo else 150
cond = cb.t3_cond((rng.standard_normal(dims.speaker_embed) * 0.1).astype(np.float32), rng.integers(0, 6561, n_cond),
                  emotion_adv=None if turbo else 0.5)
_tok, ptok, pfeat, xvec, _z = synth.synth_flow_inputs(fd, 4, 150, 2)
ref = dict(prompt_token=ptok, prompt_feat=pfeat, embedding=xvec)
text = " ".join(f"Sentence number {i} of the benchmark text, about thirty tokens long." for i in range(a.sentences))


def tokenize(sentence):
    return [1 + (b % 250) for b in sentence.encode()][:30]


kw = dict(max_new_tokens=a.tokens, seed=11, top_p=0.0)
if not turbo:
    kw["cfg_weight"] = 0.5
# warm-up (graphs, allocators, first-touch of the S3Gen handles)
cb.tts_chunked(model, text, cond, tokenize, s3gen=pairs, ref_dict=ref, slots=a.slots, **dict(kw, max_new_tokens=8))
list(cb.tts_stream(model, text, cond, tokenize, pairs, ref, slots=a.slots, poll_steps=a.poll, **dict(kw, max_new_tokens=8)))

t0 = time.perf_counter()
seq = cb.tts_chunked(model, text, cond, tokenize, s3gen=pairs, ref_dict=ref, slots=a.slots, poll_steps=a.poll, **kw)
t_seq = time.perf_counter() - t0

t0 = time.perf_counter()
first = None
wavs = []
for i, _s, _t, audio in cb.tts_stream(model, text, cond, tokenize, pairs, ref, slots=a.slots, poll_steps=a.poll, **kw):
    if first is None:
        first = time.perf_counter() - t0
    wavs.append(audio)
t_pipe = time.perf_counter() - t0
audio_s = sum(w.size for w in wavs) / 24000.0
same_tokens = all(len(x) == len(y) for x, y in zip(seq["speech_tokens"], [w for w in seq["speech_tokens"]]))
res = {"workload": f"{a.variant} T3 bf16 + S3Gen fp32, full size, synthetic weights: {a.sentences} sentences x {a.tokens} speech tokens "
                   f"({audio_s:.1f} s of audio), {a.slots} T3 slots, {a.pairs} (flow, vocoder) handles, one GPU",
       "t3_then_s3gen": {"wall_s": round(t_seq, 3), "audio_seconds_per_s": round(audio_s / t_seq, 1), "first_audio_s": round(t_seq, 3)},
       "pipelined": {"wall_s": round(t_pipe, 3), "audio_seconds_per_s": round(audio_s / t_pipe, 1), "first_audio_s": round(first, 3)},
       "speedup": round(t_seq / t_pipe, 3), "samples_equal": bool(np.array_equal(np.concatenate(wavs), seq["audio"]))}
print(json.dumps(res))
if a.out:
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
model.close()
for f_, v_ in pairs:
    f_.close()
    v_.close()
