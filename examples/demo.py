#!/usr/bin/env python3
"""A tour of the package on synthetic checkpoints (no real weights ship with this repository): one utterance, a batch, a ragged job
with continuous batching, and text -> audio with T3 and S3Gen overlapped.  Needs an MI355X and the built library
(python -c "import __graft_entry__ as g; g.build()").  With the real files, replace the synth.ensure_* paths by t3_cfg.safetensors /
s3gen.safetensors and `tokenize` by the reference's tokenizer.
usage: python examples/demo.py [--full]      (--full: full-size modules instead of the reduced test shapes)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chatterbox_amd as cb  # noqa: E402
from chatterbox_amd import synth  # noqa: E402

full = "--full" in sys.argv
dims = synth.FULL if full else synth.TINY
fd, hd = (synth.FLOW_FULL, synth.HIFT_FULL) if full else (synth.FLOW_TINY, synth.HIFT_TINY)
model = cb.load_t3(synth.ensure_checkpoint(dims, 1234), device=0, dtype="bf16" if full else "f32",
                   start_speech_token=dims.start_speech_token, stop_speech_token=dims.stop_speech_token)
flow = cb.load_s3gen_flow(synth.ensure_flow_checkpoint(fd, 555))
voc = cb.load_hift_vocoder(synth.ensure_hift_checkpoint(hd, 999))
rng = np.random.default_rng(0)
cond = cb.t3_cond((rng.standard_normal(dims.speaker_embed) * 0.1).astype(np.float32),           # as t3_cond, R/t3.R:103-113
                  rng.integers(0, dims.start_speech_token, 150), emotion_adv=0.5)
_t, ptok, pfeat, xvec, _z = synth.synth_flow_inputs(fd, 4, 150 if full else 9, 2)
ref_dict = dict(prompt_token=ptok, prompt_feat=pfeat, embedding=xvec)                            # the voice's s3gen reference


def tokenize(sentence):          # stand-in for the reference's text tokenizer (host string work, out of scope here)
    return [1 + (b % 250) for b in sentence.encode()][:40]


# 1. one utterance: t3_inference(model, cond, text_tokens, ...)  (R/t3.R:461-605)
t0 = time.perf_counter()
toks = cb.t3_inference(model, cond, tokenize("Hello, this is a test!"), max_new_tokens=60, seed=1)
print(f"t3_inference: {len(toks)} speech tokens in {1e3 * (time.perf_counter() - t0):.1f} ms")

# 2. tokens -> waveform: s3gen$inference  (R/s3gen.R:987-1056)
audio = cb.synthesize((flow, voc), [cb.drop_invalid_tokens(toks, dims.start_speech_token)], ref_dict, seed=1)[0]
print(f"synthesize: {audio.size / 24000:.2f} s of audio")

# 3. many utterances of uneven length through 4 batch slots that never idle
texts = [tokenize(f"Sentence number {i}.") for i in range(10)]
limits = [10 + 7 * (i % 5) for i in range(10)]
st = {}
outs = cb.t3_inference_continuous(model, [cond] * 10, texts, slots=4, poll_steps=4, max_new_tokens=limits, seed=1, mask_stop=True, stats=st)
print(f"t3_inference_continuous: {[len(o) for o in outs]} tokens, {st['steps']} steps, {st['refill_calls']} refills, occupancy {st['occupancy']:.2f}")

# 4. text -> audio, sentence by sentence as soon as each exists (T3 and S3Gen overlapped on the GPU)
t0 = time.perf_counter()
for i, sentence, tk, samples in cb.tts_stream(model, "First sentence. A second, longer one follows! And a third?", cond, tokenize,
                                              [(flow, voc)], ref_dict, slots=2, poll_steps=4, max_new_tokens=40, seed=1,
                                              drop_invalid=dims.start_speech_token):
    print(f"  [{1e3 * (time.perf_counter() - t0):7.1f} ms] sentence {i}: {len(tk)} tokens -> {samples.size / 24000:.2f} s of audio")
for h in (model, flow, voc):
    h.close()
