#!/usr/bin/env python3
"""One batch as two half-batches on two streams of one GPU (two handles, one process), against one session:
does the HBM-bound attention of one half overlap the GEMMs of the other?
usage: python scripts/two_halves.py [--utt 512] [--steps 200] [--s0 64] [--variant llama|turbo]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (same HIP runtime as bench.py)
import chatterbox_amd as cb  # noqa: E402
from chatterbox_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--utt", type=int, default=512)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--s0", type=int, default=64)
ap.add_argument("--variant", default="llama")
ap.add_argument("--chunk", type=int, default=10)
a = ap.parse_args()
turbo = a.variant == "turbo"
dims = synth.TURBO_FULL if turbo else synth.FULL
path = synth.ensure_turbo_checkpoint(dims, 4321) if turbo else synth.ensure_checkpoint(dims, 1234)
load = cb.load_t3_turbo if turbo else cb.load_t3
rpu = 1 if turbo else 2
embs = [synth.synth_embeds(rpu, a.s0, dims.hidden_size, 1000 + u) for u in range(a.utt)]
kw = dict(max_new_tokens=a.steps + 8, top_p=0.0, mask_stop=True)

m0 = load(path, dtype="bf16")
s = cb.Session(m0, embs, **kw)
s.sync()
t0 = time.perf_counter()
for _ in range(a.steps // a.chunk):
    s.decode(a.chunk)
s.sync()
one = (time.perf_counter() - t0) * 1e3 / a.steps
s.end()

m1 = load(path, dtype="bf16")
h = a.utt // 2
sa = cb.Session(m0, embs[:h], **kw)
sb = cb.Session(m1, embs[h:], first_utterance_id=h, **kw)
sa.sync(); sb.sync()
t0 = time.perf_counter()
for _ in range(a.steps // a.chunk):
    sa.decode(a.chunk)
    sb.decode(a.chunk)
sa.sync(); sb.sync()
two = (time.perf_counter() - t0) * 1e3 / a.steps
sa.end(); sb.end()
print(f"{a.variant} {a.utt} utterances, {a.steps} steps: one session {one:.4f} ms/step ({a.utt / one * 1e3:.0f} tok/s); "
      f"two halves on two streams {two:.4f} ms/step ({a.utt / two * 1e3:.0f} tok/s)")
m0.close(); m1.close()
