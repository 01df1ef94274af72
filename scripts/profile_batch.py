#!/usr/bin/env python3
"""Per-kernel-kind time of one decode step (event timing, eager) for a batched session.
usage: python scripts/profile_batch.py [--variant llama|turbo] [--utt 64] [--s0 64] [--dtype bf16] [--steps 200]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chatterbox_amd as cb  # noqa: E402
from chatterbox_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="llama")
ap.add_argument("--utt", type=int, default=64)
ap.add_argument("--s0", type=int, default=64)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
if a.variant == "turbo":
    dims = synth.TURBO_FULL
    model = cb.load_t3_turbo(synth.ensure_turbo_checkpoint(dims, 4321), dtype=a.dtype)
    embs = [synth.synth_embeds(1, a.s0, dims.hidden_size, 2000 + u) for u in range(a.utt)]
    kw = {}
else:
    dims = synth.FULL
    model = cb.load_t3(synth.ensure_checkpoint(dims, 1234), dtype=a.dtype)
    embs = [synth.synth_embeds(2, a.s0, dims.hidden_size, 1000 + u) for u in range(a.utt)]
    kw = {}
s = cb.Session(model, embs, max_new_tokens=a.steps + 40, top_p=0.0, mask_stop=True, **kw)
s.decode(a.steps // 2)
ms = s.decode_timed(a.steps // 2)
print(f"{a.variant} utt={a.utt} s0={a.s0} {a.dtype}: {ms / (a.steps // 2):.4f} ms/step (graph), {a.utt * (a.steps // 2) / ms * 1e3:.0f} tok/s")
acc = {}
for _ in range(5):
    for k in s.profile_step():
        d = acc.setdefault(k["name"], [0, 0.0])
        d[0] = k["launches"]
        d[1] += k["total_us"] / 5
tot = sum(v[1] for v in acc.values())
for n, (l, us) in acc.items():
    print(f"  {n:16s} launches {l:4d}  total {us:8.1f} us  per launch {us / max(l, 1):7.2f} us  {100 * us / tot:5.1f} %")
print(f"  eager step total {tot:.1f} us")
s.end()
model.close()
