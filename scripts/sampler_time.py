#!/usr/bin/env python3
"""Sampler kernel time per decode step by sampling mode (event timing, eager), batch 1 with the CFG pair and 64 utterances."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chatterbox_amd as cb  # noqa: E402
from chatterbox_amd import synth  # noqa: E402

dims = synth.FULL
model = cb.load_t3(synth.ensure_checkpoint(dims, 1234, head_scale=8.0), dtype="f32")
for utt in (1, 64):
    embs = [synth.synth_embeds(2, 52, dims.hidden_size, 1000 + u) for u in range(utt)]
    for name, kw in (("greedy top_p=0", dict(top_p=0.0)), ("default top_p=.95 min_p=.05", dict()),
                     ("top_p=.95 min_p=0 (top-p prefilter)", dict(top_p=0.95, min_p=0.0)),
                     ("top_p=1 min_p=0 (full support)", dict(top_p=1.0, min_p=0.0))):
        s = cb.Session(model, embs, max_new_tokens=64, mask_stop=True, **kw)
        s.decode(8)
        acc = {}
        for _ in range(5):
            for k in s.profile_step():
                acc[k["name"]] = acc.get(k["name"], 0.0) + k["total_us"] / 5
        tot = sum(acc.values())
        print(f"utt={utt:3d} {name:38s} sampler {acc['sampler']:7.2f} us of {tot:8.1f} us eager step")
        s.end()
model.close()
