#!/usr/bin/env python3
"""Four batch-1 fp32 sessions (S0 52) back to back, nothing decoded: for `rocprofv3 --kernel-trace --stats` of the prefill."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chatterbox_amd as cb  # noqa: E402
from chatterbox_amd import synth  # noqa: E402

dims = synth.FULL
model = cb.load_t3(synth.ensure_checkpoint(dims, 1234), dtype=sys.argv[1] if len(sys.argv) > 1 else "f32")
embs = [synth.synth_embeds(2, 52, dims.hidden_size, 1000)]
for _ in range(4):
    s = cb.Session(model, embs, max_new_tokens=328, top_p=0.0, mask_stop=True)
    s.sync()
    s.end()
model.close()
