#!/usr/bin/env python3
"""Host wall time of t3mi_session_begin (allocation + cache zero-fill + prompt upload + prefill + first head) by shape."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chatterbox_amd as cb  # noqa: E402
from chatterbox_amd import synth  # noqa: E402

dims = synth.FULL
for dtype in ("f32", "bf16"):
    model = cb.load_t3(synth.ensure_checkpoint(dims, 1234), dtype=dtype)
    for utt, s0, new in ((1, 52, 328), (1, 52, 1000), (8, 52, 328), (64, 64, 400)):
        embs = [synth.synth_embeds(2, s0, dims.hidden_size, 1000 + u) for u in range(utt)]
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            s = cb.Session(model, embs, max_new_tokens=new, top_p=0.0, mask_stop=True)
            s.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
            s.end()
        print(f"{dtype} utt={utt:3d} s0={s0} max_new={new}: session_begin {min(ts):.2f} ms (runs: {' '.join(f'{t:.2f}' for t in ts)})")
    model.close()
