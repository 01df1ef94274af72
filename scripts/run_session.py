#!/usr/bin/env python3
"""One batched session decoded for --steps steps (all timed, replayed graph); for A/B timing and for
`rocprofv3 --kernel-trace --stats -- python3 scripts/run_session.py ...`.
usage: python scripts/run_session.py [--variant llama|turbo] [--utt 64] [--s0 64] [--dtype bf16] [--steps 400] [--reps 1]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("T3MI_NO_TORCH", "") != "1":
    # bench.py and the tests run with torch's bundled HIP runtime already loaded; keep the same runtime here (under
    # rocprofv3 the system runtime's graph launch crashed inside the profiler's kernel-trace layer)
    import torch  # noqa: F401,E402
import chatterbox_amd as cb  # noqa: E402
from chatterbox_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="llama")
ap.add_argument("--utt", type=int, default=64)
ap.add_argument("--s0", type=int, default=64)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--reps", type=int, default=1)
ap.add_argument("--tag", default="")
ap.add_argument("--chunk", type=int, default=0, help="> 0: enqueue this many steps at a time with a sync between (under rocprofv3: "
                                                     "400 queued graph replays crashed its kernel-trace buffer)")
a = ap.parse_args()
if a.variant == "turbo":
    dims = synth.TURBO_FULL
    model = cb.load_t3_turbo(synth.ensure_turbo_checkpoint(dims, 4321), dtype=a.dtype)
    embs = [synth.synth_embeds(1, a.s0, dims.hidden_size, 2000 + u) for u in range(a.utt)]
else:
    dims = synth.FULL
    model = cb.load_t3(synth.ensure_checkpoint(dims, 1234), dtype=a.dtype)
    embs = [synth.synth_embeds(2, a.s0, dims.hidden_size, 1000 + u) for u in range(a.utt)]
best = None
for _ in range(a.reps):
    s = cb.Session(model, embs, max_new_tokens=a.steps + 8, top_p=0.0, mask_stop=True)
    s.sync()
    if a.chunk > 0:
        ms, left = 0.0, a.steps
        while left > 0:
            ms += s.decode_timed(min(a.chunk, left))
            left -= a.chunk
    else:
        ms = s.decode_timed(a.steps)
    tk = int(sum(int(s.tokens(u).sum()) for u in range(0, a.utt, max(1, a.utt // 8))))
    s.end()
    best = ms if best is None else min(best, ms)
print(f"{a.tag} {a.variant} utt={a.utt} s0={a.s0} {a.dtype} steps={a.steps}: {best / a.steps:.4f} ms/step, "
      f"{a.utt * a.steps / best * 1e3:.0f} tok/s, token checksum {tk}")
model.close()
