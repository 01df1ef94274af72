#!/usr/bin/env python3
"""Ragged utterance lengths: fixed batches (t3_inference_batch: every batch runs until its longest utterance ends) against continuous
batching (t3_inference_continuous: a slot takes the next utterance as soon as its own ends).  Full-size Llama T3, synthetic weights,
bf16, CFG pairs, greedy, stop token masked; utterance i wants lim[i] tokens, uniform in [lo, hi].  Wall time of the whole job
(prefills and refills included); useful tokens = sum(lim).
usage: python scripts/bench_continuous.py [--n 256] [--slots 64] [--lo 100] [--hi 400] [--poll 16] [--out file.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (the HIP runtime the tests and bench.py run with)
import chatterbox_amd as cb  # noqa: E402
from chatterbox_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--slots", type=int, default=64)
ap.add_argument("--lo", type=int, default=100)
ap.add_argument("--hi", type=int, default=400)
ap.add_argument("--s0", type=int, default=64)
ap.add_argument("--poll", type=int, default=16)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--out", default="")
a = ap.parse_args()

dims = synth.FULL
model = cb.load_t3(synth.ensure_checkpoint(dims, 1234), dtype=a.dtype)
rng = np.random.default_rng(7)
lim = [int(v) for v in rng.integers(a.lo, a.hi + 1, a.n)]
embs = [synth.synth_embeds(2, a.s0, dims.hidden_size, 1000 + u) for u in range(a.n)]
kw = dict(top_p=0.0, mask_stop=True)
useful = sum(lim)

# warm-up of both paths (graph instantiation, allocator)
cb.t3_inference_batch(model, embs[: a.slots], max_new_tokens=8, **kw)
cb.t3_inference_continuous(model, embs[: a.slots + 2], slots=a.slots, poll_steps=4, max_new_tokens=8, **kw)

t0 = time.perf_counter()
fixed = []
for b in range(0, a.n, a.slots):
    part = cb.t3_inference_batch(model, embs[b: b + a.slots], max_new_tokens=max(lim[b: b + a.slots]), **kw)
    fixed += [t[: lim[b + i]] for i, t in enumerate(part)]
t_fixed = time.perf_counter() - t0
steps_fixed = sum(max(lim[b: b + a.slots]) for b in range(0, a.n, a.slots))

st = {}
t0 = time.perf_counter()
cont = cb.t3_inference_continuous(model, embs, slots=a.slots, poll_steps=a.poll, max_new_tokens=lim, stats=st, **kw)
t_cont = time.perf_counter() - t0
assert all(len(c) == l for c, l in zip(cont, lim))
st2 = {}
t0 = time.perf_counter()
cont2 = cb.t3_inference_continuous(model, embs, slots=a.slots, poll_steps=a.poll, max_new_tokens=lim, longest_first=True, stats=st2, **kw)
t_lf = time.perf_counter() - t0
assert all(len(c) == l for c, l in zip(cont2, lim))
agree = float(np.mean([np.mean(np.asarray(c) == np.asarray(f)) for c, f in zip(cont, fixed)]))
res = {"workload": f"{a.n} utterances x CFG pair, S0 {a.s0}, {a.lo}..{a.hi} tokens each (uniform, {useful} in all), {a.slots} slots, {a.dtype}, greedy",
       "fixed_batches": {"wall_s": round(t_fixed, 3), "useful_tokens_per_s": round(useful / t_fixed, 1), "decode_steps": steps_fixed,
                         "slot_occupancy": round(useful / (steps_fixed * a.slots), 3)},
       "continuous": {"wall_s": round(t_cont, 3), "useful_tokens_per_s": round(useful / t_cont, 1), "decode_steps": st["steps"],
                      "refill_calls": st["refill_calls"], "slot_occupancy": round(useful / (st["steps"] * a.slots), 3), "poll_steps": a.poll},
       "continuous_longest_first": {"wall_s": round(t_lf, 3), "useful_tokens_per_s": round(useful / t_lf, 1), "decode_steps": st2["steps"],
                                    "refill_calls": st2["refill_calls"], "slot_occupancy": round(useful / (st2["steps"] * a.slots), 3),
                                    "note": "the limits ARE the lengths here, so this order is an oracle's; real lengths are only correlated with the text length"},
       "speedup": round(t_fixed / t_cont, 3), "greedy_token_agreement_with_fixed": round(agree, 4)}
print(json.dumps(res))
if a.out:
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
model.close()
