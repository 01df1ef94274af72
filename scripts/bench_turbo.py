#!/usr/bin/env python3
"""BASELINE.json configs[4] (SURVEY §8d cfg 5): Chatterbox-Turbo T3 (GPT-2 medium shapes, synthetic weights) on
k = 8 sentences x (prompt 1+375+30+1 = 407 rows, 300 decode steps), one row per sentence (no CFG):
(i) one sentence after the other, as the reference's tts_chunked does (R/tts.R:462-474), (ii) all eight batched.
Greedy (top_p = 0), stop token masked so every sentence runs all 300 steps.  Prints one JSON line.
usage: python scripts/bench_turbo.py [--dtype f32|bf16] [--sentences 8] [--steps 300]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chatterbox_amd as cb  # noqa: E402
from chatterbox_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--sentences", type=int, default=8)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--s0", type=int, default=407)
    a = ap.parse_args()
    dims = synth.TURBO_FULL
    model = cb.load_t3_turbo(synth.ensure_turbo_checkpoint(dims, 4321), dtype=a.dtype)
    embs = [synth.synth_embeds(1, a.s0, dims.hidden_size, 2000 + u) for u in range(a.sentences)]
    K = a.steps
    out = {"workload": f"configs[4]: Turbo T3 (24 x d1024, V6563), {a.sentences} sentences x S0={a.s0} x {K} steps, {a.dtype} weights",
           "dtype": a.dtype}
    # (i) sequential, like tts_chunked
    t_pref = t_dec = 0.0
    for e in embs:
        t0 = time.perf_counter()
        s = cb.Session(model, [e], max_new_tokens=K + 4, top_p=0.0, mask_stop=True)
        s.sync()
        t1 = time.perf_counter()
        ms = s.decode_timed(K)
        s.end()
        t_pref += t1 - t0
        t_dec += ms * 1e-3
    out["sequential"] = {"decode_tokens_per_s": round(a.sentences * K / t_dec, 1), "decode_ms_per_step": round(t_dec * 1e3 / (a.sentences * K), 4),
                         "prefill_ms_per_sentence": round(t_pref * 1e3 / a.sentences, 2)}
    # (ii) batched
    t0 = time.perf_counter()
    s = cb.Session(model, embs, max_new_tokens=K + 4, top_p=0.0, mask_stop=True)
    s.sync()
    t1 = time.perf_counter()
    ms = s.decode_timed(K)
    s.end()
    out["batched"] = {"decode_tokens_per_s": round(a.sentences * K / (ms * 1e-3), 1), "decode_ms_per_step": round(ms / K, 4),
                      "prefill_ms_total": round((t1 - t0) * 1e3, 2)}
    # algorithmic bytes per step: P_turbo * b_w + KV + logits
    d, F, V, L = dims.hidden_size, 4 * dims.hidden_size, dims.speech_vocab, dims.n_layer
    P = L * (4 * d * d + 2 * d * F + 9 * d + F) + 2 * d + V * d + V
    bw = 4 if a.dtype == "f32" else 2
    Lavg = a.s0 + K / 2
    for key, rows in (("sequential", 1), ("batched", a.sentences)):
        bkv = 2 if (a.dtype == "bf16" and rows > 4) else 4   # the many-row bf16 regime keeps a bf16 cache
        b = P * bw + rows * L * 2 * d * (Lavg + 1) * bkv + rows * V * 4
        out[key]["kv_bytes_per_element"] = bkv
        out[key]["hbm_frac_of_8TBs"] = round(b / (out[key]["decode_ms_per_step"] * 1e-3) / 8e12, 4)
    model.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
