#!/usr/bin/env python3
"""Several S3Gen utterances at once on ONE GPU: `--workers` host threads, each with its own flow + vocoder handle (own stream,
own workspace; weights replicated, 0.37 GB per worker), each running `--utterances` utterances of `--tokens` speech tokens.
One utterance is a chain of ~5000 launches of 4 - 16 us that leave much of the chip idle in their ramps and tails; kernels of
independent streams fill those gaps.  Prints one JSON line: wall time, utterances/s and audio seconds per second for each
worker count.  (ctypes releases the GIL during the library calls.)
usage: python scripts/bench_s3gen_concurrent.py [--workers 1,2,4] [--utterances 4] [--tokens 100] [--prompt 150]"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import chatterbox_amd as cb  # noqa: E402
from chatterbox_amd import synth  # noqa: E402


def run(workers=(1, 2, 4), utterances=4, tokens=100, prompt=150):
    fd, hd = synth.FLOW_FULL, synth.HIFT_FULL
    fpath, hpath = synth.ensure_flow_checkpoint(fd, 555), synth.ensure_hift_checkpoint(hd, 999)
    tok, ptok, pfeat, xvec, z = synth.synth_flow_inputs(fd, tokens, prompt, 4)
    rng = np.random.default_rng(1)
    frames = 2 * tokens
    phase = rng.uniform(-np.pi, np.pi, hd.n_harm).astype(np.float32)
    snoise = rng.standard_normal((hd.n_harm, frames * 480)).astype(np.float32)
    ref = dict(prompt_token=ptok, prompt_feat=pfeat, embedding=xvec)
    out = {"workload": f"s3gen$inference x N concurrent handles on one GPU, {tokens} tokens after a {prompt}-token prompt each ({frames / 50.0:.1f} s of audio), fp32",
           "runs": []}
    first = None
    for w in workers:
        handles = [(cb.load_s3gen_flow(fpath), cb.load_hift_vocoder(hpath)) for _ in range(w)]
        results = [None] * w

        def work(i):
            flow, voc = handles[i]
            a = None
            for _ in range(utterances):
                a, _m = cb.s3gen_inference(flow, voc, tok, ref, z, phase, snoise)
            results[i] = a

        for i in range(w):          # warm-up: workspaces, kernel attributes
            cb.s3gen_inference(handles[i][0], handles[i][1], tok, ref, z, phase, snoise)
        ts = [threading.Thread(target=work, args=(i,)) for i in range(w)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        wall = time.perf_counter() - t0
        if first is None:
            first = results[0]
        same = all(np.array_equal(r, first) for r in results)
        n = w * utterances
        out["runs"].append({"workers": w, "utterances": n, "wall_ms": round(wall * 1e3, 1), "ms_per_utterance": round(wall * 1e3 / n, 2),
                            "audio_seconds_per_s": round(n * frames / 50.0 / wall, 1), "identical_audio": bool(same)})
        for f, v in handles:
            f.close()
            v.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", default="1,2,4")
    ap.add_argument("--utterances", type=int, default=4)
    ap.add_argument("--tokens", type=int, default=100)
    ap.add_argument("--prompt", type=int, default=150)
    a = ap.parse_args()
    print(json.dumps(run(tuple(int(x) for x in a.workers.split(",")), a.utterances, a.tokens, a.prompt)))
