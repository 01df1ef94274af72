#!/usr/bin/env python3
"""S3Gen on the GPU at full size: speech tokens -> mel (conformer encoder + 10-step CFG solve) -> waveform
(HiFT), synthetic weights in the reference's key layouts.  One utterance: `--tokens` speech tokens (25 per second of audio)
after a 150-token voice prompt.  Prints one JSON line with the device time of each stage and the real-time factor.
usage: python scripts/bench_s3gen.py [--tokens 100] [--prompt 150] [--reps 3]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import chatterbox_amd as cb  # noqa: E402
from chatterbox_amd import synth  # noqa: E402


def run(tokens=100, prompt=150, reps=3):
    fd, hd = synth.FLOW_FULL, synth.HIFT_FULL
    tok, ptok, pfeat, xvec, z = synth.synth_flow_inputs(fd, tokens, prompt, 4)
    rng = np.random.default_rng(1)
    frames = 2 * tokens
    phase = rng.uniform(-np.pi, np.pi, hd.n_harm).astype(np.float32)
    snoise = rng.standard_normal((hd.n_harm, frames * 480)).astype(np.float32)
    best = None
    with cb.load_s3gen_flow(synth.ensure_flow_checkpoint(fd, 555)) as flow, cb.load_hift_vocoder(synth.ensure_hift_checkpoint(hd, 999)) as voc:
        for i in range(reps + 1):
            tm = {}
            audio, mel = cb.s3gen_inference(flow, voc, tok, dict(prompt_token=ptok, prompt_feat=pfeat, embedding=xvec), z, phase, snoise, timing=tm)
            if i > 0 and (best is None or tm["flow_ms"] + tm["vocoder_ms"] < best["flow_ms"] + best["vocoder_ms"]):
                best = tm
    secs = frames / 50.0
    total = best["flow_ms"] + best["vocoder_ms"]
    return {"workload": f"s3gen$inference, full size: {tokens} speech tokens after a {prompt}-token prompt -> {frames} mel frames ({2 * (tokens + prompt)} in the "
                        f"solve) -> {frames * 480} samples ({secs:.1f} s at 24 kHz); conformer encoder 6 + 4 layers, CFM 10 steps x CFG pair, HiFT; fp32",
            "flow_device_ms": round(best["flow_ms"], 2), "vocoder_device_ms": round(best["vocoder_ms"], 2), "total_device_ms": round(total, 2),
            "audio_seconds": secs, "real_time_factor": round(secs * 1e3 / total, 1), "finite": bool(np.isfinite(audio).all())}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=100)
    ap.add_argument("--prompt", type=int, default=150)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    print(json.dumps(run(a.tokens, a.prompt, a.reps)))
