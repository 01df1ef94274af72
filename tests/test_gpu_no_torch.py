"""The path the R glue and plain C hosts take: libt3mi.so on the SYSTEM HIP runtime, no torch in the process.

bench.py and several GPU tests import torch before the library is loaded, so the HIP runtime in those processes is the
copy bundled with torch; an R session never loads torch.  A fresh child process here imports only
numpy + chatterbox_amd, checks that no torch module and no torch-bundled libamdhip64 is mapped, and runs sessions whose
decode steps are REPLAYED HIP GRAPHS (the default) on both kernel families; its tokens and first-step logits must equal
those of a child that imported torch first, bit for bit (same kernels, same fixed summation orders)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, json, os, sys
sys.path.insert(0, {root!r})
if {with_torch}:
    import torch  # noqa: F401
import numpy as np
import chatterbox_amd as cb
from chatterbox_amd import synth
assert ('torch' in sys.modules) == bool({with_torch})
out = {{}}
dims = synth.TINY
model = cb.load_t3(synth.ensure_checkpoint(dims, 1234), device=0, start_speech_token=dims.start_speech_token,
                   stop_speech_token=dims.stop_speech_token)
for name, n_utt in (("gemv_2_rows", 1), ("gemm_12_rows", 6)):
    embs = [synth.synth_embeds(2, 9 + u, dims.hidden_size, 300 + u) for u in range(n_utt)]
    with cb.Session(model, embs, max_new_tokens=48, top_p=0.0, mask_stop=True) as s:
        lg = s.logits()
        for _ in range(3):
            s.decode(16)          # 48 graph replays, queued 16 at a time
            s.sync()
        toks = [list(map(int, s.tokens(u))) for u in range(n_utt)]
    out[name] = dict(tokens=toks, logits_sha=hashlib.sha256(lg.tobytes()).hexdigest())
# sampled decode through the one-call entry point (graph replay inside t3mi_generate_batch)
e = [synth.synth_embeds(2, 10, dims.hidden_size, 900 + u) for u in range(3)]
out["sampled"] = [list(map(int, t)) for t in cb.t3_inference_batch(model, e, max_new_tokens=24, seed=4, mask_stop=True)]
model.close()
maps = open('/proc/self/maps').read().splitlines()
out["hip_runtime"] = sorted({{l.split()[-1] for l in maps if 'libamdhip64' in l}})
out["torch_loaded"] = 'torch' in sys.modules
print("RESULT " + json.dumps(out))
"""


def _run(with_torch: bool) -> dict:
    env = dict(os.environ)
    env.pop("T3MI_GRAPH", None)       # graphs on (the default)
    code = CHILD.format(root=ROOT, with_torch=1 if with_torch else 0)
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, f"child (torch={with_torch}) exited {p.returncode}:\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}"
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_graph_replay_without_torch_equals_the_torch_loaded_process():
    plain = _run(False)
    assert plain["torch_loaded"] is False
    assert plain["hip_runtime"], "no libamdhip64 mapped?"
    assert not any("torch" in p for p in plain["hip_runtime"]), plain["hip_runtime"]
    for k in ("gemv_2_rows", "gemm_12_rows"):
        assert all(len(t) == 48 for t in plain[k]["tokens"])
    withtorch = _run(True)
    assert withtorch["torch_loaded"] is True
    print("HIP runtime without torch:", plain["hip_runtime"], "| with torch:", withtorch["hip_runtime"])
    for k in ("gemv_2_rows", "gemm_12_rows", "sampled"):
        assert plain[k] == withtorch[k], k
