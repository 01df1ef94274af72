#!/usr/bin/env python3
"""t3mi_open wall time for the full-size Llama checkpoint: fp32 storage, bf16 storage (conversion on device), bf16 with
the packed cache being written, and bf16 read back from the cache."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (same HIP runtime as bench.py)
import chatterbox_amd as cb  # noqa: E402
from chatterbox_amd import synth  # noqa: E402

path = synth.ensure_checkpoint(synth.FULL, 1234)
cache = path + ".t3mi-bf16"
if os.path.exists(cache):
    os.remove(cache)
for tag, kw in [("fp32", dict(dtype="f32")), ("fp32 again (page cache warm)", dict(dtype="f32")), ("bf16, converted", dict(dtype="bf16")),
                ("bf16, cache written", dict(dtype="bf16", cache=True)), ("bf16, from cache", dict(dtype="bf16", cache=True)),
                ("bf16, from cache again", dict(dtype="bf16", cache=True))]:
    t0 = time.perf_counter()
    m = cb.load_t3(path, device=0, **kw)
    dt = time.perf_counter() - t0
    print(f"{tag:32s} {dt * 1e3:8.1f} ms   cache_state {m.info.cache_state}")
    m.close()
print(f"checkpoint {os.path.getsize(path) / 1e9:.2f} GB, cache {os.path.getsize(cache) / 1e9:.2f} GB")
