"""Wall time of the on-device prompt builder at the real T3 shapes (d 1024, 150 conditioning tokens, 15 text
tokens, CFG pair): t3mi_prepare_input_embeds = build + sync + 426 KB device-to-host copy."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import chatterbox_amd as cb  # noqa: E402
from chatterbox_amd import synth  # noqa: E402

dims = synth.T3Dims(num_hidden_layers=1)
m = cb.load_t3(synth.ensure_checkpoint(dims))
rng = np.random.default_rng(0)
spk = (rng.standard_normal(256) * 0.1).astype(np.float32)
cond = cb.t3_cond(spk, rng.integers(0, 6561, 150))
text = rng.integers(1, 255, 15)
for _ in range(3):
    cb.prepare_input_embeds(m, cond, text, 0.5)
ts = []
for _ in range(20):
    t0 = time.perf_counter()
    e, lc = cb.prepare_input_embeds(m, cond, text, 0.5)
    ts.append(time.perf_counter() - t0)
print(f"prepare_input_embeds {e.shape} len_cond {lc}: median {1e3 * np.median(ts):.3f} ms, min {1e3 * min(ts):.3f} ms "
      f"(build + sync + D2H copy of {e.nbytes / 1e3:.0f} KB)")
m.close()
