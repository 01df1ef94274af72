"""Per-wave timeline of one decode step from the diagnostic stamps build (libt3mi_stamps.so).
Run on the GPU box:  python scripts/stamp_timeline.py [bf16]
Prints, per launch kind: start skew, phase times (median / max over waves), and the gap from the
previous launch's last stamp to this launch's first wave start (100 MHz clock, 10 ns ticks)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["T3MI_GRAPH"] = "0"
from chatterbox_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "chatterbox_amd", "libt3mi_stamps.so")
import chatterbox_amd as cb  # noqa: E402
from chatterbox_amd import synth  # noqa: E402

dtype = "bf16" if "bf16" in sys.argv else "f32"
dims = synth.FULL
model = cb.load_t3(synth.ensure_checkpoint(dims, 1234), dtype=dtype)
emb = synth.synth_embeds(2, 52, 1024, 1000)
sess = cb.Session(model, [emb], max_new_tokens=400, top_p=0.0, mask_stop=True)
sess.decode(150)
sess.sync()
lib = _lib.load()
W, S, NL = 8192, 8, 160
lib.t3mi_debug_stamp_step.restype = C.c_int
lib.t3mi_debug_stamp_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
buf = np.zeros((NL, W, S), dtype=np.uint64)
for rep in range(3):
    n = lib.t3mi_debug_stamp_step(model._h, buf.ctypes.data, NL)
assert n > 0, lib.t3mi_last_error()
names = ["qkv", "attn", "o_proj", "gate_up", "down"]
prev_end = None
rows = {}
t0 = None
for i in range(n):
    st = buf[i].astype(np.int64)
    live = st[:, 0] > 0
    if not live.any():
        continue
    s0 = st[live, 0]
    last = np.where(st[live] > 0, st[live], 0).max(axis=1)
    kind = names[i % 5] if i < n - 1 else "head"
    first, end = s0.min(), last.max()
    if t0 is None:
        t0 = first
    rec = dict(waves=int(live.sum()), skew=(s0.max() - first) / 100.0, dur=(end - first) / 100.0,
               gap=(first - prev_end) / 100.0 if prev_end is not None else 0.0)
    for j, nm in [(7, "issued"), (5, "x_landed"), (6, "barrier"), (1, "x_ready"), (2, "w_fma"), (3, "reduced"), (4, "stored")]:
        v = st[live, j]
        ok = v > 0
        if ok.any():
            rel = (v[ok] - first) / 100.0
            rec[nm + "_med"] = float(np.median(rel))
            rec[nm + "_max"] = float(rel.max())
    ok7 = st[live, 7] > 0
    if ok7.any():   # per wave: own start -> all loads issued
        d = (st[live, 7][ok7] - st[live, 0][ok7]) / 100.0
        rec["own_issue_med"] = float(np.median(d))
        rec["own_issue_max"] = float(d.max())
    rows.setdefault(kind, []).append(rec)
    prev_end = end
print(f"dtype={dtype}  launches={n}  step span={(prev_end - t0) / 100.0:.1f} us")
for kind, rs in rows.items():
    keys = [k for k in rs[0] if k != "waves"]
    print(f"{kind:8s} n={len(rs):3d} waves={rs[0]['waves']:5d} " + " ".join(
        f"{k}={np.median([r.get(k, np.nan) for r in rs]):.2f}" for k in keys))
