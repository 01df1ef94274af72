"""Host logic of bench.py on CPU: the rank count it reports is the rank count that ran, and the
configs[3] strong-scaling path (partition -> per-rank session -> shard.decode_sharded gather) under two gloo ranks
with the GPU session replaced by a stand-in."""
import json
import os
import socket
import subprocess
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402


def test_world_must_match_gpus():
    assert bench.resolve_world(1, env={}) == (1, 0, 0, False)
    assert bench.resolve_world(4, env={}) == (1, 0, 0, True)          # no launcher: the parent spawns 4 ranks
    assert bench.resolve_world(2, env={"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}) == (2, 1, 1, False)
    with pytest.raises(SystemExit) as e:
        bench.resolve_world(8, env={"WORLD_SIZE": "1"})
    assert e.value.code == 2


def test_mismatch_exits_nonzero_as_a_process():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, timeout=120)
    assert r.returncode == 2 and b"WORLD_SIZE=1" in r.stderr and r.stdout.strip() == b""


class _FakeSession:
    """Stands in for chatterbox_amd.Session: utterance with global id g emits token (7 * g + step) % 6561."""

    def __init__(self, model, embs, max_new_tokens, first_utterance_id=0, **kw):
        self.n, self.base, self.steps = len(embs), first_utterance_id, 0

    def sync(self):
        pass

    def decode(self, n):
        self.steps += n

    def decode_timed(self, n):
        self.steps += n
        return 1.0 * n

    def tokens(self, i):
        return (7 * (self.base + i) + np.arange(self.steps, dtype=np.int64)) % 6561

    def end(self):
        pass


def _fake_info():
    return types.SimpleNamespace(d_model=1024, d_ff=4096, vocab=8194, n_layers=30)


def _expected_checksum(total, K, W):
    return int(sum(int(((7 * u + np.arange(W, W + K)) % 6561).sum()) for u in range(total)) % (1 << 31))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    rk = bench.Ranks(world, rank, rank, "gloo")
    try:
        cb = types.SimpleNamespace(Session=_FakeSession)
        model = types.SimpleNamespace(info=_fake_info())
        r = bench.sharded_workload(rk, cb, model, 11, 8, "bf16", 5, 2, 64)
        q.put((rank, r))
    finally:
        rk.close()


def test_strong_scaling_path_two_gloo_ranks():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res.values():
        assert r["ranks"] == 2 and r["utterances"] == 11 and r["scaling"] == "strong"
        assert r["rows_on_busiest_rank"] == 12                    # 6 + 5 utterances, CFG pairs
        assert r["token_checksum"] == _expected_checksum(11, 5, 2)
    assert json.dumps(res[0])                                      # the record is plain JSON


def test_strong_scaling_path_single_process():
    rk = bench.Ranks(1, 0, 0, "gloo")
    cb = types.SimpleNamespace(Session=_FakeSession)
    r = bench.sharded_workload(rk, cb, types.SimpleNamespace(info=_fake_info()), 5, 8, "bf16", 4, 0, 64)
    assert r["ranks"] == 1 and r["token_checksum"] == _expected_checksum(5, 4, 0)


def _force_dist_worker(q):
    for k in ("MASTER_PORT", "RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    rk = bench.Ranks(1, 0, 0, "gloo", force_dist=True)
    try:
        assert rk.dist is not None and rk.dist.get_world_size() == 1
        assert rk.max(3.0, 1.5) == [3.0, 1.5]                     # a real all-reduce in a one-rank group
        cb = types.SimpleNamespace(Session=_FakeSession)
        r = bench.sharded_workload(rk, cb, types.SimpleNamespace(info=_fake_info()), 5, 8, "bf16", 4, 0, 64)
        q.put(r)
    finally:
        rk.close()


def test_force_dist_runs_the_group_path_with_one_rank():
    """--force-dist: init_process_group, barrier, MAX all-reduce and all_gather_object all execute in a
    one-rank group (gloo here; `nccl` on the GPU box, profiles/r03_bench_nccl_world1.json)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_force_dist_worker, args=(q,))
    p.start()
    r = q.get(timeout=180)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert r["ranks"] == 1 and r["token_checksum"] == _expected_checksum(5, 4, 0)


def test_spawn_parent_fails_when_a_rank_dies(tmp_path, monkeypatch):
    """Without a launcher the parent starts the ranks: one rank exiting non-zero must end the run non-zero promptly (the
    others would wait at a barrier), and no JSON line is printed."""
    fake = tmp_path / "fake_rank.py"
    fake.write_text("import os, sys, time\n"
                    "if os.environ['RANK'] == '1':\n    sys.exit(7)\n"
                    "time.sleep(120)\nprint('{\"never\": 1}')\n")
    monkeypatch.setattr(bench, "__file__", str(fake))
    import io
    import contextlib
    import time as _t
    out = io.StringIO()
    t0 = _t.time()
    with contextlib.redirect_stdout(out):
        rc = bench.spawn_ranks(3, [])
    assert rc == 7 and out.getvalue() == "" and _t.time() - t0 < 60
