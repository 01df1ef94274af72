"""N > 1 host path on CPU: two `gloo` ranks shard utterances (no data-path collective) and gather
the token vectors in utterance order.  The decode itself is stubbed (the HIP path needs a GPU);
what is covered is the partition, the ragged gather and that every rank sees every utterance."""
import os
import socket

import numpy as np
import pytest

from chatterbox_amd import shard

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402


def test_partition_is_contiguous_and_balanced():
    for n_utt, world in [(512, 8), (512, 4), (7, 2), (3, 8), (64, 1)]:
        owners = [shard.owner(u, n_utt, world) for u in range(n_utt)]
        assert owners == sorted(owners) and 0 <= min(owners) and max(owners) < world
        sizes = [len(shard.local_indices(n_utt, world, r)) for r in range(world)]
        assert sum(sizes) == n_utt and max(sizes) - min(sizes) <= 1


def _fake_decode(ids):
    # ragged, deterministic per-utterance "tokens"
    return [np.arange(3 + (u % 5), dtype=np.int64) * (u + 1) for u in ids]


def _worker(rank, world, port, n_utt, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = shard.decode_sharded(_fake_decode, n_utt, world, rank, dist)
        ok = all(np.array_equal(out[u], _fake_decode([u])[0]) for u in range(n_utt))
        # whole-job token count via a reduce, as bench.py aggregates throughput
        t = torch.tensor([sum(len(x) for x in _fake_decode(shard.local_indices(n_utt, world, rank)))])
        dist.all_reduce(t)
        q.put((rank, ok, int(t.item())))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_utt, world = 11, 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_utt, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = sum(len(x) for x in _fake_decode(range(n_utt)))
    assert sorted(r for r, _, _ in res) == [0, 1]
    assert all(ok for _, ok, _ in res) and all(t == total for _, _, t in res)


def test_single_process_path():
    out = shard.decode_sharded(_fake_decode, 5, 1, 0, None)
    assert all(np.array_equal(out[u], _fake_decode([u])[0]) for u in range(5))
