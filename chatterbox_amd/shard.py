"""Utterance-level sharding over the GPUs of one node.

Each utterance (its CFG row pair, KV cache, penalty set and RNG stream) is independent, so the
path shards with NO data-path collective: weights are replicated (2.05 GB fp32 of 288 GB HBM),
utterance u goes to rank floor(u * G / U) (contiguous blocks), every rank decodes its block with
its own t3mi handle, and the only exchange is the host-side gather of int token vectors in
utterance order.  One process per GPU (`torch.distributed`; backend "nccl" = RCCL on the GPU
box, "gloo" in the CPU tests) -- the collectives carry results only, never activations.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import numpy as np


def owner(u: int, n_utt: int, world: int) -> int:
    """Rank that decodes utterance u: floor(u * G / U)."""
    return (u * world) // n_utt


def local_indices(n_utt: int, world: int, rank: int) -> List[int]:
    return [u for u in range(n_utt) if owner(u, n_utt, world) == rank]


def decode_sharded(decode_fn: Callable[[Sequence[int]], List[np.ndarray]], n_utt: int, world: int, rank: int,
                   dist=None) -> List[np.ndarray]:
    """Run `decode_fn(my_utterance_ids) -> [tokens per utterance]` on this rank's block and
    return ALL utterances' tokens in utterance order on every rank.  `dist` is an initialised
    torch.distributed module (or None for a single process)."""
    mine = local_indices(n_utt, world, rank)
    outs = decode_fn(mine) if mine else []
    if len(outs) != len(mine):
        raise RuntimeError("decode_fn returned the wrong number of utterances")
    local = [(u, np.asarray(t, dtype=np.int64)) for u, t in zip(mine, outs)]
    if dist is None:      # no process group (a single process); with a group the gather runs even for one rank
        gathered = [local]
    else:
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
    result: List[np.ndarray] = [np.zeros(0, np.int64)] * n_utt
    seen = 0
    for part in gathered:
        for u, t in part:
            result[u] = t
            seen += 1
    if seen != n_utt:
        raise RuntimeError(f"gathered {seen} utterances, expected {n_utt}")
    return result


def t3_inference_sharded(model, prompts: Sequence, text_tokens_list=None, dist=None, world: int = 1, rank: int = 0,
                         turbo: bool = False, slots: int = 0, **kw) -> List[np.ndarray]:
    """t3_inference_batch (or t3_inference_turbo_batch) over the ranks of one node: every rank passes the FULL
    list of prompts (embedding arrays, or T3Cond objects with `text_tokens_list`), decodes its own contiguous
    block on its own GPU through `model` (a T3Model on this rank's device) and receives all utterances' tokens
    in utterance order.  Each utterance keeps the RNG stream of its global index (first_utterance_id = first
    index of the block), so sampled output does not depend on the number of ranks.  `slots` > 0: every rank runs its block
    through that many batch slots with continuous batching (t3.t3_inference_continuous; `max_new_tokens` may then be a list with
    one limit per utterance of the WHOLE job)."""
    from . import t3

    n_utt = len(prompts)
    fn = t3.t3_inference_turbo_batch if turbo else t3.t3_inference_batch
    if kw.get("seed") is None:
        # no seed given: rank 0 draws one (t3._seed) and every rank uses it, so the streams (seed, u) are those of one call
        box = [t3._seed(None) if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(box, src=0)
        kw["seed"] = box[0]

    def decode(mine):
        tt = None if text_tokens_list is None else [text_tokens_list[u] for u in mine]
        if slots > 0:
            k2 = dict(kw)
            mx = k2.get("max_new_tokens", 1000)
            if not np.isscalar(mx):
                k2["max_new_tokens"] = [mx[u] for u in mine]
            return t3.t3_inference_continuous(model, [prompts[u] for u in mine], tt, slots=slots, first_utterance_id=mine[0], **k2)
        return fn(model, [prompts[u] for u in mine], tt, first_utterance_id=mine[0], **kw)

    return decode_sharded(decode, n_utt, world, rank, dist)
