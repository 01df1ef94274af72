#!/usr/bin/env python3
"""T3 speech-token decode benchmark (BASELINE.json metric: speech tokens/s; % of HBM roofline).

A "step" is one decode step of the hot path: sampling tail -> 30 Llama layers -> speech head,
for every live utterance of the batch (one new speech token per utterance per step).

Default workload (BASELINE.json configs[1]): ONE utterance with its CFG cond/uncond pair
(2 activation rows) per GPU, greedy (top_p = 0), prompt S0 = 52, stop token masked so exactly K
steps run, Llama-520M shapes with fp32 weights in the reference's safetensors layout (synthetic
weights: no checkpoint offline; SURVEY §8d cfg 2).  With --gpus N every rank decodes its own
utterance ("scaling": "weak", no data-path collective).

--total-utterances T (BASELINE.json configs[3], SURVEY §8d cfg 4): T utterances are partitioned
over the ranks with chatterbox_amd.shard (u -> floor(u*G/T)), every rank decodes its block as one
batched session and the token vectors are gathered in utterance order ("scaling": "strong").

The rank count is the one the process group actually has: under a launcher (torchrun) WORLD_SIZE
must equal --gpus or the run exits non-zero; without a launcher the parent process (which makes no
GPU call) starts --gpus fresh rank processes itself.

The N = 1 default run also attaches, under extra keys of the SAME JSON line, the 64-utterance
bf16 session of configs[2] ("batch64"), the 512-utterance workload of configs[3] on this
rank count ("cfg4_512utt") and configs[4]'s 8 Turbo sentences batched ("turbo8"), each measured
after the headline region.

Prefill and weight upload happen before the timed region (inputs resident in HBM); the timed
region is exactly K decode steps bracketed by barrier + synchronize on both sides.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--utterances", type=int, default=1, help="utterances per GPU (1 = configs[1]); weak scaling")
    ap.add_argument("--total-utterances", type=int, default=0,
                    help="> 0: strong scaling, this many utterances partitioned over the ranks (512 = configs[3])")
    ap.add_argument("--s0", type=int, default=0, help="prompt length (default 52; 64 with --total-utterances)")
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16"],
                    help="weight storage type (default f32; bf16 with --total-utterances)")
    ap.add_argument("--cpu-steps", type=int, default=600, help="decode steps of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-attach", action="store_true", help="skip the batch64 / ragged256 / cfg4_512utt / turbo8 / S3Gen attachments")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barriers (gloo: rehearsal of the N > 1 path on a box with fewer GPUs "
                         "than ranks; ranks then share devices round-robin)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group even with ONE rank, so that the N > 1 code path (init_process_group, "
                         "barrier, MAX all-reduce on a device tensor, all_gather_object) runs on a one-GPU box")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------
# rank bookkeeping (no GPU call in this part)

def resolve_world(gpus: int, env=None):
    """(world, rank, local_rank, spawn): the rank count this process group really has.  Under a launcher
    WORLD_SIZE must equal --gpus (raises SystemExit(2) otherwise: a line that says "n_gpus": 8 for a
    one-rank run would be fabricated scaling).  Without a launcher and --gpus > 1, spawn = True: the parent
    starts the ranks itself."""
    env = os.environ if env is None else env
    if "WORLD_SIZE" not in env:
        return (1, 0, 0, gpus > 1)
    world = int(env["WORLD_SIZE"])
    if world != gpus:
        sys.stderr.write(f"bench.py: --gpus {gpus} but the launcher started WORLD_SIZE={world} ranks; "
                         f"refusing to report a rank count that did not run\n")
        raise SystemExit(2)
    return (world, int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")), False)


def spawn_ranks(gpus: int, argv) -> int:
    """Parent without a launcher: start `gpus` fresh rank processes of this script (one per GPU), relay rank 0's
    stdout, return the worst exit code.  The parent never touches the GPU (it does not even import torch)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies leaves the others waiting at a barrier: as soon as one exits non-zero, end the rest (by PID) so
    # the whole run fails instead of hanging until the collective's own timeout
    import threading
    buf = []
    rd = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    rc = 0
    live = set(range(gpus))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                sys.stderr.write(f"bench.py: rank {r} exited with status {code}; stopping the other ranks\n")
                for q in sorted(live):
                    procs[q].terminate()
        if live:
            time.sleep(0.2)
    rd.join(timeout=10)
    if rc == 0 and buf:
        sys.stdout.write(buf[0].decode())
        sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------------------------------
# workload pieces

def prompt_embeds(s0: int, d: int, seed: int) -> np.ndarray:
    """cfg-2 stand-in: the reference's real prompt activations
    (tests/golden/ref_prompt_embeds.npz, [2,43,1024]) tiled/padded to [2,s0,1024];
    other utterances get N(0,.53^2) prompts (seed 1000 + u)."""
    from chatterbox_amd import synth
    g = os.path.join(ROOT, "tests", "golden", "ref_prompt_embeds.npz")
    if seed == 0 and os.path.exists(g) and d == 1024:
        e = np.load(g)["input_embeds"].astype(np.float32)
        reps = -(-s0 // e.shape[1])
        return np.ascontiguousarray(np.tile(e, (1, reps, 1))[:, :s0, :])
    return synth.synth_embeds(2, s0, d, 1000 + seed)


def step_bytes(info, rows: int, L: float, b_w: int, b_kv: int = 4) -> float:
    """Algorithmic bytes of one decode step, SURVEY §8d:
    B_step = P*b_w + R*n_layers*2*d*(L+1)*b_kv + R*V*4 ;  b_kv = 4 (fp32 cache) or 2 (bf16 cache)."""
    d, F, V, nl = info.d_model, info.d_ff, info.vocab, info.n_layers
    P = nl * (4 * d * d + 3 * d * F + 2 * d) + d + V * d
    return P * b_w + rows * nl * 2 * d * (L + 1) * b_kv + rows * V * 4


def kv_is_bf16(dtype: str, rows: int) -> bool:
    # the library keeps a bf16 KV cache when bf16 weights meet more than 4 activation rows
    return dtype == "bf16" and rows > 4 and os.environ.get("T3MI_KV", "")[:1] != "f"


def cpu_baseline(n_steps: int, s0: int, emb: np.ndarray):
    """The oracle (numpy fp32 restatement, BLAS threads) timed on this box's host cores on a
    bounded sample of the same workload: prefill of the same prompt, then n_steps greedy
    decode steps.  A reported baseline, not the target."""
    from chatterbox_amd import synth
    from oracle import t3_oracle as O
    dims = synth.FULL
    w = synth.synth_llama_weights(dims, 1234)
    om = O.LlamaOracle(w, O.LlamaConfig())
    cores = os.cpu_count()
    try:
        from threadpoolctl import threadpool_info
        th = [p.get("num_threads") for p in threadpool_info() if p.get("user_api") == "blas"]
        if th:
            cores = int(max(th))
    except Exception:
        pass
    p = O.SampleParams(max_new_tokens=n_steps + 1, top_p=0.0)
    # time decode steps only (prefill excluded, as on the GPU side)
    hidden, past = om.forward(emb)
    last = hidden[:, -1, :]
    seen = [dims.start_speech_token - 1]
    t0 = time.perf_counter()
    for step in range(n_steps):
        lg = O.cfg_combine(om.speech_logits(last), p.cfg_weight)
        lg[dims.stop_speech_token] = -np.inf
        tail = O.sampling_tail(lg, seen, p, 0.0)
        seen.append(tail.token)
        e = om.next_embed(tail.token, step + 1)
        hidden, past = om.forward(np.broadcast_to(e[None, None, :], (2, 1, e.shape[0])).astype(np.float32), past)
        last = hidden[:, -1, :]
    dt = time.perf_counter() - t0
    return {"value": round(n_steps / dt, 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"numpy fp32 oracle (oracle/t3_oracle.py), {n_steps} greedy decode steps after an S0={s0} prefill, "
                      f"1 utterance + CFG pair, Llama-520M synthetic weights; {dt:.1f} s of CPU work"}


class Ranks:
    """The process group as this run sees it: barrier, max-over-ranks, and the module for shard.decode_sharded."""

    def __init__(self, world, rank, local_rank, backend, force_dist=False):
        import torch
        self.torch, self.world, self.rank, self.backend = torch, world, rank, backend
        self.dist = None
        ndev = max(torch.cuda.device_count(), 1)
        self.local_rank = local_rank % ndev if backend == "gloo" else local_rank
        self.gpu = torch.cuda.is_available()      # False only in the CPU tests of this file's host logic
        if self.gpu:
            torch.cuda.set_device(self.local_rank)
        if world > 1 or force_dist:
            import torch.distributed as dist_mod
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:      # --force-dist without a launcher: a one-rank group of our own
                sk = socket.socket()
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
                sk.close()
            os.environ.setdefault("RANK", str(rank))
            os.environ.setdefault("WORLD_SIZE", str(world))
            if backend == "nccl":
                dist_mod.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist_mod.init_process_group("gloo")
            self.dist = dist_mod
            if dist_mod.get_world_size() != world:
                raise SystemExit(f"process group has {dist_mod.get_world_size()} ranks, expected {world}")

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def sync_barrier(self):
        if self.gpu:
            self.torch.cuda.synchronize()
        self.barrier()

    def max(self, *vals):
        if self.dist is None:
            return [float(v) for v in vals]
        t = self.torch.tensor(list(vals), dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(x) for x in t]

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def timed_decode(rk: Ranks, sess, K: int, W: int):
    """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides.
    Returns (max-over-ranks wall seconds, max-over-ranks HIP-event milliseconds on the library's stream)."""
    if W > 0:
        sess.decode(W)
    sess.sync()
    rk.sync_barrier()
    t0 = time.perf_counter()
    ev_ms = sess.decode_timed(K)          # HIP events on the library's own stream + stream sync
    rk.sync_barrier()
    dt = time.perf_counter() - t0
    return rk.max(dt, ev_ms)


def sharded_workload(rk: Ranks, cb, model, total: int, s0: int, dtype: str, K: int, W: int, d: int):
    """configs[3]: `total` utterances (prompt seeds 1000 + u) partitioned over the ranks by shard.local_indices; every
    rank decodes its block as ONE batched session for W + K steps; the token vectors come back in utterance order
    through shard.decode_sharded (the only exchange: int vectors, no activation ever crosses a GPU boundary).
    The timed region holds the K decode steps AND the gather."""
    from chatterbox_amd import shard
    mine = shard.local_indices(total, rk.world, rk.rank)
    embs = [prompt_embeds(s0, d, u) for u in mine]
    tp0 = time.perf_counter()
    sess, err = None, None
    try:
        if mine:
            sess = cb.Session(model, embs, max_new_tokens=K + W + 8, top_p=0.0, mask_stop=True, first_utterance_id=mine[0])
            sess.sync()
            if W > 0:
                sess.decode(W)
                sess.sync()
    except Exception as e:      # e.g. out of memory on one rank: every rank must learn of it before the first barrier
        err = e
    prefill_ms = (time.perf_counter() - tp0) * 1e3
    if rk.max(1.0 if err is not None else 0.0)[0] > 0:
        if sess is not None:
            sess.end()
        raise RuntimeError(f"session set-up failed on a rank: {err}")
    ev = [0.0]

    def decode(ids):
        ev[0] = sess.decode_timed(K)
        return [sess.tokens(i)[W:W + K] for i in range(len(ids))]

    rk.sync_barrier()
    t0 = time.perf_counter()
    toks = shard.decode_sharded(decode, total, rk.world, rk.rank, rk.dist)
    rk.sync_barrier()
    dt = time.perf_counter() - t0
    dt, ev_ms = rk.max(dt, ev[0])
    if sess is not None:
        sess.end()
    assert len(toks) == total and all(len(t) == K for t in toks), "sharded decode lost tokens"
    rows = 2 * max(len(shard.local_indices(total, rk.world, r)) for r in range(rk.world))
    kv16 = kv_is_bf16(dtype, rows)
    L_avg = s0 + W + (K - 1) / 2.0
    bstep = step_bytes(model.info, rows, L_avg, 4 if dtype == "f32" else 2, 2 if kv16 else 4)   # busiest rank
    step_us = ev_ms * 1e3 / K
    return {"utterances": total, "ranks": rk.world, "rows_on_busiest_rank": rows, "s0": s0, "steps": K, "warmup": W,
            "dtype": f"{dtype}-weights/{'bf16' if kv16 else 'fp32'}-kv", "scaling": "strong",
            "tokens_per_s": round(total * K / dt, 1), "ms_per_step": round(dt * 1e3 / K, 4),
            "hip_event_ms_per_step": round(ev_ms / K, 4), "prefill_ms": round(prefill_ms, 1),
            "bytes_per_step_per_gpu": int(bstep), "GBps_per_gpu": round(bstep / (step_us * 1e-6) / 1e9, 1),
            "frac": round(bstep / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            # bit-reproducible run to run for one rank count; across rank counts the rows per GPU change the GEMM tiling,
            # and with it the bf16 rounding of a few activations, so near-tied greedy picks (and this sum) may differ
            "token_checksum": int(sum(int(t.sum()) for t in toks) % (1 << 31))}


def ragged_workload(cb, model, d: int, n: int = 256, slots: int = 64, lo: int = 100, hi: int = 400, s0: int = 64):
    """Ragged utterance lengths (what real text gives; the reference's loop is serial, R/tts.R:466-474): n utterances that want
    lo..hi tokens each (uniform, seeded), greedy, through `slots` batch slots -- as consecutive fixed batches (each runs until its
    longest utterance ends) and with continuous batching (t3mi_generate_continuous: a slot takes the next utterance when its own has
    ended).  Whole-job wall time, prefills and refills included; useful tokens = the sum of the utterances' own limits."""
    rng = np.random.default_rng(7)
    lim = [int(v) for v in rng.integers(lo, hi + 1, n)]
    embs = [prompt_embeds(s0, d, 1 + u) for u in range(n)]
    kw = dict(top_p=0.0, mask_stop=True)
    cb.t3_inference_batch(model, embs[:slots], max_new_tokens=4, **kw)
    cb.t3_inference_continuous(model, embs[: slots + 2], slots=slots, poll_steps=2, max_new_tokens=4, **kw)
    t0 = time.perf_counter()
    steps_fixed = 0
    for b in range(0, n, slots):
        cb.t3_inference_batch(model, embs[b: b + slots], max_new_tokens=max(lim[b: b + slots]), **kw)
        steps_fixed += max(lim[b: b + slots])
    t_fixed = time.perf_counter() - t0
    st = {}
    t0 = time.perf_counter()
    got = cb.t3_inference_continuous(model, embs, slots=slots, poll_steps=16, max_new_tokens=lim, stats=st, **kw)
    t_cont = time.perf_counter() - t0
    st2 = {}
    t0 = time.perf_counter()
    cb.t3_inference_continuous(model, embs, slots=slots, poll_steps=16, max_new_tokens=lim, longest_first=True, stats=st2, **kw)
    t_lf = time.perf_counter() - t0
    useful = sum(lim)
    if [len(t) for t in got] != lim:
        raise RuntimeError("continuous batching returned other token counts than asked for")
    return {"workload": f"{n} utterances x CFG pair, S0 {s0}, {lo}..{hi} tokens each (uniform, {useful} in all), {slots} slots, bf16 weights + "
                        "bf16 KV, greedy; whole job incl. prefills",
            "fixed_batches": {"useful_tokens_per_s": round(useful / t_fixed, 1), "wall_s": round(t_fixed, 3), "decode_steps": steps_fixed},
            "continuous": {"useful_tokens_per_s": round(useful / t_cont, 1), "wall_s": round(t_cont, 3), "decode_steps": st["steps"],
                           "refill_calls": st["refill_calls"]},
            "continuous_longest_first": {"useful_tokens_per_s": round(useful / t_lf, 1), "wall_s": round(t_lf, 3), "decode_steps": st2["steps"],
                                         "note": "queue ordered by the limits, which ARE the lengths here (an oracle's order)"},
            "speedup": round(t_fixed / t_cont, 3)}


def turbo_workload(rk: Ranks, cb, synth, sentences: int = 8, s0: int = 407, K: int = 300):
    """configs[4] (SURVEY §8d cfg 5): Chatterbox-Turbo T3 (GPT-2 medium shapes, synthetic weights), `sentences` sentences of one
    row each (no CFG pair), prompt 1 + 375 + 30 + 1 rows, K greedy decode steps, batched in one session -- what the
    reference's tts_chunked does one sentence after the other (R/tts.R:462-474).  scripts/bench_turbo.py has both orders."""
    dims = synth.TURBO_FULL
    model = cb.load_t3_turbo(synth.ensure_turbo_checkpoint(dims, 4321), device=rk.local_rank, dtype="bf16")
    try:
        embs = [synth.synth_embeds(1, s0, dims.hidden_size, 2000 + u) for u in range(sentences)]
        tp0 = time.perf_counter()
        sess = cb.Session(model, embs, max_new_tokens=K + 4, top_p=0.0, mask_stop=True)
        sess.sync()
        prefill_ms = (time.perf_counter() - tp0) * 1e3
        dt, ev_ms = timed_decode(rk, sess, K, 0)
        sess.end()
    finally:
        model.close()
    d, F, V, L = dims.hidden_size, 4 * dims.hidden_size, dims.speech_vocab, dims.n_layer
    P = L * (4 * d * d + 2 * d * F + 9 * d + F) + 2 * d + V * d + V          # decode-path parameters incl. biases
    bkv = 2 if sentences > 4 else 4                                            # many-row bf16 regime: bf16 cache
    bstep = P * 2 + sentences * L * 2 * d * (s0 + (K - 1) / 2.0 + 1) * bkv + sentences * V * 4
    us = ev_ms * 1e3 / K
    return {"workload": f"configs[4]: Turbo T3 (24 x d1024, V{V}), {sentences} sentences batched x (S0={s0}, {K} greedy steps), bf16 weights + "
                        f"{'bf16' if bkv == 2 else 'fp32'} KV cache, synthetic weights",
            "tokens_per_s": round(sentences * K / dt, 1), "ms_per_step": round(dt * 1e3 / K, 4), "hip_event_ms_per_step": round(ev_ms / K, 4),
            "steps": K, "prefill_ms": round(prefill_ms, 1), "bytes_per_step": int(bstep),
            "GBps": round(bstep / (us * 1e-6) / 1e9, 1), "frac": round(bstep / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}


def run(args, world, rank, local_rank):
    rk = Ranks(world, rank, local_rank, args.backend, args.force_dist)
    torch = rk.torch
    import chatterbox_amd as cb
    from chatterbox_amd import synth

    dims = synth.FULL
    if rank == 0:
        synth.ensure_checkpoint(dims, 1234)
    rk.barrier()
    path = synth.checkpoint_path(dims, 1234)
    if not os.path.exists(path):          # ranks on another filesystem view
        path = synth.ensure_checkpoint(dims, 1234, root=f"/tmp/t3mi_synth_r{rank}")
    strong = args.total_utterances > 0
    dtype = args.dtype or ("bf16" if strong else "f32")
    s0 = args.s0 or (64 if strong else 52)
    K, W = args.steps, args.warmup
    model = cb.load_t3(path, device=rk.local_rank, dtype=dtype)
    info = model.info
    d = dims.hidden_size
    b_w = 4 if dtype == "f32" else 2
    out = None

    if strong:
        r = sharded_workload(rk, cb, model, args.total_utterances, s0, dtype, K, W, d)
        if rank == 0:
            bstep, step_us = r["bytes_per_step_per_gpu"], r["hip_event_ms_per_step"] * 1e3
            out = {
                "metric": "T3 speech tokens/sec (whole node)", "value": r["tokens_per_s"], "unit": "tokens/s",
                "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": r["ms_per_step"], "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "bf16-weights/f32-accumulate" if dtype == "bf16" else "f32",
                "data": "synthetic",
                "config": {"workload": f"configs[3]: {args.total_utterances} synthetic utterances x CFG pair sharded over {world} rank(s) "
                                       f"(u -> floor(u*G/U)), greedy top_p=0, S0={s0}, {K} timed steps, Llama-520M synthetic weights, {r['dtype']}",
                           "total_utterances": args.total_utterances, "rows_on_busiest_rank": r["rows_on_busiest_rank"], "s0": s0,
                           "parallelism": f"utterance-sharded x{world} (no collectives; token vectors gathered by shard.decode_sharded)"},
                "world_size_observed": world, "backend": args.backend if rk.dist is not None else None,
                "hip_event_ms_per_step": r["hip_event_ms_per_step"], "prefill_ms": r["prefill_ms"],
                "token_checksum": r["token_checksum"],
                "roofline": {"bound": "hbm", "kernel": "decode_step of the busiest rank (replayed HIP graph)", "achieved": r["GBps_per_gpu"],
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r["frac"], "traffic": None,
                             "bytes_per_launch": bstep, "avg_launch_us": round(step_us, 2)},
                "cpu_baseline": None,
            }
        model.close()
        return rk, out

    # ---- weak scaling: every rank decodes U utterances of its own (U = 1: configs[1]) ----
    U = args.utterances
    rows = 2 * U
    embs = [prompt_embeds(s0, d, rank * U + u) for u in range(U)]
    torch.cuda.synchronize()
    tp0 = time.perf_counter()
    sess = cb.Session(model, embs, max_new_tokens=K + W + 8, top_p=0.0, mask_stop=True, first_utterance_id=rank * U)
    sess.sync()
    prefill_ms = (time.perf_counter() - tp0) * 1e3   # prompt upload + cache allocation + prefill, host wall clock
    dt, ev_ms = timed_decode(rk, sess, K, W)
    ntok, _ = sess.status()
    assert int(ntok.min()) == W + K, (ntok, W, K)

    L_avg = s0 + W + (K - 1) / 2.0          # cache length before the step, averaged over the timed steps
    kv16 = kv_is_bf16(dtype, rows)
    bstep = step_bytes(info, rows, L_avg, b_w, 2 if kv16 else 4)
    step_us = ev_ms * 1e3 / K
    achieved = bstep / (step_us * 1e-6) / 1e9
    n_launch = 2 + 5 * info.n_layers
    roof = {"bound": "hbm", "kernel": "decode_step (1 sampler + 30x[qkv_rope_kv, attn_decode, o_proj_resid, "
            "gate_up_swiglu, down_resid] + head; replayed HIP graph)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "bytes_per_launch": int(bstep), "avg_launch_us": round(step_us, 2)}
    if rows <= 4:
        # the step is a chain of n_launch DEPENDENT launches; a dependent boundary costs 1.2-1.9 us on this part
        # (MI355X_MICROARCH.md price list, row "boundary") and a kernel needs ~2 us from wave start to its last
        # load's return before its first byte of output, so the chain has a latency floor beside its HBM time
        hbm_us = bstep / (HBM_PEAK_GBS * 1e3)
        floor_us = n_launch * 3.0
        roof["note"] = (f"{n_launch} dependent launches per step, {bstep / n_launch / 1e6:.1f} MB each on average: HBM time at peak "
                        f"{hbm_us:.0f} us + per-launch fixed cost (~1.5 us boundary + ~1.5 us ramp/drain) {floor_us:.0f} us "
                        f"=> floor ~{hbm_us + floor_us:.0f} us = {hbm_us / (hbm_us + floor_us):.2f} of peak; the 0.50 target of "
                        f"north_star is not reachable with one launch per all-to-all edge at {rows} rows (DESIGN.md §4)")
    # HBM traffic per step from the committed rocprofv3 PMC passes of this same command
    # (scripts/profile_round.sh; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM); bench.py cannot run
    # the profiler on itself, so the number is only attached when the workload matches.
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
        wl = tj.get("workload", {})
        if wl.get("dtype") == dtype and wl.get("utterances_per_gpu") == U and wl.get("s0") == s0:
            roof["traffic"] = int(tj["hbm_bytes_per_step"])
            roof["traffic_source"] = "profiles/traffic_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, " + str(wl.get("steps")) + ")"
    except Exception:
        pass
    if not args.no_profile and rank == 0:
        try:
            acc = {}
            reps = 5
            for _ in range(reps):
                for k in sess.profile_step():
                    a = acc.setdefault(k["name"], dict(launches=k["launches"], us=0.0, bytes=k["bytes"]))
                    a["us"] += k["total_us"] / reps
            roof["per_kernel_note"] = ("eager launches bracketed by HIP events, one step at a time: each figure includes the "
                                       "launch gap, so they sum to more than avg_launch_us of the replayed graph; the "
                                       "in-graph kernel durations are in profiles/r02_kernel_stats.csv")
            roof["per_kernel"] = [
                {"name": n, "launches_per_step": a["launches"], "avg_us": round(a["us"] / max(a["launches"], 1), 2),
                 "GB/s": round(a["bytes"] / max(a["us"], 1e-9) / 1e3, 1)} for n, a in acc.items()]
        except Exception as e:  # profiling is informational
            roof["per_kernel_error"] = str(e)
    sess.end()
    model.close()
    if not args.no_profile and rank == 0 and world == 1:
        # practical ceiling beside the 8 TB/s vendor peak: a 1 GiB device copy (read + write)
        # and a 1 GiB read-only reduction through torch's own kernels, HIP-event timed; informational only
        try:
            src = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()
            dst = torch.empty_like(src)

            def timed(fn, reps=10):
                fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e-3 / reps
            nbytes = src.numel() * 4
            roof["measured_copy_GBps"] = round(2 * nbytes / timed(lambda: dst.copy_(src)) / 1e9, 1)
            roof["measured_read_GBps"] = round(nbytes / timed(lambda: src.sum()) / 1e9, 1)
            del src, dst
        except Exception as e:
            roof["measured_copy_error"] = str(e)

    # ---- attachments: the other regimes of BASELINE's metric, measured after the headline region ----
    attach = {}
    if not args.no_attach and U == 1 and dtype == "f32":
        try:
            m16 = cb.load_t3(path, device=rk.local_rank, dtype="bf16")
            try:
                if world == 1:
                    # configs[2]: 64 utterances x 400 tokens batched on one GPU, bf16 weights + bf16 KV, all 400 steps timed
                    e64 = [prompt_embeds(64, d, u) for u in range(64)]
                    s64 = cb.Session(m16, e64, max_new_tokens=408, top_p=0.0, mask_stop=True)
                    s64.sync()
                    dt64, ev64 = timed_decode(rk, s64, 400, 0)
                    s64.end()
                    b64 = step_bytes(m16.info, 128, 64 + 399 / 2.0, 2, 2)
                    us64 = ev64 * 1e3 / 400
                    attach["batch64"] = {
                        "workload": "configs[2]: 64 synthetic utterances x CFG pair (128 rows) x 400 speech tokens on one MI355X, greedy, "
                                    "S0=64 (L 64 -> 464), bf16 weights + bf16 KV cache, fp32 accumulate",
                        "tokens_per_s": round(64 * 400 / dt64, 1), "ms_per_step": round(dt64 * 1e3 / 400, 4),
                        "hip_event_ms_per_step": round(ev64 / 400, 4), "steps": 400, "bytes_per_step": int(b64),
                        "GBps": round(b64 / (us64 * 1e-6) / 1e9, 1), "frac": round(b64 / (us64 * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
                    try:
                        attach["ragged256"] = ragged_workload(cb, m16, d)
                    except Exception as e:
                        attach["ragged256_error"] = f"{type(e).__name__}: {e}"
                # configs[3]: 512 utterances sharded over the ranks of this run
                attach["cfg4_512utt"] = sharded_workload(rk, cb, m16, 512, 64, "bf16", 400, 0, d)
            finally:
                m16.close()
        except Exception as e:   # an attachment must never cost the headline line
            attach["attach_error"] = f"{type(e).__name__}: {e}"
        if world == 1:
            try:
                attach["turbo8"] = turbo_workload(rk, cb, synth)
            except Exception as e:
                attach["turbo8_error"] = f"{type(e).__name__}: {e}"

        if world == 1:
            # SURVEY §8 f4, first piece: the S3Gen flow decoder's 10-step CFG solve on the full-size CFM estimator (fp32)
            try:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                import bench_cfm
                attach["cfm_solve"] = bench_cfm.run(frames=500, steps=10, reps=3)
                import bench_s3gen
                attach["s3gen"] = bench_s3gen.run(tokens=100, prompt=150, reps=2)
            except Exception as e:
                attach["cfm_solve_error"] = f"{type(e).__name__}: {e}"

    if rank == 0:
        cpu = None
        if world == 1 and args.cpu_steps > 0:
            cpu = cpu_baseline(args.cpu_steps, s0, embs[0])
        total_tokens = world * U * K
        out = {
            "metric": "T3 speech tokens/sec (whole node)",
            "value": round(total_tokens / dt, 2),
            "unit": "tokens/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt * 1e3 / K, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if dtype == "f32" else "bf16-weights/f32-accumulate",
            "data": "synthetic",
            "config": {"workload": f"{'configs[1]' if U == 1 else 'configs[2]-shaped'}: {U} utterance(s)/GPU x CFG cond/uncond pair "
                                   f"({rows} rows), greedy top_p=0, S0={s0}, Llama-520M (30x d1024/F4096, V8194) synthetic "
                                   f"weights in the reference safetensors layout, {dtype} weights, "
                                   f"{'bf16' if kv16 else 'fp32'} KV cache",
                       "utterances_per_gpu": U, "rows_per_gpu": rows, "s0": s0,
                       "parallelism": f"utterance-sharded x{world} (no collectives)"},
            "world_size_observed": world,
            "backend": args.backend if rk.dist is not None else None,      # the process group the barriers ran on (None: one rank, no group)
            "hip_event_ms_per_step": round(ev_ms / K, 5),
            "prefill_ms": round(prefill_ms, 3),   # rank 0: session set-up + prefill of its utterances (not in `value`)
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        out.update(attach)
    return rk, out


def main():
    args = parse()
    world, rank, local_rank, spawn = resolve_world(args.gpus)
    if spawn:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    # stdout carries exactly ONE JSON line: whatever libraries print there while the run lasts (gloo's "[Gloo] Rank 0 is
    # connected ..." goes to stdout) is sent to stderr, and the line is written to the real stdout at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rk, out = run(args, world, rank, local_rank)
    rk.close()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if rank == 0 and out is not None:
        print(json.dumps(out))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
