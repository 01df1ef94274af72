#!/usr/bin/env python3
"""T3 speech-token decode benchmark (BASELINE.json metric: speech tokens/s; % of HBM roofline).

A "step" is one decode step of the hot path: sampling tail -> 30 Llama layers -> speech head,
for every live utterance of the batch (one new speech token per utterance per step).

Workload at N=1 (BASELINE.json configs[1]): ONE utterance with its CFG cond/uncond pair
(2 activation rows) on one MI355X, greedy (top_p = 0), prompt S0 = 52, stop token masked so
exactly K steps run, Llama-520M shapes with fp32 weights in the reference's safetensors
layout (synthetic weights: no checkpoint offline; SURVEY §8d cfg 2).  With --gpus N every
rank decodes its own utterance (weak scaling, no data-path collective).

Prefill and weight upload happen before the timed region (inputs resident in HBM); the timed
region is exactly K decode steps bracketed by barrier + synchronize on both sides.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--utterances", type=int, default=1, help="utterances per GPU (1 = configs[1])")
    ap.add_argument("--s0", type=int, default=52)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="weight storage type")
    ap.add_argument("--cpu-steps", type=int, default=600, help="decode steps of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barriers (gloo: rehearsal of the N > 1 path on a box with fewer GPUs "
                         "than ranks; ranks then share devices round-robin)")
    return ap.parse_args()


def prompt_embeds(s0: int, d: int, seed: int) -> np.ndarray:
    """cfg-2 stand-in: the reference's real prompt activations
    (tests/golden/ref_prompt_embeds.npz, [2,43,1024]) tiled/padded to [2,s0,1024];
    other utterances get N(0,.53^2) prompts."""
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


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    import torch
    dist = None
    if args.backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist_mod.init_process_group("gloo")
        dist = dist_mod
    else:
        torch.cuda.set_device(local_rank)

    def barrier():
        if dist is not None:
            dist.barrier()

    import chatterbox_amd as cb
    from chatterbox_amd import synth

    dims = synth.FULL
    if rank == 0:
        path = synth.ensure_checkpoint(dims, 1234)
    barrier()
    path = synth.checkpoint_path(dims, 1234)
    if not os.path.exists(path):          # ranks on another filesystem view
        path = synth.ensure_checkpoint(dims, 1234, root=f"/tmp/t3mi_synth_r{rank}")
    model = cb.load_t3(path, device=local_rank, dtype=args.dtype)
    info = model.info
    U = args.utterances
    rows = 2 * U
    K, W = args.steps, args.warmup
    embs = [prompt_embeds(args.s0, dims.hidden_size, rank * U + u) for u in range(U)]
    torch.cuda.synchronize()
    tp0 = time.perf_counter()
    sess = cb.Session(model, embs, max_new_tokens=K + W + 8, top_p=0.0, mask_stop=True)
    sess.sync()
    prefill_ms = (time.perf_counter() - tp0) * 1e3   # prompt upload + cache allocation + prefill, host wall clock

    if W > 0:
        sess.decode(W)
    sess.sync()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    ev_ms = sess.decode_timed(K)          # HIP events on the library's own stream + stream sync
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt, ev_ms], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, ev_ms = float(t[0]), float(t[1])
    ntok, _ = sess.status()
    assert int(ntok.min()) == W + K, (ntok, W, K)

    b_w = 4 if args.dtype == "f32" else 2
    L_avg = args.s0 + W + (K - 1) / 2.0          # cache length before the step, averaged over the timed steps
    # the library keeps a bf16 KV cache when bf16 weights meet more than 4 activation rows
    kv_bf16 = args.dtype == "bf16" and rows > 4 and os.environ.get("T3MI_KV", "")[:1] != "f"
    bstep = step_bytes(info, rows, L_avg, b_w, 2 if kv_bf16 else 4)
    step_us = ev_ms * 1e3 / K
    achieved = bstep / (step_us * 1e-6) / 1e9
    roof = {"bound": "hbm", "kernel": "decode_step (1 sampler + 30x[qkv_rope_kv, attn_decode, o_proj_resid, "
            "gate_up_swiglu, down_resid] + head; replayed HIP graph)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "bytes_per_launch": int(bstep), "avg_launch_us": round(step_us, 2)}
    # HBM traffic per step from the committed rocprofv3 PMC passes of this same command
    # (scripts/profile_round.sh; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM); bench.py cannot run
    # the profiler on itself, so the number is only attached when the workload matches.
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
        wl = tj.get("workload", {})
        if wl.get("dtype") == args.dtype and wl.get("utterances_per_gpu") == U and wl.get("s0") == args.s0:
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
                                       "in-graph kernel durations are in profiles/r01_kernel_stats.csv")
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
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record(); torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e-3 / reps
            nbytes = src.numel() * 4
            roof["measured_copy_GBps"] = round(2 * nbytes / timed(lambda: dst.copy_(src)) / 1e9, 1)
            roof["measured_read_GBps"] = round(nbytes / timed(lambda: src.sum()) / 1e9, 1)
            del src, dst
        except Exception as e:
            roof["measured_copy_error"] = str(e)

    if rank == 0:
        cpu = None
        if world == 1 and args.cpu_steps > 0:
            cpu = cpu_baseline(args.cpu_steps, args.s0, embs[0])
        total_tokens = n_gpus * U * K
        out = {
            "metric": "T3 speech tokens/sec (whole node)",
            "value": round(total_tokens / dt, 2),
            "unit": "tokens/s",
            "n_gpus": n_gpus, "steps": K, "warmup": W,
            "ms_per_step": round(dt * 1e3 / K, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.dtype == "f32" else "bf16-weights/f32-accumulate",
            "data": "synthetic",
            "config": {"workload": f"{'configs[1]' if U == 1 else 'configs[2]-shaped'}: {U} utterance(s)/GPU x CFG cond/uncond pair "
                                   f"({rows} rows), greedy top_p=0, S0={args.s0}, Llama-520M (30x d1024/F4096, V8194) synthetic "
                                   f"weights in the reference safetensors layout, {args.dtype} weights, "
                                   f"{'bf16' if kv_bf16 else 'fp32'} KV cache",
                       "utterances_per_gpu": U, "rows_per_gpu": rows, "s0": args.s0,
                       "parallelism": f"utterance-sharded x{n_gpus} (no collectives)"},
            "hip_event_ms_per_step": round(ev_ms / K, 5),
            "prefill_ms": round(prefill_ms, 3),   # rank 0: session set-up + prefill of its utterances (not in `value`)
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
