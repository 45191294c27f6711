#!/usr/bin/env python3
"""Headline benchmark: mel-frames/sec of one Reformer-TTS training step (forward + loss + reversible
backward + gradient all-reduce + clip + AdamW) on synthetic LJSpeech-shaped batches,
``config/baseline.yml`` (BASELINE.json configs[1]): B=12 per GPU, text 200 -> 256, mel 1024 x 80.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement): whole-job mel-frames/s, plus
  roofline      the dominant kernel (LSH chunk-attention backward, MFMA-bound): algorithmic FLOP per
                launch / its average duration measured with HIP events on the launch stream
  cpu_baseline  the CPU oracle (eager fp32 PyTorch restatement of the reference's step) timed on
                this host's cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=12, help="per-GPU batch (baseline.yml: 12)")
    ap.add_argument("--mel-len", type=int, default=1024)
    ap.add_argument("--text-len", type=int, default=200)
    ap.add_argument("--config", default="baseline", choices=["baseline", "long"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a hipGraph")
    ap.add_argument("--cpu-budget-s", type=float, default=25.0)
    ap.add_argument("--recompute", default="stash", choices=["stash", "attention-stash", "full"],
                    help="what the reversible backward recomputes: 'full' = everything, as the reference does; "
                         "'attention-stash' = attention outputs kept; 'stash' (default) = attention outputs and block outputs "
                         "f(x) kept, LayerNorm and the input projections recomputed, streams reconstructed by subtraction")
    return ap.parse_args()


def cpu_baseline(model_cfg, mel_len, text_len, budget_s):
    """Oracle forward + loss + backward on the host cores, bounded sample: B=1 at the bench's
    sequence lengths (frames/s is size-normalised), 1 warm-up-free pass repeated while time remains."""
    from oracle import model_ref, synth
    from reformer_tts_amd.model.config import as_kwargs
    cores = min(len(os.sched_getaffinity(0)), 16)      # the GPU box grants 16 host cores per GPU
    torch.set_num_threads(cores)
    cfg = as_kwargs(model_cfg)
    from reformer_tts_amd.training import build_model
    shapes = {k: tuple(v.shape) for k, v in build_model(model_cfg).state_dict().items()}
    sd = {k: v.requires_grad_(v.dtype.is_floating_point and not k.endswith("inv_freq"))
          for k, v in synth.synth_state_dict(shapes, seed=0).items()}
    batch = model_ref.synthetic_batch(1, text_len, mel_len, seed=42)
    g = torch.Generator().manual_seed(1)
    rots = [torch.randn(s, generator=g) for s in model_ref.rotation_shapes(cfg, text_len, mel_len)]
    times = []
    t_end = time.perf_counter() + budget_s
    while True:
        t0 = time.perf_counter()
        loss = model_ref.training_forward(sd, cfg, batch, rots)[0]
        loss.backward()
        times.append(time.perf_counter() - t0)
        for v in sd.values():
            v.grad = None
        if time.perf_counter() + times[-1] > t_end or len(times) >= 3:
            break
    best = min(times)
    return dict(value=round(mel_len / best, 2), unit="mel-frames/s", cores=cores, kind="port",
                sample=f"B=1, text {text_len}, mel {mel_len}: forward+loss+backward of oracle/model_ref.py "
                       f"(fp32 eager, {len(times)} pass(es), best {best:.2f} s; no optimizer step)")


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("RTTS_DIST_BACKEND", "nccl")     # "gloo": single-GPU rehearsal of the N>1 code path
    if backend == "gloo":
        local_rank %= max(torch.cuda.device_count(), 1)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from reformer_tts_amd import engine, ops
    engine.STASH_ATTENTION = args.recompute != "full"
    engine.STASH_BLOCK_OUTPUT = args.recompute == "stash"
    from reformer_tts_amd.model.config import (baseline_model_config, baseline_training_config,
                                               long_sequence_model_config)
    from reformer_tts_amd.training import Trainer, build_model, synthetic_batch

    model_cfg = baseline_model_config() if args.config == "baseline" else long_sequence_model_config()
    tcfg = baseline_training_config()
    tcfg.batch_size = args.batch
    model = build_model(model_cfg, dev, seed=42)          # identical init on every rank
    trainer = Trainer(model, tcfg, dev)
    batch = synthetic_batch(args.batch, args.text_len, args.mel_len, seed=42 + rank, device=dev)

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    dec_bucket = int(model_cfg.dec_reformer_kwargs.self_attn_kwargs.bucket_size)
    tkey = f"rtts_lsh_attn_bwd/bs{dec_bucket}"        # the decoder's LSH backward: the dominant kernel of the step
    note(f"model built ({trainer.n_params} parameters), warming up")
    # N == 1: the whole step is one hipGraph.  N > 1: four graphs (fwd + decoder-side bwd | encoder stack bwd | encoder
    # prenet bwd | clip+AdamW) around three eager all-reduces of parts of the flat gradient buffer, so no collective is
    # ever captured;
    # RTTS_GRAPH_DP=1 opts into a single graph with the per-block RCCL all-reduces captured inside (overlapped with the
    # backward; not exercisable on a 1-GPU box).
    use_graph = not args.no_graph
    one_graph = world == 1 or (os.environ.get("RTTS_GRAPH_DP") == "1" and backend == "nccl")
    if use_graph:
        ok = 1
        try:
            trainer.capture(batch, segmented=not one_graph)
        except Exception as exc:  # noqa: BLE001  (a capture that fails must not cost the whole measurement)
            ok = 0
            print(f"[bench] rank {rank}: graph capture failed ({type(exc).__name__}: {exc}); falling back to eager launches",
                  file=sys.stderr, flush=True)
        if world > 1:           # every rank takes the same path: collectives are issued outside the graphs
            flag = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        use_graph = bool(ok)
    if use_graph:
        step_fn = trainer.replay
        note("step captured into " + ("one hipGraph" if one_graph else "four hipGraphs around the three gradient all-reduces"))
    else:
        trainer._bulk_allreduce = False               # eager: per-block all-reduce overlapped with the backward
        step_fn = lambda: trainer.train_step(batch)   # noqa: E731
    for i in range(args.warmup):
        step_fn()
        torch.cuda.synchronize()
        note(f"warm-up step {i} done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if not use_graph:
        ops.TIMING.enable(tkey)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step_fn()[0]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    note(f"timed region done: {1e3 * elapsed / args.steps:.2f} ms/step")
    if use_graph:
        # HIP events cannot be read back from inside a replayed graph: time the dominant kernel on the same stream over
        # the same number of eager steps of the same workload, directly after the timed region
        ops.TIMING.enable(tkey)
        for _ in range(args.steps):
            trainer.train_step(batch)
        torch.cuda.synchronize()
    avg_ms, launches, flops_per_launch = ops.TIMING.summary(tkey)
    ops.TIMING.disable()

    if not (float(loss) == float(loss)):
        raise RuntimeError("training loss is not finite")
    if rank == 0:
        frames = world * args.batch * args.mel_len * args.steps
        out = {
            "metric": "mel-frames/sec training step, LJSpeech-shape batch",
            "value": round(frames / elapsed, 1), "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"config/baseline.yml full Reformer-TTS training step (enc 3 / dec 3 layers, d=512, "
                                   f"LSH 8 rounds, buckets 64/128), per-GPU batch {args.batch}, text {args.text_len}->256, "
                                   f"mel {args.mel_len}x80" if args.config == "baseline" else
                                   f"config/bucket-size-64-18-06.yml, per-GPU batch {args.batch}, mel {args.mel_len}",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}", "final_loss": round(float(loss), 4), "reversible_recompute": args.recompute,
                       "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2**30, 2),
                       "launch": ("hipGraph replay" if one_graph else "hipGraph replay (fwd+dec bwd | all-reduce | enc stack bwd | all-reduce | enc prenet bwd | all-reduce | optimizer)")
                       if use_graph else "eager"},
        }
        if launches:
            traffic = None      # HBM bytes per launch from the committed rocprofv3 PMC passes (collected outside this process)
            try:
                with open(os.path.join(ROOT, "profiles", "r01k_pmc_lsh_attn_bwd.json")) as fh:
                    traffic = json.load(fh)["traffic_bytes"] if (args.batch, args.mel_len, args.config) == (12, 1024, "baseline") else None
            except OSError:
                pass
            ach = flops_per_launch / (avg_ms * 1e-3) / 1e12
            out["roofline"] = {"kernel": "lsh_attn_bwd_kernel", "bound": "mfma", "achieved": round(ach, 2),
                               "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / BF16_DENSE_PEAK_TFLOPS, 4),
                               "traffic": traffic, "avg_launch_ms": round(avg_ms, 4), "launches_timed": launches}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model_cfg, args.mel_len, args.text_len, args.cpu_budget_s)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
