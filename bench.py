#!/usr/bin/env python3
"""Headline benchmark: mel-frames/sec of one Reformer-TTS training step (forward + loss + reversible
backward + gradient all-reduce + clip + AdamW) on synthetic LJSpeech-shaped batches,
``config/baseline.yml`` (BASELINE.json configs[1]): B=12 per GPU, text 200 -> 256, mel 1024 x 80.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement).  WHICH MODE IS THE HEADLINE: ``value`` / ``ms_per_step`` are
the step with the reference's reversible ACTIVATION RECOMPUTE (``--recompute full``, the default): only each stack's output
and the LSH sort permutations survive the forward; the backward reconstructs the streams by subtraction and re-runs
LayerNorm, projections, attention and feed-forward of every block, as ``reformer_tts/model/reversible.py:69-98`` does and as
BASELINE.json's north_star words it.  The memory-for-time mode this package also offers (every activation the backward
needs kept in HBM, ~1 GB of 288 GB, same gradients to 3e-4: DESIGN.md section 4) is timed in the same run and reported
under the top-level key ``stash`` (its own value / ms_per_step / peak memory) -- never mixed into ``value``.  Also:
  roofline        the dominant kernel (LSH chunk-attention backward, MFMA-bound): algorithmic FLOP per launch / its average
                  duration measured with HIP events on the launch stream; ``traffic`` = HBM bytes per launch from the
                  committed rocprofv3 PMC passes (profiles/<round>_pmc_traffic.json) IF that file names the kernel and
                  shape this run launches -- otherwise null and ``traffic_stale`` says why (``--strict`` raises instead)
  rooflines       that entry plus lsh_attn_fwd (MFMA), lsh_hash_sort (HBM-bound, SURVEY.md 8(d) bytes) and rtts_gemm_nt (every
                  projection / feed-forward / convolution GEMM of the step, MFMA-bound), each against the spec peak AND the
                  peak measured on this box by a stream copy / an MFMA loop (``peak_measured``)
  cpu_baseline    the CPU oracle (eager fp32 PyTorch restatement of the reference's step) timed on this host's cores on a
                  bounded sample of the same workload: one warm-up pass, then up to 3 timed passes (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA (spec)
HBM_PEAK_GBS = 8000.0             # 8 TB/s HBM3E (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (baseline.yml: 12; long: 4)")
    ap.add_argument("--mel-len", type=int, default=None)
    ap.add_argument("--text-len", type=int, default=200)
    ap.add_argument("--config", default="baseline", choices=["baseline", "long"],
                    help="baseline = config/baseline.yml (BASELINE.json configs[1]); long = config/bucket-size-64-18-06.yml at "
                         "mel 4096 (configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the full-recompute leg and the peak probes")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a hipGraph")
    ap.add_argument("--cpu-budget-s", type=float, default=25.0)
    ap.add_argument("--strict", action="store_true", help="raise when the committed PMC traffic file does not match the kernel / shape of this run")
    ap.add_argument("--recompute", default="full", choices=["stash", "projection-stash", "output-stash", "attention-stash", "full"],
                    help="TTSTrainingConfig.recompute for the headline leg -- what the reversible backward recomputes: 'full' (the "
                         "default HERE, the headline: BASELINE.json's north_star names the activation recompute) = everything, as the "
                         "reference does; 'attention-stash' = attention outputs kept; 'output-stash' = block outputs f(x) kept too; "
                         "'projection-stash' = and the projections, streams still reconstructed by subtraction; 'stash' (the "
                         "package's own default for training, the secondary number reported here) = and the streams: every sublayer's "
                         "LayerNorm input / output and projections stay in HBM (~1 GB of 288 GB), the backward recomputes nothing.  "
                         "The other end of the range is timed in the same run (top-level key 'stash' or 'full_recompute')")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 12 if args.config == "baseline" else 4
    if args.mel_len is None:
        args.mel_len = 1024 if args.config == "baseline" else 4096
    return args


def cpu_baseline(model_cfg, mel_len, text_len, budget_s):
    """Oracle forward + loss + backward on the host cores, bounded sample: B=1 at the bench's sequence lengths (frames/s is
    size-normalised); ONE warm-up pass, then timed passes while the budget lasts (at most 3)."""
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

    def one_pass():
        t0 = time.perf_counter()
        loss = model_ref.training_forward(sd, cfg, batch, rots)[0]
        loss.backward()
        dt = time.perf_counter() - t0
        for v in sd.values():
            v.grad = None
        return dt

    warm = one_pass()
    times = []
    t_end = time.perf_counter() + max(budget_s - warm, 0.0)
    while len(times) < 3 and (not times or time.perf_counter() + times[-1] <= t_end):
        times.append(one_pass())
    mean = sum(times) / len(times)
    return dict(value=round(mel_len / mean, 2), unit="mel-frames/s", cores=cores, kind="port",
                sample=f"B=1, text {text_len}, mel {mel_len}: forward+loss+backward of oracle/model_ref.py (fp32 eager; 1 warm-up "
                       f"pass of {warm:.2f} s, then {len(times)} timed pass(es), mean {mean:.2f} s, best {min(times):.2f} s; no "
                       f"optimizer step)")


def measure_peaks(dev):
    """Stream-copy GB/s and bf16 MFMA TFLOP/s this box reaches (HIP events on the current stream)."""
    from reformer_tts_amd import _lib
    s = torch.cuda.current_stream().cuda_stream
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    src.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        _lib.call("rtts_peak_copy", src.data_ptr(), dst.data_ptr(), n, s)
    a.record()
    for _ in range(5):
        _lib.call("rtts_peak_copy", src.data_ptr(), dst.data_ptr(), n, s)
    b.record()
    torch.cuda.synchronize()
    copy_gbs = 5 * 2 * n / (a.elapsed_time(b) * 1e-3) / 1e9
    del src, dst
    sink = torch.zeros(64, dtype=torch.float32, device=dev)
    wgs, iters = 256, 20000                        # one 8-wave workgroup per CU = two waves per SIMD
    _lib.call("rtts_peak_mfma", sink.data_ptr(), wgs, 2000, s)
    a.record()
    _lib.call("rtts_peak_mfma", sink.data_ptr(), wgs, iters, s)
    b.record()
    torch.cuda.synchronize()
    mfma_tf = wgs * 8 * iters * 8 * 16384.0 / (a.elapsed_time(b) * 1e-3) / 1e12
    return round(copy_gbs, 1), round(mfma_tf, 1)


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

    from reformer_tts_amd.model.config import (baseline_model_config, baseline_training_config,
                                               long_sequence_model_config)
    from reformer_tts_amd.training import Trainer, build_model, synthetic_batch

    model_cfg = baseline_model_config() if args.config == "baseline" else long_sequence_model_config()
    tcfg = baseline_training_config()
    tcfg.batch_size = args.batch
    tcfg.recompute = args.recompute                       # TTSTrainingConfig.recompute: a configuration, read by Trainer

    def set_mode(mode):
        """The second leg of the run re-uses the trainer in the other mode: its configuration field, nothing else."""
        trainer.cfg.recompute = trainer.recompute = mode
        trainer._recompute_for.clear()

    model = build_model(model_cfg, dev, seed=42)          # identical init on every rank ...
    trainer = Trainer(model, tcfg, dev)                   # ... rank-dependent rotations and dropout (Trainer._decorrelate_replicas)
    batch = synthetic_batch(args.batch, args.text_len, args.mel_len, seed=42 + rank, device=dev)

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    dec_bucket = int(model_cfg.dec_reformer_kwargs.self_attn_kwargs.bucket_size)
    heads_dec = int(model_cfg.dec_reformer_kwargs.self_attn_kwargs.heads)
    n_hashes_dec = int(model_cfg.dec_reformer_kwargs.self_attn_kwargs.n_hashes)
    t_dec = -(-args.mel_len // 256) * 256
    tags = dict(bwd=f"rtts_lsh_attn_bwd/bs{dec_bucket}", fwd=f"rtts_lsh_attn_fwd/bs{dec_bucket}",
                hash=f"rtts_lsh_hash_sort/nb{t_dec // dec_bucket}", gemm="rtts_gemm_nt")
    note(f"model built ({trainer.n_params} parameters), warming up")
    # N == 1: the whole step is one hipGraph.  N > 1: a chain of graphs (fwd + heads/postnet bwd | one per decoder layer |
    # encoder stack bwd | encoder prenet bwd | clip+AdamW) with the all-reduce of each graph's slice of the flat gradient
    # buffer issued between them, so no collective is ever captured.
    use_graph = not args.no_graph
    one_graph = world == 1

    def capture():
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
        return bool(ok)

    def settle():
        """Host garbage out of the way before the warm-up steps of a timed region (see `timed`)."""
        torch.cuda.synchronize()
        gc.collect()

    def timed(step_fn, steps):
        # A collection of the host's garbage inside the timed region can free a replaced hipGraph and its memory pool (a second
        # capture on the same trainer leaves one behind): hipFree synchronises the device -- seen as a one-off 5-9 ms stall in a
        # 10-step region (profiles/r04b_stash_variance.log).  Collect now, and keep the collector off while the clock runs.
        # (The collection itself runs BEFORE the warm-up steps -- `settle` -- so that the device does not sit idle through it right
        #  in front of the clock: a 10-step region started from an idle chip read 6.39 ms/step where 100 steps read 6.31.)
        gc.disable()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            for _ in range(steps):
                last = step_fn()[0]
            torch.cuda.synchronize()
        finally:
            gc.enable()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, last

    def graph_timed(fn, calls, replays):
        """-> (average ms per call, calls timed): ``calls`` invocations of ``fn`` captured into one hipGraph, replayed."""
        from reformer_tts_amd._graphs import capturing
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with capturing(g):
            for _ in range(calls):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(replays):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / (calls * replays), calls * replays

    if use_graph:
        use_graph = capture()
    if use_graph:
        step_fn = trainer.replay
        n_graphs = len(trainer._segments) + len(getattr(trainer, "_tail_main", ())) + len(getattr(trainer, "_tail_side", ()))
        note("step captured into " + ("one hipGraph" if one_graph else f"{n_graphs + 1} hipGraphs around {len(trainer.segment_plan())} gradient all-reduces"))
    else:
        trainer._bulk_allreduce = False               # eager: per-block all-reduce overlapped with the backward
        step_fn = lambda: trainer.train_step(batch)   # noqa: E731
    settle()
    for i in range(args.warmup):          # back to back, like the timed steps: the region starts from a busy chip, not from an idle one
        step_fn()
    note(f"{args.warmup} warm-up step(s) launched")
    elapsed, loss = timed(step_fn, args.steps)
    note(f"timed region done: {1e3 * elapsed / args.steps:.2f} ms/step")

    # HIP events cannot be read back from inside a replayed graph: time the roofline kernels on the same stream over eager
    # steps of the same workload, directly after the timed region
    n_eager = min(args.steps, 10)
    ops.TIMING.enable(*tags.values())
    for _ in range(n_eager):
        trainer.train_step(batch)
    torch.cuda.synchronize()
    timing = {k: ops.TIMING.summary(t) for k, t in tags.items()}
    gemm_shapes = {k.split("/")[1]: ops.TIMING.summary(k) for k in ops.TIMING.records if k.startswith("rtts_gemm_nt/")}
    ops.TIMING.disable()
    if not (float(loss) == float(loss)):
        raise RuntimeError("training loss is not finite")

    # the other end of the recompute range, in the same run (a second capture on the same trainer): the memory-for-time
    # mode when the headline is the reference's full recompute, and the other way round
    peak_mem_gb = round(torch.cuda.max_memory_allocated(dev) / 2**30, 2)
    other_mode = "stash" if args.recompute != "stash" else "full"
    other = None
    if not args.no_extra:
        torch.cuda.reset_peak_memory_stats(dev)
        set_mode(other_mode)
        ok = capture() if use_graph else True
        fn = trainer.replay if (use_graph and ok) else (lambda: trainer.train_step(batch))
        settle()
        for _ in range(2):
            fn()
        k = max(3, min(args.steps, 10))
        dt, _ = timed(fn, k)
        other = {"ms_per_step": round(1e3 * dt / k, 3), "value": round(world * args.batch * args.mel_len * k / dt, 1),
                 "unit": "mel-frames/s", "steps": k, "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2**30, 2),
                 "reversible_recompute": other_mode,
                 "what": ("memory-for-time mode: every sublayer's LayerNorm input/output, projections, attention outputs and block "
                          "outputs are kept in HBM and the backward recomputes nothing; same gradients as the headline mode to 3e-4 "
                          "(tests/test_model_hip.py::test_attention_stash_matches_pure_recompute)") if other_mode == "stash" else
                         "the reference's pure activation recompute (reversible.py:69-98)"}
        set_mode(args.recompute)
        note(f"{other_mode} mode: {other['ms_per_step']} ms/step")
    peaks = (None, None)
    if rank == 0 and not args.no_extra:
        peaks = measure_peaks(dev)
        note(f"measured on this box: stream copy {peaks[0]} GB/s, bf16 MFMA loop {peaks[1]} TFLOP/s")

    if rank == 0:
        frames = world * args.batch * args.mel_len * args.steps
        n_par = trainer.n_params
        launch = ("hipGraph replay (one graph; the encoder is a parallel branch beside the decoder: Trainer.train_step_overlapped)" if one_graph else
                  "hipGraph replay (fwd + heads/postnet bwd | one graph per decoder layer bwd, the lowest cut at the keys' gradient | behind the cut two lanes side by side: rest of that layer + decoder prenet || one graph per encoder block + encoder prenet | optimizer; the "
                  "all-reduce of a graph's gradient range runs while the next graph replays: config.dist.schedule)") \
            if use_graph else "eager"
        workload = (f"config/baseline.yml full Reformer-TTS training step (enc 3 / dec 3 layers, d=512, LSH 8 rounds, buckets 64/128), "
                    f"per-GPU batch {args.batch}, text {args.text_len}->256, mel {args.mel_len}x80" if args.config == "baseline" else
                    f"config/bucket-size-64-18-06.yml full Reformer-TTS training step (enc 6 / dec 6 layers, d=512, LSH 8 rounds, "
                    f"buckets 64/64, dropouts 0.15), per-GPU batch {args.batch}, text {args.text_len}->256, mel {args.mel_len}x80")
        out = {
            "metric": "mel-frames/sec training step, LJSpeech-shape batch",
            "value": round(frames / elapsed, 1), "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": workload, "global_batch": world * args.batch, "parallelism": f"dp{world}",
                       "final_loss": round(float(loss), 4), "reversible_recompute": args.recompute,
                       "peak_hbm_gb": peak_mem_gb, "launch": launch,
                       # what a multi-GPU record can be checked against
                       "dist": {"world_size": world, "backend": (dist.get_backend() if world > 1 else None),
                                "backend_ranks": (dist.get_world_size() if world > 1 else 1),
                                "allreduce_bytes_per_step": (4 * n_par if world > 1 else 0),
                                "collectives_per_step": (len(trainer.segment_plan()) if (world > 1 and use_graph) else (0 if world == 1 else "per block")),
                                "schedule": (trainer.segment_plan() if (world > 1 and use_graph) else None),
                                "rank_seeds": "rotations and dropout seeded with seed + rank",
                                # the communication policy (one-GPU probe: profiles/r04_comm_overlap_probe.log)
                                "policy": ("all-reduces on the collective library's own stream beside the chain of graphs: one graph per "
                                           "decoder layer and per encoder block, a graph's gradient range exchanged while the next graph "
                                           "replays, only the encoder prenet's 17 MB has nothing to hide behind; NO CU mask on the compute "
                                           "stream (grids are sized for 256 CUs: 248 cost +26 % in the probe); fp32 messages of 7-17 MB")
                                if world > 1 else None}},
        }
        if other is not None:
            out["stash" if other_mode == "stash" else "full_recompute"] = other
        rooflines = []
        from reformer_tts_amd import _lib
        shape_key = {"BH": args.batch * heads_dec, "T": t_dec, "bucket": dec_bucket, "rounds": n_hashes_dec}

        def pmc_traffic(kernel_prefix):
            """HBM bytes per launch of ``kernel_prefix`` at ``shape_key`` from the newest committed profiles/*_pmc_traffic.json
            (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, collected outside this process; scripts/pmc_summary.py), or
            (None, reason) when no entry names this kernel at this shape -- a changed kernel must not inherit old counters."""
            import glob
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
            if not files:
                return None, "no profiles/*_pmc_traffic.json committed"
            with open(files[-1]) as fh:
                doc = json.load(fh)
            for e in doc.get("entries", []):
                if e["kernel"].startswith(kernel_prefix) and e.get("shape") == shape_key:
                    return int(e["traffic_bytes"]), os.path.basename(files[-1])
            return None, (f"{os.path.basename(files[-1])} has no entry for {kernel_prefix}* at {shape_key} "
                          f"(it holds: {[e['kernel'] for e in doc.get('entries', [])]})")

        def mfma_entry(kernel, avg_ms, launches, flops, extra=None):
            ach = flops / (avg_ms * 1e-3) / 1e12
            traffic, src = pmc_traffic(kernel)
            if traffic is None:
                if args.strict:
                    raise RuntimeError(f"roofline.traffic: {src}")
                note(f"ROOFLINE TRAFFIC NOT REPORTED for {kernel}: {src}")
            e = {"kernel": kernel, "bound": "mfma", "achieved": round(ach, 2), "peak": BF16_DENSE_PEAK_TFLOPS,
                 "unit": "TFLOP/s", "frac": round(ach / BF16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic,
                 ("traffic_source" if traffic is not None else "traffic_stale"): src,
                 "algorithmic_flop_per_launch": flops, "avg_launch_ms": round(avg_ms, 4), "launches_timed": launches,
                 "peak_measured": peaks[1], "frac_of_measured": (round(ach / peaks[1], 4) if peaks[1] else None)}
            e.update(extra or {})
            return e

        avg_ms, launches, flops = timing["bwd"]
        if launches:
            run = _lib.load().rtts_lsh_attn_bwd_run_length(args.batch, heads_dec, t_dec, n_hashes_dec, dec_bucket)
            entry = mfma_entry(f"lsh_attn_bwd_walk_kernel<{dec_bucket}" if run > 0 else f"lsh_attn_bwd_kernel<{dec_bucket}",
                               avg_ms, launches, flops, {"chunks_per_workgroup": max(run, 1)})
            out["roofline"] = entry
            rooflines.append(entry)
        avg_ms, launches, flops = timing["fwd"]
        if launches:
            run = _lib.load().rtts_lsh_attn_fwd_run_length(args.batch, heads_dec, t_dec, n_hashes_dec, dec_bucket)
            rooflines.append(mfma_entry(f"lsh_attn_fwd_walk_kernel<{dec_bucket}" if run > 0 else f"lsh_attn_fwd_kernel<{dec_bucket}",
                                        avg_ms, launches, flops, {"chunks_per_workgroup": max(run, 1)}))
        avg_ms, launches, nbytes = timing["hash"]
        if launches:
            # a 10-15 us launch is host-bound when launched eagerly (Python + ctypes per call): its device time comes from a
            # hipGraph of 20 calls at the same shape, replayed, with HIP events on the replay stream
            # ... over a ROTATION of distinct inputs: 24 qk buffers of 12.6 MB + their 3 MB outputs = 0.38 GB per pass, more than the
            # 256 MB Infinity Cache holds, so every call fetches its rows from HBM (one buffer re-hashed 200 times -- round 3 -- is an
            # Infinity-Cache figure; it is reported beside this one as `achieved_cached`)
            n_rot = 24
            rot_probe = torch.randn(1, 64, n_hashes_dec, t_dec // dec_bucket // 2, device=dev)
            qk_probes = [torch.randn(args.batch, t_dec, heads_dec * 64, device=dev).bfloat16() for _ in range(n_rot)]
            it = [0]

            def hash_next():
                ops.lsh_hash_sort(qk_probes[it[0] % n_rot], rot_probe, heads_dec, dec_bucket)
                it[0] += 1
            avg_ms, launches = graph_timed(hash_next, 2 * n_rot, 5)
            cached_ms, _ = graph_timed(lambda: ops.lsh_hash_sort(qk_probes[0], rot_probe, heads_dec, dec_bucket), 20, 10)
            del qk_probes
            two = _lib.load().rtts_lsh_hash_sort_launches(t_dec) == 2
            ach = nbytes / (avg_ms * 1e-3) / 1e9
            # counters of the launch(es) one call makes (the hash of all rounds and the sorts are two kernels from T = 1024 on)
            parts = [pmc_traffic(k) for k in (("lsh_hash_rounds_kernel", "lsh_sort_ids_kernel") if two else ("lsh_hash_sort_kernel",))]
            h_traffic = sum(p[0] for p in parts) if all(p[0] is not None for p in parts) else None
            rooflines.append({"kernel": "lsh_hash_rounds_kernel + lsh_sort_ids_kernel" if two else "lsh_hash_sort_kernel", "bound": "hbm",
                              "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": h_traffic,
                              ("traffic_source" if h_traffic is not None else "traffic_stale"): parts[0][1] if h_traffic is not None
                              else next(p[1] for p in parts if p[0] is None),
                              "avg_launch_ms": round(avg_ms, 4),
                              "launches_timed": launches, "timed": f"hipGraph of {2 * n_rot} calls over a rotation of {n_rot} distinct inputs "
                              "(0.38 GB per pass: beyond the 256 MB Infinity Cache, so the rows come from HBM), replayed 5 times (device time; "
                              "eager launches of a kernel this short are host-bound)",
                              "achieved_cached": round(nbytes / (cached_ms * 1e-3) / 1e9, 1), "avg_launch_ms_cached": round(cached_ms, 4),
                              "peak_measured": peaks[0],
                              "frac_of_measured": (round(ach / peaks[0], 4) if peaks[0] else None),
                              "algorithmic_bytes_per_launch": int(nbytes),
                              "note": "SURVEY.md 8(d): 224 B per token and head (hash 128 + 32, sort 64); decoder shape; one rtts_lsh_hash_sort "
                                      "call = " + ("two launches (hash of all rounds, then the sorts)" if two else "one launch")})
        avg_ms, launches, flops = timing["gemm"]
        if launches:
            ach = flops / (avg_ms * 1e-3) / 1e12
            big = max(gemm_shapes.items(), key=lambda kv: kv[1][2]) if gemm_shapes else None
            e = {"kernel": "gemm_nt_kernel (all shapes of the step)", "bound": "mfma", "achieved": round(ach, 2),
                 "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / BF16_DENSE_PEAK_TFLOPS, 4), "traffic": None,
                 "avg_launch_ms": round(avg_ms, 4), "launches_timed": launches, "peak_measured": peaks[1],
                 "frac_of_measured": (round(ach / peaks[1], 4) if peaks[1] else None)}
            if big is not None:
                bms, bn, bfl = big[1]
                e["largest_shape"] = {"MxNxK": big[0], "avg_launch_ms": round(bms, 4), "launches_timed": bn,
                                      "achieved": round(bfl / (bms * 1e-3) / 1e12, 2)}
            rooflines.append(e)
        out["rooflines"] = rooflines
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model_cfg, args.mel_len, args.text_len, args.cpu_budget_s)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
