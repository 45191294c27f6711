#!/usr/bin/env python3
"""Generation throughput of ReformerTTS.infer on one MI355X, config/baseline.yml, random-init
weights, B utterances of 200 phonemes, ``--frames`` mel frames each (stop tokens ignored so every run has the same
length), next to the CPU oracle's infer on the host cores for a bounded number of frames.

    python scripts/infer_bench.py --batch 1 --frames 300 > gpurun_out/infer_bench.json
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from reformer_tts_amd.model.config import as_kwargs, baseline_model_config  # noqa: E402
from reformer_tts_amd.training import build_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--cpu-frames", type=int, default=4)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = baseline_model_config()
    model = build_model(cfg, dev, seed=42)
    ph = torch.randint(1, 77, (args.batch, 200), generator=torch.Generator().manual_seed(0))
    out = {"workload": f"config/baseline.yml ReformerTTS.infer, B={args.batch}, 200 phonemes, {args.frames} frames, concat strategy",
           "unit": "mel-frames/s (all utterances)"}
    for name, kw in (("reference_semantics", dict()), ("cache_encoder", dict(cache_encoder=True)),
                     ("graph_per_frame", dict(use_graph=True)), ("graph_per_frame_cache_encoder", dict(use_graph=True, cache_encoder=True))):
        model.infer(ph, max_len=90, stop_at_stop_token=False, **kw)          # warm-up (allocator, lazy tables)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        spec, _ = model.infer(ph, max_len=args.frames, stop_at_stop_token=False, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name] = {"frames_per_s": round(args.batch * spec.shape[2] / dt, 1), "ms_per_frame": round(1e3 * dt / spec.shape[2], 3),
                     "frames": int(spec.shape[2])}
    # CPU oracle: the same loop (full fp32 forwards), bounded
    from oracle import model_ref, synth
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    c = as_kwargs(cfg)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0)
    for k, v in model.state_dict().items():
        if "running" in k:
            sd[k] = v.detach().cpu().clone()
    n_layers = model_ref.count_lsh_layers(c)
    g = torch.Generator().manual_seed(1)

    def rots():
        while True:
            for s in model_ref.rotation_shapes(c, 200, 256):
                yield torch.randn(s, generator=g)
    t0 = time.perf_counter()
    with torch.no_grad():
        # max_len must exceed n_mels for the loop to run at all (reference guard): count frames from 81 on
        spec, _ = model_ref.infer(sd, c, ph[:1], rots(), max_len=80 + args.cpu_frames, stop_at_stop_token=False)
    dt = time.perf_counter() - t0
    out["cpu_oracle"] = {"frames_per_s": round(spec.shape[2] / dt, 2), "ms_per_frame": round(1e3 * dt / spec.shape[2], 1),
                         "frames": int(spec.shape[2]), "cores": cores, "note": "B=1, fp32 eager restatement of the reference loop"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
