#!/usr/bin/env python3
"""BASELINE config #5 on one MI355X: SqueezeWave vocoder throughput (mel -> audio) and end-to-end text -> mel -> audio with
ReformerTTS.infer in front of it (random-init weights of the default configurations; 22.05 kHz audio, 256 samples per mel
frame), next to the CPU oracle of the vocoder on a bounded mel length.

    python scripts/vocoder_bench.py > gpurun_out/vocoder_bench.json
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from reformer_tts_amd.model.config import baseline_model_config  # noqa: E402
from reformer_tts_amd.squeeze_wave import SqueezeWave, WNConfig  # noqa: E402
from reformer_tts_amd.training import build_model  # noqa: E402

SR = 22050


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mel-len", type=int, default=1024)
    ap.add_argument("--tts-frames", type=int, default=200)
    ap.add_argument("--cpu-mel-len", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    sw = SqueezeWave(12, 128, 80, 2, 16, WNConfig()).to(dev).eval()
    for wn in sw.wn_layers:                                   # end_conv is zero-initialised: give the flow something to do
        wn.end_conv.weight.data.normal_(0, 0.01)
    out = {"unit": "audio samples/s", "sample_rate": SR}
    for b in (1, 8):
        mel = (torch.randn(b, 80, args.mel_len) * 2 - 5).clamp(-11.5, 2.0).to(dev)
        dt, audio = timed(lambda: sw.infer(mel), 5)
        out[f"vocoder_B{b}"] = {"mel_len": args.mel_len, "ms": round(1e3 * dt, 3), "samples_per_s": round(audio.numel() / dt, 0),
                                "x_realtime": round(audio.numel() / dt / SR, 1)}
        run = sw.capture(b, args.mel_len)
        dt, audio = timed(lambda: run(mel), 20)
        out[f"vocoder_B{b}_graph"] = {"mel_len": args.mel_len, "ms": round(1e3 * dt, 3), "samples_per_s": round(audio.numel() / dt, 0),
                                      "x_realtime": round(audio.numel() / dt / SR, 1)}
    # CPU oracle of the vocoder (fp32 eager restatement of the reference), bounded
    from oracle import squeezewave_ref as sw_ref
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu().float() for k, v in sw.state_dict().items()}
    melc = (torch.randn(1, 80, args.cpu_mel_len) * 2 - 5).clamp(-11.5, 2.0)
    t0 = time.perf_counter()
    with torch.no_grad():
        a = sw_ref.infer(sd, sw_ref.default_cfg(), melc)
    dtc = time.perf_counter() - t0
    out["vocoder_cpu_oracle"] = {"mel_len": args.cpu_mel_len, "ms": round(1e3 * dtc, 1), "samples_per_s": round(a.numel() / dtc, 0),
                                 "x_realtime": round(a.numel() / dtc / SR, 2), "cores": cores}
    # end to end: text -> mel (ReformerTTS.infer, encoder cached) -> audio
    tts = build_model(baseline_model_config(), dev, seed=42)
    ph = torch.randint(1, 77, (1, 200), generator=torch.Generator().manual_seed(0))

    def e2e():
        spec, _ = tts.infer(ph, max_len=args.tts_frames, stop_at_stop_token=False, cache_encoder=True)
        return sw.infer(spec)
    def e2e_graph():
        spec, _ = tts.infer(ph, max_len=args.tts_frames, stop_at_stop_token=False, cache_encoder=True, use_graph=True)
        return sw.infer(spec)
    dtg, audio = timed(e2e_graph, 1)
    out["text_to_audio_B1_graph"] = {"frames": args.tts_frames, "ms": round(1e3 * dtg, 1), "samples_per_s": round(audio.numel() / dtg, 0),
                                     "x_realtime": round(audio.numel() / dtg / SR, 2), "note": "one hipGraph replay per frame (captures included)"}
    dt, audio = timed(e2e, 1)
    out["text_to_audio_B1"] = {"frames": args.tts_frames, "ms": round(1e3 * dt, 1), "samples_per_s": round(audio.numel() / dt, 0),
                               "x_realtime": round(audio.numel() / dt / SR, 2),
                               "note": "ReformerTTS.infer (one full decoder forward per frame, encoder cached) dominates"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
