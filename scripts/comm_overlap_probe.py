#!/usr/bin/env python3
"""One-GPU probe for the data-parallel schedule: how much of a ring all-reduce hides behind the backward?

On N GPUs the gradient range of every captured segment (forward + heads/postnet backward | one graph per decoder layer backward |
encoder stack backward | encoder prenet backward) is all-reduced by RCCL on a second queue while the NEXT segment replays.  RCCL's
kernel is not a chip-filling grid: a few resident workgroups move the message at the pace of the xGMI links.  This script stands
such a kernel in (``rtts_comm_probe``: 8-32 workgroups, 16 KB pieces, write-through stores, paced so that a 17 MB message takes
what a ring over 7 links of ~50 GB/s effective would: 2 * 7/8 * bytes / 175 GB/s) and replays the REAL chain of graphs with it:

  chain alone | comm alone | chain + comm on a second stream | chain on a CU-MASKED stream (the collective's CUs left free) + comm

and reports the step time, the time the last message finishes after the last segment (exposed communication) and what the chain
pays.  (GPU box only.)   python scripts/comm_overlap_probe.py > gpurun_out/r04_comm_overlap_probe.log"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from reformer_tts_amd import _lib
from reformer_tts_amd.model.config import baseline_model_config, baseline_training_config
from reformer_tts_amd.training import Trainer, build_model, synthetic_batch

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
BUS_GBS = float(os.environ.get("PROBE_BUS_GBS", "175"))        # effective ring bandwidth the pacing emulates
WORLD = 8


def masked_stream(free_cus: int):
    """A HIP stream whose kernels may use all CUs but `free_cus` (hipExtStreamCreateWithCUMask), as a torch ExternalStream."""
    hip = None
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            hip = C.CDLL(name)
            break
        except OSError:
            continue
    if hip is None:
        return None, "libamdhip64 not loadable by name"
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    words = (ncu + 31) // 32
    mask = (C.c_uint32 * words)()
    # leave `free_cus` CUs out, spread over the mask (one bit in every ncu / free_cus)
    skip = set(range(0, ncu, max(1, ncu // free_cus))) if free_cus else set()
    skip = set(sorted(skip)[:free_cus])
    for cu in range(ncu):
        if cu not in skip:
            mask[cu // 32] |= 1 << (cu % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(words), mask)
    if rc != 0:
        return None, f"hipExtStreamCreateWithCUMask failed with {rc}"
    return torch.cuda.ExternalStream(st.value, device=dev), f"{ncu - len(skip)} of {ncu} CUs"


def main():
    model = build_model(baseline_model_config(), dev, seed=42)
    tcfg = baseline_training_config()
    tcfg.batch_size = 12
    tcfg.recompute = os.environ.get("PROBE_RECOMPUTE", "full")
    tr = Trainer(model, tcfg, dev)
    tr.two_lane_tail = False        # this probe walks ONE lane of (graph, range) pairs: the chain before the two-lane tail
    batch = synthetic_batch(12, 200, 1024, seed=42, device=dev)
    tr.capture(batch, segmented=True)
    plan = tr.segment_plan()
    segs = tr._segments
    msgs = [4 * (e - s) for _, (s, e) in segs]
    print(f"chain: {len(segs)} graphs + optimizer; messages (MB): " + ", ".join(f"{m / 1e6:.1f}" for m in msgs), flush=True)
    for p in plan:
        print(f"   after '{p['after']}': {p['allreduce_bytes'] / 1e6:.1f} MB, overlaps {p['overlaps']}")
    src = torch.zeros(max(msgs) // 4, dtype=torch.int32, device=dev)
    dst = torch.zeros_like(src)
    main_s = torch.cuda.current_stream()
    comm_s = torch.cuda.Stream()

    def comm(nbytes, wgs, sleep, stream):
        _lib.call("rtts_comm_probe", src.data_ptr(), dst.data_ptr(), nbytes - nbytes % 16, wgs, sleep, stream.cuda_stream)

    def time_comm_alone(nbytes, wgs, sleep):
        for _ in range(2):
            comm(nbytes, wgs, sleep, main_s)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            comm(nbytes, wgs, sleep, main_s)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / 5 * 1e3

    def calibrate(wgs):
        """sleep units so that the 17 MB message takes 2 (N-1)/N bytes / BUS_GBS"""
        ref = 17.0e6
        target = 2.0 * (WORLD - 1) / WORLD * ref / (BUS_GBS * 1e9) * 1e6
        lo, hi = 0, 64
        while lo < hi:
            mid = (lo + hi) // 2
            if time_comm_alone(int(ref), wgs, mid) < target:
                lo = mid + 1
            else:
                hi = mid
        return lo, target

    def run_chain(with_comm, wgs=16, sleep=0, chain_stream=None, reps=10):
        """-> (us per step, us the last message ends after the last segment's end, us of the communication kernels alone)"""
        cs = chain_stream if chain_stream is not None else main_s
        tot, lag = 0.0, 0.0
        for rep in range(reps + 2):
            torch.cuda.synchronize()
            t0, t_seg, t_end = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            t_comm = torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(cs):
                t0.record(cs)
                tr.set_step_hyper(tr.global_step)
                tr.global_step += 1
                for (g, _), nbytes in zip(segs, msgs):
                    g.replay()
                    if with_comm:
                        ev = torch.cuda.Event()
                        ev.record(cs)
                        comm_s.wait_event(ev)
                        comm(nbytes, wgs, sleep, comm_s)
                t_seg.record(cs)
                if with_comm:
                    t_comm.record(comm_s)
                    cs.wait_stream(comm_s)
                tr._graph_opt.replay()
                t_end.record(cs)
            torch.cuda.synchronize()
            if rep >= 2:
                tot += t0.elapsed_time(t_end) * 1e3
                if with_comm:
                    lag += max(0.0, t_seg.elapsed_time(t_comm) * 1e3)
        return tot / reps, lag / reps

    base, _ = run_chain(False)
    print(f"chain alone: {base:8.1f} us per step", flush=True)
    for wgs in (8, 16, 32):
        sleep, target = calibrate(wgs)
        alone = [time_comm_alone(m, wgs, sleep) for m in msgs]
        both, lag = run_chain(True, wgs, sleep)
        print(f"comm grid {wgs:2d} workgroups, pace {sleep} units (17 MB in {time_comm_alone(17000000, wgs, sleep):6.1f} us, target {target:6.1f}): "
              f"messages alone {sum(alone):7.1f} us in all ({', '.join(f'{a:.0f}' for a in alone)}); chain + comm {both:8.1f} us per step "
              f"(+{both - base:6.1f} over the chain alone); last message ends {lag:6.1f} us after the last segment", flush=True)
    for free in (8, 16, 32):
        ms, what = masked_stream(free)
        if ms is None:
            print(f"CU-masked stream: not available ({what})")
            break
        try:
            mbase, _ = run_chain(False, chain_stream=ms)
            sleep, _ = calibrate(16)
            both, lag = run_chain(True, 16, sleep, chain_stream=ms)
            print(f"chain on a stream masked to {what}: alone {mbase:8.1f} us per step ({mbase - base:+6.1f} vs unmasked); + comm (16 workgroups) "
                  f"{both:8.1f} us ({both - base:+6.1f} vs the unmasked chain alone); last message ends {lag:6.1f} us after the last segment", flush=True)
        except Exception as exc:  # noqa: BLE001
            print(f"chain on a masked stream ({what}): failed: {type(exc).__name__}: {exc}")
            break


if __name__ == "__main__":
    main()
