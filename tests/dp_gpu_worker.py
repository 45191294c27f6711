"""Worker of tests/test_dp_gpu.py: one rank of a 2-process gloo job sharing cuda:0.  Runs three training steps of the small
model in the two data-parallel modes of the trainer and checks (a) replicas stay identical, (b) the modes agree:
    eager      per-block all-reduce issued from the block-done hooks, overlapped with the backward
    graphs     a chain of hipGraphs (fwd + heads/postnet bwd | one per decoder layer | one per encoder block | encoder prenet bwd | clip + AdamW)
               with the all-reduce of each graph's gradient range issued while the next one replays
(parameters after three steps, loss of the third step)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import model_ref  # noqa: E402
from reformer_tts_amd.model.config import TTSTrainingConfig, model_config_from_dict  # noqa: E402
from reformer_tts_amd.model.lsh_attention import LSHSelfAttention  # noqa: E402
from reformer_tts_amd.training import Trainer, build_model, synthetic_batch  # noqa: E402


def run(mode, rank, dev):
    cfg = model_ref.small_cfg()
    cfg["enc_reformer_kwargs"]["attn_kwargs"]["implementation"] = "hip"
    cfg["dec_reformer_kwargs"]["self_attn_kwargs"]["implementation"] = "hip"
    cfg["enc_reformer_kwargs"]["depth"] = 2            # two encoder blocks: the chain cuts the encoder stack's backward per block
    torch.manual_seed(1)
    model = build_model(model_config_from_dict(cfg), dev)
    for m in model.modules():
        if isinstance(m, LSHSelfAttention):
            m.forced_rotations = torch.randn(1, 64, 4, (128 if not m.causal else 256) // 64 // 2,
                                             generator=torch.Generator().manual_seed(5))
    tr = Trainer(model, TTSTrainingConfig(batch_size=2, learning_rate=1e-3, warmup_steps=4, gradient_clip_val=1.0), dev)
    assert tr.world == 2
    batch = synthetic_batch(2, 100, 256, seed=42 + rank, device=dev)
    losses = []
    if mode == "eager":
        for _ in range(3):
            losses.append(float(tr.train_step(batch)[0]))
    else:
        tr.capture(batch)                      # two eager steps (bulk all-reduce), then the graphs are captured
        # forward+loss+heads | one graph per decoder layer (depth 1 here) | one per encoder block (2) | encoder prenet, then the optimizer
        assert len(tr._segments) == 5 and tr._graph_opt is not None, len(tr._segments)
        plan = tr.segment_plan()
        assert sum(p["allreduce_bytes"] for p in plan) == 4 * tr.n_params and plan[-1]["overlaps"].startswith("nothing")
        assert [p["after"] for p in plan][2:4] == ["encoder block 1 backward", "encoder block 0 backward"], plan
        losses = [None, None, float(tr.replay()[0])]
    torch.cuda.synchronize()
    return tr, losses


def main():
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    res = {}
    for mode in ("eager", "graphs"):
        tr, losses = run(mode, rank, dev)
        chk = tr.flat_p.double().sum().reshape(1).cpu()
        both = [torch.zeros_like(chk) for _ in range(2)]
        dist.all_gather(both, chk)
        assert torch.equal(both[0], both[1]), f"{mode}: replicas diverged {both}"
        res[mode] = (tr.flat_p.clone(), losses)
    pe, le = res["eager"]
    pg, lg = res["graphs"]
    rel = float((pe - pg).norm() / pe.norm())
    assert rel < 2e-3, f"parameters after 3 steps differ between the modes: rel {rel}"
    assert abs(le[2] - lg[2]) <= 2e-2 * abs(le[2]), (le, lg)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("DP_GPU_OK", rel, le[2], lg[2])


if __name__ == "__main__":
    main()
