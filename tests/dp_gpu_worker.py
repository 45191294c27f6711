"""Worker of tests/test_dp_gpu.py: one rank of a 2-process gloo job sharing cuda:0.  Runs three training steps of the small
model in the two data-parallel modes of the trainer and checks (a) replicas stay identical, (b) the modes agree:
    eager      per-block all-reduce issued from the block-done hooks, overlapped with the backward
    graphs     a chain of hipGraphs (fwd + heads/postnet bwd | one per decoder layer, the lowest cut at the keys' gradient | then two lanes side
               by side: rest of that layer + decoder prenet || one per encoder block + encoder prenet | clip + AdamW) with the all-reduce of
               each graph's gradient range issued while the next one replays
(parameters after three steps, loss of the third step).  Then the same pair with ``sync_batchnorm``: the chain keeps the three pieces
that hold BatchNorm's exchanges eager between its graphs (Trainer._capture_around_sync_bn); BatchNorm's running statistics
after three steps must agree too (the eager pieces execute once more while the chain is built: that must leave no trace)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import model_ref  # noqa: E402
from reformer_tts_amd.model.config import TTSTrainingConfig, model_config_from_dict  # noqa: E402
from reformer_tts_amd.model.lsh_attention import LSHSelfAttention  # noqa: E402
from reformer_tts_amd.training import Trainer, build_model, synthetic_batch  # noqa: E402


def run(mode, rank, dev, sync_bn=False):
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
    tr = Trainer(model, TTSTrainingConfig(batch_size=2, learning_rate=1e-3, warmup_steps=4, gradient_clip_val=1.0, sync_batchnorm=sync_bn), dev)
    assert (tr._sync_bn is not None) == sync_bn
    assert tr.world == 2
    batch = synthetic_batch(2, 100, 256, seed=42 + rank, device=dev)
    losses = []
    if mode == "eager":
        for _ in range(3):
            losses.append(float(tr.train_step(batch)[0]))
    else:
        tr.capture(batch)                      # two eager steps (bulk all-reduce), then the graphs are captured
        if sync_bn:
            # encoder prenet fwd (eager) | both stacks fwd | heads + postnet + loss fwd/bwd (eager) | decoder layer | 2 encoder blocks |
            # encoder prenet bwd (eager), then the optimizer: four graphs + clip/AdamW around three eager pieces
            assert len(tr._segments) == 7 and sum(isinstance(g, torch.cuda.CUDAGraph) for g, _ in tr._segments) == 4, tr._segments
            plan = tr.segment_plan()
            assert len(plan) == 5 and sum(p["allreduce_bytes"] for p in plan) == 4 * tr.n_params and plan[-1]["overlaps"].startswith("nothing"), plan
            losses = [None, None, float(tr.replay()[0])]
            torch.cuda.synchronize()
            return tr, losses
        # forward+loss+heads | decoder layer 0 up to the keys' gradient | then two lanes side by side: the rest of the layer + the decoder
        # prenet || one graph per encoder block (2) + encoder prenet; then the optimizer
        assert len(tr._segments) == 2 and len(tr._tail_main) == 1 and len(tr._tail_side) == 3 and tr._graph_opt is not None
        assert tr._segments[1][1] is None                      # the cut graph exchanges nothing: the layer's range is final behind it
        plan = tr.segment_plan()
        assert len(plan) == 5 and sum(p["allreduce_bytes"] for p in plan) == 4 * tr.n_params, plan
        assert [p["after"] for p in plan][1:3] == ["encoder block 1 backward (lane 2)", "encoder block 0 backward (lane 2)"], plan
        losses = [None, None, float(tr.replay()[0])]
    torch.cuda.synchronize()
    return tr, losses


def main():
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    report = []
    for sync_bn in (False, True):
        res = {}
        for mode in ("eager", "graphs"):
            tr, losses = run(mode, rank, dev, sync_bn)
            chk = tr.flat_p.double().sum().reshape(1).cpu()
            both = [torch.zeros_like(chk) for _ in range(2)]
            dist.all_gather(both, chk)
            assert torch.equal(both[0], both[1]), f"{mode} sync_bn={sync_bn}: replicas diverged {both}"
            stats = torch.cat([b.detach().float().flatten() for n, b in tr.model.named_buffers() if "running_" in n])
            res[mode] = (tr.flat_p.clone(), losses, stats)
        pe, le, se = res["eager"]
        pg, lg, sg = res["graphs"]
        rel = float((pe - pg).norm() / pe.norm())
        assert rel < 2e-3, f"sync_bn={sync_bn}: parameters after 3 steps differ between the modes: rel {rel}"
        assert abs(le[2] - lg[2]) <= 2e-2 * abs(le[2]), (sync_bn, le, lg)
        rel_stats = float((se - sg).norm() / se.norm())
        assert rel_stats < 2e-3, f"sync_bn={sync_bn}: BatchNorm running statistics after 3 steps differ between the modes: rel {rel_stats}"
        report += [rel, rel_stats, le[2], lg[2]]
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("DP_GPU_OK", *report)


if __name__ == "__main__":
    main()
