"""Worker of tests/test_dp_gpu.py::test_sync_batchnorm...: one rank of a 2-process gloo job sharing cuda:0.

The reference's BatchNorm layers (reformer_tts/model/modules.py:29,127) normalise over the whole batch of the single process
it trains in.  With ``sync_batchnorm`` the data-parallel trainer must reproduce exactly that: here every rank runs the encoder
prenet's convolution stack (three Conv1d-BatchNorm-ReLU layers on edges.ConvBNAct) on ITS half of a batch with the per-channel
sums all-reduced between the two stages of the statistics / of the backward, and rank 0 compares outputs, input gradients,
parameter gradients (summed over the ranks, as the gradient all-reduce does) and running statistics with a single-process
run over the full batch.  The halves have DIFFERENT lengths of valid rows per rank only through their content; a second case
gives the ranks different batch sizes (3 + 1 samples), so the row count has to travel with the sums."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reformer_tts_amd import edges, engine  # noqa: E402
from reformer_tts_amd.model.modules import EncoderPreNet  # noqa: E402


def stack_run(prenet, x, wgt):
    for p in prenet.parameters():
        p.grad = None
    x = x.clone().requires_grad_()
    z = edges.ConvStackFn.apply(x, edges.encoder_prenet_stack(prenet), True)
    (z.float() * wgt).sum().backward()
    engine.flush_wgrad()
    torch.cuda.synchronize()
    c = prenet.convolutions
    grads = {n: p.grad.clone() for n, p in prenet.named_parameters() if p.grad is not None and "convolutions" in n}
    stats = {f"bn{i}.{k}": getattr(getattr(c, f"bn{i}"), k).clone() for i in (1, 2, 3) for k in ("running_mean", "running_var")}
    return z.detach().float(), x.grad.clone(), grads, stats


def main():
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    worst = 0.0
    for split in ((2, 2), (3, 1)):
        torch.manual_seed(7)
        b, l, c = sum(split), 128, 128
        x = torch.randn(b, l, c, device=dev)
        wgt = torch.randn(b, l, c, device=dev)
        lo = sum(split[:rank])
        mine = slice(lo, lo + split[rank])

        def fresh():
            torch.manual_seed(3)
            pn = EncoderPreNet(77, c, dropout=0.0).to(dev).train()
            with torch.no_grad():
                for i in (1, 2, 3):
                    bn = getattr(pn.convolutions, f"bn{i}")
                    bn.weight.uniform_(0.5, 1.5)
                    bn.bias.uniform_(-0.5, 0.5)
            return pn

        edges.SYNC_BN = None
        z_full, dx_full, g_full, st_full = stack_run(fresh(), x, wgt)                 # the single process of the reference
        edges.SYNC_BN = dist.group.WORLD
        z, dx, g, st = stack_run(fresh(), x[mine], wgt[mine])
        edges.SYNC_BN = None
        for n in g:                                                                    # what the gradient all-reduce does
            dist.all_reduce(g[n])

        def rel(a, r):
            return float((a - r).norm() / r.norm().clamp_min(1e-20))

        errs = {"z": rel(z, z_full[mine]), "dx": rel(dx, dx_full[mine])}
        errs.update({n: rel(g[n], g_full[n]) for n in g if float(g_full[n].norm()) > 1e-6})
        errs.update({n: rel(st[n], st_full[n]) for n in st})
        w = max(errs.values())
        worst = max(worst, w)
        # bf16 activations: a sum taken in another order moves roundings; a statistic over half the batch would be O(0.1)
        assert w < 2e-2, (split, rank, sorted(errs.items(), key=lambda kv: -kv[1])[:4])
        # and the statistics really were global: local statistics give a different result
        edges.SYNC_BN = None
        z_local = stack_run(fresh(), x[mine], wgt[mine])[0]
        assert rel(z_local, z_full[mine]) > 5 * errs["z"], (rel(z_local, z_full[mine]), errs["z"])
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("SYNCBN_OK", worst)


if __name__ == "__main__":
    main()
