"""Data-parallel gradient exchange on CPU with gloo, world size 2: the per-block buckets issued by
the block-done hooks plus the final 'rest' reduce must cover every gradient element exactly once."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import model_ref
        from reformer_tts_amd.model.config import TTSTrainingConfig, model_config_from_dict
        from reformer_tts_amd.training import Trainer, build_model
        cfg = model_ref.small_cfg()
        cfg["enc_reformer_kwargs"]["depth"] = 2
        cfg["dec_reformer_kwargs"]["depth"] = 2
        cfg["enc_reformer_kwargs"]["attn_kwargs"]["implementation"] = "hip"
        cfg["dec_reformer_kwargs"]["self_attn_kwargs"]["implementation"] = "hip"
        model = build_model(model_config_from_dict(cfg), "cpu")
        tr = Trainer(model, TTSTrainingConfig(), "cpu")
        assert tr.world == world
        # partition check: buckets + rest tile [0, n_params) without overlap
        spans = sorted(list(tr.block_bucket.values()) + tr.rest)
        cur = 0
        for s, e in spans:
            assert s == cur and e > s, (s, e, cur)
            cur = e
        assert cur == tr.n_params
        # every rank holds rank-dependent "gradients"; hooks fire in backward order like the executor does
        tr.flat_g.copy_(torch.arange(tr.flat_g.numel(), dtype=torch.float32) * (rank + 1) * 1e-3)
        for name, seq in (("dec", model.dec.reformer.layers), ("enc", model.enc.reformer.layers)):
            for i in range(len(seq.blocks) - 1, -1, -1):
                seq.block_done_hook(seq, i)
        tr.finish_allreduce()
        expect = torch.arange(tr.flat_g.numel(), dtype=torch.float32) * 1e-3 * sum(r + 1 for r in range(world))
        torch.testing.assert_close(tr.flat_g[:tr.n_params], expect[:tr.n_params], rtol=1e-6, atol=1e-6)
        # identical replicas: same seed => same parameters on every rank
        chk = tr.flat_p.double().sum().reshape(1)
        both = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(both, chk)
        assert all(torch.equal(both[0], b) for b in both)
        # ... but their own randomness (seed + rank): LSH rotation seeds and the per-step dropout seed word
        from reformer_tts_amd.model.lsh_attention import LSHSelfAttention
        mine = torch.tensor([float(tr.step_seed(5))] + [float(m.seed) for m in model.modules() if isinstance(m, LSHSelfAttention)],
                            dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        assert all(bool((every[0] != e).all()) for e in every[1:]), every
        assert tr.rank == rank
        q.put((rank, "ok"))
    except Exception as exc:  # noqa: BLE001
        q.put((rank, repr(exc)))
    finally:
        dist.destroy_process_group()


def test_block_buckets_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
