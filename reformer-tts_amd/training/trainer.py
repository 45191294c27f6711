"""Training step of the reference's ``LitReformerTTS`` (``/root/reference/reformer_tts/training/
wrappers.py:53-105,234-297``; ``train.py:77-89``) without Lightning: teacher-forced forward + loss,
reversible backward, data-parallel gradient all-reduce, global-norm clip, warm-up, AdamW.

MI355X-first mechanics:
  * all parameters (and their gradients, Adam moments) live in ONE flat fp32 buffer each, the
    modules hold views: zero_grad is one memset, clip + AdamW are two kernel launches
    (``csrc/optim.hip``), and a data-parallel bucket is a slice -- no flatten/unflatten copies;
  * data parallel = one process per GPU over RCCL (``torch.distributed`` backend "nccl"); the
    gradient all-reduce of reversible block k is issued from the block-done hook while block k-1
    recomputes; leftovers (prenets, heads, postnet) go last.  ``DistributedDataParallel`` is not
    used: the reversible backward produces parameter gradients as a side effect of nested
    backward calls, outside DDP's hook model (hard parts).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from .. import _lib, engine
from .._graphs import capturing
from ..model import ReformerTTS, TTSLoss
from ..model.config import ReformerTTSConfig, TTSTrainingConfig, as_kwargs

NO_DECAY = ("bias", "norm.weight")  # wrappers.py:240


def build_model(cfg: ReformerTTSConfig, device=None, seed: int = 42) -> ReformerTTS:
    torch.manual_seed(seed)  # train.py:16 seed_everything(42)
    model = ReformerTTS(**as_kwargs(cfg))
    return model.to(device) if device is not None else model


def synthetic_batch(batch: int, text_len: int, mel_len: int, n_mels: int = 80, dict_size: int = 76, seed: int = 42,
                    device=None) -> Dict[str, torch.Tensor]:
    """LJSpeech-shaped synthetic batch in the collate layout of ``dataset/utils.py:5-42``
    (SURVEY.md section 8d): phonemes uniform in [1, dict_size]; log-mels clamp(N(-5, 2^2), log 1e-5, 2)
    behind a zero start frame; stop one-hot at the last frame; loss mask of ones."""
    g = torch.Generator().manual_seed(seed)
    ph = torch.randint(1, dict_size + 1, (batch, text_len), generator=g)
    mel = (torch.randn(batch, mel_len, n_mels, generator=g) * 2.0 - 5.0).clamp(math.log(1e-5), 2.0)
    spec = torch.cat([torch.zeros(batch, 1, n_mels), mel], dim=1)
    stop = torch.zeros(batch, mel_len)
    stop[:, -1] = 1.0
    out = dict(phonemes=ph, spectrogram=spec, stop_tokens=stop, loss_mask=torch.ones(batch, mel_len, n_mels))
    return {k: v.to(device) for k, v in out.items()} if device is not None else out


def stop_mae(stop_logits: torch.Tensor, stop_tokens: torch.Tensor) -> torch.Tensor:
    """Mean absolute error, in frames, of the predicted end of the utterance (``wrappers.py:74-80``): the first frame whose
    stop logit is positive -- frame 0 when there is none -- against the frame of the one-hot stop token."""
    b, l = stop_tokens.shape
    frames = torch.arange(l, device=stop_logits.device).expand(b, l)
    first = torch.where(stop_logits.reshape(b, l) > 0, frames, torch.full_like(frames, l)).amin(dim=1)
    first = torch.where(first == l, torch.zeros_like(first), first)
    return (first - stop_tokens.argmax(dim=1)).abs().float().mean()


class Trainer:
    def __init__(self, model: ReformerTTS, cfg: TTSTrainingConfig, device, process_group=None):
        self.model, self.cfg, self.device = model, cfg, torch.device(device)
        self.loss = TTSLoss(torch.tensor(cfg.positive_stop_weight, device=self.device), cfg.raw_pred_loss_weight,
                            cfg.post_pred_loss_weight, cfg.stop_loss_weight, cfg.spectrogram_loss)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        self.global_step = 0
        self.epoch = 0                                   # finished epochs (drives the exponential LR schedule)
        sch = cfg.lr_scheduler
        self._base_lr = float(cfg.learning_rate if sch is None else sch.initial_lr)
        self._lr = self._base_lr                         # what the reference's optimizer.param_groups[...]["lr"] holds
        # also validated here, before any training, as the reference does in configure_optimizers (wrappers.py:258-279)
        if sch is not None:
            if sch.start_schedule_epoch < 1:
                raise AssertionError("start_schedule_epoch has to be >= 1")
            end = sch.end_schedule_epoch if sch.end_schedule_epoch is not None else cfg.max_epochs
            if end is None:
                raise ValueError("lr_scheduler: end_schedule_epoch or max_epochs must be set")
            if end <= sch.start_schedule_epoch:
                raise ValueError(f"lr_scheduler: the schedule must end after it starts (start {sch.start_schedule_epoch}, end {end})")
        from .. import edges
        # the group lives on this trainer and is (re)installed before every forward (``_apply_modes``): a later trainer of the
        # same process with the option off must not inherit it
        self._sync_bn = (process_group if process_group is not None else dist.group.WORLD) \
            if (getattr(cfg, "sync_batchnorm", False) and self.world > 1) else None
        edges.SYNC_BN = self._sync_bn
        self.recompute = str(getattr(cfg, "recompute", "stash"))
        if self.recompute not in engine.RECOMPUTE_MODES:
            raise ValueError(f"tts_training.recompute = {self.recompute!r}: expected one of {engine.RECOMPUTE_MODES}")
        self._recompute_for = {}                         # padded batch shape -> the mode that fits the free HBM
        self._decorrelate_replicas()
        self._flatten()
        self._make_buckets()
        self._pending: List = []

    def _decorrelate_replicas(self):
        """Data-parallel replicas hold identical parameters but must draw their own randomness (seed + rank):
        the LSH layers' rotation generators, the default device generator (rotations in graph mode) and the per-step dropout
        seed word (``step_seed``) all depend on the rank.  Rank 0 of any world keeps the single-GPU streams."""
        if self.rank == 0:
            return
        from ..model.lsh_attention import LSHSelfAttention
        for m in self.model.modules():
            if isinstance(m, LSHSelfAttention):
                m.seed = int(m.seed) + 1000003 * self.rank
                m._gen = None
        if self.device.type == "cuda":
            torch.cuda.manual_seed(torch.initial_seed() + 1000003 * self.rank)

    def step_seed(self, step_index: int) -> int:
        """Dropout seed word of optimizer step ``step_index`` on this rank (added to every site's constant in the kernels)."""
        return (step_index * 2246822519 + 3266489917 + self.rank * 2654435769) % (1 << 31)

    # ------------------------------------------------------------------ flat storage
    def _flatten(self):
        named = [(n, p) for n, p in self.model.named_parameters() if p.requires_grad]
        align = 8                                   # elements: 16-byte rows in the bf16 mirror, 32 B in fp32
        starts, off = [], 0
        for _, p in named:
            starts.append(off)
            off += -(-p.numel() // align) * align   # padding stays zero in every buffer
        total = off
        pad = 0
        dev = self.device
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.decay_mask = torch.zeros(total, dtype=torch.uint8, device=dev)
        self.n_params = total                       # padded length of the flat buffers
        self.offsets: Dict[str, tuple] = {}
        for (n, p), off in zip(named, starts):
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.flat_p[off:off + k].view_as(p)
            p.grad = self.flat_g[off:off + k].view_as(p)
            if not any(nd in n for nd in NO_DECAY):
                self.decay_mask[off:off + k] = 1
            self.offsets[n] = (off, off + k)
        if dev.type == "cuda":
            self.ws_partial = torch.zeros(2048, dtype=torch.float32, device=dev)
            self.ws_scale = torch.zeros(2, dtype=torch.float32, device=dev)
            self.hyper = torch.zeros(2, dtype=torch.float32, device=dev)        # {lr, bias-corrected step size} of the step
            # pinned staging of the per-step words {lr, step size | dropout seed}: a RING of slots, each guarded by an event
            # recorded behind its copies -- the host runs several steps ahead of the stream (nothing in replay/train_step
            # synchronises), so a single slot would be overwritten before the copy queued for an earlier step has read it
            self._hyper_ring = [dict(hyper=torch.zeros(2, dtype=torch.float32).pin_memory(),
                                     seed=torch.zeros(1, dtype=torch.int32).pin_memory(), ev=None) for _ in range(8)]
            self._ring_next = 0
            # bf16 mirror of every parameter, refreshed by ONE cast launch per optimizer step; modules see views
            self.flat_pb = torch.zeros(total + pad, dtype=torch.bfloat16, device=dev)
            for n, p in named:
                s, e = self.offsets[n]
                p._bf16_mirror = self.flat_pb[s:e].view_as(p)
            self.refresh_mirror()

    def refresh_mirror(self):
        _lib.call("rtts_cast_f32_bf16", self.flat_p.data_ptr(), self.flat_pb.data_ptr(), self.flat_p.numel(),
                  torch.cuda.current_stream().cuda_stream)

    def _make_buckets(self):
        """One bucket per reversible block (parameters of a block are contiguous in module order),
        keyed by (stack, block index); everything else falls into the final 'rest' all-reduce."""
        self.block_bucket: Dict[tuple, tuple] = {}
        covered = []
        for stack_name, seq in (("enc", self.model.enc.reformer.layers), ("dec", self.model.dec.reformer.layers)):
            prefix = f"{stack_name}.reformer.layers.blocks."
            for i in range(len(seq.blocks)):
                rng = [self.offsets[n] for n in self.offsets if n.startswith(f"{prefix}{i}.")]
                if rng:
                    s, e = min(r[0] for r in rng), max(r[1] for r in rng)
                    self.block_bucket[(stack_name, i)] = (s, e)
                    covered.append((s, e))
            seq.block_done_hook = self._make_hook(stack_name)
        covered.sort()
        self.rest: List[tuple] = []
        cur = 0
        for s, e in covered:
            if s > cur:
                self.rest.append((cur, s))
            cur = max(cur, e)
        if cur < self.n_params:
            self.rest.append((cur, self.n_params))

    def _make_hook(self, stack_name):
        def hook(seq, i):
            if self.world > 1 and not self._bulk_allreduce and not self._accumulating and (stack_name, i) in self.block_bucket:
                s, e = self.block_bucket[(stack_name, i)]
                self._pending.append(dist.all_reduce(self.flat_g[s:e], group=self.pg, async_op=True))
        # the stack loop flushes a block's weight gradients early only while somebody waits for them (engine.py)
        hook.active = lambda: self.world > 1 and not self._bulk_allreduce and not self._accumulating
        return hook

    # ------------------------------------------------------------------ step pieces
    use_fused_edges = True

    @staticmethod
    def _frames(batch):
        """(input frames [0, L-1), target frames [1, L)) of a batch (``wrappers.py:57-63``): views of ``spectrogram``, or the
        two padded buffers of a per-shape graph entry -- the input must be ZERO behind the batch's own length, like the tail
        ``pad_to_multiple`` appends, while the target still holds the last frame, so one padded array cannot serve both."""
        if "spectrogram_input" in batch:
            return batch["spectrogram_input"], batch["spectrogram_target"]
        spec = batch["spectrogram"]
        return spec[:, :-1], spec[:, 1:]

    def _fused_edges_ok(self, batch) -> bool:
        spec = self._frames(batch)[0]
        lm = spec.shape[1]
        lp = -(-lm // self.model.pad_base) * self.model.pad_base      # the decoder runs on the padded length
        ok = self.use_fused_edges and self.device.type == "cuda" and self.model.training and (spec.shape[0] * lp) % 64 == 0
        if not ok and self.use_fused_edges and self.device.type == "cuda" and self.model.training:
            _lib.note_general_path("heads / postnet / loss", f"batch x padded length = {spec.shape[0]} x {lp} is not a multiple of 64")
        if ok:
            # the BatchNorm kernels run over the convolutions' 128-padded width and read / write their per-channel vectors at that
            # width; the [mel | stop] heads share one 128-wide GEMM
            pn = self.model.postnet.layers
            width = pn.conv0.out_channels
            if width % 128 != 0 or self.model.num_mel_coeffs + 1 > 128 or self.model.num_mel_coeffs % 8 != 0:
                _lib.note_general_path("heads / postnet / loss", f"postnet width {width} (needs a multiple of 128) / {self.model.num_mel_coeffs} "
                                       "mel channels (a multiple of 8, at most 127)")
                ok = False
        return ok

    def _cut_at_encoder(self, keys):
        """keys_hook of the split step: the decoder reads a detached copy, the encoder's backward is run later from its grad."""
        self._enc_out = keys
        self._enc_in = keys.detach().requires_grad_(True)
        for attr in ("_rtts_bf16", "_rtts_ready"):          # the stack's own bf16 copy / the "encoder done" event travel with the values
            if hasattr(keys, attr):
                setattr(self._enc_in, attr, getattr(keys, attr))
        return self._enc_in

    def _cut_at_enc_stack(self, x):
        """enc_stack_hook of the split step: cut between the encoder prenet (+ positional encoding) and its stack."""
        self._pre_out = x
        self._pre_in = x.detach().requires_grad_(True)
        return self._pre_in

    # ------------------------------------------------------------------ what the backward recomputes
    HBM_HEADROOM = 0.85         # share of the free HBM a stash may claim (the rest: workspaces, the allocator's slack)

    def stash_estimate(self, batch, mode: str) -> int:
        """Bytes the forward of both stacks holds for the backward in ``mode`` at this batch's padded shape
        (``engine.stash_bytes``; 0 when a stack runs the general path)."""
        pb = self.model.pad_base
        b, lp = batch["phonemes"].shape
        lm = self._frames(batch)[0].shape[1]
        te, td = -(-lp // pb) * pb, -(-lm // pb) * pb
        total = 0
        for seq, rows, keys in ((self.model.enc.reformer.layers, b * te, 0), (self.model.dec.reformer.layers, b * td, b * te)):
            if not getattr(seq, "_program_built", True):          # built lazily by the stack's first forward: do it now
                seq._program, seq._program_built = engine.build_program(seq), True
            prog = getattr(seq, "_program", None)
            if prog:
                total += engine.stash_bytes(prog, rows, self.model.dec.mel_linear.in_features, mode, keys)
        return total

    def resolve_recompute(self, batch, free_bytes: Optional[int] = None) -> str:
        """The configured mode, or -- when its estimated footprint does not fit ``HBM_HEADROOM`` of the free HBM -- the
        highest lower mode that does ("full" always does: it holds nothing).  Decided once per padded batch shape, logged
        once when it differs from the configuration; a captured graph keeps the mode it was captured in."""
        key = self._shape_key(batch) if "spectrogram" in batch else tuple(batch["phonemes"].shape) + tuple(self._frames(batch)[0].shape[:2])
        hit = self._recompute_for.get(key)
        if hit is not None and free_bytes is None:
            return hit
        mode = self.recompute
        if self.device.type == "cuda" or free_bytes is not None:
            if free_bytes is None:
                free, _ = torch.cuda.mem_get_info(self.device)
                # what the caching allocator holds but does not use is available to the stash too
                free += torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)
            else:
                free = int(free_bytes)
            rank = engine.RECOMPUTE_MODES.index(mode)
            while rank > 0 and self.stash_estimate(batch, engine.RECOMPUTE_MODES[rank]) > self.HBM_HEADROOM * free:
                rank -= 1
            if engine.RECOMPUTE_MODES[rank] != mode:
                need = self.stash_estimate(batch, mode)
                mode = engine.RECOMPUTE_MODES[rank]
                _lib.log_once(f"recompute:{key}", f"tts_training.recompute = {self.recompute!r} would hold ~{need / 2**30:.2f} GiB for the "
                              f"backward at batch shape {key}; {free / 2**30:.2f} GiB of HBM are free: running this shape in mode {mode!r}")
        self._recompute_for[key] = mode
        return mode

    def _apply_modes(self, batch):
        """Process-wide switches this trainer owns, installed before each of its forwards (two trainers may alternate)."""
        from .. import edges
        edges.SYNC_BN = self._sync_bn
        engine.set_recompute(self.resolve_recompute(batch))     # cached per shape: a capture (after its warm-up passes) asks nothing of the device

    def forward_loss(self, batch, split: bool = False):
        """``wrappers.py:53-72``: input frames [0, L-1), targets [1, L), mask = loss_mask.mean(-1)."""
        self._apply_modes(batch)
        spec = self._frames(batch)[0]
        from ..model.lsh_attention import LSHSelfAttention
        if getattr(self, "_graph_rotations", False):
            # graph mode: the hash rotations of every LSH layer of this forward are slices of ONE sample (one launch, not one
            # per layer); 256 K values cover 6 + 6 layers at 64 buckets (a layer that does not fit draws its own)
            LSHSelfAttention.rotation_pool = (torch.randn(1 << 18, device=spec.device, dtype=torch.float32), [0])
        if spec.device.type == "cuda":
            from ..edges import ConvK5
            ConvK5.refresh_all(spec.device)          # every convolution's GEMM-layout weight copy, one launch
        try:
            return self._forward_loss(batch, split)
        finally:
            LSHSelfAttention.rotation_pool = None

    def _forward_loss(self, batch, split: bool = False):
        spec_in, spec_tgt = self._frames(batch)
        # the encoder on a stream of its own beside the decoder's first blocks: the overlapped one-process step, and the forward
        # of the data-parallel chain of graphs (there the encoder's backward keeps its own graphs: its gradient ranges are
        # exchanged while the next graph replays)
        side_enc = bool(split) and self.overlap_encoder and self.device.type == "cuda" and self._fused_edges_ok(batch)
        if side_enc:
            from ..model.lsh_attention import LSHSelfAttention
            if LSHSelfAttention.rotation_pool is not None:        # drawn on this stream, read by the encoder's hash kernels on theirs
                LSHSelfAttention.rotation_pool[0].record_stream(self._enc_stream())
        if self._fused_edges_ok(batch):
            from ..edges import PostnetLoss
            if getattr(self, "_postnet_loss", None) is None:
                self._postnet_loss = PostnetLoss(self.model, self.loss)
            overlap = split == "overlap"
            y = self.model.decoder_hidden(batch["phonemes"], spec_in, loss_mask=batch["loss_mask"],
                                          keys_hook=self._cut_at_encoder if split else None,
                                          enc_stack_hook=self._cut_at_enc_stack if (split and not overlap) else None,
                                          enc_stream=self._enc_stream() if side_enc else None)
            losses = self._postnet_loss.apply(y, spec_tgt, batch["stop_tokens"], batch["loss_mask"], batch.get("valid_len"))
            return losses[0], losses[1], losses[2], losses[3]
        raw, post, stop, _ = self.model(batch["phonemes"], spec_in, spectrogram_mask=batch["loss_mask"].mean(dim=-1))
        return self.loss(raw, post, stop.view(stop.shape[0], -1), spec_tgt, batch["stop_tokens"], batch["loss_mask"])

    def zero_grad(self):
        self.flat_g.zero_()

    def finish_allreduce(self):
        """Reduce what no block bucket covers (prenets, heads, postnet, ...) and wait for everything."""
        if self.world > 1 and not self._bulk_allreduce and not self._accumulating:
            for s, e in self.rest:
                self._pending.append(dist.all_reduce(self.flat_g[s:e], group=self.pg, async_op=True))
            for w in self._pending:
                w.wait()
            self._pending.clear()

    def _queue_keys(self):
        """(device index, stream handle) of the streams this trainer launches on -- the deferred-gradient queues it owns."""
        if self.device.type != "cuda":
            return None
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        keys = [(idx, torch.cuda.current_stream(self.device).cuda_stream)]
        if getattr(self, "_enc_stream_obj", None) is not None:
            keys.append((idx, self._enc_stream_obj.cuda_stream))
        return keys

    def _warm_stream(self):
        """ONE side stream per trainer for every warm-up pass in front of a capture (scratch buffers and queues are keyed by
        stream handle: a fresh stream per capture would pin a 64 MB slab each for the life of the process)."""
        if getattr(self, "_warm_stream_obj", None) is None:
            self._warm_stream_obj = torch.cuda.Stream(self.device)
        return self._warm_stream_obj

    def _run_backward(self, loss):
        """``loss.backward()`` + the flush of the deferred gradient work (weight gradients, column sums, convolution
        re-layouts are queued by the executors and launched grouped).  The executors queue from autograd's worker thread; the
        queues are keyed by (device, stream), so this flush on the calling thread drains them.  Nothing may stay queued: a
        leftover entry would be added to the NEXT step's gradients after ``zero_grad``.  A backward that raises drops its
        entries."""
        keys = self._queue_keys()             # this trainer's queues only: another trainer / thread keeps what it has pending
        one = getattr(self, "_one", None)
        if one is None or one.device != loss.device or one.dtype != loss.dtype:
            one = self._one = torch.ones((), dtype=loss.dtype, device=loss.device)     # the root gradient: a constant, not a fill per step
        try:
            loss.backward(one if loss.dim() == 0 else None)
            engine.flush_wgrad(keys=keys)
        except BaseException:
            engine.discard_pending(keys)
            raise
        left = engine.pending_all(keys)
        if left:
            engine.discard_pending(keys)
            raise RuntimeError(f"{left} deferred gradient launches were still queued after the flush")

    def backward(self, loss):
        self._run_backward(loss)
        self.finish_allreduce()

    def lr_now(self) -> float:
        return self.lr_now_for(self.global_step)

    def end_epoch(self) -> float:
        """The reference's per-epoch ``MultiplicativeLR`` step (``wrappers.py:258-279``): while start <= epoch <= end the rate is
        multiplied by exp(-gamma), gamma = (ln initial_lr - ln final_lr) / (end - start).  -> the factor applied."""
        self.epoch += 1
        sch = self.cfg.lr_scheduler
        if sch is None:
            return 1.0
        if sch.start_schedule_epoch < 1:
            raise AssertionError("start_schedule_epoch has to be >= 1")
        end = sch.end_schedule_epoch if sch.end_schedule_epoch is not None else self.cfg.max_epochs
        if end is None:
            raise ValueError("lr_scheduler: end_schedule_epoch or max_epochs must be set")
        factor = 1.0
        if sch.start_schedule_epoch <= self.epoch <= end:
            gamma = (math.log(sch.initial_lr) - math.log(sch.final_lr)) / (end - sch.start_schedule_epoch)
            factor = math.exp(-gamma)
        self._lr *= factor
        return factor

    def take_lr(self, step_index: int) -> float:
        """The rate optimizer step ``step_index`` runs with, taken: during warm-up the hook ASSIGNS it to the optimizer state
        (``wrappers.py:284-294``), so the epoch schedule continues from it.  (``lr_now_for`` only reads.)"""
        lr = self.lr_now_for(step_index)
        if self.cfg.warmup_steps is not None and step_index < self.cfg.warmup_steps:
            self._lr = lr
        return lr

    def set_step_hyper(self, step_index: int):
        """Host -> device copy of this step's {lr, lr*sqrt(1-b2^t)/(1-b1^t)} (t = step_index + 1); outside any graph."""
        lr = self.take_lr(step_index)
        t = step_index + 1
        slot = self._hyper_ring[self._ring_next % len(self._hyper_ring)]
        self._ring_next += 1
        if slot["ev"] is not None:
            slot["ev"].synchronize()                 # the copies that last read this slot have run (8 steps ago: no stall)
        slot["hyper"][0] = lr
        slot["hyper"][1] = lr * math.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t)
        self.hyper.copy_(slot["hyper"], non_blocking=True)
        from .._seeds import seed_base
        slot["seed"][0] = self.step_seed(step_index)                                # fresh dropout masks every step, per rank
        seed_base(self.device).copy_(slot["seed"], non_blocking=True)
        slot["ev"] = torch.cuda.Event()
        slot["ev"].record(torch.cuda.current_stream())

    def optimizer_step(self, update_hyper: bool = True):
        """Clip by global norm (after the all-reduce, on averaged gradients) + HF-AdamW."""
        if self.device.type != "cuda":
            raise _lib.RttsError("optimizer_step runs on the GPU only (no CPU fallback for the HIP path)")
        if update_hyper:
            self.set_step_hyper(self.global_step)
        self.global_step += 1
        n = self.flat_p.numel()
        stream = torch.cuda.current_stream().cuda_stream
        _lib.call("rtts_grad_clip_scale", self.flat_g.data_ptr(), n, 1.0 / self.world, float(self.cfg.gradient_clip_val),
                  self.ws_partial.data_ptr(), self.ws_scale.data_ptr(), stream)
        _lib.call("rtts_adamw_step", self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                  self.flat_v.data_ptr(), self.decay_mask.data_ptr(), n, self.ws_scale.data_ptr(), self.hyper.data_ptr(), 0.9, 0.999,
                  1e-6, float(self.cfg.weight_decay), self.flat_pb.data_ptr(), stream)     # writes the bf16 mirror too
        from ..engine import WEIGHT_EPOCH
        WEIGHT_EPOCH[0] += 1

    def lr_now_for(self, step: int) -> float:
        """Rate of optimizer step ``step`` (pure: a logging call changes nothing): base * min(1, (step+1)/warmup) while step <
        warmup -- the warm-up hook (``wrappers.py:284-294``) assigns that value, overwriting what the epoch schedule left, and
        ``set_step_hyper`` performs the assignment when the step is actually taken -- afterwards whatever the last assignment /
        the epoch schedule (``end_epoch``) made of it."""
        if self.cfg.warmup_steps is not None and step < self.cfg.warmup_steps:
            return min(1.0, float(step + 1) / self.cfg.warmup_steps) * self._base_lr
        return self._lr

    def train_step(self, batch, update_hyper: bool = True):
        """One micro-batch: forward + loss + backward (+ all-reduce) + optimizer step."""
        self.model.train()
        self.zero_grad()
        total, raw_l, post_l, stop_l = self.forward_loss(batch)
        self.backward(total)
        if self._bulk_allreduce:
            self.bulk_allreduce()
        self.optimizer_step(update_hyper)
        return total.detach(), raw_l.detach(), post_l.detach(), stop_l.detach()

    # ------------------------------------------------------------------ encoder beside decoder (captured steps)
    overlap_encoder = True      # Trainer.capture (one process): the encoder on a stream of its own, see train_step_overlapped

    def _enc_stream(self):
        """The encoder's side stream -- never the stream the caller is on.  ``torch.cuda.Stream()`` hands out the streams of a POOL
        (32 per device and priority, round robin), so the n-th stream object of a process can BE the stream a hipGraph capture runs
        on (torch.cuda.graph's capture stream comes from the same pool): the fork would then be a wait of a stream on itself and
        the captured graph degenerate -- seen as a segmentation fault inside hipGraphLaunch once a test session had created enough
        trainers (round 4).  Checked at every use: the current stream of a capture is not the one the warm-up ran on."""
        cur = torch.cuda.current_stream(self.device).cuda_stream
        s = getattr(self, "_enc_stream_obj", None)
        if s is None or s.cuda_stream == cur:
            # (default priority: a HIGH-priority stream for this branch -- it is the critical path of the backward's tail -- doubled
            #  the step, 12.4 against 6.2 ms; profiles/r04_enc_stream_priority_ab.log)
            for _ in range(64):
                s = torch.cuda.Stream(self.device)
                if s.cuda_stream != cur:
                    break
            else:
                raise RuntimeError("no stream other than the current one could be obtained for the encoder branch")
            self._enc_stream_obj = s
        return s

    def train_step_overlapped(self, batch, update_hyper: bool = True):
        """``train_step`` with the ENCODER ON A STREAM OF ITS OWN (meant to be captured: in a hipGraph the two streams become
        parallel branches; launched eagerly the host serialises them anyway).  Forward: encoder prenet + stack beside the
        decoder prenet and the first decoder LSH block (they do not read the encoder's output); the first cross-attention
        waits.  Backward: the decoder stack runs outside autograd (``ReversibleSequence.manual``) and its backward generator
        says when the gradient of the keys is complete -- after the lowest cross-attention block -- from where the encoder's
        backward (stack + prenet, autograd, on the encoder's stream) runs beside decoder layer 0's LSH-attention backward and
        the decoder prenet's.  Same arithmetic, same gradients as ``train_step`` (``test_overlapped_step_matches_the_serial_step``)."""
        if not self._fused_edges_ok(batch):
            return self.train_step(batch, update_hyper)
        self.model.train()
        self.zero_grad()
        total, raw_l, post_l, stop_l = self.forward_backward_overlapped(batch)
        if self._bulk_allreduce:
            self.bulk_allreduce()
        self.optimizer_step(update_hyper)
        return total.detach(), raw_l.detach(), post_l.detach(), stop_l.detach()

    handover_encoder_wgrads = os.environ.get("RTTS_HANDOVER_WGRADS", "1") != "0"     # A/B: the encoder's weight gradients on its own stream

    def forward_backward_overlapped(self, batch, loss_scale=None):
        """Forward + loss + backward of one (micro-)batch in the two-stream schedule of ``train_step_overlapped``; gradients
        accumulate into the flat buffer.  ``loss_scale``: a device scalar multiplied into the loss before the backward (1 / number
        of micro-batches under ``accumulate_grad_batches``).  -> the four losses."""
        main, side = torch.cuda.current_stream(), self._enc_stream()
        dec_seq = self.model.dec.reformer.layers
        dec_seq.manual = {}
        try:
            engine.stamp("step: first kernel of the forward")
            total, raw_l, post_l, stop_l = self.forward_loss(batch, split="overlap")
            engine.stamp("forward + loss done")
            if "call" not in dec_seq.manual:
                raise RuntimeError("the decoder stack did not take the explicit executor: the overlapped step needs it")
            self._run_backward(total if loss_scale is None else total * loss_scale)     # heads + postnet: stops at the decoder stack's output (a leaf)
            ctx, dec_x, _, dec_out = dec_seq.manual["call"]
            gen = engine.stack_backward_steps(ctx, dec_out.grad, notify_dkeys=True)
            dx = None
            idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
            steal = dict(src=(idx, side.cuda_stream), src_stream=side, pending=[]) if self.handover_encoder_wgrads else None
            while True:
                try:
                    with torch.no_grad():
                        _, item = next(gen)
                except StopIteration as fin:
                    dx = fin.value[0]
                    break
                if isinstance(item, tuple) and item[0] == "dkeys":
                    dkeys = item[1]
                    ev = torch.cuda.Event()
                    ev.record(main)
                    side.wait_event(ev)
                    dkeys.record_stream(side)
                    engine.stamp("backward: d(keys) complete (main stream)")
                    engine.STEAL = steal
                    try:
                        with torch.cuda.stream(side):
                            engine.stamp("encoder branch: backward starts")
                            self._enc_out.backward(dkeys)          # encoder stack + prenet, on the encoder's stream
                            engine.flush_wgrad()
                            engine.stamp("encoder branch: backward done")
                    finally:
                        engine.STEAL = None
            dec_x.backward(dx)                                 # decoder prenet + positional encoding
            engine.flush_wgrad()
            engine.stamp("decoder branch: backward done")
            if steal is not None and steal["pending"]:
                # the encoder stack's weight gradients, handed over by its stream: the main stream has nothing left to do, the
                # encoder's chain is still running (engine.STEAL).  (The prenet's convolution weight gradients handed over the same
                # way measured null / slightly worse -- 6.22 vs 6.24 ms, stash 5.10 vs 5.05: profiles/r04_handover_ab.log -- removed.)
                engine.run_handed_over(steal)
                engine.stamp("main stream: the encoder's weight gradients done")
            main.wait_stream(side)
            engine.stamp("backward joined")
            if engine.pending_all():
                raise RuntimeError("deferred gradient launches were still queued at the end of the overlapped step")
        finally:
            dec_seq.manual = None
            self._enc_out = self._enc_in = None
        return total, raw_l, post_l, stop_l

    _accumulating = False

    def train_accumulated(self, batches):
        """``accumulate_grad_batches`` of the reference's trainer (pytorch-lightning 0.7.6, ``training/train.py:77-89``,
        5 in ``config/baseline.yml``): the losses of the N micro-batches are scaled by 1/N, their gradients accumulate in
        the flat buffer, the gradient exchange happens once (with the last micro-batch's backward), then ONE clip + AdamW
        step; ``global_step`` -- which drives warm-up and bias correction -- counts optimizer steps.
        -> mean of the micro-batch total losses (device scalar)."""
        batches = list(batches)
        n = len(batches)
        self.model.train()
        self.zero_grad()
        total = None
        for i, batch in enumerate(batches):
            self._accumulating = i + 1 < n                     # all-reduce only with the last micro-batch
            loss = self.forward_loss(batch)[0]
            self._run_backward(loss * (1.0 / n))
            total = loss.detach() if total is None else total + loss.detach()
        self._accumulating = False
        if self.world > 1 and not self._bulk_allreduce:
            # the per-block hooks of the last backward have reduced the block buckets; what no bucket covers follows
            self.finish_allreduce()
        elif self._bulk_allreduce:
            self.bulk_allreduce()
        self.optimizer_step(True)
        return total / n

    @torch.no_grad()
    def validate(self, batch):
        """``LitReformerTTS.validation_step`` (``wrappers.py:107-140``): teacher-forced forward in eval mode (BatchNorm
        running statistics, no dropout), the four losses; the stacks run through the explicit executor's forward.
        -> (total, raw, post, stop, stop_mae) device scalars.  Parameters and optimizer state are untouched."""
        was_training = self.model.training
        stacks = (self.model.enc.reformer.layers, self.model.dec.reformer.layers)
        self.model.eval()
        for st in stacks:
            st.fused_in_eval = True
        try:
            spec = batch["spectrogram"]
            raw, post, stop, _ = self.model(batch["phonemes"], spec[:, :-1], spectrogram_mask=batch["loss_mask"].mean(dim=-1))
            stop2 = stop.view(stop.shape[0], -1)
            return (*self.loss(raw, post, stop2, spec[:, 1:], batch["stop_tokens"], batch["loss_mask"]),
                    stop_mae(stop2, batch["stop_tokens"]))
        finally:
            for st in stacks:
                st.fused_in_eval = False
            self.model.train(was_training)

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self) -> dict:
        """Everything a resumed run needs: the model's own state_dict (reference names: loads into the reference and
        vice versa), the Adam moments keyed by parameter name, and the step counter (drives warm-up and bias correction)."""
        moments = {n: (self.flat_m[s:e].detach().clone(), self.flat_v[s:e].detach().clone()) for n, (s, e) in self.offsets.items()}
        return {"model": {k: v.detach().clone() for k, v in self.model.state_dict().items()}, "adam": moments,
                "global_step": self.global_step, "epoch": self.epoch, "lr": self._lr}

    def load_state_dict(self, state: dict) -> None:
        self.model.load_state_dict(state["model"])             # copies into the flat fp32 buffer the parameters view
        for n, (m, v) in state["adam"].items():
            s, e = self.offsets[n]
            self.flat_m[s:e].copy_(m.reshape(-1))
            self.flat_v[s:e].copy_(v.reshape(-1))
        self.global_step = int(state["global_step"])
        self.epoch = int(state.get("epoch", 0))
        self._lr = float(state.get("lr", self._base_lr))
        if self.device.type == "cuda":
            self.refresh_mirror()                              # the bf16 mirror the GEMMs read
            engine.WEIGHT_EPOCH[0] += 1                        # cached re-layouts of weights (conv permutations) are stale

    def fit(self, host_batches, log_every: int = 0, graphs: Optional[bool] = None):
        """One epoch over an iterable of HOST batches (the output of ``dataset.custom_sequence_padder``; lengths vary from
        batch to batch): the next batch is copied to HBM on a copy stream while the current step runs;
        ``cfg.accumulate_grad_batches`` micro-batches make one optimizer step; the exponential LR schedule advances at the
        end (``end_epoch``).

        ``graphs`` (default: on, on the GPU): every micro-batch replays a hipGraph.  The reference's loader yields few distinct
        PADDED shapes -- the model pads text and mel to multiples of ``pad_base`` = 256 (``reformer_tts.py:119-125``) -- so one
        forward + loss + backward graph is captured per ``(B, ceil(Lp / pad_base), ceil(Lm / pad_base))`` the first time that
        shape is seen (``_shape_graph``) and batches are copied into its buffers, zero-padded exactly as ``pad_to_multiple``
        would pad them; the batch's own mel length -- the loss's denominators and cut-off (``loss.py:28-53``) -- travels as a
        device word.  Clip + AdamW is one more graph.  A shape outside the fused edges' envelope runs eagerly.
        -> list of the (mean) total loss per optimizer step (device scalars; no per-step host sync)."""
        from ..dataset import BatchPrefetcher
        losses, group = [], []
        acc = max(1, int(self.cfg.accumulate_grad_batches))
        from .. import edges
        use_graphs = (self.device.type == "cuda" and self.use_fused_edges and edges.SYNC_BN is None) if graphs is None else bool(graphs)

        def step(batches):
            if use_graphs:
                return self._train_group_graphed(batches)
            return self.train_step(batches[0])[0] if len(batches) == 1 else self.train_accumulated(batches)

        for batch in BatchPrefetcher(host_batches, self.device):
            group.append(batch)
            if len(group) == acc:
                losses.append(step(group))
                group = []
                if log_every and len(losses) % log_every == 0:
                    print(f"step {self.global_step}: loss {float(losses[-1]):.4f}", flush=True)
        if group:                                              # a trailing partial group still makes a step
            losses.append(step(group))
        self.end_epoch()
        return losses

    # ------------------------------------------------------------------ per-shape graph cache (real, ragged batches)
    MAX_SHAPE_GRAPHS = 32          # LJSpeech at pad_base 256: text 1-2 x mel 1-4 multiples x (full | last) batch size

    def _shape_key(self, batch):
        pb = self.model.pad_base
        b, lp = batch["phonemes"].shape
        lm = batch["spectrogram"].shape[1] - 1
        return (b, -(-lp // pb), -(-lm // pb))

    def _shape_graph(self, batch):
        """The captured forward + loss + backward of this batch's padded shape, or None when the shape has to run eagerly.
        Entry: device buffers in the collate layout padded to multiples of pad_base + the graph that reads them."""
        cache = self.__dict__.setdefault("_shape_graphs", {})
        key = self._shape_key(batch)
        if key in cache:
            return cache[key]
        pb, dev = self.model.pad_base, self.device
        b, kp, km = key
        nm = batch["spectrogram"].shape[2]
        bufs = dict(phonemes=torch.zeros(b, kp * pb, dtype=batch["phonemes"].dtype, device=dev),
                    spectrogram_input=torch.zeros(b, km * pb, nm, dtype=torch.float32, device=dev),
                    spectrogram_target=torch.zeros(b, km * pb, nm, dtype=torch.float32, device=dev),
                    stop_tokens=torch.zeros(b, km * pb, dtype=torch.float32, device=dev),
                    loss_mask=torch.zeros(b, km * pb, nm, dtype=torch.float32, device=dev),
                    valid_len=torch.ones(1, dtype=torch.int32, device=dev))
        self.model.train()
        if len(cache) >= self.MAX_SHAPE_GRAPHS or not self._fused_edges_ok(bufs):
            cache[key] = None
            return None
        from ..model.lsh_attention import LSHSelfAttention
        for m in self.model.modules():
            if isinstance(m, LSHSelfAttention):
                m.use_default_generator = True
        self._graph_rotations = True
        if getattr(self, "_acc_scale", None) is None:
            self._acc_scale = torch.ones((), dtype=torch.float32, device=dev)    # 1 / micro-batches of the group, a device word
            self._valid_ring = [dict(v=torch.zeros(2, dtype=torch.float32).pin_memory(), ev=None) for _ in range(8)]
            self._valid_next = 0
        entry = dict(bufs=bufs, graph=None, out=None)
        self._fill_shape_buffers(entry, batch)
        hooks = [(seq, seq.block_done_hook) for seq in (self.model.enc.reformer.layers, self.model.dec.reformer.layers)]

        def micro():
            # caches keyed by the weight epoch (the convolutions' GEMM-layout weight copies, the decoder prenet's padded weight)
            # are refreshed by launches the HOST decides on: the capture must contain them, a replay follows an optimizer step
            engine.WEIGHT_EPOCH[0] += 1
            if self.overlap_encoder:                     # the encoder as a parallel branch of the graph (train_step_overlapped)
                return self.forward_backward_overlapped(bufs, self._acc_scale)[0].detach()
            total = self.forward_loss(bufs)[0]
            self._run_backward(total * self._acc_scale)
            return total.detach()

        # the warm-up passes below must not leak into the running step: gradients and BatchNorm statistics are put back
        saved_g = self.flat_g.clone()
        saved_buffers = [(t, t.clone()) for t in self.model.buffers()]
        side = self._warm_stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            for seq, _ in hooks:
                seq.block_done_hook = None                  # no collective inside a capture: the exchange follows the last replay
            with torch.cuda.stream(side):
                for _ in range(2):                          # allocator, lazy attributes
                    micro()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            pool = next((e["graph"].pool() for e in cache.values() if e is not None), None)
            with self._capturing(graph, **({} if pool is None else {"pool": pool})):
                entry["out"] = micro()
            entry["graph"] = graph
        finally:
            for seq, h in hooks:
                seq.block_done_hook = h
            self.flat_g.copy_(saved_g)
            for t, c in saved_buffers:
                t.copy_(c)
        cache[key] = entry
        return entry

    def _fill_shape_buffers(self, entry, batch):
        """Device batch (its own lengths) -> the entry's padded buffers: same zeros as ``pad_to_multiple`` appends."""
        bufs = entry["bufs"]
        lp = batch["phonemes"].shape[1]
        lm = batch["spectrogram"].shape[1] - 1
        spec = batch["spectrogram"]
        for name, src, n in (("phonemes", batch["phonemes"], lp), ("spectrogram_input", spec[:, :-1], lm), ("spectrogram_target", spec[:, 1:], lm),
                             ("stop_tokens", batch["stop_tokens"], lm), ("loss_mask", batch["loss_mask"], lm)):
            dst = bufs[name]
            if dst.shape[1] > n:
                dst[:, n:].zero_()
            dst[:, :n].copy_(src, non_blocking=True)
        bufs["valid_len"].fill_(lm)

    def _train_group_graphed(self, batches):
        """``train_accumulated`` with every micro-batch replayed from its shape's graph (eager for shapes without one)."""
        from .._seeds import seed_base
        n = len(batches)
        self.model.train()
        entries = [self._shape_graph(b) for b in batches]      # captures first: a capture must not see a half-built gradient
        self.zero_grad()
        if getattr(self, "_acc_scale", None) is not None:
            self._acc_scale.fill_(1.0 / n)
        if getattr(self, "_graph_opt_only", None) is None and any(e is not None for e in entries):
            self.set_step_hyper(self.global_step)
            side = self._warm_stream()
            side.wait_stream(torch.cuda.current_stream())
            saved = [t.clone() for t in (self.flat_p, self.flat_m, self.flat_v, self.flat_pb)]
            with torch.cuda.stream(side):
                self.optimizer_step(update_hyper=False)        # warm-up of the two launches (their results are discarded)
            torch.cuda.current_stream().wait_stream(side)
            self.global_step -= 1
            for t, c in zip((self.flat_p, self.flat_m, self.flat_v, self.flat_pb), saved):
                t.copy_(c)
            self._graph_opt_only = torch.cuda.CUDAGraph()
            pool = next(e["graph"].pool() for e in entries if e is not None)
            with self._capturing(self._graph_opt_only, pool=pool):
                self.optimizer_step(update_hyper=False)
            self.global_step -= 1
        self.set_step_hyper(self.global_step)                  # lr, Adam step size, dropout seed of this optimizer step
        total = None
        for i, (batch, entry) in enumerate(zip(batches, entries)):
            if entry is None:
                self._accumulating = True                      # the exchange happens once, below
                loss = self.forward_loss(batch)[0]
                self._run_backward(loss * (1.0 / n))
                self._accumulating = False
                loss = loss.detach()
            else:
                self._fill_shape_buffers(entry, batch)
                if i:                                          # the graph's dropout sites are frozen constants + this device word:
                    seed_base(self.device).add_(7919)          # micro-batches of one step must not repeat each other's masks
                entry["graph"].replay()
                loss = entry["out"].clone()
            total = loss if total is None else total + loss
        if self.world > 1:
            dist.all_reduce(self.flat_g, group=self.pg)
        if getattr(self, "_graph_opt_only", None) is not None:
            self.global_step += 1
            self._graph_opt_only.replay()
            engine.WEIGHT_EPOCH[0] += 1
        else:
            self.optimizer_step(update_hyper=False)
        return total / n

    # ------------------------------------------------------------------ hipGraph replay of the whole step
    _bulk_allreduce = False

    _capturing = staticmethod(capturing)      # torch.cuda.graph with the garbage collector held off (see _graphs.py)

    def capture(self, batch, segmented: Optional[bool] = None):
        """Capture the step for THIS batch buffer into hipGraphs; afterwards ``replay()`` runs one full step per call: the
        host only writes {lr, step size, dropout seed} for the step into device memory and launches.

        * one process (or ``segmented=False``): ONE graph = forward + backward (+ per-block all-reduce) + clip + AdamW +
          mirror refresh;
        * data parallel (default when world > 1): a CHAIN of graphs with the gradient exchange between them, no collective
          inside any capture (nothing is asked of RCCL beyond plain all-reduces of slices of the flat gradient buffer):
              A    zero + forward + loss + backward of heads and postnet       -> all-reduce of their part (~7 MB)
              D_l  backward of decoder layer l = L-1 .. 0, one graph each (the stack's backward is driven by hand through
                   engine.stack_backward_steps, on this thread, so that a capture can end at a layer boundary); the bottom
                   layer's graph also holds the decoder prenet's backward     -> all-reduce of that layer (~17 MB each)
              B    backward of the encoder stack, from d(loss)/d(encoder output) -> all-reduce of the stack's part (~35 MB)
              B'   backward of the encoder prenet + positional encoding       -> all-reduce of the prenet's part (~17 MB)
              C    clip + AdamW + mirror refresh
          Every all-reduce is issued the moment its graph has been enqueued and runs while the NEXT graph replays; only the
          last one (17 MB of 108) has nothing to hide behind.  Messages of 7-35 MB: xGMI's ring is per-link bound.

        Rotations and dropout draw from the graph-safe default generator."""
        from ..model.lsh_attention import LSHSelfAttention
        if segmented is None:
            segmented = self.world > 1
        segmented = bool(segmented) and self._fused_edges_ok(batch) if segmented else False
        if self._sync_bn is not None:
            # SyncBatchNorm all-reduces per-channel sums inside the forward and the backward of the two convolution stacks: no
            # collective is captured into a hipGraph, so those pieces stay eager BETWEEN the graphs (_capture_around_sync_bn)
            if not self._fused_edges_ok(batch):
                raise _lib.RttsError("sync_batchnorm: this batch shape is outside the fused edges' envelope: such steps run eagerly "
                                     "(train_step / fit(graphs=False))")
            segmented = True
        for m in self.model.modules():
            if isinstance(m, LSHSelfAttention):
                m.use_default_generator = True
        self._graph_rotations = True
        self._bulk_allreduce = bool(segmented)
        side = self._warm_stream()
        side.wait_stream(torch.cuda.current_stream())
        step = self.train_step_overlapped if (self.overlap_encoder and self.world == 1 and not segmented) else self.train_step
        with torch.cuda.stream(side):
            for k in range(2):                           # warm-up on a side stream (allocator, lazy attributes); the second one in
                (step if k else self.train_step)(batch)  # the form that is captured (its second stream's scratch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        self._graph_opt = None
        self._segments = []
        self._tail_main, self._tail_side = [], []
        self._segment_names = None
        self.set_step_hyper(self.global_step)
        if self._sync_bn is not None:
            return self._capture_around_sync_bn(batch)
        if not segmented:
            with self._capturing(self._graph):
                self._graph_out = step(batch, update_hyper=False)
            self.global_step -= 1                        # capturing does not execute: the captured step has not run yet
            return self._graph_out
        # ---- segmented capture: [(graph, gradient range that is final when it ends)] in replay order
        enc_end, stack_begin, dec_stack_end = self._flat_layout()
        dec_seq = self.model.dec.reformer.layers
        enc_seq = self.model.enc.reformer.layers
        segs = []
        dec_seq.manual = {}
        enc_seq.manual = {}          # the encoder stack too: its backward is cut per block, so that its 35 MB of gradients leave in
        #                              three messages behind the following blocks instead of in one behind the 0.15 ms prenet backward
        try:
            with self._capturing(self._graph):
                self.model.train()
                self.zero_grad()
                total, raw_l, post_l, stop_l = self.forward_loss(batch, split=True)
                self._run_backward(total)            # heads + postnet; stops at the decoder stack's output (a leaf: driven by hand)
                self._graph_out = (total.detach(), raw_l.detach(), post_l.detach(), stop_l.detach())
            segs.append((self._graph, (dec_stack_end, self.n_params)))
            if "call" not in dec_seq.manual:
                raise RuntimeError("the decoder stack did not take the explicit executor: the segmented capture needs it")
            two_lanes = self.two_lane_tail and enc_seq.manual is not None and "call" in enc_seq.manual
            dec_segs, tail_main = self._capture_decoder_layers(dec_seq.manual["call"], enc_end, split=two_lanes)
            segs += dec_segs
        finally:
            dec_seq.manual = None
            enc_call = enc_seq.manual.get("call") if enc_seq.manual is not None else None
            enc_seq.manual = None
        tail_side = []
        if enc_call is not None:
            # Two lanes behind the keys' gradient (two_lane_tail): the encoder's graphs are captured on a stream and in a memory pool of
            # their own -- scratch buffers and deferred-gradient queues are keyed by the stream a launch is made on, graph-private memory
            # by the pool -- because they REPLAY beside the rest of decoder layer 0 (`replay`).  One lane: everything in one pool, in order.
            lane = None
            if two_lanes:
                cap_s = torch.cuda.graph.default_capture_stream
                lane = dict(stream=self._distinct_stream(avoid=[torch.cuda.current_stream(self.device)] + ([cap_s] if cap_s is not None else [])), pool=None)
            block_segs, dx_enc = self._capture_encoder_blocks(enc_call, stack_begin, lane)
            g = torch.cuda.CUDAGraph()
            with self._capturing(g, **self._lane_kw(lane)):
                self._pre_out.backward(dx_enc)
                engine.flush_wgrad()
                assert engine.pending_all() == 0
                if self.overlap_encoder:
                    torch.cuda.current_stream().wait_stream(self._enc_stream())
            block_segs.append((g, (0, stack_begin)))
            if two_lanes:
                tail_side = block_segs
            else:
                segs += block_segs
        else:
            g = torch.cuda.CUDAGraph()
            with self._capturing(g, pool=self._graph.pool()):
                self._enc_out.backward(self._enc_in.grad)
                engine.flush_wgrad()
                assert engine.pending_all() == 0
                if self.overlap_encoder:             # the encoder's nodes (and its deferred launches) ran on the encoder's stream: join
                    torch.cuda.current_stream().wait_stream(self._enc_stream())
            segs.append((g, (stack_begin, enc_end)))
            g = torch.cuda.CUDAGraph()
            with self._capturing(g, pool=self._graph.pool()):
                self._pre_out.backward(self._pre_in.grad)
                engine.flush_wgrad()
                assert engine.pending_all() == 0
                if self.overlap_encoder:
                    torch.cuda.current_stream().wait_stream(self._enc_stream())
            segs.append((g, (0, stack_begin)))
        self._enc_out = self._enc_in = self._pre_out = self._pre_in = None
        if tail_side:
            n_dec = len(segs) - 2                                            # graphs of whole decoder layers in front of the cut
            n_enc = len(tail_side) - 1
            self._segment_names = (["forward + loss + heads/postnet backward"] +
                                   [f"decoder layer {k} backward" for k in range(n_dec, 0, -1)] +
                                   ["decoder layer 0 backward up to the keys' gradient"] +
                                   [f"encoder block {k} backward (lane 2)" for k in range(n_enc - 1, -1, -1)] + ["encoder prenet backward (lane 2)"] +
                                   ["rest of decoder layer 0 + decoder prenet backward (lane 1, beside lane 2)"])
        self._finish_chain(segs, tail_main, tail_side)
        return self._graph_out

    two_lane_tail = os.environ.get("RTTS_TWO_LANE_TAIL", "1") != "0"      # A/B: the encoder's graphs behind decoder layer 0's, one lane

    def _distinct_stream(self, avoid):
        """A pool stream whose handle is none of ``avoid``'s (torch.cuda.Stream() objects share 32 streams per device)."""
        taken = {a.cuda_stream for a in avoid}
        for _ in range(64):
            s_ = torch.cuda.Stream(self.device)
            if s_.cuda_stream not in taken:
                return s_
        raise RuntimeError("no distinct stream could be obtained")

    def _lane_kw(self, lane):
        """Arguments of ``_capturing`` for a graph of the encoder lane (its own stream, its own pool) or of the one-lane chain."""
        if lane is None:
            return dict(pool=self._graph.pool())
        kw = dict(stream=lane["stream"])
        if lane["pool"] is not None:
            kw["pool"] = lane["pool"]
        return kw

    def _flat_layout(self):
        """(end of the encoder's parameters, begin of the encoder stack's, end of the decoder stack's) in the flat buffer; checks
        the order the chain's gradient ranges rely on: encoder prenet | encoder stack | decoder prenet | decoder stack | heads, postnet."""
        enc_names = [n for n in self.offsets if n.startswith("enc.")]
        enc_end = max(self.offsets[n][1] for n in enc_names)
        if min(self.offsets[n][0] for n in self.offsets if not n.startswith("enc.")) < enc_end:
            raise RuntimeError("flat buffer: encoder parameters are expected to come first")
        stack_names = [n for n in enc_names if n.startswith("enc.reformer.")]
        stack_begin = min(self.offsets[n][0] for n in stack_names)
        if any(self.offsets[n][0] >= stack_begin for n in enc_names if n not in stack_names):
            raise RuntimeError("flat buffer: the encoder prenet is expected in front of the encoder stack")
        dec_stack = [n for n in self.offsets if n.startswith("dec.reformer.")]
        dec_stack_end = max(self.offsets[n][1] for n in dec_stack)
        tail = [n for n in self.offsets if not n.startswith("enc.") and self.offsets[n][0] >= dec_stack_end]
        if any(n.startswith(("dec.prenet", "dec.positional_encoding", "dec.reformer")) for n in tail) or not tail:
            raise RuntimeError("flat buffer: heads and postnet are expected behind the decoder stack")
        return enc_end, stack_begin, dec_stack_end

    def _capture_decoder_layers(self, dec_call, enc_end, split: bool = False):
        """One graph per decoder layer, top to bottom, from d(loss)/d(stack output) = ``dec_out.grad`` (a buffer the first graph
        -- or the eager postnet island -- fills); the bottom layer's graph also holds the decoder prenet's backward and leaves
        d(loss)/d(keys) on ``self._enc_in.grad``.  ``split``: the bottom layer is cut where d(loss)/d(keys) is complete (behind its
        cross-attention's backward): the graph up to there carries no gradient range (None), what follows -- the layer's LSH
        sublayer, the decoder prenet -- is returned apart: it replays BESIDE the encoder's graphs.
        -> ([(graph, gradient range | None)], [(graph, range)] of the part behind the cut (empty without ``split``))"""
        ctx, dec_x, _, dec_out = dec_call
        gen = engine.stack_backward_steps(ctx, dec_out.grad, complete_layers=True, notify_dkeys=split)
        segs, tail, finished, behind_cut = [], [], False, False
        while not finished:
            g = torch.cuda.CUDAGraph()
            rng_out, cut = None, False
            with self._capturing(g, pool=self._graph.pool()):
                with torch.no_grad():
                    _, done = next(gen)      # one decoder layer's backward + its weight gradients and column sums
                if isinstance(done, tuple) and len(done) == 2 and done[0] == "dkeys":
                    self._enc_in.grad = done[1]          # complete: the encoder's backward may start here
                    cut = True
                else:
                    rng = [self.block_bucket[("dec", j)] for j in done if ("dec", j) in self.block_bucket]
                    lo, hi = min(r[0] for r in rng), max(r[1] for r in rng)
                    if 0 in done:                # the bottom layer: the generator ends; the decoder prenet's backward joins this graph
                        try:
                            with torch.no_grad():
                                next(gen)
                            raise RuntimeError("stack_backward_steps yielded after block 0")
                        except StopIteration as fin:
                            dx, dkeys = fin.value
                        dec_x.backward(dx)
                        self._enc_in.grad = dkeys
                        engine.flush_wgrad()
                        assert engine.pending_all() == 0
                        lo = enc_end             # decoder prenet + positional encoding sit between the encoder and the stack
                        finished = True
                    rng_out = (lo, hi)
            (tail if behind_cut else segs).append((g, rng_out))
            behind_cut = behind_cut or cut
        return segs, tail

    def _capture_encoder_blocks(self, enc_call, stack_begin, lane=None):
        """The encoder stack driven by hand like the decoder's: one graph per encoder block (LSH + feed-forward: 4 weight
        gradients, 11.6 MB of gradients at the baseline widths), each block's range exchanged while the next one replays.
        -> ([(graph, gradient range)], d(loss)/d(stack input))"""
        ctx_e, enc_x, _, _ = enc_call
        if enc_x is not self._pre_in:
            raise RuntimeError("segmented capture: the encoder stack's input is expected to be the cut behind the prenet")
        gen = engine.stack_backward_steps(ctx_e, self._enc_in.grad, complete_layers=True, flush_at=4)
        segs, finished, dx_enc = [], False, None
        while not finished:
            g = torch.cuda.CUDAGraph()
            with self._capturing(g, **self._lane_kw(lane)):
                with torch.no_grad():
                    try:
                        _, done = next(gen)
                    except StopIteration as fin:      # (only when the stack yields nothing: cannot happen with >= 1 block)
                        raise RuntimeError("encoder stack backward ended without a stop") from fin
                rng = [self.block_bucket[("enc", j)] for j in done if ("enc", j) in self.block_bucket]
                lo, hi = min(r[0] for r in rng), max(r[1] for r in rng)
                if 0 in done:
                    try:
                        with torch.no_grad():
                            next(gen)
                        raise RuntimeError("stack_backward_steps yielded after block 0")
                    except StopIteration as fin:
                        dx_enc = fin.value[0]
                    engine.flush_wgrad()
                    assert engine.pending_all() == 0
                    lo = stack_begin
                    finished = True
            segs.append((g, (lo, hi)))
            if lane is not None and lane["pool"] is None:
                lane["pool"] = g.pool()          # the lane's later graphs share this one's memory pool
        return segs, dx_enc

    def _finish_chain(self, segs, tail_main=(), tail_side=()):
        """Checks that the exchanged ranges tile the flat gradient buffer, stores the chain (``tail_main`` / ``tail_side``: the two
        lanes behind the keys' gradient, replayed side by side), captures clip + AdamW behind it."""
        cover = sorted(r for _, r in list(segs) + list(tail_main) + list(tail_side) if r is not None)
        if cover[0][0] != 0 or cover[-1][1] != self.n_params or any(a[1] != b[0] for a, b in zip(cover, cover[1:])):
            raise RuntimeError(f"segmented capture: the gradient ranges do not tile the flat buffer: {cover}")
        self._segments = segs
        self._tail_main, self._tail_side = list(tail_main), list(tail_side)
        if self._tail_side and getattr(self, "_lane_replay_stream", None) is None:
            self._lane_replay_stream = self._distinct_stream(avoid=[torch.cuda.current_stream(self.device)])
        self._graph_opt = torch.cuda.CUDAGraph()
        with self._capturing(self._graph_opt, pool=self._graph.pool()):
            self.optimizer_step(update_hyper=False)
        self.global_step -= 1

    class _Eager:
        """A piece of the chain that is not a hipGraph (it holds a collective): launched from the host on every ``replay``."""

        def __init__(self, fn):
            self.replay = fn

    def _capture_around_sync_bn(self, batch):
        """The data-parallel chain with ``sync_batchnorm``: the BatchNorm layers of the encoder prenet (3) and of the postnet (5)
        all-reduce their per-channel sums in the forward AND in the backward (edges.ConvBNAct; reference modules.py:29,127 under
        ``torch.nn.SyncBatchNorm``).  No collective is captured: those three pieces run eagerly between the graphs, everything
        else -- both reversible stacks, the decoder prenet, clip + AdamW: 85 % of the step's launches -- replays:

              E1  eager  conv weight re-layouts, encoder prenet forward (3 exchanges of 2 x 512 floats)
              F   graph  hash rotations, encoder stack forward, decoder prenet + decoder stack forward
              E2  eager  heads + postnet + loss, forward and backward (5 + 5 exchanges)      -> all-reduce of their gradients
              D_l graph  decoder layer l backward (bottom one: + decoder prenet)             -> all-reduce of the layer
              B_k graph  encoder block k backward                                            -> all-reduce of the block
              E3  eager  encoder prenet backward (3 exchanges)                               -> all-reduce of its gradients
              C   graph  clip + AdamW + mirror refresh

        What crosses a graph / eager boundary lives in buffers with fixed addresses: the encoder stack's input (E1 copies into it),
        the padded masks and frames, the decoder stack's output and ITS gradient (E2 accumulates into it; zeroed first), the
        gradient of the encoder stack's input (E3 reads it)."""
        from ..model.lsh_attention import LSHSelfAttention
        from ..edges import ConvK5, PostnetLoss
        model, dev = self.model, self.device
        enc_end, stack_begin, dec_stack_end = self._flat_layout()
        if getattr(self, "_postnet_loss", None) is None:
            self._postnet_loss = PostnetLoss(model, self.loss)
        spec_in, spec_tgt = self._frames(batch)
        st = {}

        def pre_forward():
            """E1 (also zeroes the flat gradient: the first launch of a step)."""
            model.train()
            self.zero_grad()
            self._apply_modes(batch)
            ConvK5.refresh_all(dev)
            pad_ph, ph_mask, sp_mask, pad_spec = model._encode_inputs(batch["phonemes"], spec_in, batch["loss_mask"].mean(dim=-1))
            x = model.enc.prenet(pad_ph, pe=model.enc.positional_encoding)
            if "x" not in st:                      # first call (before the capture): the hand-over buffers
                st.update(x=x.detach().clone().requires_grad_(True), ph_mask=ph_mask.clone(), sp_mask=sp_mask.clone(), pad_spec=pad_spec.clone())
            else:
                with torch.no_grad():
                    st["x"].copy_(x)
                    st["ph_mask"].copy_(ph_mask)
                    st["sp_mask"].copy_(sp_mask)
                    st["pad_spec"].copy_(pad_spec)
            st["pre_out"] = x

        def postnet():
            """E2"""
            y = st["dec_out"]
            y.grad.zero_()
            total, raw_l, post_l, stop_l = self._postnet_loss.apply(y, spec_tgt, batch["stop_tokens"], batch["loss_mask"], batch.get("valid_len"))
            self._run_backward(total)
            for dst, src in zip(self._graph_out, (total, raw_l, post_l, stop_l)):
                dst.copy_(src.detach())

        def pre_backward():
            """E3"""
            st.pop("pre_out").backward(st["dx_enc"])
            keys = self._queue_keys()
            engine.flush_wgrad(keys=keys)
            assert engine.pending_all(keys) == 0

        dec_seq, enc_seq = model.dec.reformer.layers, model.enc.reformer.layers
        self._graph_out = tuple(torch.zeros((), dtype=torch.float32, device=dev) for _ in range(4))
        engine.WEIGHT_EPOCH[0] += 1        # caches refreshed by launches the host decides on (the decoder prenet's padded weight) must be IN the capture
        # the eager pieces EXECUTE while the chain is being built (a capture does not): BatchNorm's running statistics are put back
        saved_buffers = [(t, t.clone()) for t in model.buffers()]
        pre_forward()
        segs = [(self._Eager(pre_forward), None)]
        dec_seq.manual, enc_seq.manual = {}, {}
        try:
            with self._capturing(self._graph):
                LSHSelfAttention.rotation_pool = (torch.randn(1 << 18, device=dev, dtype=torch.float32), [0])
                try:
                    keys = model.enc.reformer(st["x"], input_mask=st["ph_mask"])
                    self._pre_in = st["x"]
                    keys = self._cut_at_encoder(keys)
                    kpm = ~st["ph_mask"]
                    kpm._rtts_not = st["ph_mask"]
                    y = model.dec.hidden(st["pad_spec"], keys=keys, key_padding_mask=kpm, input_mask=st["sp_mask"])[0]
                finally:
                    LSHSelfAttention.rotation_pool = None
            segs.append((self._graph, None))
            if "call" not in dec_seq.manual or "call" not in enc_seq.manual or dec_seq.manual["call"][3] is not y:
                raise RuntimeError("sync_batchnorm chain: both stacks have to take the explicit executor")
            y.grad = torch.zeros_like(y)
            st["dec_out"] = y
            self._graph.replay()                   # real values in the hand-over buffers for the eager piece below (and its exchanges)
            postnet()
            segs.append((self._Eager(postnet), (dec_stack_end, self.n_params)))
            segs += self._capture_decoder_layers(dec_seq.manual["call"], enc_end)[0]
            n_dec = len(segs) - 3
            block_segs, st["dx_enc"] = self._capture_encoder_blocks(enc_seq.manual["call"], stack_begin)
            segs += block_segs
            n_enc = len(block_segs)
            for g, _ in segs[3:]:
                g.replay()                         # d(loss)/d(encoder stack input) of THIS batch for the eager piece below
            pre_backward()
            segs.append((self._Eager(pre_backward), (0, stack_begin)))
        finally:
            dec_seq.manual = enc_seq.manual = None
            engine.discard_pending(self._queue_keys())
            with torch.no_grad():
                for t, c in saved_buffers:
                    t.copy_(c)
        self._enc_out = self._enc_in = self._pre_out = self._pre_in = None
        self._segment_names = (["encoder prenet forward (eager: SyncBatchNorm)", "both stacks forward",
                                "heads + postnet + loss, forward and backward (eager: SyncBatchNorm)"] +
                               [f"decoder layer {k} backward" + (" + decoder prenet backward" if k == 0 else "") for k in range(n_dec - 1, -1, -1)] +
                               [f"encoder block {k} backward" for k in range(n_enc - 1, -1, -1)] + ["encoder prenet backward (eager: SyncBatchNorm)"])
        self._finish_chain(segs)
        return self._graph_out

    def segment_plan(self):
        """[(bytes of the gradient range all-reduced after segment k, launches-free description)] of the captured data-parallel
        schedule (bench.py prints it): which collective overlaps which graph."""
        if getattr(self, "_segment_names", None):
            names = self._segment_names
        else:
            n_dec = len(self.model.dec.reformer.layers.blocks) // 6          # f, swap, f, swap, f, swap per decoder layer
            n_enc = len(self._segments) - 2 - n_dec                          # encoder-stack graphs: one per block, or one for the stack
            enc_names = [f"encoder block {k} backward" for k in range(n_enc - 1, -1, -1)] if n_enc > 1 else ["encoder stack backward"]
            names = ["forward + loss + heads/postnet backward"] + [f"decoder layer {k} backward" for k in range(n_dec - 1, -1, -1)] + \
                    enc_names + ["encoder prenet backward"]
            names[n_dec] += " + decoder prenet backward"
        out = []
        side, main = list(getattr(self, "_tail_side", None) or ()), list(getattr(self, "_tail_main", None) or ())
        chain = list(self._segments) + side + main
        first_side, first_main = len(self._segments), len(self._segments) + len(side)
        for k, ((_, rng), nm) in enumerate(zip(chain, names)):
            if rng is None:
                continue
            if side and k >= first_main:
                nxt = "the encoder lane's remaining graphs"
            elif side and k >= first_side:
                nxt = ("the other lane + " + names[k + 1]) if k + 1 < first_main else "the other lane's remainder, else nothing (exposed)"
            else:
                nxt = names[k + 1] if k + 1 < len(names) else "nothing (exposed)"
            out.append(dict(after=nm, allreduce_bytes=4 * (rng[1] - rng[0]), overlaps=nxt))
        return out

    def bulk_allreduce(self):
        if self.world > 1:
            dist.all_reduce(self.flat_g, group=self.pg)

    def replay(self):
        self.set_step_hyper(self.global_step)
        self.global_step += 1
        if self._graph_opt is None:
            self._graph.replay()
            return self._graph_out
        # data parallel: every segment's gradient range is final when its graph ends -- its all-reduce is issued at once and
        # runs while the next segment replays; only the last (the encoder prenet's 17 MB) has nothing to hide behind
        works = []

        def run(lane):
            for g, rng in lane:
                g.replay()               # a hipGraph, or an eager piece that holds SyncBatchNorm's exchanges (_capture_around_sync_bn)
                if self.world > 1 and rng is not None:
                    works.append(dist.all_reduce(self.flat_g[rng[0]:rng[1]], group=self.pg, async_op=True))
        run(self._segments)
        if getattr(self, "_tail_side", None):
            # behind the keys' gradient: the encoder's graphs on a stream of their own BESIDE the rest of decoder layer 0 and the
            # decoder prenet (the one-graph step overlaps the same two pieces: Trainer.forward_backward_overlapped); each lane's
            # all-reduces are issued from its own stream context, so that each waits for its own lane only
            cur, side = torch.cuda.current_stream(self.device), self._lane_replay_stream
            if side.cuda_stream == cur.cuda_stream:
                side = self._lane_replay_stream = self._distinct_stream(avoid=[cur])
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                run(self._tail_side)
            run(self._tail_main)
            cur.wait_stream(side)
        for w in works:
            w.wait()
        self._graph_opt.replay()
        if self._sync_bn is not None:
            engine.WEIGHT_EPOCH[0] += 1  # the eager pieces' cached weight re-layouts follow the optimizer step
        return self._graph_out
