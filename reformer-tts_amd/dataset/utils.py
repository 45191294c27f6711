"""Batch format of the training step.

``custom_sequence_padder`` mirrors ``/root/reference/reformer_tts/dataset/utils.py:5-42`` -- same name, same argument
(a collection of ``{"phonemes": LongTensor(len), "spectrogram": Tensor(len, n_mels)}``), same four outputs -- but
fills preallocated (optionally pinned) host tensors in one pass instead of building them from ``pad_sequence``, ``cat``
and a Python loop of ``repeat``s, so that the host->HBM copy of a batch (12 MB at B=12, T=1024) can be asynchronous.
``BatchPrefetcher`` stages the next batch on a copy stream while the current step's graph replays.

Host-side integer/byte work: there is nothing here for a GPU kernel to do."""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, List, Optional

import torch


def custom_sequence_padder(batch: List[Dict[str, torch.Tensor]], pin_memory: bool = False) -> Dict[str, torch.Tensor]:
    """-> phonemes (B, Lp) int64 zero padded; spectrogram (B, 1 + Lm, n_mels) with a zero start frame;
    stop_tokens (B, Lm) one-hot at index len-1; loss_mask (B, Lm, n_mels) ones over the valid frames."""
    b = len(batch)
    if b == 0:
        raise ValueError("custom_sequence_padder: empty batch")
    n_mels = batch[0]["spectrogram"].shape[1]
    lens = [int(e["spectrogram"].shape[0]) for e in batch]
    lp, lm = max(int(e["phonemes"].shape[0]) for e in batch), max(lens)
    if min(lens) < 1:
        raise ValueError("custom_sequence_padder: empty spectrogram")   # the reference's one-hot would index frame -1
    kw = dict(pin_memory=pin_memory and torch.cuda.is_available())
    phonemes = torch.zeros(b, lp, dtype=batch[0]["phonemes"].dtype, **kw)
    spectrogram = torch.zeros(b, lm + 1, n_mels, dtype=batch[0]["spectrogram"].dtype, **kw)
    stop_tokens = torch.zeros(b, lm, **kw)
    loss_mask = torch.zeros(b, lm, n_mels, **kw)
    for i, (e, n) in enumerate(zip(batch, lens)):
        phonemes[i, :e["phonemes"].shape[0]] = e["phonemes"]
        spectrogram[i, 1:1 + n] = e["spectrogram"]
        stop_tokens[i, n - 1] = 1.0
        loss_mask[i, :n] = 1.0
    return {"phonemes": phonemes, "spectrogram": spectrogram, "stop_tokens": stop_tokens, "loss_mask": loss_mask}


class BatchPrefetcher:
    """Iterates device-resident batches: batch k+1 is copied host->HBM on its own HIP stream while step k runs.
    With ``into`` (a dict of preallocated device tensors, e.g. the buffers a captured hipGraph reads) every batch of
    matching shape is copied INTO those buffers instead of fresh ones."""

    def __init__(self, host_batches: Iterable[Dict[str, torch.Tensor]], device, into: Optional[Dict[str, torch.Tensor]] = None):
        self.it: Iterator = iter(host_batches)
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self.into = into
        self._next = None
        self._stage()

    def _stage(self):
        try:
            host = next(self.it)
        except StopIteration:
            self._next = None
            return
        with torch.cuda.stream(self.stream):
            # staged NEXT TO the buffers of a captured graph (they are still being read by the running step);
            # __next__ moves it in, ordered on the compute stream
            self._next = {k: v.to(self.device, non_blocking=True) for k, v in host.items()}

    def __iter__(self):
        return self

    def __next__(self) -> Dict[str, torch.Tensor]:
        if self._next is None:
            raise StopIteration
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.stream)
        batch = self._next
        for v in batch.values():
            v.record_stream(cur)
        if self.into is not None:
            for k, v in batch.items():
                if self.into[k].shape != v.shape:
                    raise ValueError(f"BatchPrefetcher: {k} has shape {tuple(v.shape)}, the captured buffers hold {tuple(self.into[k].shape)}")
                self.into[k].copy_(v, non_blocking=True)
            batch = self.into
        self._stage()
        return batch
