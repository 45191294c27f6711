"""CPU restatement of the optimizer step of the reference's training loop (TEST INFRASTRUCTURE).

PARITY UNPINNED: the update rule lives in third-party code absent from /root/reference and from
this image -- ``transformers==2.11.0`` ``optimization.AdamW`` (``requirements.txt:29``; used at
``reformer_tts/training/wrappers.py:15,252-256``) and pytorch-lightning 0.7.6's gradient clipping
(``training/train.py:77-89``).  Restated from their published behaviour:
  clip: torch.nn.utils.clip_grad_norm_(params, max_norm)  -> g *= min(1, max_norm/(|g|+1e-6))
  AdamW(betas=(0.9, 0.999), eps=1e-6, correct_bias=True): decoupled decay applied AFTER the Adam
  update, on the updated parameter, only for groups with weight_decay > 0."""
import math

import torch


def clip_coef(grads, max_norm: float) -> float:
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
    return 1.0 if max_norm <= 0 else min(1.0, max_norm / (total + 1e-6))


def adamw_step(p, g, m, v, step: int, lr: float, wd: float, beta1=0.9, beta2=0.999, eps=1e-6):
    """In-place on p, m, v (fp32 tensors); returns p."""
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    step_size = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p.addcdiv_(m, v.sqrt().add_(eps), value=-step_size)
    if wd > 0.0:
        p.add_(p, alpha=-lr * wd)
    return p


def lr_trajectory(steps_per_epoch: int, epochs: int, learning_rate: float, warmup_steps=None, lr_scheduler=None, max_epochs=None):
    """Learning rate used by every optimizer step of a run: ``reformer_tts/training/wrappers.py:258-279`` (an exponential
    ``MultiplicativeLR`` stepped once per epoch: factor exp(-gamma) while start <= epoch <= end, gamma = (ln initial - ln final)
    / (end - start), base rate ``lr_scheduler.initial_lr``) combined with the warm-up hook ``:284-294`` (which ASSIGNS
    lr = base * min(1, (step + 1) / warmup) while step < warmup, overwriting whatever the scheduler left).
    ``lr_scheduler``: dict(initial_lr, final_lr, start_schedule_epoch, end_schedule_epoch) or None.  -> list of floats."""
    base = learning_rate if lr_scheduler is None else lr_scheduler["initial_lr"]
    lr = base
    out = []
    step = 0
    for epoch in range(epochs):
        for _ in range(steps_per_epoch):
            if warmup_steps is not None and step < warmup_steps:
                lr = min(1.0, float(step + 1) / warmup_steps) * base
            out.append(lr)
            step += 1
        if lr_scheduler is not None:            # scheduler.step() at the end of the epoch: last_epoch becomes epoch + 1
            start = lr_scheduler["start_schedule_epoch"]
            assert start >= 1, "start_schedule_epoch has to be >= 1"
            end = lr_scheduler.get("end_schedule_epoch")
            end = max_epochs if end is None else end
            gamma = (math.log(lr_scheduler["initial_lr"]) - math.log(lr_scheduler["final_lr"])) / (end - start)
            current = epoch + 1
            if start <= current <= end:
                a = math.exp(-gamma * float(current)) * lr_scheduler["initial_lr"]
                b = math.exp(-gamma * float(current - 1)) * lr_scheduler["initial_lr"]
                lr *= a / b
    return out


def stop_mae(stop_out: torch.Tensor, stop_tokens: torch.Tensor) -> torch.Tensor:
    """``wrappers.py:74-80``: first frame whose stop logit is positive (0 if none: the argmax of the all-zero row minus the
    differentiator is index 0) against the frame of the one-hot stop token; mean absolute difference."""
    differentiator = (torch.arange(stop_tokens.shape[1]) / 10000).unsqueeze(0).repeat(stop_tokens.shape[0], 1)
    stop_idx = torch.argmax((stop_out.view(stop_out.shape[0], -1) > 0).float() - differentiator, dim=1)
    stop_err = stop_idx - torch.argmax(stop_tokens, dim=1)
    return stop_err.abs().float().mean()
