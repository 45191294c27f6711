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
