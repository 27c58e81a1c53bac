"""Per-step learning-rate schedule of the reference (cosine.py:6-25): linear warm-up over
``warmup_epochs * steps_per_epoch`` steps, then a half cosine down to zero at ``total_epochs * steps_per_epoch``."""
import math

from torch.optim.lr_scheduler import LRScheduler


def warmup_cosine_factor(step, warmup_steps, total_steps):
    """LR multiplier at optimizer step `step` (1-based, as torch's `_step_count`)."""
    if step < warmup_steps:
        return step / warmup_steps
    return 0.5 * (1.0 + math.cos(math.pi * (step - warmup_steps) / (total_steps - warmup_steps)))


class WarmupCosineScheduler(LRScheduler):
    def __init__(self, optimizer, warmup_epochs: int, total_epochs: int, steps_per_epoch: int, last_epoch=-1,
                 verbose=False):
        self.warmup_steps = warmup_epochs * steps_per_epoch
        self.total_steps = total_epochs * steps_per_epoch
        super().__init__(optimizer, last_epoch=last_epoch)

    def get_lr(self):
        f = warmup_cosine_factor(self._step_count, self.warmup_steps, self.total_steps)
        return [f * base for base in self.base_lrs]
