"""LinearWarmupCosineAnnealingLR [ref: scheduler/warmup_cosine.py:8-50]: linear warm-up from 0, then cos^2 decay to 0."""
import math
from typing import Union

from torch.optim import Optimizer
from torch.optim.lr_scheduler import LambdaLR


class LinearWarmupCosineAnnealingLR(LambdaLR):
    def __init__(self, optimizer: Optimizer, total_steps: int, warmup_steps: Union[int, float], last_epoch: int = -1, **kw):
        assert warmup_steps < total_steps, "Warmup steps should be less than total steps."
        self.tsteps = total_steps
        self.wsteps = math.ceil(total_steps * warmup_steps) if isinstance(warmup_steps, float) else warmup_steps
        super().__init__(optimizer, self._lr_multiplier, last_epoch)

    def _lr_multiplier(self, step: int) -> float:
        if step < self.wsteps:
            return max(0, step / float(max(1, self.wsteps)))
        frac = (step - self.wsteps) / (self.tsteps - self.wsteps)
        return max(0, math.cos(frac * (math.pi / 2)) ** 2)


def build_scheduler(optimizer, sched_config, total_steps=None, steps_per_epoch=None):
    name = sched_config["name"].lower()
    if name == "cosine":
        cfg = sched_config["config"]
        total = total_steps if total_steps is not None else cfg["total_epochs"] * steps_per_epoch
        warm = cfg.get("warmup_steps", cfg.get("warmup_epochs", 0) * (steps_per_epoch or 0))
        return LinearWarmupCosineAnnealingLR(optimizer, total_steps=total, warmup_steps=warm)
    if name == "constant":
        return LambdaLR(optimizer, lambda s: 1.0)
    raise NotImplementedError(f"Not implemented scheduler : {name}")
