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
    """[ref: scheduler/__init__.py:8-16] plus the epoch -> step resolution the reference's trainer performs on the config
    right before the call [ref: trainer_ddp.py:146-153], so both a resolved config ({"total_steps", "warmup_steps"})
    and the shipped epoch-based configs ({"total_epochs", "warmup_epochs"} + ``steps_per_epoch``) build the same
    schedule.  An int ``warmup_epochs`` is a number of epochs; a float is passed through unchanged as a FRACTION of
    the total steps (the reference's rule)."""
    name = sched_config["name"].lower()
    cfg = dict(sched_config.get("config", {}) or {})
    if name == "cosine":
        if total_steps is not None:
            total = total_steps
        elif "total_epochs" in cfg and steps_per_epoch is not None:
            total = steps_per_epoch * cfg["total_epochs"]
        elif "total_steps" in cfg:
            total = cfg["total_steps"]
        else:
            raise ValueError("cosine scheduler: 'total_epochs' needs steps_per_epoch (= len(train_dataloader), what the "
                             "reference resolves it with, trainer_ddp.py:146-153), or give 'total_steps'")
        if "warmup_epochs" in cfg and isinstance(cfg["warmup_epochs"], float):
            warm = cfg["warmup_epochs"]
        elif "warmup_epochs" in cfg and steps_per_epoch is not None:
            warm = steps_per_epoch * cfg["warmup_epochs"]
        elif "warmup_epochs" in cfg and "warmup_steps" not in cfg:
            raise ValueError("cosine scheduler: an integer 'warmup_epochs' needs steps_per_epoch (= len(train_dataloader)); "
                             "silently training without warm-up is not what the config asks for")
        else:
            warm = cfg.get("warmup_steps", 0)
        return LinearWarmupCosineAnnealingLR(optimizer, total_steps=total, warmup_steps=warm)
    if name == "constant":
        from torch.optim.lr_scheduler import ConstantLR
        return ConstantLR(optimizer, **cfg)
    raise NotImplementedError(f"got not implemented scheduler : {name}")
