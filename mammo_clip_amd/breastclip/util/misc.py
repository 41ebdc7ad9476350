"""seed_everything [ref: util/utils.py:11-17] and a rank-0-only scalar writer [ref: util/dist_summery_writer.py:27-31].
tensorboard is not a dependency of the hot path: ``DistSummaryWriter`` wraps any object with ``add_scalar`` (or
records to memory), never forcing a device sync -- values are kept as tensors until someone reads them."""
import random

import numpy as np
import torch

from .global_env import GlobalEnv


def seed_everything(seed: int):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    return seed


class DistSummaryWriter:
    def __init__(self, sink=None, keep_last: int = 64):
        self.sink = sink
        self.records = []
        self.keep_last = keep_last

    def add_scalar(self, tag, value, global_step=None, **kw):
        if not GlobalEnv.get().master:
            return
        if self.sink is not None:
            self.sink.add_scalar(tag, value, global_step, **kw)
            return
        self.records.append((tag, value.detach() if torch.is_tensor(value) else value, global_step))
        if len(self.records) > self.keep_last:
            del self.records[: len(self.records) - self.keep_last]
