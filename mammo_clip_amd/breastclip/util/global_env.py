"""Process-wide distributed environment [ref: util/global_env.py:8-34].

Same contract as the reference's singleton: ``GlobalEnv.get()`` returns one cached record with the fields
``world_size, world_rank, local_rank, num_gpus, master, summary_writer`` (the loss reads ``world_size`` /
``world_rank`` for the label offset and ``summary_writer.train.add_scalar`` for its logging hook), the record is
taken from ``torch.distributed`` at the FIRST call -- so call it after ``init_process_group`` -- and constructing
``GlobalEnv`` a second time raises.  ``reset()`` is an addition for tests and re-initialisation."""
import os
from dataclasses import dataclass, field
from typing import Any, Optional

import torch
import torch.distributed as dist


class SummaryWriter:
    """holder for the (optional) tensorboard writers and the global step counter the loss hook logs against"""

    def __init__(self):
        self.train: Optional[Any] = None
        self.valid: Optional[Any] = None
        self.global_step: int = 0


@dataclass(frozen=True)
class DistEnv:
    world_size: int
    world_rank: int
    local_rank: int
    num_gpus: int
    master: bool
    summary_writer: SummaryWriter = field(default_factory=SummaryWriter)

    def __iter__(self):                      # tuple-style unpacking, like the reference's namedtuple
        return iter((self.world_size, self.world_rank, self.local_rank, self.num_gpus, self.master, self.summary_writer))


def _probe() -> DistEnv:
    if dist.is_available() and dist.is_initialized():
        rank = dist.get_rank()
        return DistEnv(world_size=dist.get_world_size(), world_rank=rank, local_rank=int(os.environ.get("LOCAL_RANK", 0)),
                       num_gpus=1, master=(rank == 0))
    return DistEnv(world_size=1, world_rank=0, local_rank=0, num_gpus=torch.cuda.device_count(), master=True)


class GlobalEnv:
    _instance: Optional[DistEnv] = None

    def __init__(self):
        if type(self)._instance is not None:
            raise Exception("This class is a singleton")
        type(self)._instance = _probe()

    @classmethod
    def get(cls) -> DistEnv:
        if cls._instance is None:
            cls()
        return cls._instance

    @classmethod
    def reset(cls) -> None:
        cls._instance = None
