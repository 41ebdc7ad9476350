"""Process-wide distributed environment singleton [ref: util/global_env.py:8-34]: world size / rank are cached
at the first ``GlobalEnv.get()`` -- call it after ``init_process_group``.  ``reset()`` is an addition for tests."""
import collections
import os

import torch
import torch.distributed as dist


class SummaryWriter:
    def __init__(self):
        self.train = None
        self.valid = None
        self.global_step = 0


_Env = collections.namedtuple("DistEnv", ["world_size", "world_rank", "local_rank", "num_gpus", "master", "summary_writer"])


class GlobalEnv:
    _instance = None

    @staticmethod
    def get():
        if GlobalEnv._instance is None:
            GlobalEnv()
        return GlobalEnv._instance

    @staticmethod
    def reset():
        GlobalEnv._instance = None

    def __init__(self):
        if GlobalEnv._instance is not None:
            raise Exception("This class is a singleton")
        if dist.is_available() and dist.is_initialized():
            GlobalEnv._instance = _Env(dist.get_world_size(), dist.get_rank(), int(os.environ.get("LOCAL_RANK", 0)), 1,
                                       dist.get_rank() == 0, SummaryWriter())
        else:
            GlobalEnv._instance = _Env(1, 0, 0, torch.cuda.device_count(), True, SummaryWriter())
