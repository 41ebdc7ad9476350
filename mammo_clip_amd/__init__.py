"""mammo_clip_amd -- MI355X-native (gfx950) implementation of Mammo-CLIP's contrastive pre-training hot path.

(``mammo-clip_amd`` at the repo root is a symlink to this directory: the layout name of the task, not importable.)

  lib.py         ctypes binding of libmammoclip_hip.so (C ABI: include/mammoclip_hip.h)
  ops.py         tensor-level wrappers (device memory + streams from torch, compute from HIP kernels)
  breastclip/    host-side mirror of the reference's ``breastclip`` model / loss API
  engine.py      data-parallel training step (RCCL) used by bench.py
"""
from . import lib  # noqa: F401

__all__ = ["lib"]
