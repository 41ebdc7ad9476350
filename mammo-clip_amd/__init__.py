"""mammo_clip_amd -- MI355X-native (gfx950) implementation of Mammo-CLIP's contrastive pre-training hot path.

The directory is named ``mammo-clip_amd`` (not importable as-is); ``mammo_clip_amd.py`` at the repo root
registers it under the module name ``mammo_clip_amd``.

  lib.py         ctypes binding of libmammoclip_hip.so (C ABI: include/mammoclip_hip.h)
  ops.py         tensor-level wrappers (device memory + streams from torch, compute from HIP kernels)
  breastclip/    host-side mirror of the reference's ``breastclip`` model / loss API
  engine.py      data-parallel training step (RCCL) used by bench.py
"""
from . import lib  # noqa: F401

__all__ = ["lib"]
