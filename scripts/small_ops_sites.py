"""Which Python call sites launch the torch fill / copy kernels of a training step?  (GPU; developer tool)
Wraps torch.zeros / zeros_like / Tensor.zero_ / fill_ / copy_ / clone / contiguous (only when it copies) and counts
by the nearest caller frame inside the package, over one B5 step at a small image size."""
import collections
import os
import sys
import traceback
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mammo_clip_amd  # noqa: F401
from mammo_clip_amd import engine
from mammo_clip_amd.breastclip import util
from mammo_clip_amd.breastclip.loss import build_loss
from mammo_clip_amd.breastclip.model import build_model
from mammo_clip_amd.breastclip.optimizer import build_optimizer
import bench

DEV = torch.device("cuda:0")
util.GlobalEnv.reset()
model = build_model(bench.model_cfg("tf_efficientnet_b5_ns-detect"), bench.LOSS_CFG, types.SimpleNamespace(vocab_size=28996)).to(DEV)
opt = build_optimizer(model, {"name": "adamw", "config": {"lr": 5e-5, "weight_decay": 1e-4}})
tr = engine.Trainer(model, build_loss(bench.LOSS_CFG), opt, None, DEV)
batch = bench.synth_batch_gpu(2, 320, 192, 64, DEV, 1)
for _ in range(2):
    tr.step(batch)
torch.cuda.synchronize()

sites = collections.Counter()
ON = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "mammo_clip_amd" in fr.filename and "small_ops_sites" not in fr.filename:
            return f"{os.path.relpath(fr.filename)}:{fr.lineno} {fr.name}"
    return "<outside>"


def wrap(owner, name, cond=None):
    orig = getattr(owner, name)

    def f(*a, **k):
        if ON[0] and (cond is None or cond(*a, **k)):
            sites[(name, site())] += 1
        return orig(*a, **k)
    setattr(owner, name, f)


def on_gpu(*a, **k):
    t = a[0] if a and isinstance(a[0], torch.Tensor) else None
    return t is None or t.is_cuda


wrap(torch, "zeros"); wrap(torch, "zeros_like"); wrap(torch, "ones"); wrap(torch, "full")
wrap(torch.Tensor, "zero_", on_gpu); wrap(torch.Tensor, "fill_", on_gpu); wrap(torch.Tensor, "copy_", on_gpu)
wrap(torch.Tensor, "clone", on_gpu); wrap(torch.Tensor, "contiguous", lambda t, *a, **k: t.is_cuda and not t.is_contiguous())
wrap(torch.Tensor, "float", lambda t, *a, **k: t.is_cuda and t.dtype != torch.float32)
wrap(torch.Tensor, "to", on_gpu); wrap(torch.Tensor, "new_zeros", on_gpu)
wrap(torch, "cat"); wrap(torch, "stack")
ON[0] = True
tr.step(batch)
torch.cuda.synchronize()
ON[0] = False
for (name, where), n in sorted(sites.items(), key=lambda kv: -kv[1])[:50]:
    print(f"{n:5d} {name:12s} {where}")
