"""Where do the at::native fill / copy / add launches of a training step come from?  (GPU; developer tool)
torch profiler with Python stacks over one config-#1 step; prints the call sites of aten::fill_ / zero_ / copy_ / add."""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity

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
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
sites = {}
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::zeros", "aten::ones", "aten::clone", "aten::contiguous"):
        st = [f for f in (ev.stack or []) if "mammo_clip_amd" in f or "bench.py" in f or "torch/autograd" in f]
        key = (ev.name, st[0] if st else "<no python frame>")
        sites[key] = sites.get(key, 0) + 1
for (name, where), n in sorted(sites.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{n:5d} {name:18s} {where}")
