"""which Python sites issue the small torch launches (memcpy D2D, fill, elementwise) of one train step (developer tool)"""
import os, sys, types, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import mammo_clip_amd
from mammo_clip_amd import engine
from mammo_clip_amd.breastclip.model import build_model
from mammo_clip_amd.breastclip.loss import build_loss
from mammo_clip_amd.breastclip.optimizer import build_optimizer
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
model = build_model(bench.model_cfg("tf_efficientnet_b5_ns-detect"), bench.LOSS_CFG, types.SimpleNamespace(vocab_size=28996)).to(dev)
lossf = build_loss(bench.LOSS_CFG)
opt = build_optimizer(model, {"name": "adamw", "config": {"lr": 5e-5, "weight_decay": 1e-4}})
tr = engine.Trainer(model, lossf, opt, None, dev)
batch = bench.synth_batch_gpu(8, 1520, 912, 256, dev, 1)
for _ in range(2):
    tr.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.name in (
            "aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::add_", "aten::mul_", "aten::cat", "aten::to", "aten::_to_copy", "aten::zeros", "aten::ones", "aten::index", "aten::flip"):
        # only ops that actually launched something
        if not ev.kernels:
            continue
        st = [s for s in (ev.stack or []) if ("mammo_clip_amd" in s or "bench.py" in s or "/root/repo" in s)]
        sites[(ev.name, (st[0] if st else " | ".join((ev.stack or ["?"])[:3]))[-110:])] += 1
for (name, site), n in sites.most_common(40):
    print(f"{n:5d} {name:18s} {site}")
