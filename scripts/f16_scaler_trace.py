"""f16 build: loss / loss scale / skipped steps over the first N steps of a workload (GPU; developer tool)
usage: MC_STORAGE=f16 python scripts/f16_scaler_trace.py [pairs] [H] [W] [T] [steps] [arch]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_amd  # noqa
from mammo_clip_amd import engine
from mammo_clip_amd.breastclip import util
from mammo_clip_amd.breastclip.loss import build_loss
from mammo_clip_amd.breastclip.model import build_model
from mammo_clip_amd.breastclip.optimizer import build_optimizer
import bench

a = sys.argv[1:]
nb, H, W, T, steps = (int(a[i]) if len(a) > i else d for i, d in enumerate((32, 1520, 912, 256, 24)))
arch = a[5] if len(a) > 5 else "tf_efficientnet_b5_ns-detect"
DEV = torch.device("cuda:0")
util.GlobalEnv.reset()
model = build_model(bench.model_cfg(arch), bench.LOSS_CFG, types.SimpleNamespace(vocab_size=28996)).to(DEV)
opt = build_optimizer(model, {"name": "adamw", "config": {"lr": 5e-5, "weight_decay": 1e-4}})
tr = engine.Trainer(model, build_loss(bench.LOSS_CFG), opt, None, DEV)
batch = bench.synth_batch_gpu(nb, H, W, T, DEV, 1)
for i in range(steps):
    out = tr.step(batch)
    sc = tr.scaler
    print(f"step {i:3d} loss {float(out['total']):9.5f} scale {sc.scale if sc else 1:g} skipped {sc.skipped if sc else 0}", flush=True)
