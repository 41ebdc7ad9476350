"""probe: peak device memory of the cfg3 training step"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import mammo_clip_amd
from mammo_clip_amd import engine
from mammo_clip_amd.breastclip.model import build_model
from mammo_clip_amd.breastclip.loss import build_loss
from mammo_clip_amd.breastclip.optimizer import build_optimizer
dev = torch.device("cuda:0")
model = build_model(bench.model_cfg("tf_efficientnet_b5_ns-detect"), bench.LOSS_CFG, types.SimpleNamespace(vocab_size=28996)).to(dev)
opt = build_optimizer(model, {"name": "adamw", "config": {"lr": 5e-5, "weight_decay": 1e-4}})
tr = engine.Trainer(model, build_loss(bench.LOSS_CFG), opt, None, dev)
batch = bench.synth_batch_gpu(32, 1520, 912, 256, dev, 1)
for _ in range(2): tr.step(batch)
torch.cuda.synchronize()
print("peak allocated GB", torch.cuda.max_memory_allocated() / 2**30, "reserved GB", torch.cuda.max_memory_reserved() / 2**30)
