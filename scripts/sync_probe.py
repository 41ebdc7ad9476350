"""probe: list host<->device synchronisation points inside one training step (torch sync debug mode)"""
import os, sys, types, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_amd  # noqa
from mammo_clip_amd import engine
from mammo_clip_amd.breastclip.model import build_model
from mammo_clip_amd.breastclip.loss import build_loss
import bench

dev = torch.device("cuda:0")
model = build_model(bench.model_cfg("tf_efficientnet_b5_ns-detect"), bench.LOSS_CFG, types.SimpleNamespace(vocab_size=28996)).to(dev)
lossf = build_loss(bench.LOSS_CFG)
opt = torch.optim.AdamW(model.parameters(), lr=1e-5, fused=True)
tr = engine.Trainer(model, lossf, opt, None, dev)
batch = bench.synth_batch_gpu(4, 512, 512, 64, dev, 1)
tr.step(batch); torch.cuda.synchronize()
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
import traceback
_orig = warnings.showwarning
def show(message, category, filename, lineno, file=None, line=None):
    if "synchron" in str(message).lower():
        print("SYNC:", str(message)[:100])
        for l in traceback.format_stack()[-12:-2]:
            if "mammo" in l or "bench" in l: print("   ", l.strip().splitlines()[0])
warnings.showwarning = show
tr.step(batch)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print("done")
