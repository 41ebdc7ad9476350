"""cProfile of the host side of one config-#1 step (launch-bound parity case): where do the ~15 us per launch go?"""
import os, sys, types, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import mammo_clip_amd
from mammo_clip_amd import engine
from mammo_clip_amd.breastclip import util
from mammo_clip_amd.breastclip.loss import build_loss
from mammo_clip_amd.breastclip.model import build_model
from mammo_clip_amd.breastclip.optimizer import build_optimizer
dev = torch.device("cuda:0")
enc, arch, b, H, W, T = "tf_efficientnetv2-detect", "efficientnet-b2", 4, 224, 224, 64
util.GlobalEnv.reset()
model = build_model(bench.model_cfg(enc), bench.LOSS_CFG, types.SimpleNamespace(vocab_size=28996)).to(dev)
opt = build_optimizer(model, {"name": "adamw", "config": {"lr": 5e-5, "weight_decay": 1e-4}})
tr = engine.Trainer(model, build_loss(bench.LOSS_CFG), opt, None, dev)
batch = bench.synth_batch_gpu(b, H, W, T, dev, 1)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.step(batch)
torch.cuda.synchronize()
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(28)
print(st.getvalue()[:6000])
