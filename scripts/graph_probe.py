"""Feasibility probe: capture one whole training step of the launch-bound config #1 (B2, 4 pairs, 224x224, T=64) in a
HIP graph (torch.cuda.graph) and replay it.  Dropout off / constant hyper-parameters here: the probe only answers whether
the ctypes launches, the autograd backward and the multi-tensor AdamW capture and replay, and what a replay costs."""
import sys, time, types, torch
sys.path.insert(0, "/root/repo")
import bench
from mammo_clip_amd import lib as L, engine
from mammo_clip_amd.breastclip import util
from mammo_clip_amd.breastclip.loss import build_loss
from mammo_clip_amd.breastclip.model import build_model
from mammo_clip_amd.breastclip.optimizer import build_optimizer
L.load()
dev = torch.device("cuda:0")
enc_name, arch_name, b, H, W, T = bench.WORKLOADS["cfg1"]
util.GlobalEnv.reset(); torch.manual_seed(10)
model = build_model(bench.model_cfg(enc_name), bench.LOSS_CFG, types.SimpleNamespace(vocab_size=28996)).to(dev)
enc = model.image_encoder
enc._dropout_p = 0.0
enc._global_params = enc._global_params._replace(drop_connect_rate=0.0)
for lyr in model.text_encoder.text_encoder.encoder.layer:
    lyr.p_attn = lyr.p_hidden = 0.0
model.text_encoder.text_encoder.config.hidden_dropout_prob = 0.0
lossf = build_loss(bench.LOSS_CFG)
opt = build_optimizer(model, {"name": "adamw", "config": {"lr": 5e-5, "weight_decay": 1e-4}})
tr = engine.Trainer(model, lossf, opt, None, dev)
batch = bench.synth_batch_gpu(b, H, W, T, dev, seed=10)
for _ in range(5):
    ld = tr.step(batch, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    ld = tr.step(batch, 1)
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 20 * 1e3, "loss", float(ld["total"]))
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
t0 = time.perf_counter()
with torch.cuda.graph(g):
    ld_static = tr.step(batch, 1)
torch.cuda.synchronize()
print("capture s", time.perf_counter() - t0)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print("graph ms/step", (time.perf_counter() - t0) / 20 * 1e3, "loss", float(ld_static["total"]))
