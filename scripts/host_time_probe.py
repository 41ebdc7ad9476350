"""probe: host issue time vs GPU time per training step at cfg3, and when the two view streams actually run"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_amd  # noqa
from mammo_clip_amd import engine
from mammo_clip_amd.breastclip.model import build_model
from mammo_clip_amd.breastclip.loss import build_loss
import bench

dev = torch.device("cuda:0")
torch.manual_seed(10)
model = build_model(bench.model_cfg("tf_efficientnet_b5_ns-detect"), bench.LOSS_CFG, types.SimpleNamespace(vocab_size=28996)).to(dev)
lossf = build_loss(bench.LOSS_CFG)
opt = torch.optim.AdamW(model.parameters(), lr=1e-5, fused=True)
tr = engine.Trainer(model, lossf, opt, None, dev)
batch = bench.synth_batch_gpu(32, 1520, 912, 256, dev, 1)
marks = []
def ev(tag):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((tag, e, time.perf_counter()))
p0, s0 = model._embed_pair, model._embed_second
def pair(*a):
    ev("pair.begin"); r = p0(*a); ev("pair.end"); return r
def second(*a):
    ev("second.begin"); r = s0(*a); ev("second.end"); return r
model._embed_pair, model._embed_second = pair, second
def step(batch):
    model.train(); opt.zero_grad(set_to_none=True)
    ev("fwd.begin"); out = model(batch, dev)
    ev("loss.begin"); ld = lossf(**out, is_train=True)
    ev("bwd.begin"); ld["total"].backward()
    ev("opt.begin"); opt.step(); ev("opt.end")
tr.step = step
for ov in (False, True):
    model.overlap_views = ov
    tr.step(batch); tr.step(batch); torch.cuda.synchronize()
    marks.clear()
    t0 = time.perf_counter()
    for _ in range(3):
        ev("step.begin")
        tr.step(batch)
    ev("end")
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 3 * 1e3
    print(f"overlap={ov}: wall per step {tot:.1f} ms")
    base, hb = marks[0][1], marks[0][2]
    for tag, e, h in marks:
        print(f"   {tag:14s} gpu {base.elapsed_time(e):8.1f} ms   host {(h - hb) * 1e3:8.1f} ms")
