"""Which host call sites issue the small aten launches (fills, adds, copies) of one cfg3 step."""
import collections, sys, types, torch
sys.path.insert(0, "/root/repo")
import bench
from mammo_clip_amd import lib as L, engine
from mammo_clip_amd.breastclip import util
from mammo_clip_amd.breastclip.loss import build_loss
from mammo_clip_amd.breastclip.model import build_model
from mammo_clip_amd.breastclip.optimizer import build_optimizer
from mammo_clip_amd.breastclip.scheduler import LinearWarmupCosineAnnealingLR
L.load()
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
enc_name, arch_name, b, H, W, T = bench.WORKLOADS[wl]
util.GlobalEnv.reset(); torch.manual_seed(10)
model = build_model(bench.model_cfg(enc_name), bench.LOSS_CFG, types.SimpleNamespace(vocab_size=28996)).to(dev)
lossf = build_loss(bench.LOSS_CFG)
opt = build_optimizer(model, {"name": "adamw", "config": {"lr": 5e-5, "weight_decay": 1e-4}})
sched = LinearWarmupCosineAnnealingLR(opt, total_steps=10000, warmup_steps=100)
tr = engine.Trainer(model, lossf, opt, sched, dev)
batch = bench.synth_batch_gpu(b, H, W, T, dev, seed=10)
for _ in range(2):
    tr.step(batch, 1)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    tr.step(batch, 1)
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::mul", "aten::_foreach_add_"):
        st = [f for f in (ev.stack or []) if "mammo_clip_amd" in f or "bench.py" in f]
        key = (ev.name, st[0] if st else "<autograd engine / no python frame>")
        agg[key] += 1
for (name, where), n in agg.most_common(45):
    print(f"{n:5d} {name:16s} {where[-110:]}")
