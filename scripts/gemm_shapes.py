"""Per-shape GPU time of the plain NT tile-GEMM launches of one cfg3 step (which problems the 128x128 kernel spends its
time on).  python scripts/gemm_shapes.py"""
import collections, sys, types, torch
sys.path.insert(0, "/root/repo")
import bench
from mammo_clip_amd import lib as L, ops, engine
from mammo_clip_amd.breastclip import util
from mammo_clip_amd.breastclip.loss import build_loss
from mammo_clip_amd.breastclip.model import build_model
from mammo_clip_amd.breastclip.optimizer import build_optimizer
from mammo_clip_amd.breastclip.scheduler import LinearWarmupCosineAnnealingLR

L.load()
dev = torch.device("cuda:0")
enc_name, arch_name, b, H, W, T = bench.WORKLOADS["cfg3"]
util.GlobalEnv.reset()
torch.manual_seed(10)
model = build_model(bench.model_cfg(enc_name), bench.LOSS_CFG, types.SimpleNamespace(vocab_size=28996)).to(dev)
lossf = build_loss(bench.LOSS_CFG)
opt = build_optimizer(model, {"name": "adamw", "config": {"lr": 5e-5, "weight_decay": 1e-4}})
sched = LinearWarmupCosineAnnealingLR(opt, total_steps=10000, warmup_steps=100)
tr = engine.Trainer(model, lossf, opt, sched, dev)
batch = bench.synth_batch_gpu(b, H, W, T, dev, seed=10)
for _ in range(2):
    tr.step(batch, 1)
torch.cuda.synchronize()
rec = []
orig = ops.gemm
def spy(A, B, C_out, M, N, K, *a, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(A, B, C_out, M, N, K, *a, **kw)
    e1.record()
    rec.append((kw.get("kind"), M, N, K, kw.get("batch", 1), kw.get("a_kmajor", 0), kw.get("b_kmajor", 0), kw.get("splits", 1), e0, e1))
    return r
ops.gemm = spy
tr.step(batch, 1)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for kind, M, N, K, bt, ak, bk, sp, e0, e1 in rec:
    key = (kind, M, N, K, bt, ak, bk, sp)
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print("total gemm ms", tot)
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    kind, M, N, K, bt, ak, bk, sp = key
    fl = 2.0 * M * N * K * bt * n
    print(f"{str(kind):14s} M={M:7d} N={N:5d} K={K:6d} b={bt:4d} tn={ak}{bk} sp={sp:3d} n={n:3d} {ms:7.2f} ms {fl / ms / 1e9:7.1f} TF")
