"""f16 build, cfg3 shape: per-step loss and the first non-finite parameter gradients (GPU; MC_STORAGE=f16)"""
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

DEV = torch.device("cuda:0")
util.GlobalEnv.reset()
model = build_model(bench.model_cfg("tf_efficientnet_b5_ns-detect"), bench.LOSS_CFG, types.SimpleNamespace(vocab_size=28996)).to(DEV)
lossf = build_loss(bench.LOSS_CFG)
nb = int(os.environ.get("NB", "32"))
batch = bench.synth_batch_gpu(nb, 1520, 912, 256, DEV, 1)
model.train()
out = model(batch, DEV)
ld = lossf(**out, is_train=True)
print("forward loss", float(ld["total"]), {k: bool(torch.isfinite(v).all()) for k, v in out.items() if torch.is_tensor(v)})
scale = float(os.environ.get("MC_PROBE_SCALE", "1"))
(ld["total"] * scale).backward()
bad = [(n, float(p.grad.float().abs().max())) for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
print("non-finite grads:", len(bad), bad[:12])
names = [n for n, p in model.named_parameters() if p.grad is not None]
print("last finite / first bad by order:", [n for n, _ in bad][-3:])
g1 = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
if os.environ.get("MC_PROBE_SCALE2"):
    s2 = float(os.environ["MC_PROBE_SCALE2"])
    for p in model.parameters():
        p.grad = None
    # same seeds: reset the counter-based RNG call counters so that the second forward draws the same masks
    from mammo_clip_amd import engine as _e
    irng, trng = _e._rng_counters(model)
    irng.calls, trng._calls = 0, 0
    for m in model.modules():
        if hasattr(m, "track_update"):
            m.track_update = False
    out = model(batch, DEV)
    ld = lossf(**out, is_train=True)
    (ld["total"] * s2).backward()
    cosmin, worst = 2.0, None
    import torch.nn.functional as F
    for n, p in model.named_parameters():
        if p.grad is None or n not in g1:
            continue
        a, b_ = g1[n].reshape(-1).double() / scale, p.grad.float().reshape(-1).double() / s2
        c = float((a @ b_) / (a.norm() * b_.norm() + 1e-300))
        r = float(a.norm() / (b_.norm() + 1e-300))
        if c < cosmin:
            cosmin, worst = c, (n, c, r)
    print(f"grads at scale {scale} vs {s2}: min cosine", worst, "loss", float(ld["total"]))
