"""Poison the caching allocator's free blocks with NaN before forward / backward: any kernel that reads memory it (or a
predecessor) did not write shows up as NaN / changed gradients."""
import sys, types, torch
sys.path.insert(0, "/root/repo")
import mammo_clip_amd
from mammo_clip_amd import engine
from mammo_clip_amd.breastclip import util as U
from mammo_clip_amd.breastclip.loss import build_loss
from mammo_clip_amd.breastclip.model import build_model
from oracle import weights as ow
dev = torch.device("cuda:0")
cfg = {"name": "clip_custom", "temperature": 0.07,
       "image_encoder": {"source": "cnn", "name": "tf_efficientnetv2-detect", "pretrained": False, "model_type": "cnn"},
       "text_encoder": {"source": "huggingface", "name": "emilyalsentzer/Bio_ClinicalBERT", "pretrained": False,
                        "gradient_checkpointing": False, "pooling": "eos", "cache_dir": "", "trust_remote_code": True},
       "projection_head": {"name": "linear", "dropout": 0.1, "proj_dim": 512}}
loss_cfg = {"breast_clip": dict(label_smoothing=0.0, i2i_weight=1.0, t2t_weight=0.5, loss_ratio=1.0)}
H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
batch = ow.synth_batch(2, H, H, 16, seed=5)
bt = {"images": batch["images"].to(dev), "image_views": batch["image_views"].to(dev),
      "text_tokens": {k: v.to(dev) for k, v in batch["text_tokens"].items()},
      "text_tokens2": {k: v.to(dev) for k, v in batch["text_tokens2"].items()}}
def poison(val):
    torch.cuda.synchronize()
    keep = []
    for sz in [2 ** i for i in range(8, 27)] + [3 * 2 ** i for i in range(8, 25)]:
        for _ in range(6 if sz < 2 ** 20 else 2):
            keep.append(torch.full((sz // 4,), val, dtype=torch.float32, device=dev))
    del keep
    torch.cuda.synchronize()
def run(val_f, val_b):
    U.GlobalEnv.reset(); torch.manual_seed(0)
    m = build_model(cfg, loss_cfg, types.SimpleNamespace(vocab_size=28996)).to(dev)
    lossf = build_loss(loss_cfg); m.train()
    if val_f is not None: poison(val_f)
    out = m(bt, dev)
    ld = lossf(**out, is_train=True)
    if val_b is not None: poison(val_b)
    ld["total"].backward()
    return float(ld["total"]), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
l0, g0 = run(None, None)
for tag, vf, vb in (("poison fwd 1e30", 1e30, None), ("poison bwd NaN", None, float("nan")), ("poison bwd 1e30", None, 1e30), ("poison both 7.0", 7.0, 7.0)):
    l1, g1 = run(vf, vb)
    bad = []
    for n in g0:
        d = (g1[n] - g0[n]).abs().max()
        if not torch.isfinite(g1[n]).all() or float(d) > 1e-3 * float(g0[n].abs().max() + 1e-12):
            bad.append((n, float(d), bool(torch.isfinite(g1[n]).all())))
    print(tag, "loss", l0, l1, "params changed:", len(bad), bad[:6])
