import sys, types, torch, os
sys.path.insert(0, "/root/repo")
import mammo_clip_amd
from mammo_clip_amd import engine
from mammo_clip_amd.breastclip import util as U
from mammo_clip_amd.breastclip.loss import build_loss
from mammo_clip_amd.breastclip.model import build_model
from oracle import weights as ow
out_path, r_ = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
cfg = {"name": "clip_custom", "temperature": 0.07,
       "image_encoder": {"source": "cnn", "name": "tf_efficientnetv2-detect", "pretrained": False, "model_type": "cnn"},
       "text_encoder": {"source": "huggingface", "name": "emilyalsentzer/Bio_ClinicalBERT", "pretrained": False,
                        "gradient_checkpointing": False, "pooling": "eos", "cache_dir": "", "trust_remote_code": True},
       "projection_head": {"name": "linear", "dropout": 0.1, "proj_dim": 512}}
loss_cfg = {"breast_clip": dict(label_smoothing=0.0, i2i_weight=1.0, t2t_weight=0.5, loss_ratio=1.0)}
U.GlobalEnv.reset(); torch.manual_seed(0)
model = build_model(cfg, loss_cfg, types.SimpleNamespace(vocab_size=28996)).to(dev)
batch = ow.synth_batch(4, 64, 64, 16, seed=5)
bt = {"images": batch["images"].to(dev), "image_views": batch["image_views"].to(dev),
      "text_tokens": {k: v.to(dev) for k, v in batch["text_tokens"].items()},
      "text_tokens2": {k: v.to(dev) for k, v in batch["text_tokens2"].items()}}
mbs, b = engine._split_batch(bt, 2)
KS = ("image_embeddings", "text_embeddings", "text_embeddings2", "image_view_embeddings")
variant = sys.argv[3] if len(sys.argv) > 3 else ""
if "V" in variant:
    torch.autograd.graph.increment_version(list(model.parameters()))
if "H" in variant:
    for p_ in model.parameters():
        p_.register_post_accumulate_grad_hook(lambda q: None)
if "B" in variant:      # what _broadcast_flat does to parameters and buffers: flatten, copy back
    with torch.no_grad():
        for t_ in list(model.parameters()) + list(model.buffers()):
            t_.data.copy_(t_.data.clone())
if "F" in variant:      # a no-grad forward first (fills the derived-weight caches before the broadcast)
    with torch.no_grad():
        model.train(); model(mbs[r_], dev)
    torch.autograd.graph.increment_version(list(model.parameters()))
if "C" in variant:      # fresh zero-offset copies of the micro-batch inputs (what a rank's own batch looks like)
    mbs = [{k: (v.clone() if torch.is_tensor(v) else {kk: vv.clone() for kk, vv in v.items()}) for k, v in mb.items()} for mb in mbs]
model.train()
model.image_encoder.rng.calls, model.text_encoder.text_encoder._calls = 2 * r_, r_
o = model(mbs[r_], dev)
g = torch.Generator(device="cpu").manual_seed(77)
gs = [torch.randn(o[k].shape, generator=g).to(dev) * 0.1 for k in KS]
if "P" in variant:      # perturb the embedding gradients by ~1e-7 relative (what a different fp32 summation order does)
    gs = [t * (1.0 + 1e-7 * torch.randn(t.shape, generator=g).to(dev)) for t in gs]
torch.autograd.backward([o[k] for k in KS], gs)
torch.save({"emb": {k: o[k].detach().cpu() for k in KS}, "g": {n: p.grad.cpu() for n, p in model.named_parameters() if p.grad is not None}}, out_path)
print("saved", out_path)
