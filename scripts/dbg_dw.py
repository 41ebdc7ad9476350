"""debug: poison the caching allocator with NaNs, then verify every dwconv_fwd(stats=True) call of a small train step"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_amd  # noqa
from mammo_clip_amd import ops
from mammo_clip_amd.breastclip.model import build_model
from mammo_clip_amd.breastclip.loss import build_loss
from oracle import arch as oarch, bert as obert, weights as ow

dev = torch.device("cuda:0")
junk = torch.randn(1 << 28, device=dev) * 3; del junk
junk = [torch.randn(1 << 20, device=dev) * 3 for _ in range(64)]; del junk
import torch.nn.functional as F

orig = ops.dwconv_fwd
bad = []
def chk(x, w_kkc, n, h, w, c, k, stride, pad_l, pad_t, oh, ow, pro=None, stats=False):
    r = orig(x, w_kkc, n, h, w, c, k, stride, pad_l, pad_t, oh, ow, pro=pro, stats=stats)
    if stats:
        y, part = r
        s = part.double().sum(0)
        yf = y.double()
        ref = torch.stack([yf.sum(0), (yf * yf).sum(0)])
        err = ((s - ref).abs() / (ref.abs() + 1.0)).max().item()
        if not (err < 1e-3):
            bad.append((n, h, w, c, k, stride, err, bool(torch.isnan(part).any())))
    else:
        y = r
    xf = x.float()
    if pro is not None:
        xf = F.silu(xf * pro[0] + pro[1])
    xi = xf.view(n, h, w, c).permute(0, 3, 1, 2)
    pr, pb = (ow - 1) * stride + k - w - pad_l, (oh - 1) * stride + k - h - pad_t
    xi = F.pad(xi, (pad_l, max(pr, 0), pad_t, max(pb, 0)))
    wt = w_kkc.view(k, k, c).permute(2, 0, 1).unsqueeze(1)
    ref = F.conv2d(xi, wt, stride=stride, groups=c)[:, :, :oh, :ow].permute(0, 2, 3, 1).reshape(n * oh * ow, c)
    e = ((y.float() - ref).abs().max() / (ref.abs().max() + 1e-6)).item()
    if not (e < 2e-2):
        bad.append(("y-mismatch", n, h, w, c, k, stride, pro is not None, round(e, 4)))
    if torch.isnan(y.float()).any():
        bad.append(("nan-y", n, h, w, c, k, stride))
    return r
ops.dwconv_fwd = chk
import mammo_clip_amd.breastclip.model.modules.efficientnet_custom as E
if hasattr(E, "ops"): E.ops.dwconv_fwd = chk

cfg = {"name": "clip_custom", "temperature": 0.07,
       "image_encoder": {"source": "cnn", "name": "tf_efficientnet_b5_ns-detect", "pretrained": True, "model_type": "cnn"},
       "text_encoder": {"source": "huggingface", "name": "emilyalsentzer/Bio_ClinicalBERT", "pretrained": False,
                        "gradient_checkpointing": False, "pooling": "eos", "cache_dir": "", "trust_remote_code": True},
       "projection_head": {"name": "linear", "dropout": 0.1, "proj_dim": 512}}
model = build_model(cfg, {"breast_clip": {}}, types.SimpleNamespace(vocab_size=28996)).to(dev)
shapes = ow.clip_shapes(oarch.build_arch("efficientnet-b5"), obert.BertShape())
model.load_state_dict(ow.synth_state_dict(shapes, seed=3), strict=True)
model.train()
enc = model.image_encoder
import mammo_clip_amd.lib as L
trace = []
origcall = L.call
def tcall(name, *args, **kw):
    r = origcall(name, *args, **kw)
    return r
recs = []
def chk2(x, w_kkc, n, h, w, c, k, stride, pad_l, pad_t, oh, ow, pro=None, stats=False):
    r = chk(x, w_kkc, n, h, w, c, k, stride, pad_l, pad_t, oh, ow, pro=pro, stats=stats)
    y = r[0] if stats else r
    recs.append((("dw", n, h, w, c, k, stride, stats), y.float().double().sum().item(), (r[1].double().sum().item() if stats else 0.0),
                 x.float().double().sum().item()))
    return r
ops.dwconv_fwd = chk2
x = torch.randn(2, 3, 96, 64, device=dev)
runs = []
for trial in range(3):
    recs.clear()
    with torch.no_grad():
        out = enc(x)
    torch.cuda.synchronize()
    runs.append((list(recs), out.float().double().sum().item()))
    print("trial", trial, "out", runs[-1][1], "bad", bad[:4])
for i, (a, b) in enumerate(zip(runs[0][0], runs[1][0])):
    if a != b:
        print("first divergence at dwconv call", i, a, b)
        break
else:
    print("all dwconv calls identical across runs")
