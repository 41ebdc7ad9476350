"""RCCL smoke of the data-parallel exchange steps on ONE GPU (world_size 1, backend "nccl" = RCCL): the collectives the
multi-GPU run issues -- fused embedding all-gather / reduce-scatter [ref: util/dist_autograd.py:5-27], bucketed
gradient all-reduce(AVG) from post-accumulate hooks [ref: trainer_ddp.py:134 DDP], validation loss all-reduce
[ref: trainer_ddp.py:384-387] -- run through the real backend with GPU tensors, and a full Trainer.step with the
buckets forced on matches the plain single-process step.  (Rank arithmetic is covered by the world_size-2 gloo tests.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, types
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=0, world_size=1, device_id=dev)
import mammo_clip_amd
from mammo_clip_amd import engine
from mammo_clip_amd.breastclip import util as U
from mammo_clip_amd.breastclip.util.dist_autograd import all_gather_fused, DistAutogradAllGatherFunction
from mammo_clip_amd.breastclip.loss import build_loss
from mammo_clip_amd.breastclip.model import build_model
from mammo_clip_amd.breastclip.optimizer import build_optimizer
from oracle import weights as ow

# 1. fused gather / reduce-scatter and the reference-style per-tensor function on RCCL
a = torch.randn(5, 512, device=dev, requires_grad=True); b = torch.randn(5, 512, device=dev, requires_grad=True)
ga, gb = all_gather_fused([a, b])
assert torch.equal(ga, a) and torch.equal(gb, b)
(ga * 2).sum().backward(retain_graph=True); (gb * 3).sum().backward()
assert torch.equal(a.grad, torch.full_like(a, 2.0)) and torch.equal(b.grad, torch.full_like(b, 3.0))
t = torch.randn(4, 8, device=dev, requires_grad=True)
(out,) = DistAutogradAllGatherFunction(partial=False).apply(t)
out.sum().backward()
assert torch.equal(out, t) and torch.equal(t.grad, torch.ones_like(t))

# 2. one training step with the gradient buckets forced on == the plain step (AVG over one rank is the identity)
cfg = {"name": "clip_custom", "temperature": 0.07,
       "image_encoder": {"source": "cnn", "name": "tf_efficientnetv2-detect", "pretrained": False, "model_type": "cnn"},
       "text_encoder": {"source": "huggingface", "name": "emilyalsentzer/Bio_ClinicalBERT", "pretrained": False,
                        "gradient_checkpointing": False, "pooling": "eos", "cache_dir": "", "trust_remote_code": True},
       "projection_head": {"name": "linear", "dropout": 0.1, "proj_dim": 512}}
loss_cfg = {"breast_clip": dict(label_smoothing=0.0, i2i_weight=1.0, t2t_weight=0.5, loss_ratio=1.0)}
batch = ow.synth_batch(2, 64, 64, 16, seed=3)
bt = {"images": batch["images"].to(dev), "image_views": batch["image_views"].to(dev),
      "text_tokens": {k: v.to(dev) for k, v in batch["text_tokens"].items()},
      "text_tokens2": {k: v.to(dev) for k, v in batch["text_tokens2"].items()}}
res = []
for buckets in (False, True):
    torch.manual_seed(0)
    U.GlobalEnv.reset()
    model = build_model(cfg, loss_cfg, types.SimpleNamespace(vocab_size=28996)).to(dev)
    opt = build_optimizer(model, {"name": "adamw", "config": {"lr": 1e-4, "weight_decay": 1e-4}})
    tr = engine.Trainer(model, build_loss(loss_cfg), opt, None, dev, bucket_mb=16)
    if buckets:
        tr.buckets = engine.GradBuckets(list(model.parameters()), 16 << 20)
        assert len(tr.buckets.buckets) > 3
    losses = [float(tr.step(bt)["total"]) for _ in range(2)]
    res.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}))
(l0, s0), (l1, s1) = res
assert l0[0] == l1[0], (l0, l1)
assert abs(l0[1] - l1[1]) < 5e-3, (l0, l1)        # float atomics in the depthwise weight gradient: round-off only
worst = max(float((s0[k].float() - s1[k].float()).abs().max()) for k in s0)
assert worst < 5e-3, worst

# 3. validation: one all-reduce per batch on the device
out = engine.validate(model, build_loss(loss_cfg), {"v": [bt, bt]}, dev)
assert abs(out["v"]["total"] - out["v"]["contrastive"]) < 1e-6 and out["v"]["total"] > 0
dist.destroy_process_group()
print("RCCL-OK")
'''


@pytest.mark.gpu
def test_rccl_world1_exchange_steps(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script), ROOT, "29877"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
