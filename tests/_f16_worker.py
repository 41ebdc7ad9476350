"""Worker of tests/test_f16_storage_gpu.py: runs in a process of its own with MC_STORAGE=f16 (the storage type is fixed when
the kernel library is loaded) and prints ONE JSON line.  usage: python tests/_f16_worker.py bn8k | scaler"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import test_fullsize_gpu as T                       # noqa: E402  (model / batch helpers, fixtures)
from mammo_clip_amd import engine, lib as L, ops    # noqa: E402
from mammo_clip_amd.breastclip import util          # noqa: E402
from mammo_clip_amd.breastclip.optimizer import build_optimizer   # noqa: E402
from oracle import weights as ow                    # noqa: E402

assert L.STORAGE == "f16" and ops.BF16 == torch.float16 and L.load().mc_storage_is_f16() == 1


def bn8k():
    """reference fixture e2e_b2_bn8k (B2 + BERT-base, b = 8, 456^2, T = 64, stochastic ops off): eval and TRAIN loss,
    embeddings, and -- through a loss-scaled backward -- the stored parameter gradients"""
    z = np.load(os.path.join(T.GOLDEN, "e2e_b2_bn8k.npz"))
    b, H, W, Tn = [int(v) for v in z["meta"]]
    model, lossf, sd, arch = T._build("tf_efficientnetv2-detect", "efficientnet-b2")
    bt = T._to_dev(ow.synth_batch(b, H, W, Tn, seed=10))
    util.GlobalEnv.reset()
    rep = {}
    model.eval()
    with torch.no_grad():
        out = model(bt, T.DEV)
        rep["eval_dloss"] = float(lossf(**out, is_train=False)["total"]) - float(z["eval/total"])
    rep["eval_min_cos"] = min(T._cos_rows(out[k], z["eval/" + k]) for k in T.EMB)
    model.train()
    out = model(bt, T.DEV)
    loss = lossf(**out, is_train=True)["total"]
    rep["train_dloss"] = float(loss) - float(z["train/total"])
    rep["train_min_cos"] = min(T._cos_rows(out[k], z["train/" + k]) for k in T.EMB)
    scale = 4096.0
    (loss * scale).backward()
    named = dict(model.named_parameters())
    cs, nr = {}, {}
    for key in z.files:
        if not key.startswith("train/grad/"):
            continue
        n = key[len("train/grad/"):]
        g = named[n].grad.float() / scale
        ref = torch.as_tensor(z[key]).to(g.device)
        cs[n] = T._cos_flat(g, ref)
        nr[n] = float(g.norm() / (ref.norm() + 1e-30))
    rep["grad_min_cos"], rep["grad_min_cos_at"] = min(cs.values()), min(cs, key=cs.get)
    rep["grad_norm_ratio_min"], rep["grad_norm_ratio_max"] = min(nr.values()), max(nr.values())
    rep["n_grads"] = len(cs)
    rep["nonfinite_grads"] = sum(int(not torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
    return rep


def scaler():
    """Trainer under the dynamic loss scale: a clean step updates the parameters and leaves UNSCALED finite gradients; a step
    whose scale overflows f16 is skipped (parameters and optimizer state untouched) and halves the scale"""
    model, lossf, sd, arch = T._build("tf_efficientnetv2-detect", "efficientnet-b2")
    bt = T._to_dev(ow.synth_batch(4, 224, 224, 64, seed=10))
    util.GlobalEnv.reset()
    opt = build_optimizer(model, {"name": "adamw", "config": {"lr": 1e-4, "weight_decay": 1e-4}})
    tr = engine.Trainer(model, lossf, opt, None, T.DEV)
    rep = {"auto_scaler": type(tr.scaler).__name__, "init_scale": tr.scaler.scale}
    tr.scaler = engine.LossScaler(init_scale=1024.0)
    w0 = {n: p.detach().clone() for n, p in model.named_parameters()}
    out = tr.step(bt)
    rep["clean_loss"] = float(out["total"])
    rep["clean_skipped"] = tr.scaler.skipped
    rep["clean_changed"] = sum(int(not torch.equal(w0[n], p)) for n, p in model.named_parameters() if p.grad is not None)
    rep["clean_grads_finite"] = all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
    # unscaled: the same step without a scaler on a fresh model gives the same gradient (to f16 underflow of the unscaled run)
    g_scaled = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    model2, lossf2, _, _ = T._build("tf_efficientnetv2-detect", "efficientnet-b2")
    util.GlobalEnv.reset()
    tr2 = engine.Trainer(model2, lossf2, torch.optim.SGD(model2.parameters(), lr=0.0), None, T.DEV, loss_scale=None)
    tr2.step(bt)
    # (one cosine over ALL gradients: parameters whose exact gradient is zero -- a bias in front of a BatchNorm -- hold
    # round-off only, their own cosine means nothing)
    names = [n for n, p in model2.named_parameters() if p.grad is not None and n in g_scaled]
    a = torch.cat([g_scaled[n].reshape(-1).double() for n in names])
    b_ = torch.cat([dict(model2.named_parameters())[n].grad.reshape(-1).double() for n in names])
    rep["scaled_vs_unscaled_cos"] = float((a @ b_) / (a.norm() * b_.norm()))
    # overflow: 2^40 x gradients of O(1e-2) do not fit f16
    tr.scaler = engine.LossScaler(init_scale=2.0 ** 40)
    w1 = {n: p.detach().clone() for n, p in model.named_parameters()}
    steps1 = [int(v["step"]) for v in opt.state_dict()["state"].values()]      # APPLIED steps (host count - device skip count)
    out = tr.step(bt)
    rep["overflow_reported_skipped"] = int(out["skipped_steps"])                 # (the step's own report, a device scalar)
    rep["overflow_reported_scale"] = float(out["loss_scale"])
    rep["overflow_skipped"] = tr.scaler.skipped
    rep["overflow_scale_after"] = tr.scaler.scale
    rep["overflow_params_unchanged"] = all(torch.equal(w1[n], p) for n, p in model.named_parameters())
    rep["overflow_steps_unchanged"] = [int(v["step"]) for v in opt.state_dict()["state"].values()] == steps1
    # ... and recovers: the next steps at the halved scales skip until the scale fits f16 again, then apply (dynamic policy,
    # no host decision anywhere: the AdamW kernel reads the flag)
    tr.scaler = engine.LossScaler(init_scale=2.0 ** 24)          # overflows f16 (stored gradients x 1.7e7); 2^16 does not (r04 trace)
    seq = []
    for _ in range(14):
        o = tr.step(bt)
        seq.append((float(o["loss_scale"]), int(o["skipped_steps"])))
    rep["recover_seq"] = seq
    rep["recover_applied"] = [int(v["step"]) for v in opt.state_dict()["state"].values()][0] - steps1[0]
    rep["recover_params_finite"] = all(bool(torch.isfinite(p).all()) for p in model.parameters())
    # a STATIC scale (Trainer(loss_scale=number) / MC_LOSS_SCALE): the flag still skips a bad step, the scale never moves
    tr3 = engine.Trainer(model, lossf, opt, None, T.DEV, loss_scale=512.0)
    o3 = tr3.step(bt)
    rep["static_dynamic_flag"], rep["static_scale_after"], rep["static_skipped"] = tr3.scaler.dynamic, float(o3["loss_scale"]), int(o3["skipped_steps"])
    sd_ = tr.scaler.state_dict()
    sc2 = engine.LossScaler()
    sc2.load_state_dict(sd_)
    rep["scaler_state_roundtrip"] = sc2.state_dict() == sd_
    return rep


def shapes():
    """BASELINE per-sample shapes against the fp32 oracle run on the same weights and batch: cfg3 / cfg4 (B5, 1520 x 912,
    T = 256, 2 pairs) and cfg2 (B2, 912 x 912, T = 256, 4 pairs), eval and train mode (stochastic ops off)"""
    rep = {}
    for tag, enc, arch_name, b, H, W, Tn in (("cfg3", "tf_efficientnet_b5_ns-detect", "efficientnet-b5", 2, 1520, 912, 256),
                                             ("cfg2", "tf_efficientnetv2-detect", "efficientnet-b2", 4, 912, 912, 256)):
        model, lossf, sd, arch = T._build(enc, arch_name)
        bt = T._to_dev(ow.synth_batch(b, H, W, Tn, seed=10))
        util.GlobalEnv.reset()
        for mode in ("eval", "train"):
            train = mode == "train"
            model.train(train)
            model.load_state_dict(sd, strict=True)
            with torch.set_grad_enabled(False):
                out = model(bt, T.DEV)
                lh = float(lossf(**out, is_train=train)["total"])
            lo, eo, _ = T._oracle(sd, bt, arch, b, train, grad_keys=("logit_scale",))
            rep[f"{tag}/{mode}_dloss"] = lh - lo
            rep[f"{tag}/{mode}_min_cos"] = min(T._cos_rows(out[k], eo[k]) for k in T.EMB)
        if tag == "cfg3":
            # BACKWARD at the production shape under the DYNAMIC loss scale at its default 65536 (this is where
            # unscaled f16 gradients flush to zero): gradients of the fullsize test's parameter sample against the fp32 oracle
            keys = T.GRAD_KEYS_B5
            model.train()
            model.load_state_dict(sd, strict=True)
            model.zero_grad(set_to_none=True)
            scaler = engine.LossScaler()
            out = model(bt, T.DEV)
            loss = lossf(**out, is_train=True)["total"]
            (loss * scaler.scale_tensor(T.DEV)).backward()
            del out, loss
            params = [p for p in model.parameters() if p.grad is not None]
            finite = scaler.unscale_(params, sync=True)
            scaler.update()
            pd = dict(model.named_parameters())
            gh = {k: pd[k].grad.detach().clone() for k in keys}
            model.zero_grad(set_to_none=True)
            torch.cuda.empty_cache()
            _, _, go = T._oracle(sd, bt, arch, b, True, keys)
            cs = {k: T._cos_flat(gh[k], go[k]) for k in keys if k != "logit_scale"}
            nr = {k: float(gh[k].float().norm() / (go[k].norm() + 1e-30)) for k in keys}
            rep["cfg3/bwd_finite"], rep["cfg3/bwd_scale_after"], rep["cfg3/bwd_skipped"] = bool(finite), scaler.scale, scaler.skipped
            rep["cfg3/grad_min_cos"], rep["cfg3/grad_min_cos_at"] = min(cs.values()), min(cs, key=cs.get)
            rep["cfg3/grad_cos"] = {k.replace("image_encoder.", "img.").replace("text_encoder.text_encoder.encoder.", "txt."): round(v, 5) for k, v in cs.items()}
            rep["cfg3/grad_norm_ratio"] = {k.replace("image_encoder.", "img.").replace("text_encoder.text_encoder.encoder.", "txt."): round(v, 4) for k, v in nr.items()}
            del go, gh
        del model
        torch.cuda.empty_cache()
    return rep


if __name__ == "__main__":
    print("F16-WORKER " + json.dumps({"bn8k": bn8k, "scaler": scaler, "shapes": shapes}[sys.argv[1]]()))
