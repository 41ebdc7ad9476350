"""f16 storage build against the reference's bn8k / config-#1 fixtures (GPU; run with MC_STORAGE=f16 or bf16).
Prints eval / train |loss - reference|, embedding cosines and -- with a backward -- gradient cosines."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fullsize_gpu as T          # noqa: E402  (helpers: _build, _to_dev, fixtures)
from mammo_clip_amd import lib as L, ops          # noqa: E402
from mammo_clip_amd.breastclip import util        # noqa: E402
from oracle import weights as ow                   # noqa: E402

print("storage", L.STORAGE, ops.BF16, L.LIB_PATH)
z = np.load(os.path.join(T.GOLDEN, "e2e_b2_bn8k.npz"))
b, H, W, Tn = [int(v) for v in z["meta"]]
model, lossf, sd, arch = T._build("tf_efficientnetv2-detect", "efficientnet-b2")
bt = T._to_dev(ow.synth_batch(b, H, W, Tn, seed=10))
scale = float(os.environ.get("MC_PROBE_SCALE", "1"))
for hilo in (False, True):
    model.text_encoder.set_hilo_weights(hilo)
    util.GlobalEnv.reset()
    model.eval()
    with torch.no_grad():
        out = model(bt, T.DEV)
        le = float(lossf(**out, is_train=False)["total"])
    rep = {"eval dloss": le - float(z["eval/total"])}
    for k in T.EMB:
        rep["eval cos " + k] = T._cos_rows(out[k], z["eval/" + k])
    model.train()
    out = model(bt, T.DEV)
    loss = lossf(**out, is_train=True)["total"]
    rep["train dloss"] = float(loss) - float(z["train/total"])
    for k in T.EMB:
        rep["train cos " + k] = T._cos_rows(out[k], z["train/" + k])
    for p in model.parameters():
        p.grad = None
    (loss * scale).backward()
    gk = [k[len("train/grad/"):] for k in z.files if k.startswith("train/grad/")]
    named = dict(model.named_parameters())
    cs, nr = {}, {}
    for k in gk:
        g = named[k].grad
        if g is None:
            cs[k] = float("nan"); continue
        g = g.float() / scale
        ref = torch.as_tensor(z["train/grad/" + k]).to(g.device)
        cs[k] = T._cos_flat(g, ref)
        nr[k] = float(g.norm() / (ref.norm() + 1e-30))
    rep["grad cos min"] = min(cs.values()); rep["grad cos argmin"] = min(cs, key=cs.get)
    rep["grad norm ratio range"] = (min(nr.values()), max(nr.values()))
    rep["nonfinite grads"] = sum(int(not torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
    print("hilo" if hilo else "plain", {k: (round(v, 6) if isinstance(v, float) else v) for k, v in rep.items()}, flush=True)
