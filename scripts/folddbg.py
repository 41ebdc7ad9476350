import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import tests.test_model_gpu as tm
from mammo_clip_amd.breastclip.model.modules import efficientnet_custom as enc
from mammo_clip_amd import ops
z = np.load(os.path.join(tm.GOLDEN, "e2e_b5_small.npz"))
b, H, W, T = [int(v) for v in z["meta"]]
model, lossf, sd = tm._build("tf_efficientnet_b5_ns-detect", "efficientnet-b5")
batch = tm.ow.synth_batch(b, H, W, T, seed=41)
res = {}
for tag, thr in (("explicit", 1 << 62), ("folded", 0)):
    enc.BN_FOLD_MIN_BYTES = thr
    model.zero_grad(set_to_none=True)
    out, ld = tm._run(model, lossf, batch, True)
    ld["total"].backward()
    res[tag] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
for n, g in res["explicit"].items():
    if not any(n.startswith("image_encoder._blocks.%d." % i) for i in (0, 1, 2, 3, 4)) and "stem" not in n and "image_encoder._bn0" not in n: continue
    g2 = res["folded"][n]
    cos = float(torch.nn.functional.cosine_similarity(g.flatten().double(), g2.flatten().double(), dim=0))
    print(n, round(cos, 5), tuple(g.shape), float(g.abs().max()), float((g - g2).abs().max()))
