"""debug: determinism bisect - run the same train-mode forward twice, checksum the result of every ops.* call."""
import os, sys, types, inspect
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import mammo_clip_amd  # noqa
from mammo_clip_amd import ops
from oracle import weights as ow
import test_model_gpu as T

dev = torch.device("cuda:0")
junk = torch.randn(1 << 28, device=dev) * 3; del junk
recs = []
def wrap(name, fn):
    def f(*a, **k):
        r = fn(*a, **k)
        outs = r if isinstance(r, (tuple, list)) else (r,)
        cs = tuple(float(o.detach().double().sum()) for o in outs if torch.is_tensor(o) and o.is_floating_point())
        shp = tuple(tuple(o.shape) for o in outs if torch.is_tensor(o))
        recs.append((name, shp, cs))
        return r
    return f
for name, fn in list(vars(ops).items()):
    if inspect.isfunction(fn) and not name.startswith("_") and fn.__module__ == ops.__name__ and name not in ("empty",):
        setattr(ops, name, wrap(name, fn))

tag = sys.argv[1] if len(sys.argv) > 1 else "b5"
enc, arch = ("tf_efficientnet_b5_ns-detect", "efficientnet-b5") if tag == "b5" else ("tf_efficientnetv2-detect", "efficientnet-b2")
model, lossf, sd = T._build(enc, arch)
import numpy as np
z = np.load(os.path.join(T.GOLDEN, ("e2e_b5_small" if tag == "b5" else "e2e_b2_cfg1") + ".npz"))
b, H, W, Tn = [int(v) for v in z["meta"]]
batch = ow.synth_batch(b, H, W, Tn, seed=10)
runs = []
for trial in range(3):
    recs.clear()
    with torch.no_grad():
        out, ld = T._run(model, lossf, batch, True)
    torch.cuda.synchronize()
    runs.append((list(recs), float(ld["total"])))
    print("trial", trial, "loss", runs[-1][1], "ops", len(recs))
for t in (1, 2):
    for i, (a, b_) in enumerate(zip(runs[0][0], runs[t][0])):
        if a != b_:
            print("run0 vs run%d: first divergence at op #%d" % (t, i), a, b_)
            print("   previous op:", runs[0][0][i - 1])
            break
    else:
        print("run0 vs run%d identical" % t)
