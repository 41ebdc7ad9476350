"""Shape fuzz of the whole model (GPU; developer tool): tiny / odd images, odd batches, report lengths 1..512.
Eval-mode embeddings against the fp32 oracle (cosine), one train-mode forward + backward for crashes / NaN."""
import os, sys, types, itertools, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_fullsize_gpu as T
from mammo_clip_amd.breastclip import util
from oracle import weights as ow
cases = [("tf_efficientnetv2-detect", "efficientnet-b2", b, h, w, t) for (b, h, w, t) in
         [(1, 32, 32, 1), (2, 33, 65, 2), (5, 64, 48, 7), (3, 47, 131, 33), (2, 96, 96, 300), (2, 64, 64, 512), (7, 35, 35, 17)]]
cases += [("tf_efficientnet_b5_ns-detect", "efficientnet-b5", b, h, w, t) for (b, h, w, t) in [(1, 32, 32, 3), (3, 65, 33, 25), (2, 129, 97, 257)]]
bad = 0
for enc, arch_name, b, h, w, t in cases:
    try:
        model, lossf, sd, arch = T._build(enc, arch_name)
        bt = T._to_dev(ow.synth_batch(b, h, w, t, seed=3))
        util.GlobalEnv.reset()
        model.eval()
        with torch.no_grad():
            out = model(bt, T.DEV)
            lh = float(lossf(**out, is_train=False)["total"])
        lo, eo, _ = T._oracle(sd, bt, arch, b, False)
        cs = min(T._cos_rows(out[k], eo[k]) for k in T.EMB)
        model.train()
        out = model(bt, T.DEV)
        l = lossf(**out, is_train=True)["total"]
        l.backward()
        gn = sum(float(p.grad.float().pow(2).sum()) for p in model.parameters() if p.grad is not None) ** 0.5
        ok = cs >= 0.999 and abs(lh - lo) < 5e-3 and gn == gn and float(l) == float(l)
        bad += 0 if ok else 1
        print(f"{'ok ' if ok else 'BAD'} {arch_name} b={b} {h}x{w} T={t}: eval min cos {cs:.6f} dloss {lh - lo:+.2e}; train loss {float(l):.4f} |grad| {gn:.3e}", flush=True)
    except Exception as e:
        bad += 1
        print(f"EXC {arch_name} b={b} {h}x{w} T={t}: {type(e).__name__}: {str(e)[:300]}", flush=True)
        tb = traceback.extract_tb(sys.exc_info()[2])
        print("     at", [f"{os.path.basename(f.filename)}:{f.lineno}" for f in tb if "mammo_clip_amd" in f.filename][-4:])
print("bad cases:", bad)
