"""Which STORED tensor class carries the train-mode loss deviation?  ("what's weak" #1 / next #1b.)

CPU experiment on the oracle (no GPU): the fp32 restatement is run in train mode (dropout / drop-connect off) with the
values of ONE tensor class at a time rounded to bf16 exactly where the HIP path stores that class (or feeds an MFMA
with it) -- oracle.efficientnet.ROUND -- and |loss - fp32 loss| / min image-embedding cosine are tabulated.  Classes:
E expand-conv output, D depthwise output, P project-conv output, Y block output (residual stream), A1 the gated
activation operand of the project GEMM, W 1x1/stem weights, H head-conv output, IN stem patches.  "all" = the HIP path's
storage set, "all-fp16" the same set rounded to fp16 (the reference's AMP dtype, trainer.py:271-278).

usage: python scripts/rounding_ablation.py [bn8k|cfg1|b5small] [--eval] [--out FILE.json]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import arch as oarch, bert as obert, clip as oclip, efficientnet as oeff, loss as oloss, weights as ow  # noqa: E402

CASES = {"bn8k": ("efficientnet-b2", 8, 456, 456, 64), "cfg1": ("efficientnet-b2", 4, 224, 224, 64),
         "b5small": ("efficientnet-b5", 4, 320, 192, 32)}
CLASSES = ["IN", "W", "E", "D", "A1", "P", "Y", "H"]


def run(case, rounds, train=True):
    an, b, H, W, T = CASES[case]
    arch = oarch.build_arch(an)
    sd = ow.synth_state_dict(ow.clip_shapes(arch, obert.BertShape()), seed=10)
    batch = ow.synth_batch(b, H, W, T, seed=10)
    txt = {}

    def go(tags, dtype=torch.bfloat16):
        oeff.ROUND, oeff.ROUND_DTYPE = (set(tags) if tags else None), dtype
        try:
            with torch.no_grad():
                if not txt:                                          # the text side is identical in every arm: once
                    for k in ("text_tokens", "text_tokens2"):
                        h = oclip.encode_text(sd, batch[k], obert.BertShape())
                        txt[k] = oclip._project_norm(sd, "text_projection", h)
                img = oclip._project_norm(sd, "image_projection",
                                          oeff.forward(sd, batch["images"], arch, train, "image_encoder."))
                view = oclip._project_norm(sd, "image_projection",
                                           oeff.forward(sd, batch["image_views"], arch, train, "image_encoder."))
                loss = oloss.breast_clip_rank(img, txt["text_tokens"], txt["text_tokens2"], view, sd["logit_scale"].exp(),
                                              0, b)["loss"]
            return float(loss), img, view
        finally:
            oeff.ROUND = None

    t0 = time.time()
    l0, i0, v0 = go(None)
    print(f"{case}: fp32 {'train' if train else 'eval'}-mode loss {l0:.6f}  ({time.time() - t0:.1f} s per arm)", flush=True)
    rows = []
    for name, tags, dt in rounds:
        l, i, v = go(tags, dt)
        cos = min(float(torch.nn.functional.cosine_similarity(i, i0, dim=1).min()),
                  float(torch.nn.functional.cosine_similarity(v, v0, dim=1).min()))
        rows.append(dict(arm=name, dloss=l - l0, min_cos=cos))
        print(f"  {name:>12s}: dloss {l - l0:+.2e}   min image cos {cos:.6f}", flush=True)
    return dict(case=case, mode='train' if train else 'eval', fp32_loss=l0, rows=rows)


if __name__ == "__main__":
    torch.set_num_threads(8)
    case = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "bn8k"
    rounds = [(c, [c], torch.bfloat16) for c in CLASSES]
    rounds += [("all", CLASSES, torch.bfloat16), ("all-but-Y", [c for c in CLASSES if c != "Y"], torch.bfloat16),
               ("all-but-E,D", [c for c in CLASSES if c not in "ED"], torch.bfloat16),
               ("E,D only", ["E", "D"], torch.bfloat16), ("all-fp16", CLASSES, torch.float16)]
    res = run(case, rounds, train="--eval" not in sys.argv)
    if "--out" in sys.argv:
        with open(sys.argv[sys.argv.index("--out") + 1], "w") as f:
            json.dump(res, f, indent=1)
