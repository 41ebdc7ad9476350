"""GPU diagnostic: HIP path vs fp32 oracle vs the oracle under torch bf16 autocast (what the reference's own AMP
path would do), train mode, stochastic ops off.  Shows how much of the deviation is inherent to bf16 storage."""
import sys, types, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_amd
from mammo_clip_amd.breastclip.model import build_model
from mammo_clip_amd.breastclip.loss import build_loss
from mammo_clip_amd.breastclip import util
from oracle import arch as oarch, bert as obert, weights as ow, clip as oclip, loss as oloss

DEV = torch.device("cuda:0")

def build(enc_name, arch_name):
    cfg = {"name": "clip_custom", "temperature": 0.07,
           "image_encoder": {"source": "cnn", "name": enc_name, "pretrained": True, "model_type": "cnn"},
           "text_encoder": {"source": "huggingface", "name": "x", "pretrained": False, "gradient_checkpointing": False,
                            "pooling": "eos", "cache_dir": "", "trust_remote_code": True},
           "projection_head": {"name": "linear", "dropout": 0.1, "proj_dim": 512}}
    loss_cfg = {"breast_clip": dict(label_smoothing=0.0, i2i_weight=1.0, t2t_weight=0.5, loss_ratio=1.0)}
    model = build_model(cfg, loss_cfg, types.SimpleNamespace(vocab_size=28996))
    arch = oarch.build_arch(arch_name)
    sd = ow.synth_state_dict(ow.clip_shapes(arch, obert.BertShape()), seed=10)
    model.load_state_dict(sd, strict=True)
    enc = model.image_encoder
    enc._dropout_p = 0.0
    enc._global_params = enc._global_params._replace(drop_connect_rate=0.0)
    for lyr in model.text_encoder.text_encoder.encoder.layer:
        lyr.p_attn = lyr.p_hidden = 0.0
    model.text_encoder.text_encoder.config.hidden_dropout_prob = 0.0
    return model.to(DEV), build_loss(loss_cfg), {k: v.to(DEV) for k, v in sd.items()}, arch

def cos(a, b):
    return float(torch.nn.functional.cosine_similarity(a.float(), b.float(), dim=1).min())

def oracle_loss(out, b):
    return oloss.breast_clip_rank(out["image_embeddings"], out["text_embeddings"], out["text_embeddings2"],
                                  out["image_view_embeddings"], out["logit_scale"], 0, b)["loss"]

for (enc, an, b, H, W, T) in [("tf_efficientnetv2-detect", "efficientnet-b2", 4, 224, 224, 64),
                              ("tf_efficientnetv2-detect", "efficientnet-b2", 16, 224, 224, 64),
                              ("tf_efficientnetv2-detect", "efficientnet-b2", 8, 448, 448, 64),
                              ("tf_efficientnet_b5_ns-detect", "efficientnet-b5", 2, 160, 96, 32),
                              ("tf_efficientnet_b5_ns-detect", "efficientnet-b5", 8, 320, 192, 32)]:
    model, lossf, sd, arch = build(enc, an)
    batch = ow.synth_batch(b, H, W, T, seed=10)
    bt = {"images": batch["images"].to(DEV), "image_views": batch["image_views"].to(DEV),
          "text_tokens": {k: v.to(DEV) for k, v in batch["text_tokens"].items()},
          "text_tokens2": {k: v.to(DEV) for k, v in batch["text_tokens2"].items()}}
    for train in (False, True):
        util.GlobalEnv.reset()
        model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
        model.train(train)
        with torch.no_grad():
            out = model(bt, DEV)
            l_hip = float(lossf(**out, is_train=train)["total"])
            o32 = oclip.forward(sd, bt, arch, obert.BertShape(), train=train)
            o32["labels"] = o32["labels"].to(DEV)
            l32 = float(oracle_loss(o32, b))
            with torch.autocast("cuda", dtype=torch.bfloat16):
                o16 = oclip.forward(sd, bt, arch, obert.BertShape(), train=train)
            l16 = float(oracle_loss({k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in o16.items()}, b))
        print(f"{an} b={b} {H}x{W} train={train}: loss fp32 {l32:.5f} | hip {l_hip:.5f} (d {l_hip-l32:+.5f}) | "
              f"autocast-bf16 oracle {l16:.5f} (d {l16-l32:+.5f}) | cos img hip {cos(out['image_embeddings'], o32['image_embeddings']):.5f} "
              f"ac {cos(o16['image_embeddings'], o32['image_embeddings']):.5f} | cos txt hip {cos(out['text_embeddings'], o32['text_embeddings']):.5f} "
              f"ac {cos(o16['text_embeddings'], o32['text_embeddings']):.5f}", flush=True)
    del model
    torch.cuda.empty_cache()
