"""Loss-deviation probe: |loss(HIP path) - loss(fp32 oracle)| on identical weights and inputs at a workload's per-sample shape.

TEST INFRASTRUCTURE (the checker side of ``bench.py``'s ``parity`` / ``parity_build`` objects and of
tests/test_fullsize_gpu.py's comparisons): the oracle runs torch fp32 on the same GPU after the HIP work has drained; it is
never the thing measured.  north_star's tolerance is |loss - reference| <= 1e-3; the oracle is pinned to the reference by the
golden vectors of tests/golden (tests/test_oracle_golden.py)."""
import torch

from . import arch as oarch, bert as obert, clip as oclip, loss as oloss, weights as ow

EMB = ("image_embeddings", "text_embeddings", "text_embeddings2", "image_view_embeddings")


def _stochastic_off(model):
    enc = model.image_encoder
    enc._dropout_p = 0.0
    enc._global_params = enc._global_params._replace(drop_connect_rate=0.0)
    for lyr in model.text_encoder.text_encoder.encoder.layer:
        lyr.p_attn = lyr.p_hidden = 0.0
    model.text_encoder.text_encoder.config.hidden_dropout_prob = 0.0


def loss_deviation(model, loss_func, arch_name, H, W, T, pairs=2, device="cuda:0", seed=10):
    """``model`` / ``loss_func``: the HIP-path model (any weights: they are overwritten with the probe's synthetic state dict)
    and its loss.  Returns {"eval_dloss", "train_dloss", "eval_min_cos", "train_min_cos", "pairs", "shape"}: eval mode and train
    mode (BatchNorm batch statistics; dropout / drop-connect off on both sides) at ``pairs`` pairs of H x W images and T-token
    reports [ref: model/clip.py:80-114, loss/breast_clip.py:29-127]."""
    from mammo_clip_amd.breastclip import util
    dev = torch.device(device)
    arch = oarch.build_arch(arch_name)
    sd = ow.synth_state_dict(ow.clip_shapes(arch, obert.BertShape()), seed=seed)
    model.load_state_dict(sd, strict=True)
    _stochastic_off(model)
    batch = ow.synth_batch(pairs, H, W, T, seed=seed)
    bt = {"images": batch["images"].to(dev), "image_views": batch["image_views"].to(dev),
          "text_tokens": {k: v.to(dev) for k, v in batch["text_tokens"].items()},
          "text_tokens2": {k: v.to(dev) for k, v in batch["text_tokens2"].items()}}
    sdd = {k: v.to(dev) for k, v in sd.items()}
    rep = {"pairs": pairs, "shape": f"{arch_name} {H}x{W}, T={T}"}
    for train in (False, True):
        util.GlobalEnv.reset()
        model.train(train)
        model.load_state_dict(sd, strict=True)             # (train mode moved the running statistics)
        with torch.no_grad():
            out = model(bt, dev)
            lh = float(loss_func(**out, is_train=train)["total"])
        emb = {k: out[k].detach().float() for k in EMB}
        del out
        torch.cuda.synchronize()
        with torch.backends.cudnn.flags(enabled=False), torch.no_grad():
            oo = oclip.forward(sdd, bt, arch, obert.BertShape(), train=train)
            lo = float(oloss.breast_clip_rank(oo["image_embeddings"], oo["text_embeddings"], oo["text_embeddings2"],
                                              oo["image_view_embeddings"], oo["logit_scale"], 0, pairs)["loss"])
        tag = "train" if train else "eval"
        rep[tag + "_dloss"] = lh - lo
        rep[tag + "_loss"] = (lh, lo)
        rep[tag + "_min_cos"] = min(float(torch.nn.functional.cosine_similarity(emb[k], oo[k].float(), dim=1).min()) for k in EMB)
        del oo
        torch.cuda.empty_cache()
    return rep
