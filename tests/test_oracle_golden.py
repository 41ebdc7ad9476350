"""Pins the CPU oracle (oracle/) against golden vectors produced by the REFERENCE implementation
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import arch as oarch, bert as obert, clip as oclip, efficientnet as oeff, loss as oloss, weights as ow

TOL = dict(rtol=2e-4, atol=2e-5)


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_arch_tables_match_reference(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "arch_tables.json")))
    for name, tab in ref.items():
        a = oarch.build_arch(name)
        assert a.stem_out == tab["stem_out"] and list(a.stem_pad) == tab["stem_pad"]
        assert a.head_in == tab["head_in"] and a.head_out == tab["head_out"]
        assert abs(a.dropout - tab["dropout"]) < 1e-12
        assert len(a.blocks) == len(tab["blocks"])
        for b, rb in zip(a.blocks, tab["blocks"]):
            got = dict(idx=b.idx, expand=b.expand, k=b.k, s=b.s, cin=b.cin, cexp=b.cexp, cout=b.cout,
                       cse=b.cse, pad=list(b.pad), skip=b.skip)
            assert got == rb, (name, got, rb)
        shapes = ow.efficientnet_shapes(a)
        n_params = sum(int(np.prod(s)) for k, s in shapes.items()
                       if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
        assert n_params == tab["n_params"] and len(shapes) == tab["n_state"]


def test_spatial_chain_b5_matches_survey():
    a = oarch.build_arch("efficientnet-b5")
    chain = oarch.spatial_chain(a, 1520, 912)
    assert chain[0] == (760, 456) and chain[-1] == (48, 29)
    assert a.blocks[8].pad == (1, 2, 1, 2) and a.blocks[3].pad == (0, 1, 0, 1)
    a2 = oarch.build_arch("efficientnet-b2")
    assert oarch.spatial_chain(a2, 224, 224)[-1] == (7, 7)
    assert oarch.spatial_chain(a2, 912, 912)[-1] == (29, 29)


def _mbconv_cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "mbconv_kats.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    return z, names


def test_mbconv_kats(golden_dir):
    z, names = _mbconv_cases(golden_dir)
    assert len(names) >= 6
    for n in names:
        e, k, s, cin, cout, nominal, H, W, b = [int(v) for v in z[f"{n}/meta"]]
        pad = tuple(int(v) for v in z[f"{n}/pad"])
        assert pad == oarch.same_pad((nominal, nominal), k, s)
        blk = oarch.Block(idx=0, expand=e, k=k, s=s, cin=cin, cexp=cin * e, cout=cout,
                          cse=max(1, int(cin * 0.25)), pad=pad, skip=(s == 1 and cin == cout))
        sd = {kk[len(n) + 3:]: _t(z[kk]) for kk in z.files if kk.startswith(n + "/w/")}
        sd = {"blk." + kk: v for kk, v in sd.items()}
        x = _t(z[f"{n}/x"])
        y = oeff.mbconv(sd, "blk", x, blk, train=False)
        np.testing.assert_allclose(y.numpy(), z[f"{n}/y_eval"], **TOL)
        # train mode + grads
        sdg = {kk: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in kk else v)
               for kk, v in sd.items()}
        xr = x.clone().requires_grad_(True)
        nb = {}
        yt = oeff.mbconv(sdg, "blk", xr, blk, train=True, new_buffers=nb)
        np.testing.assert_allclose(yt.detach().numpy(), z[f"{n}/y_train"], **TOL)
        (yt * _t(z[f"{n}/r"])).sum().backward()
        np.testing.assert_allclose(xr.grad.numpy(), z[f"{n}/dx"], rtol=1e-3, atol=1e-4)
        for kk in z.files:
            if kk.startswith(n + "/g/"):
                g = sdg["blk." + kk[len(n) + 3:]].grad
                np.testing.assert_allclose(g.numpy(), z[kk], rtol=2e-3, atol=2e-4, err_msg=kk)
            if kk.startswith(n + "/buf/"):
                np.testing.assert_allclose(nb["blk." + kk[len(n) + 5:]].numpy(), z[kk], **TOL)


def test_bert_kat(golden_dir):
    z = np.load(os.path.join(golden_dir, "bert_kat.npz"))
    vocab, hidden, layers, heads, inter, max_pos, tv = [int(v) for v in z["meta"]]
    cfg = obert.BertShape(vocab=vocab, hidden=hidden, layers=layers, heads=heads, inter=inter,
                          max_pos=max_pos, type_vocab=tv)
    sd = {k[2:]: _t(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith("w/")}
    assert list(ow.bert_shapes(cfg).keys()) == [k for k in sd.keys()]
    tok = dict(input_ids=_t(z["ids"]), attention_mask=_t(z["mask"]))
    out = obert.forward(sd, tok, cfg)
    m = z["mask"][..., None].astype(bool)
    np.testing.assert_allclose(out.detach().numpy() * m, z["out"] * m, **TOL)
    (out * _t(z["r"])).sum().backward()
    for k in z.files:
        if k.startswith("g/"):
            np.testing.assert_allclose(sd[k[2:]].grad.numpy(), z[k], rtol=2e-3, atol=2e-4, err_msg=k)


@pytest.mark.parametrize("cls_name", ["breast_clip", "breast_clip_contrastive"])
@pytest.mark.parametrize("W", [1, 2, 4])
def test_loss_kats(golden_dir, cls_name, W):
    z = np.load(os.path.join(golden_dir, "loss_kats.npz"))
    emb = {k: _t(z[k]).clone().requires_grad_(True) for k in ("img", "txt", "txt2", "view")}
    lsp = _t(z["logit_scale_param"]).clone().requires_grad_(True)
    N = emb["img"].shape[0]
    b = N // W
    total = 0.0
    for r in range(W):
        if cls_name == "breast_clip":
            l = oloss.breast_clip_rank(emb["img"], emb["txt"], emb["txt2"], emb["view"], lsp.exp(), r, b)["loss"]
        else:
            l = oloss.contrastive_rank(emb["img"], emb["txt"], lsp.exp(), r, b)["loss"]
        np.testing.assert_allclose(float(l.detach()), float(z[f"{cls_name}/W{W}/r{r}/total"]), rtol=1e-5, atol=1e-6)
        total = total + l
    # each rank backpropagates ITS OWN loss; reduce_scatter(SUM) hands rank r the sum over ranks of
    # d loss_q / d emb[r-th slice]  == d (sum_q loss_q) / d emb[slice r]
    total.backward()
    for r in range(W):
        sl = slice(r * b, (r + 1) * b)
        for k in emb:
            key = f"{cls_name}/W{W}/r{r}/d{k}"
            g = emb[k].grad if emb[k].grad is not None else torch.zeros_like(emb[k])
            np.testing.assert_allclose(g[sl].numpy(), z[key], rtol=1e-4, atol=1e-6, err_msg=key)
    dsum = sum(float(z[f"{cls_name}/W{W}/r{r}/dscale"]) for r in range(W))
    np.testing.assert_allclose(float(lsp.grad), dsum, rtol=1e-4, atol=1e-6)


def _e2e(golden_dir, tag, arch_name):
    z = np.load(os.path.join(golden_dir, tag + ".npz"))
    b, H, W, T = [int(v) for v in z["meta"]]
    arch = oarch.build_arch(arch_name)
    cfg = obert.BertShape()
    shapes = ow.clip_shapes(arch, cfg)
    assert len(shapes) == int(z["n_state"])
    sd = ow.synth_state_dict(shapes, seed=10)
    batch = ow.synth_batch(b, H, W, T, seed=10)
    return z, arch, cfg, sd, batch, b


def _loss_from(out, b):
    return oloss.breast_clip_rank(out["image_embeddings"], out["text_embeddings"], out["text_embeddings2"],
                                  out["image_view_embeddings"], out["logit_scale"], 0, b)["loss"]


@pytest.mark.parametrize("tag,arch_name", [("e2e_b2_cfg1", "efficientnet-b2"), ("e2e_b5_small", "efficientnet-b5")])
def test_e2e_eval(golden_dir, tag, arch_name):
    z, arch, cfg, sd, batch, b = _e2e(golden_dir, tag, arch_name)
    with torch.no_grad():
        taps = {}
        out = oclip.forward(sd, batch, arch, cfg, train=False, taps=taps)
        loss = _loss_from(out, b)
    for k in ("image_embeddings", "text_embeddings", "text_embeddings2", "image_view_embeddings"):
        np.testing.assert_allclose(out[k].numpy(), z["eval/" + k], rtol=1e-3, atol=2e-5, err_msg=k)
    np.testing.assert_allclose(float(loss), float(z["eval/total"]), rtol=0, atol=1e-5)
    ref_taps = z["eval/block_taps_view0"]
    for i in range(len(arch.blocks)):
        t = taps[f"block{i}"]
        np.testing.assert_allclose([float(t.mean()), float(t.abs().max())], ref_taps[i], rtol=1e-3, atol=1e-4)


def test_e2e_eval_further_seeds(golden_dir):
    """config #1, eval mode, on the two further (weights, inputs) seeds of e2e_b2_cfg1_eval_seeds.npz (reference-generated)"""
    z = np.load(os.path.join(golden_dir, "e2e_b2_cfg1_eval_seeds.npz"))
    b, H, W, T = [int(v) for v in z["meta"]]
    arch, cfg = oarch.build_arch("efficientnet-b2"), obert.BertShape()
    for s_ in [int(v) for v in z["seeds"]]:
        sd = ow.synth_state_dict(ow.clip_shapes(arch, cfg), seed=s_)
        batch = ow.synth_batch(b, H, W, T, seed=s_)
        with torch.no_grad():
            out = oclip.forward(sd, batch, arch, cfg, train=False)
            loss = _loss_from(out, b)
        for k in ("image_embeddings", "text_embeddings", "text_embeddings2", "image_view_embeddings"):
            np.testing.assert_allclose(out[k].numpy(), z[f"s{s_}/eval/{k}"], rtol=1e-3, atol=2e-5, err_msg=f"seed {s_} {k}")
        np.testing.assert_allclose(float(loss), float(z[f"s{s_}/eval/total"]), rtol=0, atol=1e-5)


def test_e2e_train_b2_cfg1(golden_dir):
    z, arch, cfg, sd, batch, b = _e2e(golden_dir, "e2e_b2_cfg1", "efficientnet-b2")
    sdg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v)
           for k, v in sd.items()}
    nb = {}
    out = oclip.forward(sdg, batch, arch, cfg, train=True, new_buffers=nb)
    loss = _loss_from(out, b)
    np.testing.assert_allclose(float(loss), float(z["train/total"]), rtol=0, atol=2e-5)
    loss.backward()
    for k in z.files:
        if k.startswith("train/grad/"):
            g = sdg[k[len("train/grad/"):]].grad
            ref = z[k]
            scale = max(1e-8, float(np.abs(ref).max()))
            assert float(np.abs(g.numpy() - ref).max()) <= 5e-3 * scale + 1e-7, k
        if k.startswith("train/buf/") and "num_batches" not in k:
            np.testing.assert_allclose(nb[k[len("train/buf/"):]].numpy(), z[k], rtol=1e-3, atol=1e-5, err_msg=k)
    names, norms = list(z["train/grad_names"]), z["train/grad_norms"]
    bad = []
    for n, ref in zip(names, norms):
        g = sdg[str(n)].grad
        if ref < 0:
            assert g is None or float(g.abs().max()) == 0.0      # pooler.* is unused (text_encoder.py:49)
            continue
        got = float(g.norm())
        if abs(got - ref) > 5e-3 * ref + 3e-6:   # bn2.bias grads are pure round-off (a train-mode BN follows)
            bad.append((str(n), got, float(ref)))
    assert not bad, bad[:5]


def test_input_pipeline_vs_reference(golden_dir):
    """Row N4 pin: raw uint8 pixels -> the float32 batch the reference's ImageTextDataset.__getitem__ + collate_fn +
    trainer permute produced (make_golden.py gen_input_pipeline): the oracle's three-line restatement must reproduce
    it bit for bit, in the [b,3,H,W] permuted layout the model receives."""
    from oracle import inputs as oin
    z = np.load(os.path.join(golden_dir, "input_pipeline.npz"))
    mean, std = float(z["mean_std"][0]), float(z["mean_std"][1])
    for key in ("images", "image_views"):
        raw, ref = z["raw/" + key], z["out/" + key]
        assert raw.dtype == np.uint8 and ref.dtype == np.float32 and ref.shape == (raw.shape[0], 3) + raw.shape[1:3]
        got = np.stack([oin.normalize_u8(raw[i], mean, std) for i in range(raw.shape[0])]).transpose(0, 3, 1, 2)
        assert np.array_equal(got, ref), float(np.abs(got - ref).max())
        # each image is min-max scaled on its own: the per-image extremes map to (0 - mean)/std and (1 - mean)/std
        for i in range(raw.shape[0]):
            assert np.isclose(ref[i].min(), (0.0 - mean) / std) and np.isclose(ref[i].max(), (1.0 - mean) / std)


def test_bf16_operand_floor_of_train_mode_loss():
    """DESIGN.md "Tolerances": north_star's |loss - reference| <= 1e-3 is below what ANY bf16-MFMA implementation can
    hold in TRAIN mode at random initialisation.  Evidence on the fp32 oracle itself (config #1: B2, b = 4, 224^2,
    T = 64, stochastic ops off): rounding ONLY the 1x1-convolution / stem weights to bf16 -- every activation, every
    statistic and the whole text encoder still fp32 -- moves the loss by more than 1e-3, and so does each stored
    activation class on its own, while fp16 storage (the reference's AMP dtype, trainer.py:271-278) of ALL classes stays
    inside 1e-3.  (Full table: scripts/rounding_ablation.py, profiles/r03_rounding_ablation_*.json.)"""
    import torch
    from oracle import arch as oarch, bert as obert, clip as oclip, efficientnet as oeff, loss as oloss, weights as ow
    arch = oarch.build_arch("efficientnet-b2")
    b = 4
    sd = ow.synth_state_dict(ow.clip_shapes(arch, obert.BertShape()), seed=10)
    batch = ow.synth_batch(b, 224, 224, 64, seed=10)
    with torch.no_grad():
        txt = [oclip._project_norm(sd, "text_projection", oclip.encode_text(sd, batch[k], obert.BertShape()))
               for k in ("text_tokens", "text_tokens2")]

        def loss(tags, dtype=torch.bfloat16):
            oeff.ROUND, oeff.ROUND_DTYPE = tags, dtype
            try:
                img = [oclip._project_norm(sd, "image_projection", oeff.forward(sd, batch[k], arch, True, "image_encoder."))
                       for k in ("images", "image_views")]
            finally:
                oeff.ROUND, oeff.ROUND_DTYPE = None, torch.bfloat16
            return float(oloss.breast_clip_rank(img[0], txt[0], txt[1], img[1], sd["logit_scale"].exp(), 0, b)["loss"])

        l0 = loss(None)
        dev = {t: abs(loss({t}) - l0) for t in ("W", "E", "D", "Y")}
        all16 = abs(loss({"IN", "W", "E", "D", "A1", "P", "Y", "H"}, torch.float16) - l0)
    assert all(v > 1e-3 for v in dev.values()), dev
    assert all16 < 1e-3, all16
