#!/usr/bin/env python3
"""DEV-ONLY golden-vector generator.  Runs in the build container only (needs /root/reference).

Imports the REFERENCE implementation (``/root/reference/src/codebase/breastclip``) with empty stubs
for the third-party packages that are absent here (recipe: SURVEY.md section 8c), drives it on seeded
synthetic inputs and weights (``oracle/weights.py``) and stores the INPUT/OUTPUT VECTORS it produced
as small fixtures next to this file.  Nothing of the reference travels: the fixtures are data only.

    python tests/golden/make_golden.py            # regenerate everything (~2-4 min on 8 cores)

Fixtures (checked by tests/test_oracle_golden.py and by the GPU parity tests):
  arch_tables.json  per-block channel/kernel/stride/SE/static-pad tables of the reference modules
  mbconv_kats.npz   MBConvBlock known-answer tests (eval + train, outputs + grads)
  bert_kat.npz      small BertModel known-answer test (padding mask, outputs + grads)
  loss_kats.npz     both loss classes, W in {1,2,4} ranks over gloo (losses + grads)
  e2e_b2_cfg1_eval_seeds.npz   config #1, eval mode, two further (weights, inputs) seeds: embeddings + loss
  e2e_b2_cfg1.npz   BASELINE config #1 (B2 + BERT-base, b=4, 224^2, T=64): embeddings, losses, grads
  e2e_b5_small.npz  B5 + BERT-base, b=2, 160x96, T=32: same (pins the B5 table end to end)
  e2e_b2_bn8k.npz   B2 + BERT-base, b=8, 456^2, T=64: train-mode fixture in which every BatchNorm sees >= 1800 samples
                    per channel (the bf16 train-mode tolerance is stated on this one)
  input_pipeline.npz  the reference's ImageTextDataset.__getitem__ + collate + trainer permute driven on generated PNGs:
                    raw uint8 pixels in, the normalised float32 batch tensor out (pins row N4)
  ref_cpu_timing.json  the reference's own config-#1 training step timed on this container's cores
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src/codebase"
sys.path.insert(0, ROOT)


def import_reference():
    """SURVEY.md section 8c steps 1-4."""
    from transformers import AutoConfig, AutoModel, BertConfig, BertModel, SwinModel, ViTModel  # noqa: F401

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

    stub("timm")
    tv = stub("torchvision")
    tvm = stub("torchvision.models")
    tvr = stub("torchvision.models.resnet", resnet50=None, resnet101=None, resnet152=None)
    tv.models, tvm.resnet = tvm, tvr
    stub("omegaconf", OmegaConf=_Dummy, DictConfig=dict)
    stub("nltk", download=lambda *a, **k: None, tokenize=types.ModuleType("tokenize"))
    stub("albumentations", __all__=[])
    stub("cv2")
    stub("tensorboard")
    stub("torch.utils.tensorboard", SummaryWriter=_Dummy)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import breastclip  # noqa: F401
    from breastclip.model.modules import efficientnet_custom, text_encoder
    efficientnet_custom.load_pretrained_weights = lambda *a, **k: None

    def fake_cfg(name, **kw):
        return BertConfig(vocab_size=28996, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                          intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
                          attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
                          layer_norm_eps=1e-12, pad_token_id=0)

    text_encoder.AutoConfig.from_pretrained = staticmethod(fake_cfg)
    from breastclip import util
    util.GlobalEnv.get().summary_writer.train = _Dummy()
    return breastclip


def np_(t):
    return t.detach().cpu().numpy()


# ---------------------------------------------------------------------------------------------
def gen_arch_tables():
    from breastclip.model.modules.efficientnet_custom import EfficientNet
    out = {}
    for name in ("efficientnet-b2", "efficientnet-b5"):
        m = EfficientNet.from_name(name, num_classes=1)

        def pad_of(conv):
            sp = conv.static_padding
            return list(sp.padding) if hasattr(sp, "padding") else [0, 0, 0, 0]

        blocks = []
        for i, b in enumerate(m._blocks):
            a = b._block_args
            blocks.append(dict(idx=i, expand=a.expand_ratio, k=a.kernel_size,
                               s=a.stride if isinstance(a.stride, int) else a.stride[0],
                               cin=a.input_filters, cexp=a.input_filters * a.expand_ratio,
                               cout=a.output_filters, cse=b._se_reduce.out_channels,
                               pad=pad_of(b._depthwise_conv),
                               skip=bool(a.id_skip and (a.stride in (1, [1])) and a.input_filters == a.output_filters)))
        out[name] = dict(stem_out=m._conv_stem.out_channels, stem_pad=pad_of(m._conv_stem),
                         head_in=m._conv_head.in_channels, head_out=m._conv_head.out_channels,
                         dropout=m._global_params.dropout_rate, n_params=sum(p.numel() for p in m.parameters()),
                         n_state=len(m.state_dict()), blocks=blocks)
    json.dump(out, open(os.path.join(HERE, "arch_tables.json"), "w"), indent=1)
    print("arch_tables.json", {k: len(v["blocks"]) for k, v in out.items()})


# ---------------------------------------------------------------------------------------------
MBCONV_CASES = [
    # name, expand, k, s, cin, cout, nominal_image_size, H, W, batch
    ("e1_k3_s1_skip", 1, 3, 1, 16, 16, 130, 9, 7, 2),
    ("e1_k3_s1_noskip", 1, 3, 1, 32, 16, 130, 8, 8, 2),
    ("e6_k3_s2_pad01", 6, 3, 2, 16, 24, 130, 10, 8, 2),      # (0,1,0,1) asymmetric
    ("e6_k5_s2_pad12", 6, 5, 2, 24, 40, 114, 11, 9, 2),      # (1,2,1,2) asymmetric (B5 blk8 case)
    ("e6_k5_s1_skip", 6, 5, 1, 40, 40, 29, 7, 6, 3),
    ("e6_k3_s2_pad11", 6, 3, 2, 40, 80, 57, 9, 9, 2),        # (1,1,1,1) with stride 2
]


def gen_mbconv_kats():
    from breastclip.model.modules.efficientnet_custom import MBConvBlock
    from breastclip.model.modules.efficient_net_custom_utils import BlockArgs, GlobalParams
    from oracle.weights import synth_tensor
    store = {}
    gp = GlobalParams(batch_norm_momentum=0.99, batch_norm_epsilon=1e-3, drop_connect_rate=0.2,
                      depth_divisor=8, width_coefficient=1.0, depth_coefficient=1.0, image_size=224,
                      dropout_rate=0.2, num_classes=1, min_depth=None, include_top=True)
    for (name, e, k, s, cin, cout, nominal, H, W, b) in MBCONV_CASES:
        ba = BlockArgs(num_repeat=1, kernel_size=k, stride=s, expand_ratio=e, input_filters=cin,
                       output_filters=cout, se_ratio=0.25, id_skip=True)
        blk = MBConvBlock(ba, gp, image_size=[nominal, nominal])
        sd = {kk: synth_tensor(f"{name}.{kk}", tuple(v.shape), 7) for kk, v in blk.state_dict().items()}
        blk.load_state_dict(sd, strict=True)
        g = torch.Generator().manual_seed(1234)
        x = torch.randn(b, cin, H, W, generator=g)
        store[f"{name}/meta"] = np.array([e, k, s, cin, cout, nominal, H, W, b], dtype=np.int64)
        store[f"{name}/pad"] = np.array(getattr(blk._depthwise_conv.static_padding, "padding", (0, 0, 0, 0)),
                                        dtype=np.int64)
        for kk, v in sd.items():
            store[f"{name}/w/{kk}"] = np_(v)
        store[f"{name}/x"] = np_(x)
        blk.eval()
        with torch.no_grad():
            store[f"{name}/y_eval"] = np_(blk(x, drop_connect_rate=0.1))   # eval: drop_connect is identity
        blk.train()
        xr = x.clone().requires_grad_(True)
        y = blk(xr, drop_connect_rate=None)
        r = torch.randn(y.shape, generator=g)
        store[f"{name}/y_train"] = np_(y)
        store[f"{name}/r"] = np_(r)
        (y * r).sum().backward()
        store[f"{name}/dx"] = np_(xr.grad)
        for kk, p in blk.named_parameters():
            store[f"{name}/g/{kk}"] = np_(p.grad)
        for kk, v in blk.state_dict().items():
            if "running" in kk:
                store[f"{name}/buf/{kk}"] = np_(v)
    np.savez_compressed(os.path.join(HERE, "mbconv_kats.npz"), **store)
    print("mbconv_kats.npz", len(store), "arrays")


# ---------------------------------------------------------------------------------------------
def gen_bert_kat():
    from transformers import BertConfig, BertModel
    from oracle.weights import synth_tensor
    cfg = BertConfig(vocab_size=300, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                     intermediate_size=160, hidden_act="gelu", hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0, max_position_embeddings=40, type_vocab_size=2,
                     layer_norm_eps=1e-12, pad_token_id=0)
    m = BertModel(cfg)
    sd = {k: synth_tensor("bertkat." + k, tuple(v.shape), 3) for k, v in m.state_dict().items()}
    m.load_state_dict(sd, strict=True)
    m.train()   # dropout p = 0 -> deterministic; exercises the train-mode code path
    g = torch.Generator().manual_seed(99)
    b, T = 3, 24
    ids = torch.randint(5, 300, (b, T), generator=g)
    lens = torch.tensor([24, 9, 17])
    mask = (torch.arange(T)[None] < lens[:, None]).long()
    ids = ids * mask
    tok = dict(input_ids=ids, token_type_ids=torch.zeros_like(ids), attention_mask=mask)
    out = m(**tok)["last_hidden_state"]
    r = torch.randn(out.shape, generator=g) * mask[..., None]
    (out * r).sum().backward()
    store = {"meta": np.array([300, 64, 2, 4, 160, 40, 2], dtype=np.int64), "ids": np_(ids), "mask": np_(mask),
             "out": np_(out), "r": np_(r)}
    for k, v in sd.items():
        store["w/" + k] = np_(v)
    for k, p in m.named_parameters():
        if p.grad is not None:
            store["g/" + k] = np_(p.grad)
    np.savez_compressed(os.path.join(HERE, "bert_kat.npz"), **store)
    print("bert_kat.npz", len(store), "arrays; unused-grad params:",
          [k for k, p in m.named_parameters() if p.grad is None])


# ---------------------------------------------------------------------------------------------
def _loss_worker(rank, W, port, emb, b, cls_name, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if W > 1:
        dist.init_process_group("gloo", rank=rank, world_size=W)
    import_reference()
    from breastclip.loss import build_loss
    cfg = {cls_name: dict(label_smoothing=0.0, i2i_weight=1.0, t2t_weight=0.5, loss_ratio=1.0)}
    lf = build_loss(cfg)
    sl = slice(rank * b, (rank + 1) * b)
    loc = {k: v[sl].clone().requires_grad_(True) for k, v in emb.items() if k != "logit_scale_param"}
    lsp = emb["logit_scale_param"].clone().requires_grad_(True)
    out = dict(image_embeddings=loc["img"], text_embeddings=loc["txt"], text_embeddings2=loc["txt2"],
               image_view_embeddings=loc["view"], labels=torch.arange(b), logit_scale=lsp.exp())
    ld = lf(**out, is_train=True)
    ld["total"].backward()
    res = {"contrastive": float(ld["contrastive"].detach()), "total": float(ld["total"].detach()),
           "dscale": float(lsp.grad)}
    for k in loc:
        res["d" + k] = np_(loc[k].grad) if loc[k].grad is not None else np.zeros_like(np_(loc[k]))
    ret[rank] = res
    if W > 1:
        dist.destroy_process_group()


def gen_loss_kats():
    import torch.multiprocessing as mp
    g = torch.Generator().manual_seed(2024)
    N, D = 16, 48
    emb = {}
    base = torch.randn(N, D, generator=g)
    for k in ("img", "txt", "txt2", "view"):
        e = base + 0.7 * torch.randn(N, D, generator=g)
        emb[k] = e / e.norm(dim=1, keepdim=True)
    emb["logit_scale_param"] = torch.tensor(float(np.log(1 / 0.07)))
    store = {k: np_(v).copy() for k, v in emb.items()}   # copy: mp.spawn moves the tensors to shm
    port = 29650
    for cls_name in ("breast_clip", "breast_clip_contrastive"):
        for W in (1, 2, 4):
            b = N // W
            mgr = mp.Manager()
            ret = mgr.dict()
            if W == 1:
                _loss_worker(0, 1, port, emb, b, cls_name, ret)
            else:
                port += 1
                mp.spawn(_loss_worker, args=(W, port, emb, b, cls_name, ret), nprocs=W, join=True)
            for r in range(W):
                for k, v in ret[r].items():
                    store[f"{cls_name}/W{W}/r{r}/{k}"] = np.asarray(v)
            print(cls_name, "W", W, [round(ret[r]["total"], 6) for r in range(W)])
    np.savez_compressed(os.path.join(HERE, "loss_kats.npz"), **store)


# ---------------------------------------------------------------------------------------------
GRAD_KEYS_FULL = [
    "logit_scale", "image_projection.projection.bias", "text_projection.projection.bias",
    "image_encoder._conv_stem.weight", "image_encoder._bn0.weight",
    "image_encoder._blocks.0._depthwise_conv.weight", "image_encoder._blocks.0._se_reduce.weight",
    "image_encoder._blocks.2._expand_conv.weight", "image_encoder._blocks.3._expand_conv.weight",
    "image_encoder._blocks.2._bn1.bias",
    "image_encoder._bn1.weight",
    "text_encoder.text_encoder.embeddings.LayerNorm.weight",
    "text_encoder.text_encoder.embeddings.position_embeddings.weight",
    "text_encoder.text_encoder.encoder.layer.0.attention.self.query.bias",
    "text_encoder.text_encoder.encoder.layer.11.output.LayerNorm.weight",
    "text_encoder.text_encoder.encoder.layer.11.intermediate.dense.bias",
]


def gen_e2e(tag, enc_name, arch_name, b, H, W, T):
    from breastclip.model import build_model
    from breastclip.loss import build_loss
    from breastclip.model.modules import efficient_net_custom_utils as U
    from oracle import arch as oarch, weights as ow
    from oracle.bert import BertShape
    model_cfg = {"name": "clip_custom", "temperature": 0.07,
                 "image_encoder": {"source": "cnn", "name": enc_name, "pretrained": True, "model_type": "cnn"},
                 "text_encoder": {"source": "huggingface", "name": "emilyalsentzer/Bio_ClinicalBERT",
                                  "pretrained": False, "gradient_checkpointing": False, "pooling": "eos",
                                  "cache_dir": "/tmp/none", "trust_remote_code": True, "mlm_head": True},
                 "projection_head": {"name": "linear", "dropout": 0.1, "proj_dim": 512}}
    loss_cfg = {"breast_clip": dict(label_smoothing=0.0, i2i_weight=1.0, t2t_weight=0.5, loss_ratio=1.0)}
    tok = types.SimpleNamespace(vocab_size=28996)
    torch.manual_seed(10)
    model = build_model(model_cfg, loss_cfg, tok)
    arch = oarch.build_arch(arch_name)
    shapes = ow.clip_shapes(arch, BertShape())
    ref_sd = model.state_dict()
    assert list(ref_sd.keys()) == list(shapes.keys()), "state_dict key inventory/order mismatch"
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), (k, v.shape, shapes[k])
    sd = ow.synth_state_dict(shapes, seed=10)
    model.load_state_dict(sd, strict=True)
    # all stochastic ops off
    model.image_encoder._dropout.p = 0.0
    model.image_encoder._global_params = model.image_encoder._global_params._replace(drop_connect_rate=0.0)
    for mod in model.text_encoder.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if hasattr(mod, "dropout") and isinstance(getattr(mod, "dropout"), float):
            mod.dropout = 0.0
    model.text_encoder.text_encoder.config.attention_probs_dropout_prob = 0.0
    model.text_encoder.text_encoder.config.hidden_dropout_prob = 0.0
    batch = ow.synth_batch(b, H, W, T, seed=10)
    lf = build_loss(loss_cfg)
    store = {"meta": np.array([b, H, W, T], dtype=np.int64),
             "n_params": np.array(sum(p.numel() for p in model.parameters())),
             "n_state": np.array(len(ref_sd))}

    class BE(dict):
        def to(self, device):
            return self

    def run(train):
        model.train(train)
        model.load_state_dict(sd, strict=True)
        bt = {"images": batch["images"], "image_views": batch["image_views"],
              "text_tokens": BE(batch["text_tokens"]), "text_tokens2": BE(batch["text_tokens2"])}
        model.zero_grad(set_to_none=True)
        out = model(bt, torch.device("cpu"))
        ld = lf(**out, is_train=train)
        return out, ld

    # block taps (eval) via forward hooks for bisecting
    taps = {}
    hooks = []
    for i, blk in enumerate(model.image_encoder._blocks):
        hooks.append(blk.register_forward_hook(
            lambda m, a, o, i=i: taps.setdefault(f"block{i}", []).append((float(o.mean()), float(o.abs().max())))))
    with torch.no_grad():
        out, ld = run(False)
    for h in hooks:
        h.remove()
    for k in ("image_embeddings", "text_embeddings", "text_embeddings2", "image_view_embeddings"):
        store["eval/" + k] = np_(out[k])
    store["eval/logit_scale"] = np_(out["logit_scale"])
    store["eval/contrastive"] = np_(ld["contrastive"])
    store["eval/total"] = np_(ld["total"])
    store["eval/block_taps_view0"] = np.array([taps[f"block{i}"][0] for i in range(len(model.image_encoder._blocks))])
    print(tag, "eval loss", float(ld["total"]))

    out, ld = run(True)
    ld["total"].backward()
    for k in ("image_embeddings", "text_embeddings", "text_embeddings2", "image_view_embeddings"):
        store["train/" + k] = np_(out[k])
    store["train/contrastive"] = np_(ld["contrastive"])
    store["train/total"] = np_(ld["total"])
    names, norms = [], []
    for k, p in model.named_parameters():
        names.append(k)
        norms.append(float(p.grad.norm()) if p.grad is not None else -1.0)
    store["train/grad_names"] = np.array(names)
    store["train/grad_norms"] = np.array(norms)
    pd = dict(model.named_parameters())
    for k in GRAD_KEYS_FULL:
        if k in pd:
            store["train/grad/" + k] = np_(pd[k].grad)
    wg = pd["text_encoder.text_encoder.embeddings.word_embeddings.weight"].grad
    rows = torch.tensor([0, 101, 102, int(batch["text_tokens"]["input_ids"][0, 1])])
    store["train/grad_word_rows_idx"] = np_(rows)
    store["train/grad_word_rows"] = np_(wg[rows])
    newsd = model.state_dict()
    for k in ("image_encoder._bn0.running_mean", "image_encoder._bn0.running_var",
              "image_encoder._blocks.3._bn1.running_var", "image_encoder._bn1.running_mean",
              "image_encoder._bn0.num_batches_tracked"):
        store["train/buf/" + k] = np_(newsd[k])
    print(tag, "train loss", float(ld["total"].detach()), "params", int(store["n_params"]), "state", int(store["n_state"]))
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **store)


def gen_e2e_eval_seeds(seeds=(11, 12), enc_name="tf_efficientnetv2-detect", arch_name="efficientnet-b2", b=4, H=224, W=224, T=64):
    """BASELINE config #1 in EVAL mode on further (weights, inputs) seeds (VERDICT r3 #6b: the 1e-3 loss bound held with 11 %
    margin on ONE seed): embeddings + loss of the reference for each seed -> e2e_b2_cfg1_eval_seeds.npz"""
    from breastclip.model import build_model
    from breastclip.loss import build_loss
    from oracle import arch as oarch, weights as ow
    from oracle.bert import BertShape
    model_cfg = {"name": "clip_custom", "temperature": 0.07,
                 "image_encoder": {"source": "cnn", "name": enc_name, "pretrained": True, "model_type": "cnn"},
                 "text_encoder": {"source": "huggingface", "name": "emilyalsentzer/Bio_ClinicalBERT",
                                  "pretrained": False, "gradient_checkpointing": False, "pooling": "eos",
                                  "cache_dir": "/tmp/none", "trust_remote_code": True, "mlm_head": True},
                 "projection_head": {"name": "linear", "dropout": 0.1, "proj_dim": 512}}
    loss_cfg = {"breast_clip": dict(label_smoothing=0.0, i2i_weight=1.0, t2t_weight=0.5, loss_ratio=1.0)}
    torch.manual_seed(10)
    model = build_model(model_cfg, loss_cfg, types.SimpleNamespace(vocab_size=28996))
    shapes = ow.clip_shapes(oarch.build_arch(arch_name), BertShape())
    lf = build_loss(loss_cfg)
    store = {"meta": np.array([b, H, W, T], dtype=np.int64), "seeds": np.array(seeds, dtype=np.int64)}

    class BE(dict):
        def to(self, device):
            return self
    for sd_ in seeds:
        model.load_state_dict(ow.synth_state_dict(shapes, seed=sd_), strict=True)
        model.eval()
        batch = ow.synth_batch(b, H, W, T, seed=sd_)
        bt = {"images": batch["images"], "image_views": batch["image_views"],
              "text_tokens": BE(batch["text_tokens"]), "text_tokens2": BE(batch["text_tokens2"])}
        with torch.no_grad():
            out = model(bt, torch.device("cpu"))
            ld = lf(**out, is_train=False)
        for k in ("image_embeddings", "text_embeddings", "text_embeddings2", "image_view_embeddings"):
            store[f"s{sd_}/eval/{k}"] = np_(out[k])
        store[f"s{sd_}/eval/total"] = np_(ld["total"])
        print("eval seeds", sd_, float(ld["total"]))
    np.savez_compressed(os.path.join(HERE, "e2e_b2_cfg1_eval_seeds.npz"), **store)


def gen_traj(tag="traj_b5_small", enc_name="tf_efficientnet_b5_ns-detect", arch_name="efficientnet-b5", b=2, H=160, W=96, T=32,
             steps=4):
    """Row H (the hot loop): the reference's own build_optimizer / build_scheduler / call order for a few steps on a
    fixed batch with every stochastic op off -> per-step losses and parameter deltas [ref: trainer_ddp.py:279-308]."""
    from breastclip.model import build_model
    from breastclip.loss import build_loss
    from breastclip.optimizer import build_optimizer
    from breastclip.scheduler import build_scheduler
    from oracle import arch as oarch, weights as ow
    from oracle.bert import BertShape
    model_cfg = {"name": "clip_custom", "temperature": 0.07,
                 "image_encoder": {"source": "cnn", "name": enc_name, "pretrained": True, "model_type": "cnn"},
                 "text_encoder": {"source": "huggingface", "name": "emilyalsentzer/Bio_ClinicalBERT",
                                  "pretrained": False, "gradient_checkpointing": False, "pooling": "eos",
                                  "cache_dir": "/tmp/none", "trust_remote_code": True, "mlm_head": True},
                 "projection_head": {"name": "linear", "dropout": 0.1, "proj_dim": 512}}
    loss_cfg = {"breast_clip": dict(label_smoothing=0.0, i2i_weight=1.0, t2t_weight=0.5, loss_ratio=1.0)}
    torch.manual_seed(10)
    model = build_model(model_cfg, loss_cfg, types.SimpleNamespace(vocab_size=28996))
    shapes = ow.clip_shapes(oarch.build_arch(arch_name), BertShape())
    sd = ow.synth_state_dict(shapes, seed=10)
    model.load_state_dict(sd, strict=True)
    model.image_encoder._dropout.p = 0.0
    model.image_encoder._global_params = model.image_encoder._global_params._replace(drop_connect_rate=0.0)
    for mod in model.text_encoder.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    model.text_encoder.text_encoder.config.attention_probs_dropout_prob = 0.0
    model.text_encoder.text_encoder.config.hidden_dropout_prob = 0.0
    batch = ow.synth_batch(b, H, W, T, seed=10)
    lf = build_loss(loss_cfg)
    lr, wd, total, warm = 1e-5, 1e-4, 10, 2          # large enough that the steps visibly move the loss, small enough to stay smooth
    opt = build_optimizer(model, {"name": "adamw", "config": {"lr": lr, "weight_decay": wd}})
    sch = build_scheduler(opt, {"name": "cosine", "config": {"total_steps": total, "warmup_steps": warm}})

    class BE(dict):
        def to(self, device):
            return self

    watch = ["logit_scale", "image_projection.projection.weight", "text_projection.projection.bias",
             "image_encoder._bn0.weight", "image_encoder._conv_head.weight",
             "text_encoder.text_encoder.encoder.layer.11.output.dense.bias"]
    p0 = {k: v.detach().clone() for k, v in model.named_parameters() if k in watch}
    losses, lrs = [], []
    model.train()
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        bt = {"images": batch["images"], "image_views": batch["image_views"],
              "text_tokens": BE(batch["text_tokens"]), "text_tokens2": BE(batch["text_tokens2"])}
        out = model(bt, torch.device("cpu"))
        ld = lf(**out, is_train=True)
        ld["total"].backward()
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
        losses.append(float(ld["total"].detach()))
    store = {"meta": np.array([b, H, W, T, steps], dtype=np.int64), "hyper": np.array([lr, wd, total, warm]),
             "losses": np.array(losses), "lrs": np.array(lrs)}
    pd = dict(model.named_parameters())
    for k in watch:
        store["delta/" + k] = np_((pd[k].detach() - p0[k]).reshape(-1)[:4096])     # leading slice: small fixture
    print(tag, "losses", losses, "lrs", lrs)
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **store)


def gen_input_pipeline():
    """Row N4: drive the reference's own ``ImageTextDataset.__getitem__`` [ref: data/datasets/imagetext.py:67-212] and
    ``collate_fn`` (:214-234) on generated PNG files, then the trainer's permute [ref: trainer_ddp.py:288-291]; store
    the raw uint8 pixels and the float32 batch tensor the model receives.  ``__init__`` opens a hard-coded absolute
    path of the authors' cluster (:54-57), so the object is allocated without it and given the attributes
    ``__getitem__`` reads; no transform (``tfms = None``, the 'valid' split of the shipped configs has none)."""
    import tempfile
    import pandas as pd
    from pathlib import Path
    from PIL import Image
    from breastclip.data.datasets.imagetext import ImageTextDataset
    rng = np.random.default_rng(1234)
    H, W = 76, 48
    mean, std = 0.3089279, 0.25053555408335154            # configs/pre_train_b5_clip.yaml:23-24
    cases = {"full_range": (0, 256), "narrow": (17, 201), "dark": (3, 40)}
    store = {"meta": np.array([H, W]), "mean_std": np.array([mean, std], dtype=np.float64)}
    with tempfile.TemporaryDirectory() as td:
        rows = []
        for pid, (name, (lo, hi)) in enumerate(cases.items()):
            d = Path(td) / "img" / str(pid)
            d.mkdir(parents=True)
            files = []
            for v in range(2):
                g = rng.integers(lo, hi, size=(H, W), dtype=np.uint8)
                Image.fromarray(g, mode="L").save(d / f"v{v}.png")               # 8-bit grey; the dataset opens it as RGB
                files.append(f"v{v}.png")
            rows.append({"patient_id": pid, "image": str(files), "text": str(["no mass .", "benign calcification ."])})
        ds = object.__new__(ImageTextDataset)
        ds.df = pd.DataFrame(rows)
        ds.root_dir, ds.img_dir, ds.dataset = Path(td), "img", "upmc"
        ds.split, ds.tfms, ds.mean, ds.std = "valid", None, mean, std
        ds.image_encoder_type = "tf_efficientnet_b5_ns-detect"
        ds.image_aug_other_image = ds.image_view_aug = True
        ds.has_backtranslated = False
        ds.text_max_length = 8
        ds.tokenizer = lambda texts, **kw: {"input_ids": torch.zeros(len(texts), kw["max_length"], dtype=torch.long)}
        import random
        random.seed(0); np.random.seed(0)
        items = [ds[i] for i in range(len(rows))]
        batch = ds.collate_fn(items)
        assert batch["images"].shape == (len(rows), 1, H, W, 3) and batch["images"].dtype == torch.float32
        for key in ("images", "image_views"):
            x = batch[key].squeeze(1).permute(0, 3, 1, 2)                        # trainer_ddp.py:288-291
            store["out/" + key] = np_(x.contiguous())
        raw = np.stack([np.stack([np.array(Image.open(Path(td) / "img" / str(pid) / f"v{v}.png").convert("RGB"))
                                  for pid in range(len(rows))]) for v in range(2)])   # [2 views, b, H, W, 3] uint8
        store["raw/images"], store["raw/image_views"] = raw[0], raw[1]
    np.savez_compressed(os.path.join(HERE, "input_pipeline.npz"), **store)
    print("input_pipeline.npz", {k: v.shape for k, v in store.items()})


def time_reference_cfg1(warmup=3, reps=10):
    """the REFERENCE's own training step (its model, loss, AdamW, scheduler, call order of
    trainer_ddp.py:279-308) at config #1 (B2 + BERT-base, b=4, 224^2, T=64, fp32 CPU) on this container's cores:
    >= 3 warm-up steps, median of >= 10.  Dropout / drop-connect stay ON (it is a timing run)."""
    import time
    from breastclip.model import build_model
    from breastclip.loss import build_loss
    from breastclip.optimizer import build_optimizer
    from breastclip.scheduler import build_scheduler
    from oracle import weights as ow
    model_cfg = {"name": "clip_custom", "temperature": 0.07,
                 "image_encoder": {"source": "cnn", "name": "tf_efficientnetv2-detect", "pretrained": True, "model_type": "cnn"},
                 "text_encoder": {"source": "huggingface", "name": "emilyalsentzer/Bio_ClinicalBERT",
                                  "pretrained": False, "gradient_checkpointing": False, "pooling": "eos",
                                  "cache_dir": "/tmp/none", "trust_remote_code": True, "mlm_head": True},
                 "projection_head": {"name": "linear", "dropout": 0.1, "proj_dim": 512}}
    loss_cfg = {"breast_clip": dict(label_smoothing=0.0, i2i_weight=1.0, t2t_weight=0.5, loss_ratio=1.0)}
    torch.manual_seed(10)
    model = build_model(model_cfg, loss_cfg, types.SimpleNamespace(vocab_size=28996))
    lf = build_loss(loss_cfg)
    opt = build_optimizer(model, {"name": "adamw", "config": {"lr": 5e-5, "weight_decay": 1e-4}})
    sch = build_scheduler(opt, {"name": "cosine", "config": {"total_steps": 10000, "warmup_steps": 100}})
    b, H, W, T = 4, 224, 224, 64
    batch = ow.synth_batch(b, H, W, T, seed=10)

    class BE(dict):
        def to(self, device):
            return self

    model.train()
    times = []
    for it in range(warmup + reps):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        bt = {"images": batch["images"], "image_views": batch["image_views"],
              "text_tokens": BE(batch["text_tokens"]), "text_tokens2": BE(batch["text_tokens2"])}
        out = model(bt, torch.device("cpu"))
        ld = lf(**out, is_train=True)
        ld["total"].backward()
        opt.step()
        sch.step()
        dt = time.perf_counter() - t0
        if it >= warmup:
            times.append(dt)
    med = float(np.median(times))
    res = {"what": "reference breastclip training step (fwd + breast_clip loss + bwd + AdamW + scheduler), its own code, CPU fp32",
           "config": "BASELINE configs[0]: EfficientNet-B2 + BERT-base(BioClinicalBERT shape), batch 4, 224x224, 64 tokens, "
                     "2 views + 2 texts per pair",
           "warmup_steps": warmup, "timed_steps": reps, "median_s_per_step": round(med, 4),
           "min_s_per_step": round(float(min(times)), 4), "max_s_per_step": round(float(max(times)), 4),
           "pairs_per_s": round(b / med, 4), "threads": torch.get_num_threads(), "host_cores": os.cpu_count(),
           "torch": torch.__version__, "where": "build container (no GPU); the reference never runs on the GPU box"}
    json.dump(res, open(os.path.join(HERE, "ref_cpu_timing.json"), "w"), indent=1)
    print("ref_cpu_timing.json", res)


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["arch", "mbconv", "bert", "loss", "e2e_b2", "e2e_seeds", "e2e_b5", "traj", "e2e_bn8k", "inputs", "timing"]
    import_reference()
    if "arch" in which:
        gen_arch_tables()
    if "mbconv" in which:
        gen_mbconv_kats()
    if "bert" in which:
        gen_bert_kat()
    if "loss" in which:
        gen_loss_kats()
    if "e2e_b2" in which:
        gen_e2e("e2e_b2_cfg1", "tf_efficientnetv2-detect", "efficientnet-b2", 4, 224, 224, 64)
    if "e2e_seeds" in which:
        gen_e2e_eval_seeds()
    if "traj" in which:
        gen_traj()
    if "e2e_b5" in which:
        gen_e2e("e2e_b5_small", "tf_efficientnet_b5_ns-detect", "efficientnet-b5", 2, 160, 96, 32)
    if "e2e_bn8k" in which:
        gen_e2e("e2e_b2_bn8k", "tf_efficientnetv2-detect", "efficientnet-b2", 8, 456, 456, 64)
    if "inputs" in which:
        gen_input_pipeline()
    if "timing" in which:
        time_reference_cfg1()
