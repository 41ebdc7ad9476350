"""Deterministic synthetic weights + inputs, keyed by state_dict name (oracle side).

There are no pretrained checkpoints in the build container or on the GPU box (no network), so parity
is asserted on random-init networks.  Weights are generated from a counter-free recipe
``torch.Generator(seed ^ crc32(name))`` so the SAME tensors can be rebuilt anywhere (golden
generation against the reference here, parity tests on the GPU box) without committing 100+ M floats.
The key inventory reproduces ``BreastClip(...).state_dict()`` of the reference (SURVEY.md section 8b:
710 entries for the B2 model), checked by ``load_state_dict(strict=True)`` in make_golden.py.
"""
import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch

from .arch import Arch
from .bert import BertShape


def _bn_keys(d, p, c):
    d[p + ".weight"] = (c,)
    d[p + ".bias"] = (c,)
    d[p + ".running_mean"] = (c,)
    d[p + ".running_var"] = (c,)
    d[p + ".num_batches_tracked"] = ()


def efficientnet_shapes(arch: Arch, prefix: str = "") -> "OrderedDict[str, Tuple[int, ...]]":
    d = OrderedDict()
    p = prefix
    d[p + "_conv_stem.weight"] = (arch.stem_out, 3, 3, 3)
    _bn_keys(d, p + "_bn0", arch.stem_out)
    for b in arch.blocks:
        q = f"{p}_blocks.{b.idx}"
        if b.expand != 1:
            d[q + "._expand_conv.weight"] = (b.cexp, b.cin, 1, 1)
            _bn_keys(d, q + "._bn0", b.cexp)
        d[q + "._depthwise_conv.weight"] = (b.cexp, 1, b.k, b.k)
        _bn_keys(d, q + "._bn1", b.cexp)
        d[q + "._se_reduce.weight"] = (b.cse, b.cexp, 1, 1)
        d[q + "._se_reduce.bias"] = (b.cse,)
        d[q + "._se_expand.weight"] = (b.cexp, b.cse, 1, 1)
        d[q + "._se_expand.bias"] = (b.cexp,)
        d[q + "._project_conv.weight"] = (b.cout, b.cexp, 1, 1)
        _bn_keys(d, q + "._bn2", b.cout)
    d[p + "_conv_head.weight"] = (arch.head_out, arch.head_in, 1, 1)
    _bn_keys(d, p + "_bn1", arch.head_out)
    return d


def bert_shapes(cfg: BertShape, prefix: str = "") -> "OrderedDict[str, Tuple[int, ...]]":
    d = OrderedDict()
    p, H = prefix, cfg.hidden
    d[p + "embeddings.word_embeddings.weight"] = (cfg.vocab, H)
    d[p + "embeddings.position_embeddings.weight"] = (cfg.max_pos, H)
    d[p + "embeddings.token_type_embeddings.weight"] = (cfg.type_vocab, H)
    d[p + "embeddings.LayerNorm.weight"] = (H,)
    d[p + "embeddings.LayerNorm.bias"] = (H,)
    for i in range(cfg.layers):
        q = f"{p}encoder.layer.{i}."
        for n in ("attention.self.query", "attention.self.key", "attention.self.value",
                  "attention.output.dense"):
            d[q + n + ".weight"] = (H, H)
            d[q + n + ".bias"] = (H,)
        d[q + "attention.output.LayerNorm.weight"] = (H,)
        d[q + "attention.output.LayerNorm.bias"] = (H,)
        d[q + "intermediate.dense.weight"] = (cfg.inter, H)
        d[q + "intermediate.dense.bias"] = (cfg.inter,)
        d[q + "output.dense.weight"] = (H, cfg.inter)
        d[q + "output.dense.bias"] = (H,)
        d[q + "output.LayerNorm.weight"] = (H,)
        d[q + "output.LayerNorm.bias"] = (H,)
    d[p + "pooler.dense.weight"] = (H, H)
    d[p + "pooler.dense.bias"] = (H,)
    return d


def clip_shapes(arch: Arch, cfg: BertShape, proj_dim: int = 512):
    """Key order of BreastClip.state_dict() (clip.py:15-44): logit_scale is a Parameter registered
    last in __init__ but nn.Module lists direct parameters before sub-modules, so it comes first."""
    d = OrderedDict()
    d["logit_scale"] = ()
    d.update(efficientnet_shapes(arch, "image_encoder."))
    d.update(bert_shapes(cfg, "text_encoder.text_encoder."))
    d["image_projection.projection.weight"] = (proj_dim, arch.head_out)
    d["image_projection.projection.bias"] = (proj_dim,)
    d["text_projection.projection.weight"] = (proj_dim, cfg.hidden)
    d["text_projection.projection.bias"] = (proj_dim,)
    return d


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator()
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
    return g


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int = 10) -> torch.Tensor:
    """One deterministic fp32 tensor; distribution chosen per role so activations stay O(1)."""
    g = _gen(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    if name == "logit_scale":
        return torch.tensor(2.6592600369327783)            # log(1/0.07), clip.py:39-41
    if leaf == "num_batches_tracked":
        return torch.tensor(0, dtype=torch.int64)
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if leaf == "running_var":
        return 0.5 + torch.rand(shape, generator=g)
    norm_like = ("_bn" in name) or ("LayerNorm" in name)
    if leaf == "weight" and norm_like:
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if leaf == "bias":
        return 0.1 * torch.randn(shape, generator=g)
    if "embeddings" in name:
        return 0.05 * torch.randn(shape, generator=g)
    # conv / linear weight: He-style fan-in scaling
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return torch.randn(shape, generator=g) * (1.5 / max(fan_in, 1)) ** 0.5


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 10) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k, synth_tensor(k, tuple(v), seed)) for k, v in shapes.items())


def synth_batch(b: int, h: int, w: int, T: int, vocab: int = 28996, seed: int = 10,
                full_length: bool = False, two_views: bool = True) -> Dict:
    """Synthetic batch in the layout the reference's model receives:
    images NCHW fp32 ~ N(0,1); tokens [CLS]=101 ... [SEP]=102 then zero padding, len ~ U[8,T]."""
    g = _gen(f"batch{b}x{h}x{w}x{T}", seed)

    def toks(tag):
        gg = _gen(f"tok{tag}{b}x{T}", seed)
        ids = torch.randint(1000, vocab, (b, T), generator=gg)
        lens = torch.full((b,), T) if full_length else torch.randint(min(8, T), T + 1, (b,), generator=gg)
        mask = (torch.arange(T)[None, :] < lens[:, None]).long()
        ids[:, 0] = 101
        ids[torch.arange(b), lens - 1] = 102
        ids = ids * mask
        return {"input_ids": ids, "token_type_ids": torch.zeros_like(ids), "attention_mask": mask}

    batch = {"images": torch.randn(b, 3, h, w, generator=g), "text_tokens": toks("a")}
    if two_views:
        batch["image_views"] = torch.randn(b, 3, h, w, generator=g)
        batch["text_tokens2"] = toks("b")
    return batch
