"""EfficientNet image encoder, functional torch-fp32 restatement (oracle side, CPU).

Follows model/modules/efficientnet_custom.py: MBConvBlock.forward :91-132, EfficientNet.extract_features
:262-285, EfficientNet.forward :287-313.  Parameters come from a ``state_dict``-style mapping with the
reference's key names (``_conv_stem.weight``, ``_blocks.{i}._expand_conv.weight`` ...), so golden
weights load into the reference with ``strict=True`` and into this function unchanged.

Stochastic ops (drop_connect efficient_net_custom_utils.py:129-154, pooled-feature dropout
efficientnet_custom.py:310-312) are only reproducible with p = 0 or in eval mode:
``drop_connect`` / ``dropout`` must be 0.0 when ``train`` is True.
"""
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .arch import Arch, BN_EPS, BN_MOMENTUM

# Storage-rounding emulation (tests / scripts/rounding_ablation.py only): a set of tensor-class tags whose members are
# rounded to ``ROUND_DTYPE`` and back where the HIP path STORES (or feeds an MFMA with) that class -- "E" expand-conv
# output, "D" depthwise output, "P" project-conv output, "Y" block output / residual stream (also the stem output),
# "A1" the gated activation operand of the project GEMM, "W" the 1x1 / stem convolution weights, "H" the head-conv
# output, "IN" the stem's input patches, "A0" the activated depthwise input silu(bn0(e)) (rounded when it is staged in LDS),
# "DW" the depthwise filter taps (bf16 operands of the packed dot-product form of the depthwise kernels).  None = the plain fp32 restatement.  Straight-through in backward (the rounding has zero-width gradient
# support otherwise).
ROUND = None
ROUND_DTYPE = torch.bfloat16
# BASELINE config #5 emulation: {"expand": {block idx}, "project": {block idx}, "head": bool} -- the 1x1 convolutions whose
# operands the HIP path quantises to per-tensor-scaled OCP e4m3 (mammo_clip_amd ... set_fp8); None = off.
FP8 = None


def _q8(x):
    """per-tensor-scaled e4m3 quantise / dequantise (scale 448 / amax), the arithmetic of csrc/fp8.hip"""
    amax = x.detach().abs().max().clamp_min(1e-30)
    xb = x.to(torch.bfloat16).float()                      # the HIP path quantises the stored bf16 tensor
    q = (xb * (448.0 / amax)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * (amax / 448.0)
    return x + (q - x).detach()


ROUND_SR = None     # torch.Generator: STOCHASTIC rounding to bf16 instead of round-to-nearest (independent draws of the same noise)


def _q(tag, x):
    if ROUND is None or tag not in ROUND:
        return x
    if ROUND_SR is not None and ROUND_DTYPE == torch.bfloat16 and x.dtype == torch.float32:
        # unbiased random rounding: add uniform noise below the bf16 ulp, truncate the low 16 bits
        bits = x.detach().contiguous().view(torch.int32)
        noise = torch.randint(0, 1 << 16, bits.shape, generator=ROUND_SR, device=bits.device, dtype=torch.int32)
        r = ((bits + noise) & -65536).view(torch.float32)
        return x + (r - x).detach()
    return x + (x.to(ROUND_DTYPE).to(x.dtype) - x).detach()


def swish(x):
    """efficient_net_custom_utils.py:64-69 (forward of SwishImplementation)."""
    return x * torch.sigmoid(x)


def _bn(sd: Dict[str, torch.Tensor], p: str, x, train: bool, new_buffers: Optional[dict]):
    """nn.BatchNorm2d(momentum=0.01, eps=1e-3): batch statistics in train mode, running stats in eval."""
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    if train:
        rm, rv = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], True, BN_MOMENTUM, BN_EPS)
        if new_buffers is not None:
            new_buffers[p + ".running_mean"] = rm
            new_buffers[p + ".running_var"] = rv
        return y
    return F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], False, BN_MOMENTUM, BN_EPS)


def _conv_static_same(x, w, bias, stride, pad, groups=1):
    """Conv2dStaticSamePadding.forward, efficient_net_custom_utils.py:273-276:
    ZeroPad2d((l, r, t, b)) then conv2d(padding=0)."""
    if any(pad):
        x = F.pad(x, pad)
    return F.conv2d(x, w, bias, stride, 0, 1, groups)


def mbconv(sd, p: str, x, blk, train: bool, new_buffers=None, taps: Optional[dict] = None):
    """MBConvBlock.forward (efficientnet_custom.py:91-132) with drop_connect disabled."""
    inp = x
    if blk.expand != 1:
        if FP8 is not None and blk.idx in FP8["expand"]:
            x = _q("E", F.conv2d(_q8(x), _q8(sd[p + "._expand_conv.weight"])))
        else:
            x = _q("E", F.conv2d(x, _q("W", sd[p + "._expand_conv.weight"])))
        if taps is not None:
            taps[p + ".expand_out"] = x
        x = swish(_bn(sd, p + "._bn0", x, train, new_buffers))
    x = _q("D", _conv_static_same(_q("A0", x), _q("DW", sd[p + "._depthwise_conv.weight"]), None, blk.s, blk.pad, groups=blk.cexp))
    if taps is not None:
        taps[p + ".dw_out"] = x
    x = swish(_bn(sd, p + "._bn1", x, train, new_buffers))
    # squeeze & excite (:114-119)
    sq = F.adaptive_avg_pool2d(x, 1)
    sq = F.conv2d(sq, sd[p + "._se_reduce.weight"], sd[p + "._se_reduce.bias"])
    sq = swish(sq)
    sq = F.conv2d(sq, sd[p + "._se_expand.weight"], sd[p + "._se_expand.bias"])
    if FP8 is not None and blk.idx in FP8["project"]:
        # the HIP path folds the SE gate into per-image copies of the weight and quantises (activation, gated weight)
        gate = torch.sigmoid(sq)
        wq = torch.stack([_q8((sd[p + "._project_conv.weight"] * gate[i][None]).to(torch.bfloat16).float()) for i in range(x.shape[0])])
        xq = _q8(x.to(torch.bfloat16).float())
        x = _q("P", torch.cat([F.conv2d(xq[i:i + 1], wq[i]) for i in range(x.shape[0])]))
    else:
        x = _q("A1", torch.sigmoid(sq) * x)
        x = _q("P", F.conv2d(x, _q("W", sd[p + "._project_conv.weight"])))
    if taps is not None:
        taps[p + ".project_out"] = x
    x = _bn(sd, p + "._bn2", x, train, new_buffers)
    if blk.skip:
        x = x + inp
    return _q("Y", x)


def extract_features(sd, x, arch: Arch, train: bool, prefix: str = "", new_buffers=None,
                     taps: Optional[dict] = None):
    """EfficientNet.extract_features (efficientnet_custom.py:262-285)."""
    p = prefix
    x = _q("E", _conv_static_same(_q("IN", x), _q("W", sd[p + "_conv_stem.weight"]), None, 2, arch.stem_pad))
    x = _q("Y", swish(_bn(sd, p + "_bn0", x, train, new_buffers)))
    if taps is not None:
        taps["stem"] = x
    for blk in arch.blocks:
        x = mbconv(sd, f"{p}_blocks.{blk.idx}", x, blk, train, new_buffers, taps)
        if taps is not None:
            taps[f"block{blk.idx}"] = x
    if FP8 is not None and FP8["head"]:
        x = _q("H", F.conv2d(_q8(x), _q8(sd[p + "_conv_head.weight"])))
    else:
        x = _q("H", F.conv2d(x, _q("W", sd[p + "_conv_head.weight"])))
    x = swish(_bn(sd, p + "_bn1", x, train, new_buffers))
    return x


def forward(sd, x, arch: Arch, train: bool, prefix: str = "", new_buffers=None, taps=None,
            return_map: bool = False):
    """EfficientNet.forward (efficientnet_custom.py:287-313): features -> global avg pool -> flatten
    (-> dropout, identity here).  Returns [b, head_out]."""
    fmap = extract_features(sd, x, arch, train, prefix, new_buffers, taps)
    pooled = F.adaptive_avg_pool2d(fmap, 1).flatten(1)
    return (pooled, fmap) if return_map else pooled
