"""``torch.library`` registration of the C-ABI kernels ("exposed as custom ops").

The model in ``breastclip/`` drives the kernels through one ``torch.autograd.Function`` per stem / MBConv block / BERT
layer (ops.py wrappers, many launches per node, hand-derived backward).  The same kernels are ALSO registered here as
dispatcher-visible operators under the ``mammoclip::`` namespace, so they show up in ``torch.ops``, in the profiler, and
can be traced / captured like any other operator:

    torch.ops.mammoclip.linear(x, w, bias, residual)          y = x . w^T (+bias)(+residual)    bf16, differentiable
    torch.ops.mammoclip.dwconv(x, w_kkc, n, h, w, k, stride, pad_l, pad_t, oh, ow)   depthwise conv, differentiable
    torch.ops.mammoclip.linear_dgrad / linear_wgrad / dwconv_bwd_data / dwconv_bwd_weight      the explicit backward ops
    torch.ops.mammoclip.gelu / gelu_bwd / softmax / l2norm / cross_entropy_ (in place)         pointwise / row kernels

Every operator has a fake (meta) implementation, so ``torch.library.opcheck`` and tracing work without a GPU; the real
implementations exist for device type "cuda" only -- a CPU tensor raises (there is no fallback).
Importing this module performs the registration (``import mammo_clip_amd.custom_ops``)."""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops

_lib = torch.library.Library("mammoclip", "FRAGMENT")        # keeps the namespace alive for the process


def _op(name, mutates=()):
    return torch.library.custom_op(f"mammoclip::{name}", mutates_args=mutates, device_types="cuda")


# ------------------------------------------------------------------------------------------------ 1x1 conv / linear
@_op("linear")
def linear(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, residual: Optional[Tensor] = None) -> Tensor:
    return ops.linear_fwd(x, w, bias=bias, residual=residual)


@linear.register_fake
def _(x, w, bias=None, residual=None):
    return x.new_empty((x.shape[0], w.shape[0]))


@_op("linear_dgrad")
def linear_dgrad(dy: Tensor, w: Tensor) -> Tensor:
    return ops.linear_dgrad(dy, w)


@linear_dgrad.register_fake
def _(dy, w):
    return dy.new_empty((dy.shape[0], w.shape[1]))


@_op("linear_wgrad")
def linear_wgrad(dy: Tensor, x: Tensor) -> Tensor:
    return ops.linear_wgrad(dy, x)


@linear_wgrad.register_fake
def _(dy, x):
    return dy.new_empty((dy.shape[1], x.shape[1]), dtype=torch.float32)


def _linear_setup(ctx, inputs, output):
    x, w, bias, residual = inputs
    ctx.save_for_backward(x, w)
    ctx.has_bias, ctx.has_res = bias is not None, residual is not None


def _linear_backward(ctx, dy):
    x, w = ctx.saved_tensors
    dy = dy.contiguous()
    dx = torch.ops.mammoclip.linear_dgrad(dy, w)
    dw = torch.ops.mammoclip.linear_wgrad(dy, x).to(w.dtype)
    db = ops.colsum(dy) if ctx.has_bias else None
    return dx, dw, db, (dy if ctx.has_res else None)


linear.register_autograd(_linear_backward, setup_context=_linear_setup)


# ------------------------------------------------------------------------------------------------ depthwise conv
@_op("dwconv")
def dwconv(x: Tensor, w_kkc: Tensor, n: int, h: int, w: int, k: int, stride: int, pad_l: int, pad_t: int, oh: int,
           ow: int) -> Tensor:
    return ops.dwconv_fwd(x, w_kkc, n, h, w, x.shape[1], k, stride, pad_l, pad_t, oh, ow)


@dwconv.register_fake
def _(x, w_kkc, n, h, w, k, stride, pad_l, pad_t, oh, ow):
    return x.new_empty((n * oh * ow, x.shape[1]))


@_op("dwconv_bwd_data")
def dwconv_bwd_data(dy: Tensor, w_kkc: Tensor, n: int, h: int, w: int, k: int, stride: int, pad_l: int, pad_t: int,
                    oh: int, ow: int) -> Tensor:
    flipped = w_kkc.flip(0).contiguous() if stride == 1 else None
    return ops.dwconv_bwd_data(dy, w_kkc, n, h, w, dy.shape[1], k, stride, pad_l, pad_t, oh, ow, w_kkc_flipped=flipped)


@dwconv_bwd_data.register_fake
def _(dy, w_kkc, n, h, w, k, stride, pad_l, pad_t, oh, ow):
    return dy.new_empty((n * h * w, dy.shape[1]))


@_op("dwconv_bwd_weight")
def dwconv_bwd_weight(x: Tensor, dy: Tensor, n: int, h: int, w: int, k: int, stride: int, pad_l: int, pad_t: int,
                      oh: int, ow: int) -> Tensor:
    return ops.dwconv_bwd_weight(x, dy, n, h, w, x.shape[1], k, stride, pad_l, pad_t, oh, ow)


@dwconv_bwd_weight.register_fake
def _(x, dy, n, h, w, k, stride, pad_l, pad_t, oh, ow):
    return x.new_empty((k * k, x.shape[1]), dtype=torch.float32)


def _dw_setup(ctx, inputs, output):
    x, w_kkc, *geo = inputs
    ctx.save_for_backward(x, w_kkc)
    ctx.geo = tuple(geo)


def _dw_backward(ctx, dy):
    x, w_kkc = ctx.saved_tensors
    dy = dy.contiguous()
    dx = torch.ops.mammoclip.dwconv_bwd_data(dy, w_kkc, *ctx.geo)
    dw = torch.ops.mammoclip.dwconv_bwd_weight(x, dy, *ctx.geo)
    return (dx, dw) + (None,) * 9


dwconv.register_autograd(_dw_backward, setup_context=_dw_setup)


# ------------------------------------------------------------------------------------------------ row / pointwise kernels
@_op("gelu")
def gelu(x: Tensor) -> Tensor:
    return ops.gelu_fwd(x)


@gelu.register_fake
def _(x):
    return torch.empty_like(x)


@_op("gelu_bwd")
def gelu_bwd(dy: Tensor, x: Tensor) -> Tensor:
    return ops.gelu_bwd(dy, x)


@gelu_bwd.register_fake
def _(dy, x):
    return torch.empty_like(x)


gelu.register_autograd(lambda ctx, dy: torch.ops.mammoclip.gelu_bwd(dy.contiguous(), ctx.saved_tensors[0]),
                       setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0]))


@_op("softmax")
def softmax(scores: Tensor) -> Tensor:
    return ops.softmax_fwd(scores, 0.0, 0, 0)[0]


@softmax.register_fake
def _(scores):
    return scores.new_empty(scores.shape, dtype=torch.bfloat16)


@_op("l2norm")
def l2norm(x: Tensor) -> Tuple[Tensor, Tensor]:
    return ops.l2norm_fwd(x)


@l2norm.register_fake
def _(x):
    return torch.empty_like(x), x.new_empty((x.shape[0],))


@_op("cross_entropy_", mutates=("logits", "loss"))
def cross_entropy_(logits: Tensor, loss: Tensor, label_offset: int, weight: float, smoothing: float,
                   labels: Optional[Tensor] = None) -> None:
    """loss[0] += weight * mean CE(logits, labels + label_offset); logits <- d loss / d logits (in place)"""
    ops.ce_fwd_bwd(logits, label_offset, weight, loss, smoothing, labels=labels)


OPS = ("linear", "linear_dgrad", "linear_wgrad", "dwconv", "dwconv_bwd_data", "dwconv_bwd_weight", "gelu", "gelu_bwd",
       "softmax", "l2norm", "cross_entropy_")
