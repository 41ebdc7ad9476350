"""``torch.library`` registration of the C-ABI kernels ("exposed as custom ops").

The model in ``breastclip/`` is one ``torch.autograd.Function`` per stem / MBConv block / BERT layer with a hand-derived
backward; INSIDE those functions every 1x1-convolution / linear and every depthwise-convolution launch goes through the
dispatcher-visible operators registered here (``ops.linear_fwd`` / ``linear_dgrad`` / ``linear_wgrad`` / ``dwconv_fwd`` /
``dwconv_bwd_data`` / ``dwconv_bwd_weight`` are thin wrappers over them):

    torch.ops.mammoclip.conv1x1(x, w, bias?, residual?, pro_scale?, pro_shift?, pro_gate?, rows_per_img, stats, has_pro)
                                  -> (y, BatchNorm column-statistic partials)      forward, fused BN+SiLU(+SE gate) prologue
    torch.ops.mammoclip.conv1x1_dgrad(dy, w, residual?, w_t?) -> dx
    torch.ops.mammoclip.conv1x1_wgrad(dy, x, pro_scale?, pro_shift?, pro_gate?, rows_per_img, has_pro) -> dw (fp32)
    torch.ops.mammoclip.dwconv_bn(x, w_kkc, n, h, w, k, stride, pad_l, pad_t, oh, ow, pro_scale?, pro_shift?, stats) -> (y, partials)
    torch.ops.mammoclip.dwconv_dgrad(dy, w_kkc, n, h, w, k, stride, pad_l, pad_t, oh, ow, w_kkc_flipped?) -> dx
    torch.ops.mammoclip.dwconv_wgrad(x, dy, n, h, w, k, stride, pad_l, pad_t, oh, ow, pro_scale?, pro_shift?) -> dw (fp32)

(registered with ``torch.library.Library.define`` / ``impl``: ~2.5 us of dispatcher per call; they are called under the
functions' own backward, so they carry no autograd formula of their own).  A second, self-contained set is differentiable
on its own and usable outside the model:

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
    return ops._linear_fwd_impl(x, w, bias=bias, residual=residual)


@linear.register_fake
def _(x, w, bias=None, residual=None):
    return x.new_empty((x.shape[0], w.shape[0]))


@_op("linear_dgrad")
def linear_dgrad(dy: Tensor, w: Tensor) -> Tensor:
    return ops._linear_dgrad_impl(dy, w)


@linear_dgrad.register_fake
def _(dy, w):
    return dy.new_empty((dy.shape[0], w.shape[1]))


@_op("linear_wgrad")
def linear_wgrad(dy: Tensor, x: Tensor) -> Tensor:
    return ops._linear_wgrad_impl(dy, x)


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
    return ops._dwconv_fwd_impl(x, w_kkc, n, h, w, x.shape[1], k, stride, pad_l, pad_t, oh, ow)


@dwconv.register_fake
def _(x, w_kkc, n, h, w, k, stride, pad_l, pad_t, oh, ow):
    return x.new_empty((n * oh * ow, x.shape[1]))


@_op("dwconv_bwd_data")
def dwconv_bwd_data(dy: Tensor, w_kkc: Tensor, n: int, h: int, w: int, k: int, stride: int, pad_l: int, pad_t: int,
                    oh: int, ow: int) -> Tensor:
    flipped = w_kkc.flip(0).contiguous() if stride == 1 else None
    return ops._dwconv_bwd_data_impl(dy, w_kkc, n, h, w, dy.shape[1], k, stride, pad_l, pad_t, oh, ow, w_kkc_flipped=flipped)


@dwconv_bwd_data.register_fake
def _(dy, w_kkc, n, h, w, k, stride, pad_l, pad_t, oh, ow):
    return dy.new_empty((n * h * w, dy.shape[1]))


@_op("dwconv_bwd_weight")
def dwconv_bwd_weight(x: Tensor, dy: Tensor, n: int, h: int, w: int, k: int, stride: int, pad_l: int, pad_t: int,
                      oh: int, ow: int) -> Tensor:
    return ops._dwconv_bwd_weight_impl(x, dy, n, h, w, x.shape[1], k, stride, pad_l, pad_t, oh, ow)


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
    return scores.new_empty(scores.shape, dtype=ops.BF16)


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


# ------------------------------------------------------------------------------------------------ the model's operators
_EMPTY = {}


def _empty_f32(like):
    key = like.device
    t = _EMPTY.get(key)
    if t is None:
        t = _EMPTY[key] = torch.empty(0, dtype=torch.float32, device=like.device)
    return t


def _pro4(ps, pf, pg, rpi, has_pro):
    return (ps, pf, pg, rpi) if has_pro else None


def _conv1x1(x, w, bias, residual, ps, pf, pg, rpi, stats, has_pro):
    r = ops._linear_fwd_impl(x, w, bias=bias, residual=residual, stats=stats, pro=_pro4(ps, pf, pg, rpi, has_pro))
    return r if stats else (r, _empty_f32(x))


def _conv1x1_meta(x, w, bias, residual, ps, pf, pg, rpi, stats, has_pro):
    return x.new_empty((x.shape[0], w.shape[0])), x.new_empty((1 if stats else 0, 2, w.shape[0]), dtype=torch.float32)


def _conv1x1_dgrad(dy, w, residual, w_t):
    return ops._linear_dgrad_impl(dy, w, residual=residual, w_t=w_t)


def _conv1x1_wgrad(dy, x, ps, pf, pg, rpi, has_pro):
    return ops._linear_wgrad_impl(dy, x, pro=_pro4(ps, pf, pg, rpi, has_pro))


def _dwconv_bn(x, w_kkc, n, h, w, k, stride, pad_l, pad_t, oh, ow, ps, pf, stats):
    r = ops._dwconv_fwd_impl(x, w_kkc, n, h, w, x.shape[1], k, stride, pad_l, pad_t, oh, ow,
                             pro=(ps, pf) if ps is not None else None, stats=stats)
    return r if stats else (r, _empty_f32(x))


def _dwconv_dgrad(dy, w_kkc, n, h, w, k, stride, pad_l, pad_t, oh, ow, w_flip):
    return ops._dwconv_bwd_data_impl(dy, w_kkc, n, h, w, dy.shape[1], k, stride, pad_l, pad_t, oh, ow, w_kkc_flipped=w_flip)


def _dwconv_wgrad(x, dy, n, h, w, k, stride, pad_l, pad_t, oh, ow, ps, pf):
    return ops._dwconv_bwd_weight_impl(x, dy, n, h, w, x.shape[1], k, stride, pad_l, pad_t, oh, ow,
                                       pro=(ps, pf) if ps is not None else None)


_GEO = "int n, int h, int w, int k, int stride, int pad_l, int pad_t, int oh, int ow"
_MODEL_OPS = {
    "conv1x1": ("(Tensor x, Tensor w, Tensor? bias, Tensor? residual, Tensor? pro_scale, Tensor? pro_shift, Tensor? pro_gate, "
                "int rows_per_img, bool stats, bool has_pro) -> (Tensor, Tensor)", _conv1x1, _conv1x1_meta),
    "conv1x1_dgrad": ("(Tensor dy, Tensor w, Tensor? residual, Tensor? w_t) -> Tensor", _conv1x1_dgrad,
                      lambda dy, w, residual, w_t: dy.new_empty((dy.shape[0], w.shape[1]))),
    "conv1x1_wgrad": ("(Tensor dy, Tensor x, Tensor? pro_scale, Tensor? pro_shift, Tensor? pro_gate, int rows_per_img, bool has_pro) "
                      "-> Tensor", _conv1x1_wgrad,
                      lambda dy, x, ps, pf, pg, rpi, hp: dy.new_empty((dy.shape[1], x.shape[1]), dtype=torch.float32)),
    "dwconv_bn": (f"(Tensor x, Tensor w_kkc, {_GEO}, Tensor? pro_scale, Tensor? pro_shift, bool stats) -> (Tensor, Tensor)", _dwconv_bn,
                  lambda x, wk, n, h, w, k, s, pl, pt, oh, ow, ps, pf, st: (
                      x.new_empty((n * oh * ow, x.shape[1])), x.new_empty((1 if st else 0, 2, x.shape[1]), dtype=torch.float32))),
    "dwconv_dgrad": (f"(Tensor dy, Tensor w_kkc, {_GEO}, Tensor? w_kkc_flipped) -> Tensor", _dwconv_dgrad,
                     lambda dy, wk, n, h, w, k, s, pl, pt, oh, ow, wf: dy.new_empty((n * h * w, dy.shape[1]))),
    "dwconv_wgrad": (f"(Tensor x, Tensor dy, {_GEO}, Tensor? pro_scale, Tensor? pro_shift) -> Tensor", _dwconv_wgrad,
                     lambda x, dy, n, h, w, k, s, pl, pt, oh, ow, ps, pf: x.new_empty((k * k, x.shape[1]), dtype=torch.float32)),
}
for _name, (_schema, _impl, _meta) in _MODEL_OPS.items():
    _lib.define(_name + _schema)
    _lib.impl(_name, _impl, "CUDA")
    _lib.impl(_name, _meta, "Meta")

OPS = ("linear", "linear_dgrad", "linear_wgrad", "dwconv", "dwconv_bwd_data", "dwconv_bwd_weight", "gelu", "gelu_bwd",
       "softmax", "l2norm", "cross_entropy_") + tuple(_MODEL_OPS)

ops._bind_model_ops()      # ops' wrappers resolve the overloads once (either module may be imported first)
