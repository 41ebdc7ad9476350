"""EfficientNet-B2/B5 image encoder on hand-written gfx950 kernels.

Mirrors the public surface of the reference module of the same name (class names ``EfficientNet`` /
``MBConvBlock``, ``from_name`` / ``from_pretrained``, ``extract_features``, ``forward`` incl. the
``{"image": x}`` dict form, and -- crucially -- the ``state_dict`` key layout ``_conv_stem.weight``,
``_bn0.*``, ``_blocks.{i}._expand_conv.weight`` ...; reference: model/modules/efficientnet_custom.py and
efficient_net_custom_utils.py), but nothing below is a translation of it:

  * activations live in HBM as NHWC bf16 ``[n*h*w, c]``; every 1x1 conv is one MFMA GEMM over pixels
  * training-mode BatchNorm statistics come out of the producing kernel's epilogue, the BN+SiLU of the
    expand conv is applied while the depthwise kernel stages its LDS halo tile, the BN+SiLU+SE-gate of the
    depthwise output is applied while the project GEMM stages its A operand: the activated expanded
    tensors never exist in HBM, forward or backward (they are recomputed from the saved conv outputs)
  * one ``torch.autograd.Function`` per stem / MBConv block / head, with a hand-derived backward

Parameters are ordinary fp32 ``nn.Parameter``s held in ``nn.Conv2d`` / ``nn.BatchNorm2d`` containers (their
``forward`` is never called) so optimizers, DDP and ``load_state_dict(strict=True)`` of reference
checkpoints work unchanged.
"""
import math
from collections import namedtuple
from typing import List, Optional

import os

import torch
from torch import nn

from .... import ops

BN_MOMENTUM = 0.01      # 1 - batch_norm_momentum 0.99 [ref: efficientnet_custom.py:53, efficient_net_custom_utils.py:520]
BN_EPS = 1e-3           # [ref: efficient_net_custom_utils.py:521]

VALID_MODELS = tuple(f"efficientnet-b{i}" for i in range(9)) + ("efficientnet-l2",)

# (width, depth, nominal resolution, dropout) [ref: efficient_net_custom_utils.py:457-479]
_COEFFS = {
    "efficientnet-b0": (1.0, 1.0, 224, 0.2), "efficientnet-b1": (1.0, 1.1, 240, 0.2),
    "efficientnet-b2": (1.1, 1.2, 260, 0.3), "efficientnet-b3": (1.2, 1.4, 300, 0.3),
    "efficientnet-b4": (1.4, 1.8, 380, 0.4), "efficientnet-b5": (1.6, 2.2, 456, 0.4),
    "efficientnet-b6": (1.8, 2.6, 528, 0.5), "efficientnet-b7": (2.0, 3.1, 600, 0.5),
    "efficientnet-b8": (2.2, 3.6, 672, 0.5), "efficientnet-l2": (4.3, 5.3, 800, 0.5),
}
# repeats, kernel, stride, expand, in, out  (se_ratio 0.25 everywhere) [ref: efficient_net_custom_utils.py:502-510]
_STAGES = ((1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
           (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320))

GlobalParams = namedtuple("GlobalParams", ["width_coefficient", "depth_coefficient", "image_size", "dropout_rate",
                                           "num_classes", "batch_norm_momentum", "batch_norm_epsilon",
                                           "drop_connect_rate", "depth_divisor", "min_depth", "include_top"])
BlockArgs = namedtuple("BlockArgs", ["num_repeat", "kernel_size", "stride", "expand_ratio", "input_filters",
                                     "output_filters", "se_ratio", "id_skip"])


def _scaled_width(filters, gp):
    if not gp.width_coefficient:
        return filters
    f = filters * gp.width_coefficient
    div = gp.depth_divisor
    out = max(gp.min_depth or div, int(f + div / 2) // div * div)
    return int(out + div) if out < 0.9 * f else int(out)


def _scaled_depth(repeats, gp):
    return int(math.ceil(gp.depth_coefficient * repeats)) if gp.depth_coefficient else repeats


def _static_pad(size_hw, k, s):
    """(left, right, top, bottom) frozen for the NOMINAL feature-map size [ref: efficient_net_custom_utils.py:262-272]."""
    ih, iw = size_hw
    ph = max((math.ceil(ih / s) - 1) * s + k - ih, 0)
    pw = max((math.ceil(iw / s) - 1) * s + k - iw, 0)
    return (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)


def _down(size_hw, s):
    return (int(math.ceil(size_hw[0] / s)), int(math.ceil(size_hw[1] / s)))


def _out_extent(i, lo, hi, k, s):
    return (i + lo + hi - k) // s + 1


class _Seeds:
    """Counter-based RNG bookkeeping: every stochastic site gets (seed, stream id); re-running a forward with
    the same seed regenerates identical dropout / drop-connect masks (used by backward)."""
    base = 0x5EED

    def __init__(self):
        self.seed = _Seeds.base
        self.calls = 0

    def next(self):
        self.calls += 1
        return self.seed * 1000003 + self.calls


# ================================================================================================
#                                      autograd functions
# ================================================================================================
class StatTape:
    """BatchNorm batch statistics and squeeze-excite pooled means of ONE micro-batch's forward, in call order.
    The micro-batched step forwards most micro-batches twice with identical inputs, weights and mask seeds (first
    without a graph for the embeddings, then with a graph for the backward): the second forward computes exactly the
    same statistics again.  A tape RECORDS them in the first forward (a few MB) and the re-forward REPLAYS them: no
    statistics epilogues, no finalize launches, no squeeze pass over the depthwise output of the early stages --
    same values, bit for bit (tested), ~13 ms of a ~90 ms re-forward at 32 x 1520x912."""
    __slots__ = ("mode", "items", "pos")

    def __init__(self):
        self.mode, self.items, self.pos = "record", [], 0

    def replay(self):
        self.mode, self.pos = "replay", 0
        return self

    def put(self, v):
        self.items.append(v)
        return v

    def get(self):
        v = self.items[self.pos]
        self.pos += 1
        return v


_TAPE = None


def set_stat_tape(tape):
    """install (or remove: None) the tape the next encoder forwards record on / replay from (engine._step_micro)"""
    global _TAPE
    _TAPE = tape


def _replaying():
    return _TAPE is not None and _TAPE.mode == "replay"


class _BNDefer:
    """Running-statistic updates of an encoder call that runs on a SIDE stream beside another call of the same encoder
    (model/clip.py: the second image view).  Both calls update the same buffers -- r <- (1 - m) r + m s, view 1 first
    [ref: clip.py:83,108 two sequential encode_image calls] -- so the side call must not touch them: its finalize launches
    update zero-filled scratch slices instead (result: m s), and ``apply`` folds them in after the join, on the main
    stream, behind view 1's updates: r <- (1 - m) r + (m s), the same two-term form (one rounding apart from the fused
    kernel expression)."""

    def __init__(self, enc):
        offs, tot = {}, 0
        for bn in enc._bn_layers:
            offs[id(bn)] = tot
            tot += bn.num_features
        self.offs, self.total = offs, tot
        self.flat = torch.zeros(2 * tot, dtype=torch.float32, device=enc._ones_b.device)
        self.used = []

    def scratch(self, bn):
        o, c = self.offs[id(bn)], bn.num_features
        self.used.append(bn)
        return self.flat[o:o + c], self.flat[self.total + o:self.total + o + c]

    @torch.no_grad()
    def apply(self):
        if not self.used:
            return
        if self.flat.is_cuda:
            self.flat.record_stream(torch.cuda.current_stream(self.flat.device))
        dst = [bn.running_mean for bn in self.used] + [bn.running_var for bn in self.used]
        src = [self.flat[self.offs[id(bn)]:self.offs[id(bn)] + bn.num_features] for bn in self.used] + \
              [self.flat[self.total + self.offs[id(bn)]:self.total + self.offs[id(bn)] + bn.num_features] for bn in self.used]
        torch._foreach_mul_(dst, 1.0 - BN_MOMENTUM)
        torch._foreach_add_(dst, src)
        self.used = []


_BN_DEFER = None


def _bn_stats(partials, count, bn: nn.BatchNorm2d, training: bool):
    if training:
        if _replaying():
            return _TAPE.get()
        if _BN_DEFER is not None and bn.track_update:
            rm, rv = _BN_DEFER.scratch(bn)
        else:
            rm, rv = bn.running_mean, bn.running_var
        st = ops.bn_finalize(partials, count, bn.weight, bn.bias, rm, rv, BN_MOMENTUM, BN_EPS, bn.track_update)
        if bn.track_update:
            if bn.defer_count is not None:
                bn.defer_count.append(bn.num_batches_tracked)     # the encoder bumps all counters with ONE launch
            else:
                bn.num_batches_tracked += 1
        if _TAPE is not None:
            _TAPE.put(st)
        return st
    return ops.bn_eval_coeffs(bn.weight, bn.bias, bn.running_mean, bn.running_var, BN_EPS)


def _conv_stats(fn, training, *a, **kw):
    """run a producer that can leave BatchNorm statistics partials (stats=True -> (y, partials)); a replaying tape needs none"""
    if training and _replaying():
        return fn(*a, **kw), None
    return fn(*a, stats=True, **kw)


# expanded-tensor bytes per call from which the BatchNorm0 backward is folded into the expand conv's gradient GEMMs
# (ops.bn_fold_expand_bwd) instead of running the apply pass; tests set it to 0 to exercise the folded path at small sizes
LINK_STEM = os.environ.get("MC_LINK_STEM", "1") != "0"                  # stem bn0 + swish inside block 0's depthwise kernels (_StemLink)
FUSE_DW_BWD = os.environ.get("MC_FUSE_DW_BWD", "1") != "0"              # stride-1 3x3 depthwise backward as one launch (ops.dwconv_bwd_fused)
FUSE_PROJ_DGRAD = os.environ.get("MC_FUSE_PROJ_DGRAD", "1") != "0"        # projection data gradient with the SE / BatchNorm1 backward in its epilogue (ops.proj_dgrad_*)
BN_FOLD_MIN_BYTES = int(os.environ.get("MC_BN_FOLD_MIN_BYTES", 250_000_000))
BN_FOLD_S2_MIN_BYTES = int(os.environ.get("MC_BN_FOLD_S2_MIN_BYTES", 0))
# round 6: blocks whose expand conv runs INSIDE the depthwise forward launch (ops.mbconv_xdw_fwd) wherever the expanded tensor
# is not stored for a backward (graph-less forwards, recompute modes >= 1, eval): input widths up to this many channels
# (measured per block shape, scripts/xdw_ab.py: the launch wins up to cin = 64; at cin = 128 it only breaks even)
XDW_MAX_CIN = int(os.environ.get("MC_XDW_MAX_CIN", 64))
# recompute mode 3 drops the depthwise output of the stride-1 3x3 blocks up to this many expanded channels (developer sweep)
MODE3_MAX_CEXP = int(os.environ.get("MC_MODE3_MAX_CEXP", 1 << 30))


class _StemFn(torch.autograd.Function):
    """_conv_stem (3x3 s2, static pad) + _bn0 + swish [ref: efficientnet_custom.py:273]."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, mod, link=None):
        """link (round 5): a ``_StemLink`` -- the stem then returns its RAW conv output e and block 0 (no expand conv: its
        depthwise stage reads the stem's output directly) applies bn0 + swish while it stages its LDS tiles, exactly like every
        other block does with its own expand conv; the activated 48-channel 760x456 tensor (the widest map of the network) is
        never written or read.  The backward mirrors it: block 0's depthwise data gradient finishes the bn0 + swish backward in
        its epilogue and leaves the BatchNorm reductions in the link."""
        n, _, h, wd = x.shape
        l, r, t, b = mod.stem_pad
        oh, ow = _out_extent(h, t, b, 3, 2), _out_extent(wd, l, r, 3, 2)
        c0 = w.shape[0]
        patches = ops.stem_im2col(x, l, t, oh, ow)
        wb = ops.stem_weight_prep(w)
        training = mod.training
        e, part = _conv_stats(ops.linear_fwd, training, patches, wb)
        st = _bn_stats(part, n * oh * ow, mod._bn0, training)
        ctx.mod, ctx.st, ctx.geo = mod, st, (n, h, wd, oh, ow, c0)
        ctx.raw = (x.mean, x.std) if isinstance(x, ops.RawImages) else None
        ctx.save_for_backward(x.data if ctx.raw else x, e)
        ctx.link = link
        mod._geo = (n, oh, ow)
        if link is not None:
            link.st, link.part = st, None
            return e.detach()               # (a new tensor object: autograd must not return a saved input as the output)
        return ops.bnact_apply(e, n, oh * ow, c0, st.scale, st.shift, 1)

    @staticmethod
    def backward(ctx, dy):
        x, e = ctx.saved_tensors
        if ctx.raw:
            x = ops.RawImages(x, *ctx.raw)
        n, h, wd, oh, ow, c0 = ctx.geo
        mod = ctx.mod
        link = ctx.link
        if link is not None:
            # dy is already dZ0 = dL/d bn0(e) (block 0's data-gradient epilogue), link.part its BatchNorm-backward reductions
            de, dgamma, dbeta = ops.bnact_bwd(e, n, oh * ow, c0, ctx.st, mod._bn0.weight, 0, g=dy.contiguous(), partials=link.part)
            link.part = None
        else:
            de, dgamma, dbeta = ops.bnact_bwd(e, n, oh * ow, c0, ctx.st, mod._bn0.weight, 1, g=dy.contiguous())
        l, r, t, b = mod.stem_pad
        patches = ops.stem_im2col(x, l, t, oh, ow)                 # recomputed, not stored
        dw = ops.linear_wgrad(de, patches)                         # [c0, 32]
        gw, gg, gb = ops.deliver_param_grads((mod._conv_stem.weight, mod._bn0.weight, mod._bn0.bias),
                                             (dw[:, :27].reshape(c0, 3, 3, 3), dgamma, dbeta))
        return (None, gw, gg, gb, None) + ((None,) if link is not None else ())


class _StemLink:
    """hand-over between the stem and block 0 when the stem's bn0 + swish lives in block 0's depthwise kernels: the stem's
    BatchNorm coefficients (forward) and the BatchNorm-backward reductions block 0's data gradient leaves (backward)"""
    __slots__ = ("st", "part")

    def __init__(self):
        self.st, self.part = None, None


def _expand_conv(blk, x, we, rows, training=False, recompute=False):
    """_expand_conv of one block: e [rows, cexp] bf16 + the BatchNorm0 column-statistic partials (deterministic: the
    backward's recompute mode reproduces the forward's tensor bit for bit)."""
    a = blk.args
    if blk.fp8 and a.cin % 16 == 0 and not ops._rows_ok(rows, a.cexp, a.cin, None, 0):
        # config #5: fp8 (e4m3, per-tensor scale) activations and weights on the fp8 MFMA; BatchNorm statistics
        # and everything downstream stay on the bf16 / fp32 path; backward uses the bf16 tensors (straight-through)
        r = ops.linear_fwd_fp8(x, we, stats=True)
        return r[0] if recompute else r
    if recompute:
        return ops.linear_fwd(x, we)
    return _conv_stats(ops.linear_fwd, training, x, we)


class _MBConvFn(torch.autograd.Function):
    """One MBConvBlock, forward + hand-derived backward [ref: efficientnet_custom.py:91-132]."""

    @staticmethod
    def forward(ctx, x, rowscale, blk, n, h, w, *params):
        a = blk.args
        training = blk.training
        k, s = a.k, a.s
        l, r, t, b = a.pad
        oh, ow = _out_extent(h, t, b, k, s), _out_extent(w, l, r, k, s)
        hw, ohw = h * w, oh * ow
        saved = {}
        rc = blk.recompute
        wkkc = ops.transpose_f32(blk._depthwise_conv.weight.view(a.cexp, k * k), cache=True)
        xdw = efree = False
        if a.expand != 1:
            we = ops.cast_bf16(blk._expand_conv.weight.view(a.cexp, a.cin))
            # round 6: where no backward needs the expanded tensor e stored (no graph, or a recompute mode that rebuilds it), the
            # expand conv runs inside the depthwise launch's staging and e never exists in HBM: BatchNorm0's batch statistics
            # then come from the Gram matrix of the block input (one pass over x, 6 x narrower than e), from the statistics
            # tape of a re-forward, or from the running statistics (eval)
            # (autograd runs Function.forward with grad mode off and reports needs_input_grad from requires_grad alone:
            # whether a graph is being recorded is noted by MBConvBlock.forward before the call)
            xdw = (rc >= 1 or not blk.__dict__.get("_recording", True)) and blk.xdw_ok(n, h, w, oh, ow)
            # ... and stride-1 3x3 blocks whose backward forms its e rows from x as well (ops.dwconv_bwd_fused with xw) and
            # folds the BatchNorm0 backward into the expand conv's gradient GEMMs never need e anywhere: fused forward always
            efree = blk.efree_ok(n, h, w, oh, ow)
            xdw = xdw or efree
            gram = None
            if xdw:
                # x^T x and colsum(x) of the Gram statistics pass are what the folded BatchNorm0 backward of this block needs
                # again: they stay in the graph (a few KB) and travel on the statistics tape to the re-forward's backward
                if training and _replaying():
                    part0, gram = None, _TAPE.get()
                elif training:
                    part0, gram = ops.bn_gram_partials(x, we, n * hw)
                    if _TAPE is not None:
                        _TAPE.put(gram)
                else:
                    part0 = None
                st0 = _bn_stats(part0, n * hw, blk._bn0, training)
                e = None
                d, part1 = _conv_stats(ops.mbconv_xdw_fwd, training, x, we, (st0.scale, st0.shift), wkkc, n, h, w, a.cexp, k, s,
                                       l, t, oh, ow)
            else:
                e, part0 = _expand_conv(blk, x, we, n * hw, training)
                st0 = _bn_stats(part0, n * hw, blk._bn0, training)
            dw_in, pro0 = e, (st0.scale, st0.shift)
            saved.update(we=we, e=None if (rc >= 1 or efree) else e, st0=st0, gram=gram)
        else:
            # block 0 behind a linked stem: x is the stem's RAW conv output, its bn0 + swish is this block's prologue
            link = blk.__dict__.pop("_in_link", None)
            dw_in, pro0 = x, ((link.st.scale, link.st.shift) if link is not None else None)
            saved.update(link=link)
        if not xdw:
            d, part1 = _conv_stats(ops.dwconv_fwd, training, dw_in, wkkc, n, h, w, a.cexp, k, s, l, t, oh, ow, pro=pro0)
        st1 = _bn_stats(part1, n * ohw, blk._bn1, training)
        # Late stages (project conv on the tiled GEMM): the squeeze pass also stores A = silu(bn1(d)); the project GEMM
        # and its weight gradient then apply only the SE gate instead of re-evaluating BN+SiLU per output tile.
        keep = not ops._rows_ok(n * ohw, a.cout, a.cexp, None, 0)
        if keep:
            pooled, act1 = ops.bnact_pool(d, n, ohw, a.cexp, st1.scale, st1.shift, 1, keep_act=True)
        elif training and _replaying():
            pooled, act1 = _TAPE.get(), None                      # (early stages: the squeeze pass only produced this)
        else:
            pooled, act1 = ops.bnact_pool(d, n, ohw, a.cexp, st1.scale, st1.shift, 1), None
            if training and _TAPE is not None:
                _TAPE.put(pooled)
        if training and _replaying():
            gate = _TAPE.get()                                     # (every block: the squeeze-excite MLP's two launches)
        else:
            gate = ops.se_fwd(pooled, blk._se_reduce.weight.view(a.cse, a.cexp), blk._se_reduce.bias,
                              blk._se_expand.weight.view(a.cexp, a.cse), blk._se_expand.bias)
            if training and _TAPE is not None:
                _TAPE.put(gate)
        wp = ops.cast_bf16(blk._project_conv.weight.view(a.cout, a.cexp))
        if keep and blk.fp8 and a.cexp % 16 == 0 and a.cout > 64:
            wg = ops.gate_weights(wp, gate)                       # [n, cout, cexp] bf16: the SE gate folded into the weights
            p, part2 = ops.linear_fwd_fp8(act1, wg, stats=True, batch_w=(n, ohw))
        elif keep:
            p, part2 = _conv_stats(ops.linear_fwd, training, act1, wp, pro=(None, None, gate, ohw))
        else:
            p, part2 = _conv_stats(ops.linear_fwd, training, d, wp, pro=(st1.scale, st1.shift, gate, ohw))
        st2 = _bn_stats(part2, n * ohw, blk._bn2, training)
        y = ops.bnact_apply(p, n, ohw, a.cout, st2.scale, st2.shift, 0,
                            rowscale=rowscale if a.skip else None, res=x if a.skip else None)
        # recompute modes (activation memory of a kept graph, see EfficientNet.set_recompute): 1 drops the expanded
        # tensor e, 2 also the depthwise output d (+ the stored activation of the late stages), 4 also the projection
        # output p; the backward rebuilds them from the block input x and the saved BatchNorm coefficients
        saved.update(x=x, d=None if rc >= 2 else d, p=None if rc >= 4 else p, wkkc=wkkc, wp=wp, st1=st1, st2=st2, pooled=pooled, gate=gate,
                     act1=None if rc >= 2 else act1, keep_act=keep, xdw=xdw, efree=(a.expand != 1 and efree),
                     rowscale=rowscale if a.skip else None, geo=(n, h, w, oh, ow))
        ctx.blk, ctx.saved = blk, saved
        blk._out_geo = (n, oh, ow)
        return y

    @staticmethod
    def backward(ctx, dy):
        blk, sv = ctx.blk, ctx.saved
        a = blk.args
        n, h, w, oh, ow = sv["geo"]
        hw, ohw = h * w, oh * ow
        k, s = a.k, a.s
        l, r, t, b = a.pad
        dy = dy.contiguous()
        x, d, p = sv["x"], sv["d"], sv["p"]
        st1, st2, gate, pooled = sv["st1"], sv["st2"], sv["gate"], sv["pooled"]
        e, act1 = sv.get("e"), sv["act1"]
        efree = bool(sv.get("efree"))
        if a.expand != 1:
            st0 = sv["st0"]
            if e is None and not efree:                      # (recompute modes: same kernels, statistics epilogues off)
                e = _expand_conv(blk, x, sv["we"], n * hw, recompute=True)
        link = sv.get("link")
        if d is None:
            if sv.get("xdw"):
                # the forward produced d with the fused launch (from the UNROUNDED expand output): rebuilt the same way, bit for bit
                d = ops.mbconv_xdw_fwd(x, sv["we"], (st0.scale, st0.shift), sv["wkkc"], n, h, w, a.cexp, k, s, l, t, oh, ow)
            else:
                d = ops.dwconv_fwd(e if a.expand != 1 else x, sv["wkkc"], n, h, w, a.cexp, k, s, l, t, oh, ow,
                                   pro=(st0.scale, st0.shift) if a.expand != 1 else ((link.st.scale, link.st.shift) if link is not None else None))
            if sv["keep_act"]:
                act1 = ops.bnact_pool(d, n, ohw, a.cexp, st1.scale, st1.shift, 1, keep_act=True)[1]
        if p is None:                                        # mode 4: the projection conv again, from the rebuilt d
            if sv["keep_act"] and blk.fp8 and a.cexp % 16 == 0 and a.cout > 64:
                p = ops.linear_fwd_fp8(act1, ops.gate_weights(sv["wp"], gate), stats=True, batch_w=(n, ohw))[0]
            elif sv["keep_act"]:
                p = ops.linear_fwd(act1, sv["wp"], pro=(None, None, gate, ohw))
            else:
                p = ops.linear_fwd(d, sv["wp"], pro=(st1.scale, st1.shift, gate, ohw))
        # y = bn2(p) * rowscale + x
        dp, dg2, db2 = ops.bnact_bwd(p, n, ohw, a.cout, st2, blk._bn2.weight, 0, g=dy, rowscale=sv["rowscale"])
        # project 1x1: p = A1 . wp^T, A1 = silu(bn1(d)) * gate   (A1 is recomputed inside the wgrad GEMM)
        wp_t = ops.cast_transpose_bf16(blk._project_conv.weight.view(a.cout, a.cexp))      # [cexp, cout]
        # Early stages (row-streaming shapes, whole 16-row groups per image): dA1 = dp . wp never goes to memory -- the
        # projection's data-gradient GEMM runs TWICE with an elementwise epilogue over d, first for the squeeze-excite /
        # BatchNorm1 sums, then (once the SE backward has produced d loss / d pooled) for the BatchNorm1 + swish backward
        # itself: 1 + 2 passes over the depthwise-site tensor instead of 1 (dgrad) + 2 (sums) + 3 (apply)
        fuse = FUSE_PROJ_DGRAD and ops.proj_dgrad_fusable(n * ohw, a.cexp, a.cout, ohw)
        da1 = None if fuse else ops.linear_dgrad(dp, sv["wp"], w_t=wp_t)
        if act1 is not None:
            dwp = ops.linear_wgrad(dp, act1, pro=(None, None, gate, ohw))
            del act1
        else:
            dwp = ops.linear_wgrad(dp, d, pro=(st1.scale, st1.shift, gate, ohw))
        # squeeze-excite
        # ONE pass over (d, dA1) yields d loss / d gate AND the ingredients of the bn1-backward reductions
        sums = ops.proj_dgrad_se_sums(dp, wp_t, d, st1, n, ohw) if fuse else ops.bnact_se_sums(d, da1, n, ohw, a.cexp, st1, 1)
        dpooled, dw1, db1, dw2, dbse2 = ops.se_bwd(pooled, gate, sums[0], blk._se_reduce.weight.view(a.cse, a.cexp),
                                                   blk._se_reduce.bias, blk._se_expand.weight.view(a.cexp, a.cse),
                                                   blk._se_expand.bias)
        # bn1 + swish: upstream of swish output = dA1 * gate + dpooled / (oh*ow)
        part1 = ops.bn_partials_from_se_sums(sums, gate, dpooled, 1.0 / ohw)
        if fuse:
            coef1, dg1, db1n = ops.bn_bwd_coefs(part1, n * ohw, st1, blk._bn1.weight)
            dd = ops.proj_dgrad_bn_apply(dp, wp_t, d, st1, coef1, gate, dpooled, 1.0 / ohw, ohw)
        else:
            dd, dg1, db1n = ops.bnact_bwd(d, n, ohw, a.cexp, st1, blk._bn1.weight, 1, g=da1, mul=gate, add=dpooled,
                                          add_scale=1.0 / ohw, partials=part1)
        del da1
        # depthwise
        del d
        if a.expand != 1:
            dw_in, pro0 = e, (st0.scale, st0.shift)
        else:
            dw_in, pro0 = x, ((link.st.scale, link.st.shift) if link is not None else None)
        wflip = ops.flipped_taps_f32(blk._depthwise_conv.weight.view(a.cexp, k * k)) if s == 1 else None   # 180 degree rotation
        # round 5: stride-1 3x3 blocks run the WHOLE depthwise backward as one launch (conv_lane.hip MODE 3): data gradient with
        # the bn0 + swish epilogue AND the weight gradient from one staging of (dd, e) -- 3 passes over the expanded tensor
        # instead of 5; where that launch is not preferred: weight gradient + data gradient as two launches
        fused_dw = (FUSE_DW_BWD and a.expand != 1 and ops.dwconv_bwd_fused_ok(n, h, w, a.cexp, k, s, l, t, oh, ow, cin=a.cin if efree else 0))
        assert fused_dw or not efree                         # (efree_ok asked the same question in the forward)
        if not fused_dw:
            dwdw = ops.dwconv_bwd_weight(dw_in, dd, n, h, w, a.cexp, k, s, l, t, oh, ow, pro=pro0)
        grads = {}
        if a.expand != 1:
            # the data-gradient kernel finishes the bn0 + swish backward in its epilogue -- it reads e at the output
            # position, writes dZ0 = dA0 * silu'(bn0(e)) and leaves the BatchNorm-backward reductions behind, so the
            # separate reduce pass over (e, dA0) is gone and the apply pass is a plain linear combination (stride 1: the
            # forward kernel on flipped taps; stride 2, round 3: the marching super-pixel kernel -- dA0 of the stride-2
            # blocks, the largest tensors of the network, is never written)
            if fused_dw:
                dz0, part0, dwdw = ops.dwconv_bwd_fused(dd, e, st0, wflip, n, h, w, a.cexp, k, l, t, oh, ow,
                                                        xw=(x, sv["we"]) if efree else None)
            else:
                dz0, part0 = ops.dwconv_bwd_data(dd, sv["wkkc"], n, h, w, a.cexp, k, s, l, t, oh, ow, w_kkc_flipped=wflip,
                                                 epi=(e, st0))
            del dd
            if efree or 2 * n * hw * a.cexp >= (BN_FOLD_MIN_BYTES if s == 1 else max(BN_FOLD_MIN_BYTES, BN_FOLD_S2_MIN_BYTES)):
                # bn0 backward is linear in (dZ0, e) and e = x We^T: it is folded into the operands of the expand conv's
                # two gradient GEMMs (ops.bn_fold_expand_bwd) -- de is never formed, e is not read again.  Three passes
                # over the expanded tensor against ~10 small launches and 6 passes over the (6x smaller) block input:
                # pays from ~0.25 GB of expanded tensor per call (B5 at 32 x 1520 x 912: blocks 4-12 from 0.4 GB on one stream; with
                # the three launch chains of round 5 the small launches overlap and the blocks at 95 x 57 / c = 3072 join)
                del e, dw_in
                coef0, dg0, db0 = ops.bn_bwd_coefs(part0, n * hw, st0, blk._bn0.weight)
                dx, dwe = ops.bn_fold_expand_bwd(dz0, x, blk._expand_conv.weight.view(a.cexp, a.cin), sv["we"], coef0,
                                                 db0, n * hw, residual=dy if a.skip else None, gram=sv.get("gram"))
                de = None
            else:
                de, dg0, db0 = ops.bnact_bwd(e, n, hw, a.cexp, st0, blk._bn0.weight, 0, g=dz0, partials=part0)
                del e, dw_in
            del dz0
        elif link is not None:
            # linked stem: this launch finishes the STEM's bn0 + swish backward (x is the stem's raw conv output): what goes
            # back is dZ0 = dL/d bn0(x), the BatchNorm reductions travel in the link
            da0, link.part = ops.dwconv_bwd_data(dd, sv["wkkc"], n, h, w, a.cexp, k, s, l, t, oh, ow, w_kkc_flipped=wflip,
                                                 epi=(x, link.st))
            del dd
        else:
            da0 = ops.dwconv_bwd_data(dd, sv["wkkc"], n, h, w, a.cexp, k, s, l, t, oh, ow, w_kkc_flipped=wflip)
            del dd
        if a.expand != 1:
            if de is not None:
                we_t = ops.cast_transpose_bf16(blk._expand_conv.weight.view(a.cexp, a.cin))     # [cin, cexp]
                if ops.xbwd_rows_ok(n * hw, a.cexp, a.cin):     # one pass over de for both gradients (round 5)
                    dx, dwe = ops.xbwd_rows(de, x, we_t, residual=dy if a.skip else None)
                else:
                    dx = ops.linear_dgrad(de, sv["we"], residual=dy if a.skip else None, w_t=we_t)
                    dwe = ops.linear_wgrad(de, x)
            grads["_expand_conv.weight"] = dwe.view(a.cexp, a.cin, 1, 1)
            grads["_bn0.weight"], grads["_bn0.bias"] = dg0, db0
        else:
            dx = da0
            if a.skip:
                dx = ops.bnact_apply(da0, n, hw, a.cin, blk._ones, blk._zeros, 0, res=dy)
        grads["_depthwise_conv.weight"] = ops.transpose_f32(dwdw).view(a.cexp, 1, k, k)
        grads["_bn1.weight"], grads["_bn1.bias"] = dg1, db1n
        grads["_se_reduce.weight"], grads["_se_reduce.bias"] = dw1.view(a.cse, a.cexp, 1, 1), db1
        grads["_se_expand.weight"], grads["_se_expand.bias"] = dw2.view(a.cexp, a.cse, 1, 1), dbse2
        grads["_project_conv.weight"] = dwp.view(a.cout, a.cexp, 1, 1)
        grads["_bn2.weight"], grads["_bn2.bias"] = dg2, db2
        ctx.saved = None
        return (dx, None, None, None, None, None) + ops.deliver_param_grads(blk._params(), [grads[nm] for nm in blk._param_names])


class _HeadFn(torch.autograd.Function):
    """_conv_head (1x1) + _bn1 + swish + global average pool [ref: efficientnet_custom.py:283, 307-309]."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, mod, n, h, wd):
        cin, cout = w.shape[1], w.shape[0]
        wb = ops.cast_bf16(w.view(cout, cin))
        if mod.fp8 and cin % 16 == 0:
            e, part = ops.linear_fwd_fp8(x, wb, stats=True)
        else:
            e, part = _conv_stats(ops.linear_fwd, mod.training, x, wb)
        st = _bn_stats(part, n * h * wd, mod._bn1, mod.training)
        pooled = ops.bnact_pool(e, n, h * wd, cout, st.scale, st.shift, 1)
        ctx.mod, ctx.st, ctx.geo = mod, st, (n, h, wd, cin, cout)
        ctx.save_for_backward(x, e, wb)
        mod._head_cache = (e, st, n, h, wd, cout)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        x, e, wb = ctx.saved_tensors
        n, h, wd, cin, cout = ctx.geo
        mod = ctx.mod
        de, dgamma, dbeta = ops.bnact_bwd(e, n, h * wd, cout, ctx.st, mod._bn1.weight, 1, add=dpooled.contiguous(),
                                          add_scale=1.0 / (h * wd))
        dx = ops.linear_dgrad(de, wb, w_t=ops.cast_transpose_bf16(mod._conv_head.weight.view(cout, cin)))
        dw = ops.linear_wgrad(de, x)
        gw, gg, gb = ops.deliver_param_grads((mod._conv_head.weight, mod._bn1.weight, mod._bn1.bias),
                                             (dw.view(cout, cin, 1, 1), dgamma, dbeta))
        return dx, gw, gg, gb, None, None, None, None


class _DropoutFn(torch.autograd.Function):
    """nn.Dropout on the pooled features [ref: efficientnet_custom.py:310-312]; Philox mask, regenerated in backward."""

    @staticmethod
    def forward(ctx, x, p, seed, sid):
        ctx.cfg = (p, seed, sid)
        return ops.dropout_f32(x.contiguous(), p, seed, sid)

    @staticmethod
    def backward(ctx, dy):
        p, seed, sid = ctx.cfg
        return ops.dropout_f32(dy.contiguous(), p, seed, sid), None, None, None


# ================================================================================================
#                                           modules
# ================================================================================================
class _BN(nn.BatchNorm2d):
    """Parameter/buffer container with the reference's BatchNorm2d keys; arithmetic happens in HIP kernels."""

    def __init__(self, c):
        super().__init__(c, momentum=BN_MOMENTUM, eps=BN_EPS)
        self.track_update = True          # engine may switch running-stat updates off for re-forward passes
        self.defer_count = None           # list collecting the counters of one encoder forward (see EfficientNet.forward)

    def forward(self, x):                 # pragma: no cover
        raise RuntimeError("BatchNorm arithmetic runs inside the fused HIP kernels, not here")


class _Conv(nn.Conv2d):
    """Parameter container (OIHW fp32) mirroring Conv2dStaticSamePadding's keys."""

    def __init__(self, cin, cout, k, stride=1, groups=1, bias=False):
        super().__init__(cin, cout, k, stride, 0, 1, groups, bias)

    def forward(self, x):                 # pragma: no cover
        raise RuntimeError("convolutions run inside the HIP kernels, not here")


_Geo = namedtuple("_Geo", ["idx", "expand", "k", "s", "cin", "cexp", "cout", "cse", "pad", "skip"])


class MBConvBlock(nn.Module):
    """Mobile inverted residual bottleneck with squeeze-excite [ref: efficientnet_custom.py:36-140]."""

    def __init__(self, block_args: BlockArgs, global_params: GlobalParams, image_size=None, idx: int = 0):
        super().__init__()
        self._block_args = block_args
        a = block_args
        s = a.stride if isinstance(a.stride, int) else a.stride[0]
        cin, cexp = a.input_filters, a.input_filters * a.expand_ratio
        cse = max(1, int(a.input_filters * a.se_ratio))
        self.has_se = True
        self.id_skip = a.id_skip
        if a.expand_ratio != 1:
            self._expand_conv = _Conv(cin, cexp, 1)
            self._bn0 = _BN(cexp)
        self._depthwise_conv = _Conv(cexp, cexp, a.kernel_size, s, groups=cexp)
        self._bn1 = _BN(cexp)
        self._se_reduce = _Conv(cexp, cse, 1, bias=True)
        self._se_expand = _Conv(cse, cexp, 1, bias=True)
        self._project_conv = _Conv(cexp, a.output_filters, 1)
        self._bn2 = _BN(a.output_filters)
        self.args = _Geo(idx, a.expand_ratio, a.kernel_size, s, cin, cexp, a.output_filters, cse,
                         _static_pad(image_size, a.kernel_size, s),
                         bool(a.id_skip and s == 1 and cin == a.output_filters))
        self._param_names = [n for n, _ in self.named_parameters()]
        self.fp8 = False
        self.recompute = 0
        self.register_buffer("_ones", torch.ones(cin), persistent=False)
        self.register_buffer("_zeros", torch.zeros(cin), persistent=False)

    def xdw_ok(self, n, h, w, oh, ow):
        """does this block's forward take the fused expand + depthwise launch (ops.mbconv_xdw_fwd) when its expanded tensor
        need not be stored?"""
        a = self.args
        return (a.expand != 1 and a.cin <= XDW_MAX_CIN and not self.fp8
                and ops.mbconv_xdw_ok(n, h, w, a.cin, a.cexp, a.k, a.s, a.pad[0], a.pad[2], oh, ow))

    def efree_ok(self, n, h, w, oh, ow):
        """may this block's expanded tensor never exist -- fused forward, fused backward with the e rows formed from the block
        input, folded BatchNorm0 backward?  (stride-1 3x3, at most 64 input channels, the shapes on which the fused backward
        launch is the preferred form, and enough expanded bytes for the fold to be the chosen BatchNorm0 backward)"""
        a = self.args
        return bool(ops.EFREE and FUSE_DW_BWD and a.k == 3 and a.s == 1 and self.training and self.xdw_ok(n, h, w, oh, ow)
                    and 2 * n * h * w * a.cexp >= BN_FOLD_MIN_BYTES
                    and ops.dwconv_bwd_fused_ok(n, h, w, a.cexp, a.k, a.s, a.pad[0], a.pad[2], oh, ow, cin=a.cin))

    def _params(self):
        """the block's Parameter objects in ``_param_names`` order (cached: ``named_parameters`` walks the module tree)"""
        # cached with the (owner dict, key) slot of every parameter: a Parameter OBJECT survives .to() / load_state_dict (both
        # write .data in place) but not ``module.weight = nn.Parameter(...)`` / ``load_state_dict(assign=True)`` / to_empty() --
        # then the slots no longer hold the cached objects and the list is rebuilt (gradients must not go to
        # orphaned Parameters)
        cached = self.__dict__.get("_plist")
        if cached is not None:
            ps, slots = cached
            for p_, (d_, k_) in zip(ps, slots):
                if d_.get(k_) is not p_:
                    cached = None
                    break
        if cached is None:
            slots = [(m._parameters, k) for m in self.modules() for k, v in m._parameters.items() if v is not None]
            ps = [d_[k_] for d_, k_ in slots]
            assert len(ps) == len(self._param_names)
            self.__dict__["_plist"] = (ps, slots)
        return ps

    def forward(self, inputs, n, h, w, rowscale=None):
        self.__dict__["_recording"] = torch.is_grad_enabled()      # is an autograd graph being recorded for this call?
        return _MBConvFn.apply(inputs, rowscale, self, n, h, w, *self._params())


class EfficientNet(nn.Module):
    """[ref: efficientnet_custom.py:143-411]"""

    def __init__(self, blocks_args: List[BlockArgs] = None, global_params: GlobalParams = None):
        super().__init__()
        assert isinstance(blocks_args, list) and len(blocks_args) > 0
        self._global_params, self._blocks_args = global_params, blocks_args
        gp = global_params
        size = (gp.image_size, gp.image_size) if isinstance(gp.image_size, int) else tuple(gp.image_size)
        c0 = _scaled_width(32, gp)
        self._conv_stem = _Conv(3, c0, 3, 2)
        self._bn0 = _BN(c0)
        self.stem_pad = _static_pad(size, 3, 2)
        size = _down(size, 2)
        self._blocks = nn.ModuleList([])
        for ba in blocks_args:
            ba = ba._replace(input_filters=_scaled_width(ba.input_filters, gp),
                             output_filters=_scaled_width(ba.output_filters, gp),
                             num_repeat=_scaled_depth(ba.num_repeat, gp))
            s = ba.stride if isinstance(ba.stride, int) else ba.stride[0]
            self._blocks.append(MBConvBlock(ba, gp, size, idx=len(self._blocks)))
            size = _down(size, s)
            if ba.num_repeat > 1:
                ba = ba._replace(input_filters=ba.output_filters, stride=1)
            for _ in range(ba.num_repeat - 1):
                self._blocks.append(MBConvBlock(ba, gp, size, idx=len(self._blocks)))
        head_out = _scaled_width(1280, gp)
        self._conv_head = _Conv(ba.output_filters, head_out, 1)
        self._bn1 = _BN(head_out)
        self._dropout_p = gp.dropout_rate if gp.include_top else 0.0
        self.out_dim = head_out
        self.fp8 = False
        self.rng = _Seeds()
        self.register_buffer("_ones_b", torch.ones(1), persistent=False)
        self._bn_layers = [m for m in self.modules() if isinstance(m, _BN)]

    # ---------------------------------------------------------------------------- construction API
    @classmethod
    def from_name(cls, model_name, in_channels=3, **override_params):
        cls._check_model_name_is_valid(model_name)
        if in_channels != 3:
            raise NotImplementedError("the gfx950 stem kernel is specialised for 3 input channels")
        w, d, res, p = _COEFFS[model_name]
        gp = GlobalParams(width_coefficient=w, depth_coefficient=d, image_size=res, dropout_rate=p, num_classes=1000,
                          batch_norm_momentum=0.99, batch_norm_epsilon=BN_EPS, drop_connect_rate=0.2, depth_divisor=8,
                          min_depth=None, include_top=True)
        if override_params:
            gp = gp._replace(**override_params)
        blocks = [BlockArgs(r, k, s, e, i, o, 0.25, True) for (r, k, s, e, i, o) in _STAGES]
        return cls(blocks, gp)

    @classmethod
    def from_pretrained(cls, model_name, weights_path=None, advprop=False, in_channels=3, num_classes=1000,
                        **override_params):
        """Builds the network; ImageNet weights are loaded only from a local ``weights_path`` (the reference
        downloads them, efficient_net_custom_utils.py:584-615 -- there is no network on the target boxes)."""
        model = cls.from_name(model_name, num_classes=num_classes, **override_params)
        if isinstance(weights_path, str):
            sd = torch.load(weights_path, map_location="cpu")
            sd.pop("_fc.weight", None)
            sd.pop("_fc.bias", None)
            ret = model.load_state_dict(sd, strict=False)
            assert not ret.unexpected_keys, ret.unexpected_keys
        return model

    @classmethod
    def get_image_size(cls, model_name):
        cls._check_model_name_is_valid(model_name)
        return _COEFFS[model_name][2]

    @classmethod
    def _check_model_name_is_valid(cls, model_name):
        if model_name not in VALID_MODELS:
            raise ValueError("model_name should be one of: " + ", ".join(VALID_MODELS))

    def set_fp8(self, on: bool = True):
        """BASELINE config #5: forward of the late-stage 1x1 convolutions (expand / project / head on the tiled MFMA path)
        with per-tensor-scaled OCP e4m3 activations and weights on gfx950's fp8 MFMA [ref: efficientnet_custom.py:104,
        122,283 are the convolutions concerned]; statistics, depthwise stages and the whole backward stay bf16 / fp32."""
        self.fp8 = bool(on)
        for blk in self._blocks:
            blk.fp8 = bool(on)
        return self

    def set_recompute(self, mode: int = 0):
        """Activation memory of a kept autograd graph against backward work (the micro-batched contrastive step keeps as
        many micro-batch graphs as fit and re-runs the forward of the others): 0 = every MBConv block stores its expanded
        tensor e and its depthwise output d (the two big ones, ~72 % of the graph); 1 = e is rebuilt in the backward by
        one more expand GEMM from the block input; 2 = e and d (and the stored late-stage activation) are rebuilt
        (expand GEMM + depthwise forward); 3 = mode 1 everywhere plus mode 2 on the blocks whose depthwise stage is a
        stride-1 3x3 (the depthwise kernels that run near HBM rate: 45 % of the d bytes of B5 for ~1.5 % more work);
        4 = mode 2 plus the projection conv's output p (needed first thing in the backward, by BatchNorm2): rebuilt from the
        rebuilt d -- the graph of a block is then its input x alone.
        Same kernels on the same operands: gradients agree across modes to the spread of the atomically reduced ones."""
        assert mode in (0, 1, 2, 3, 4)
        for blk in self._blocks:
            a = blk.args
            blk.recompute = int(mode) if mode != 3 else (2 if (a.k == 3 and a.s == 1 and a.expand != 1 and a.cexp <= MODE3_MAX_CEXP) else 1)
        return self

    def xdw_reforward_modes(self):
        """Micro-batched step (engine._step_micro): the graph-less first forward of a micro-batch runs its narrow-input blocks
        through the fused expand + depthwise launch (no expanded tensor).  Its re-forward -- same inputs, replayed statistics,
        graph recorded -- must produce the SAME embeddings, so those blocks take recompute mode 1 there (fused forward, the
        expanded tensor rebuilt by one GEMM in the backward) instead of the two-launch forward that rounds e to 16 bits first.
        Returns the list of (block, previous mode) to restore."""
        changed = []
        if not ops.XDW:
            return changed
        for blk in self._blocks:
            a = blk.args
            if blk.recompute == 0 and a.expand != 1 and a.cin <= XDW_MAX_CIN and not blk.fp8:
                changed.append((blk, blk.recompute))
                blk.recompute = 1
        return changed

    def set_swish(self, memory_efficient=True):
        """No-op: the swish is always the fused, recompute-in-backward form."""

    # ---------------------------------------------------------------------------- forward
    def _drop_connect_scales(self, n, device, seed):
        """keep/keep_prob per (block, sample) [ref: efficient_net_custom_utils.py:129-154]; rate = 0.2*idx/len."""
        rate = self._global_params.drop_connect_rate
        out = []
        nb = len(self._blocks)
        for i, blk in enumerate(self._blocks):
            p = rate * float(i) / nb if rate else 0.0
            if not (self.training and p > 0.0 and blk.args.skip):
                out.append(None)
                continue
            out.append(ops.dropout_f32(self._ones_b.expand(n).contiguous(), p, seed, 1000 + i))
        return out

    def _features_nhwc(self, inputs):
        if not inputs.is_cuda:
            raise RuntimeError("mammo_clip_amd.EfficientNet runs only on a HIP device (no CPU fallback)")
        x = inputs if isinstance(inputs, ops.RawImages) or inputs.dtype == torch.float32 else inputs.float()
        seed = self.rng.next()
        self._last_seed = seed
        # num_batches_tracked of the ~120 BatchNorm layers: collected during the forward, incremented by one
        # multi-tensor launch at its end (116 one-element add launches per forward otherwise)
        counters = [] if self.training else None
        bns = self._bn_layers if counters is not None else ()
        for bn in bns:
            bn.defer_count = counters
        try:
            b0 = self._blocks[0].args
            # block 0 has no expand conv (every EfficientNet-B*): the stem's bn0 + swish runs inside its depthwise kernels
            link = _StemLink() if (LINK_STEM and b0.expand == 1 and not b0.skip and b0.s == 1) else None
            if link is not None:
                y = _StemFn.apply(x, self._conv_stem.weight, self._bn0.weight, self._bn0.bias, self, link)
                self._blocks[0].__dict__["_in_link"] = link
            else:
                y = _StemFn.apply(x, self._conv_stem.weight, self._bn0.weight, self._bn0.bias, self)
            n, h, w = self._geo
            scales = self._drop_connect_scales(n, x.device, seed)
            for blk, rs in zip(self._blocks, scales):
                y = blk(y, n, h, w, rs)
                n, h, w = blk._out_geo
        finally:
            for bn in bns:
                bn.defer_count = None
        self._pending_counters = counters
        return y, n, h, w

    # ---------------------------------------------------------------------------- side-stream calls (model/clip.py)
    def warm_weight_images(self, backward: bool):
        """Build (or find) every cached weight image the forward -- and, with ``backward``, the backward -- of this encoder
        reads, on the CURRENT stream: two calls of the encoder on different streams then only ever hit the cache (an image
        built by one chain while the other chain is in flight would be read without an ordering between the streams).  Same
        helper calls with the same views as the autograd functions below (the cache key holds pointer, shape and strides)."""
        for blk in self._blocks:
            a = blk.args
            if a.expand != 1:
                ops.cast_bf16(blk._expand_conv.weight.view(a.cexp, a.cin))
                if backward:
                    ops.cast_transpose_bf16(blk._expand_conv.weight.view(a.cexp, a.cin))
            ops.transpose_f32(blk._depthwise_conv.weight.view(a.cexp, a.k * a.k), cache=True)
            if backward and a.s == 1:
                ops.flipped_taps_f32(blk._depthwise_conv.weight.view(a.cexp, a.k * a.k))
            ops.cast_bf16(blk._project_conv.weight.view(a.cout, a.cexp))
            if backward:
                ops.cast_transpose_bf16(blk._project_conv.weight.view(a.cout, a.cexp))
        cout, cin = self._conv_head.weight.shape[0], self._conv_head.weight.shape[1]
        ops.cast_bf16(self._conv_head.weight.view(cout, cin))
        if backward:
            ops.cast_transpose_bf16(self._conv_head.weight.view(cout, cin))

    def side_call_begin(self):
        """the next forward of this encoder runs on a side stream beside another one: BatchNorm running statistics and batch
        counters are left to ``side_call_end`` (after the join, on the main stream)"""
        global _BN_DEFER
        self._hold_counters = True
        if self.training:
            _BN_DEFER = _BNDefer(self)

    def side_call_end(self):
        global _BN_DEFER
        d, _BN_DEFER = _BN_DEFER, None
        self._hold_counters = False
        if d is not None:
            d.apply()
        self._flush_counters()

    def forward_pair(self, x1, x2, side):
        """Both image views of a batch [ref: clip.py:83,108 -- two sequential ``encode_image`` calls] as two chains issued
        BLOCK BY BLOCK in alternation: view 1 on the current stream, view 2 on ``side``.  Each chain is exactly the
        launch sequence of ``forward`` on its input (same seeds: two consecutive draws, view 1 first; per-view BatchNorm
        statistics; view 2's running-statistic updates and both views' batch counters applied after the join, view 1
        first), so every result equals two sequential calls; what changes is the ISSUE order -- the two chains are in flight
        side by side from the first launch on, in the forward and (autograd replays the creation order backwards) in the
        backward.  Returns the two pooled feature tensors."""
        global _BN_DEFER
        if not (x1.is_cuda and x2.is_cuda):
            raise RuntimeError("mammo_clip_amd.EfficientNet runs only on a HIP device (no CPU fallback)")
        main = torch.cuda.current_stream(x1.device)
        conv = lambda x: x if isinstance(x, ops.RawImages) or x.dtype == torch.float32 else x.float()   # noqa: E731
        # view 2's dtype conversion (fp16 / bf16 / integer inputs) is a launch of its own: it is issued on the SIDE stream,
        # where chain 2 reads its result (on the main stream, behind the fork, nothing ordered it before the side
        # chain's stem); the inputs themselves were produced on the main stream before the caller's fork -- the explicit wait
        # below also covers callers that built them after it (model/clip.py: ``batch["image_views"].to(device)``)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            x2c = conv(x2)
        xs = [conv(x1), x2c]
        for t_ in (x2.data if isinstance(x2, ops.RawImages) else x2,):
            if torch.is_tensor(t_) and t_.is_cuda:
                t_.record_stream(side)          # allocated on the main stream, read by the side chain (allocator reuse)
        seeds = [self.rng.next(), self.rng.next()]
        training = self.training
        counters = [[], []] if training else [None, None]
        bns = self._bn_layers if training else ()
        defer = None

        def chain(i):
            """switch the issue point to chain i (stream, counter list, running-statistic target)"""
            global _BN_DEFER
            torch.cuda.set_stream(side if i else main)
            for bn in bns:
                bn.defer_count = counters[i]
            _BN_DEFER = defer if i else None

        b0 = self._blocks[0].args
        linked = LINK_STEM and b0.expand == 1 and not b0.skip and b0.s == 1
        ys, scales, geo = [None, None], [None, None], None
        try:
            torch.cuda.set_stream(side)
            defer = _BNDefer(self) if training else None          # (zero-filled scratch: allocated and cleared on the side stream)
            links = [None, None]
            for i in (0, 1):
                chain(i)
                links[i] = _StemLink() if linked else None
                if linked:
                    ys[i] = _StemFn.apply(xs[i], self._conv_stem.weight, self._bn0.weight, self._bn0.bias, self, links[i])
                else:
                    ys[i] = _StemFn.apply(xs[i], self._conv_stem.weight, self._bn0.weight, self._bn0.bias, self)
                n, h, w = self._geo
                scales[i] = self._drop_connect_scales(n, xs[i].device, seeds[i])
            geo = (n, h, w)
            for bi, blk in enumerate(self._blocks):
                n, h, w = geo
                for i in (0, 1):
                    chain(i)
                    if bi == 0 and linked:
                        blk.__dict__["_in_link"] = links[i]
                    ys[i] = blk(ys[i], n, h, w, scales[i][bi])
                geo = blk._out_geo
            n, h, w = geo
            pooled = [None, None]
            for i in (0, 1):
                chain(i)
                pooled[i] = _HeadFn.apply(ys[i], self._conv_head.weight, self._bn1.weight, self._bn1.bias, self, n, h, w)
                if training and self._dropout_p > 0.0:
                    pooled[i] = _DropoutFn.apply(pooled[i], self._dropout_p, seeds[i], 999)
        finally:
            torch.cuda.set_stream(main)
            for bn in bns:
                bn.defer_count = None
            _BN_DEFER = None
        self._last_seed = seeds[1]
        main.wait_stream(side)
        if defer is not None:
            defer.apply()
        with torch.no_grad():
            for c in counters:
                if c:
                    torch._foreach_add_(c, 1)
        pooled[1].record_stream(main)
        return pooled[0], pooled[1]

    def _flush_counters(self):
        if getattr(self, "_hold_counters", False):
            return
        c = getattr(self, "_pending_counters", None)
        if c:
            with torch.no_grad():
                torch._foreach_add_(c, 1)
        self._pending_counters = None

    def extract_features(self, inputs):
        """Final feature map after head conv + BN + swish, NCHW fp32 (API parity; not on the training hot path)."""
        y, n, h, w = self._features_nhwc(inputs)
        _HeadFn.apply(y, self._conv_head.weight, self._bn1.weight, self._bn1.bias, self, n, h, w)
        self._flush_counters()
        e, st, n, h, w, cout = self._head_cache
        fmap = ops.bnact_apply(e, n, h * w, cout, st.scale, st.shift, 1)
        return ops.cast_f32(fmap).view(n, h, w, cout).permute(0, 3, 1, 2)

    def forward(self, inputs):
        """[ref: efficientnet_custom.py:287-313]: pooled features [b, out_dim] (fp32); the dict form
        ``{"image": x}`` returns (pooled, raw_feature_map)."""
        want_map = isinstance(inputs, dict) and "image" in inputs
        x = inputs["image"] if want_map else inputs
        y, n, h, w = self._features_nhwc(x)
        pooled = _HeadFn.apply(y, self._conv_head.weight, self._bn1.weight, self._bn1.bias, self, n, h, w)
        self._flush_counters()
        if self.training and self._dropout_p > 0.0:
            pooled = _DropoutFn.apply(pooled, self._dropout_p, self._last_seed, 999)
        if want_map:
            e, st, n, h, w, cout = self._head_cache
            fmap = ops.bnact_apply(e, n, h * w, cout, st.scale, st.shift, 1)
            return pooled, ops.cast_f32(fmap).view(n, h, w, cout).permute(0, 3, 1, 2)
        return pooled
