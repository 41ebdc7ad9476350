"""BreastClip: image encoder + text encoder + projection heads + L2 normalisation [ref: model/clip.py:14-114].
Same constructor, attributes (image_encoder, text_encoder, image_projection, text_projection, projection,
logit_scale, text_pooling) and ``forward(batch, device) -> dict`` contract as the reference."""
import logging
import math
import os
from typing import Dict

import torch
from torch import nn

from .. import _tokens
from ... import ops
from .modules import load_image_encoder, load_projection_head, load_text_encoder

log = logging.getLogger(__name__)


class _EosPoolFn(torch.autograd.Function):
    """features[b] = hidden[b, attention_mask[b].sum() - 1]  [ref: clip.py:65-68]"""

    @staticmethod
    def forward(ctx, hid, mask):
        b, t, h = hid.shape
        ctx.shape = (b, t, h)
        ctx.save_for_backward(mask)
        return ops.eos_gather(hid.contiguous(), mask, b, t, h)

    @staticmethod
    def backward(ctx, dout):
        (mask,) = ctx.saved_tensors
        b, t, h = ctx.shape
        return ops.eos_scatter(dout.contiguous(), mask, b, t, h).view(b, t, h), None


class _L2NormFn(torch.autograd.Function):
    """x / ||x||_2 per row, no epsilon [ref: clip.py:90-91]"""

    @staticmethod
    def forward(ctx, x):
        y, norm = ops.l2norm_fwd(x.contiguous())
        ctx.save_for_backward(y, norm)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, norm = ctx.saved_tensors
        return ops.l2norm_bwd(dy, y, norm)


def _record_on(t, stream):
    """a tensor (or ops.RawImages) allocated on the current stream that ``stream`` is about to read"""
    t = getattr(t, "data", t) if isinstance(t, ops.RawImages) else t
    if torch.is_tensor(t) and t.is_cuda:
        t.record_stream(stream)


_TEXT_ONE_CALL = os.environ.get("MC_TEXT_ONE_CALL", "1") != "0"      # developer switch: A/B against two encoder calls
# MC_STREAMS: encoder calls of one forward on separate HIP streams (ops.side_stream): bit 0 = the text encoder, bit 1 = the
# second image view (both views then issued block by block in alternation, EfficientNet.forward_pair; bit 2 = call after call
# instead).  Same kernels on the same operands with the same seeds and per-view statistics: same results.
_STREAMS = int(os.environ.get("MC_STREAMS", "3"))


class BreastClip(nn.Module):
    def __init__(self, model_config: Dict, all_loss_config: Dict, tokenizer=None):
        super().__init__()
        self.tokenizer = tokenizer
        self.image_encoder = load_image_encoder(model_config["image_encoder"])
        vocab = tokenizer.vocab_size if tokenizer is not None else None
        self.text_encoder = load_text_encoder(model_config["text_encoder"], vocab_size=vocab)
        self.text_pooling = model_config["text_encoder"]["pooling"]
        self.model_config = model_config
        self.loss_config = {k: v for k, v in all_loss_config.items()}
        self.projection = "projection_head" in model_config
        if self.projection:
            self.image_projection = load_projection_head(self.image_encoder.out_dim, model_config["projection_head"])
            self.text_projection = load_projection_head(self.text_encoder.out_dim, model_config["projection_head"])
        else:
            assert self.image_encoder.out_dim == self.text_encoder.out_dim, \
                "Without 'projection_head', embedding_dim of the image and text encoder must be the same."
        self.temperature = model_config["temperature"] if "temperature" in model_config else None
        if self.temperature:
            self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / self.temperature))
        else:
            self.logit_scale = torch.tensor(1, dtype=torch.float32)
            log.warning("[Mammo-CLIP] missing temperature scaling factor")

    def encode_image(self, image):
        feats = self.image_encoder(image)
        if self.model_config["image_encoder"]["model_type"].lower() == "cnn":
            return feats
        return feats[:, 0]

    def encode_image_normalized(self, image):
        emb = self.encode_image(image)
        emb = self.image_projection(emb) if self.projection else emb
        return _L2NormFn.apply(emb)

    def encode_text(self, text_tokens):
        hid = self.text_encoder(text_tokens)
        if self.text_pooling == "eos":
            return _EosPoolFn.apply(hid, text_tokens["attention_mask"])
        if self.text_pooling == "bos":
            return hid[:, 0].float()
        if self.text_pooling == "mean":
            m = text_tokens["attention_mask"].unsqueeze(-1).expand(hid.size()).float()
            return torch.sum(hid.float() * m, dim=1) / torch.clamp(m.sum(dim=1), min=1e-9)
        raise NotImplementedError("Not supported pooling method : %s", self.text_pooling)

    def forward(self, batch, device=None):
        device = batch["images"].device if device is None else device
        images = batch["images"].to(device)
        use_streams = _STREAMS if images.is_cuda else 0
        main = torch.cuda.current_stream(device) if use_streams else None
        s_txt = ops.side_stream(0, device) if use_streams & 1 else None
        two = "text_tokens2" in batch and "image_views" in batch
        enc = self.image_encoder
        s_view = ops.side_stream(1, device) if (use_streams & 2 and two and hasattr(enc, "side_call_begin")) else None
        if s_view is not None:
            enc.warm_weight_images(backward=torch.is_grad_enabled())     # both views read them: built before the fork
        if use_streams:
            ops.fork_side(device)
            ops.FORKED += 1
        try:
            return self._forward_chains(batch, device, images, two, main, s_txt, s_view)
        finally:
            if use_streams:
                ops.FORKED -= 1

    def _forward_chains(self, batch, device, images, two, main, s_txt, s_view):
        enc = self.image_encoder
        view = None
        # (an encoder with forward hooks / pre-hooks takes the call-after-call path: forward_pair is not ``__call__``, the hooks
        # would silently not fire -- ADVICE r5)
        hooked = bool(getattr(enc, "_forward_hooks", None) or getattr(enc, "_forward_pre_hooks", None))
        if s_view is not None and (_STREAMS & 4) == 0 and hasattr(enc, "forward_pair") and not hooked \
                and self.model_config["image_encoder"]["model_type"].lower() == "cnn":
            # both views block by block in alternation (EfficientNet.forward_pair): the two chains are in flight side by side
            # from the first launch on; MC_STREAMS bit 2 keeps the call-after-call issue order below (A/B)
            img, view = enc.forward_pair(images, batch["image_views"].to(device), s_view)
        else:
            img = self.encode_image(images)
        if s_view is not None and view is None:
            # the second view on its own stream, issued right behind the first (same host order of the two encoder calls as
            # without streams: seeds, statistics tapes and the gradient sink's arrival order do not change)
            views = batch["image_views"].to(device)
            s_view.wait_stream(main)                    # (a fresh H2D copy above is main-stream work issued after the fork)
            _record_on(views, s_view)
            torch.cuda.set_stream(s_view)
            enc.side_call_begin()
            try:
                view = self.encode_image(views)
            finally:
                torch.cuda.set_stream(main)
                main.wait_stream(s_view)
                enc.side_call_end()
            view.record_stream(main)
        tok = _tokens.to_device(batch["text_tokens"], device)
        txt2 = None
        if s_txt is not None:
            # token tensors moved to the device just now are main-stream allocations read by the text chain: ordered behind
            # the copy, and recorded on the consuming stream so the allocator does not hand the block out again while the
            # side stream still reads it (ops.py's crossing-tensor rule)
            s_txt.wait_stream(main)
            for t_ in tok.values():
                _record_on(t_, s_txt)
            torch.cuda.set_stream(s_txt)
        try:
            txt, txt2 = self._encode_reports(batch, tok, two, device)
        finally:
            if s_txt is not None:
                torch.cuda.set_stream(main)
        if s_txt is not None:
            main.wait_stream(s_txt)
            for t_ in (txt, txt2):
                if t_ is not None:
                    t_.record_stream(main)
        return self._finish(batch, device, img, txt, txt2, two, view)

    def _encode_reports(self, batch, tok, two, device):
        txt2 = None
        if two:
            # Both reports of a pair go through the text encoder in ONE call when their token tensors have the same
            # shape: BERT has no cross-sample interaction (LayerNorm per token, attention per sequence), so the result
            # equals two calls [ref: clip.py:92,103 calls encode_text twice] while every GEMM sees twice the rows, the
            # weight gradients are produced once and ~360 launches per step disappear.
            tok2 = _tokens.to_device(batch["text_tokens2"], device)
            if _TEXT_ONE_CALL and tok.keys() == tok2.keys() and all(torch.is_tensor(tok[k]) and tok[k].shape == tok2[k].shape for k in tok):
                nb = tok["input_ids"].shape[0]
                both = self.encode_text({k: torch.cat([tok[k], tok2[k]]) for k in tok})
                txt, txt2 = both[:nb], both[nb:]
            else:
                txt = self.encode_text(tok)
        else:
            txt = self.encode_text(tok)
        if two and txt2 is None:
            txt2 = self.encode_text(_tokens.to_device(batch["text_tokens2"], device))
        return txt, txt2

    def _finish(self, batch, device, img, txt, txt2, two, view=None):
        img_e = self.image_projection(img) if self.projection else img
        txt_e = self.text_projection(txt) if self.projection else txt
        img_e, txt_e = _L2NormFn.apply(img_e), _L2NormFn.apply(txt_e)
        out = {"image_embeddings": img_e, "text_embeddings": txt_e,
               "labels": torch.arange(img_e.shape[0], device=device), "logit_scale": self.logit_scale.exp()}
        if two:
            txt2_e = self.text_projection(txt2) if self.projection else txt      # [ref quirk: clip.py:105]
            out["text_embeddings2"] = _L2NormFn.apply(txt2_e)
            if view is None:
                view = self.encode_image(batch["image_views"].to(device))
            view_e = self.image_projection(view) if self.projection else view
            out["image_view_embeddings"] = _L2NormFn.apply(view_e)
        return out
