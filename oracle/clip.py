"""BreastClip model forward, torch-fp32 restatement (oracle side).  Follows model/clip.py:46-114:
encode_image -> projection -> x/||x||; encode_text (BERT last_hidden_state -> eos row =
attention_mask.sum(-1)-1, clip.py:65-68) -> projection -> x/||x|| (no epsilon, clip.py:90-91);
``logit_scale`` is returned already exponentiated (clip.py:100)."""
from typing import Dict

import torch
import torch.nn.functional as F

from . import bert as obert
from . import efficientnet as oeff
from .arch import Arch


def encode_text(sd, tokens, cfg: obert.BertShape, pooling: str = "eos"):
    h = obert.forward(sd, tokens, cfg, prefix="text_encoder.text_encoder.")
    if pooling == "eos":
        idx = tokens["attention_mask"].sum(dim=-1) - 1
        return h[torch.arange(h.shape[0]), idx]
    if pooling == "bos":
        return h[:, 0]
    if pooling == "mean":
        m = tokens["attention_mask"].unsqueeze(-1).expand(h.size()).float()
        return (h * m).sum(1) / torch.clamp(m.sum(1), min=1e-9)
    raise NotImplementedError(pooling)


def _project_norm(sd, name, x):
    y = F.linear(x, sd[name + ".projection.weight"], sd[name + ".projection.bias"])
    return y / y.norm(dim=1, keepdim=True)


def forward(sd: Dict[str, torch.Tensor], batch: Dict, arch: Arch, bert_cfg: obert.BertShape,
            train: bool, new_buffers=None, taps=None):
    """BreastClip.forward (clip.py:80-114).  ``batch['images']``/``['image_views']`` are NCHW fp32."""
    out = {}
    img = oeff.forward(sd, batch["images"], arch, train, "image_encoder.", new_buffers, taps)
    txt = encode_text(sd, batch["text_tokens"], bert_cfg)
    out["image_embeddings"] = _project_norm(sd, "image_projection", img)
    out["text_embeddings"] = _project_norm(sd, "text_projection", txt)
    out["labels"] = torch.arange(img.shape[0])
    out["logit_scale"] = sd["logit_scale"].exp()
    if "text_tokens2" in batch and "image_views" in batch:
        txt2 = encode_text(sd, batch["text_tokens2"], bert_cfg)
        out["text_embeddings2"] = _project_norm(sd, "text_projection", txt2)
        # second view runs through the image encoder as its own BN batch (clip.py:108)
        view = oeff.forward(sd if new_buffers is None else {**sd, **new_buffers},
                            batch["image_views"], arch, train, "image_encoder.", new_buffers, None)
        out["image_view_embeddings"] = _project_norm(sd, "image_projection", view)
    return out
