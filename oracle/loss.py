"""Contrastive losses, torch-fp32 restatement (oracle side).

``breast_clip``            -> loss/breast_clip.py:29-127  (4 x symmetric InfoNCE + ICL + TCL)
``breast_clip_contrastive`` -> loss/breast_clip_contrastive.py:28-59 (0.75 i2t + 0.25 t2i)
``combined``               -> loss/combined_loss.py:20-29

Multi-rank semantics (util/dist_autograd.py:5-27): every rank all-gathers the embeddings in rank
order, builds only its own ``b x W*b`` logits slab with ``labels = arange(b) + rank*b`` and averages
over its b rows.  Here the W ranks are simulated in one process: ``rank_loss`` takes the full
gathered tensors plus the rank index, so ``mean_r(rank_loss(r))`` and autograd through the shared
gathered tensors reproduce all_gather-forward / reduce_scatter(SUM)-backward exactly.
"""
from typing import Dict, List

import torch
import torch.nn.functional as F


def _ce(local, allv, scale, labels, smoothing=0.0):
    return F.cross_entropy(scale * local @ allv.T, labels, label_smoothing=smoothing)


def breast_clip_rank(all_img, all_txt, all_txt2, all_view, logit_scale, rank: int, b: int,
                     i2i_weight=1.0, t2t_weight=0.5, label_smoothing=0.0) -> Dict[str, torch.Tensor]:
    """loss/breast_clip.py:29-127 for one rank.  all_* are [W*b, D]; the rank's local rows are
    all_*[rank*b:(rank+1)*b]."""
    sl = slice(rank * b, (rank + 1) * b)
    img, txt, txt2, view = all_img[sl], all_txt[sl], all_txt2[sl], all_view[sl]
    labels = torch.arange(b, device=all_img.device) + rank * b
    s, ls = logit_scale, label_smoothing
    i2t = (_ce(img, all_txt, s, labels, ls) + _ce(view, all_txt, s, labels, ls)
           + _ce(img, all_txt2, s, labels, ls) + _ce(view, all_txt2, s, labels, ls)) / 4.0
    t2i = (_ce(txt, all_img, s, labels, ls) + _ce(txt, all_view, s, labels, ls)
           + _ce(txt2, all_img, s, labels, ls) + _ce(txt2, all_view, s, labels, ls)) / 4.0
    i2i = (_ce(img, all_view, s, labels) + _ce(view, all_img, s, labels)) / 2.0
    t2t = (_ce(txt2, all_txt, s, labels) + _ce(txt, all_txt2, s, labels)) / 2.0
    loss = (i2t + t2i) / 2.0 + i2i * i2i_weight + t2t * t2t_weight
    return {"loss": loss, "i2t": i2t, "t2i": t2i, "i2i": i2i, "t2t": t2t}


def contrastive_rank(all_img, all_txt, logit_scale, rank: int, b: int, label_smoothing=0.0):
    """loss/breast_clip_contrastive.py:28-59 for one rank."""
    sl = slice(rank * b, (rank + 1) * b)
    labels = torch.arange(b, device=all_img.device) + rank * b
    i2t = _ce(all_img[sl], all_txt, logit_scale, labels, label_smoothing)
    t2i = _ce(all_txt[sl], all_img, logit_scale, labels, label_smoothing)
    return {"loss": 0.75 * i2t + 0.25 * t2i, "i2t": i2t, "t2i": t2i}


def combined(named_losses: List[tuple]) -> Dict[str, torch.Tensor]:
    """loss/combined_loss.py:20-29.  named_losses = [(name, value, loss_ratio), ...]."""
    out, total = {}, 0.0
    for name, value, ratio in named_losses:
        out[name] = value
        total = total + value * ratio
    out["total"] = total
    return out
