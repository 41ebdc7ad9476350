"""build_loss [ref: loss/__init__.py:9-28]"""
from typing import Dict

from .breast_clip import BreastClip
from .breast_clip_contrastive import BreastClip_contrastive
from .combined_loss import CombinedLoss


def build_loss(all_loss_config: Dict) -> CombinedLoss:
    loss_list = []
    for name in all_loss_config:
        cfg = all_loss_config[name]
        if cfg["loss_ratio"] == 0.0:
            continue
        if name == "breast_clip":
            loss_list.append(BreastClip(**cfg))
        elif name == "breast_clip_contrastive":
            loss_list.append(BreastClip_contrastive(**cfg))
        else:
            raise KeyError(f"Unknown loss: {name}")
    return CombinedLoss(loss_list)
