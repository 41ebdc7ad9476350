"""Single-view InfoNCE variant, 0.75 * i2t + 0.25 * t2i [ref: loss/breast_clip_contrastive.py:19-59]."""
import torch.nn as nn

from .. import util
from ._infonce import InfoNCEFn
from .breast_clip import _log, all_gather


class BreastClip_contrastive(nn.Module):
    def __init__(self, label_smoothing=0.0, i2i_weight=0.0, t2t_weight=0.0, loss_ratio=1.0):
        super().__init__()
        self.name = "contrastive"
        self.label_smoothing = label_smoothing
        self.loss_ratio = loss_ratio
        self.i2i_weight = i2i_weight
        self.t2t_weight = t2t_weight

    def forward(self, image_embeddings, text_embeddings, labels, logit_scale, is_train, **kwargs):
        env = util.GlobalEnv.get()
        b = labels.size(0)
        local = [image_embeddings, text_embeddings]
        gathered = all_gather(local)
        ls = self.label_smoothing if is_train else 0.0
        total, slots = InfoNCEFn.apply(logit_scale, [(0, 1, 0.75, 0), (1, 0, 0.25, 1)], env.world_rank * b, ls, 2,
                                       *local, *gathered, labels)
        self.last_terms = slots
        _log(is_train, lambda: [("loss/contrastive/steps_i2t", slots[0] / 0.75), ("loss/contrastive/steps_t2i", slots[1] / 0.25)])
        return total
