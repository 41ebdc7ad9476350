"""Pre-training loss of the configured Mammo-CLIP recipe [ref: loss/breast_clip.py:20-127]: four symmetric InfoNCE
pairings (I1T1, I2T1, I1T2, I2T2) + image-image (ICL) + text-text (TCL) over the all-gathered embeddings."""
import torch.nn as nn

from .. import util
from ..util.dist_autograd import all_gather_fused
from ._infonce import InfoNCEFn

IMG, TXT, TXT2, VIEW = 0, 1, 2, 3
S_I2T, S_T2I, S_I2I, S_T2T = 0, 1, 2, 3


def all_gather(tensors):
    """[ref: loss/breast_clip.py:10-17], fused: one RCCL all-gather for all embedding tensors of the step."""
    if util.GlobalEnv.get().world_size > 1:
        return all_gather_fused(tensors)
    return list(tensors)


def _log(is_train, pairs):
    env = util.GlobalEnv.get()
    w = env.summary_writer.train
    if is_train and w is not None:              # the reference raises here when no writer is set; optional here
        for tag, val in pairs():
            w.add_scalar(tag, val, env.summary_writer.global_step)


class BreastClip(nn.Module):
    def __init__(self, label_smoothing=0.0, i2i_weight=0.0, t2t_weight=0.0, loss_ratio=1.0):
        super().__init__()
        self.name = "contrastive"
        self.label_smoothing = label_smoothing
        self.loss_ratio = loss_ratio
        self.i2i_weight = i2i_weight
        self.t2t_weight = t2t_weight

    def forward(self, image_embeddings, text_embeddings, text_embeddings2, image_view_embeddings, labels, logit_scale,
                is_train, **kwargs):
        env = util.GlobalEnv.get()
        b = labels.size(0)
        local = [image_embeddings, text_embeddings, text_embeddings2, image_view_embeddings]
        gathered = all_gather(local)
        ls = self.label_smoothing if is_train else 0.0
        terms = [(IMG, TXT, 0.125, S_I2T), (VIEW, TXT, 0.125, S_I2T), (IMG, TXT2, 0.125, S_I2T), (VIEW, TXT2, 0.125, S_I2T),
                 (TXT, IMG, 0.125, S_T2I), (TXT, VIEW, 0.125, S_T2I), (TXT2, IMG, 0.125, S_T2I), (TXT2, VIEW, 0.125, S_T2I)]
        terms_nosmooth = []
        if self.i2i_weight:
            terms_nosmooth += [(IMG, VIEW, 0.5 * self.i2i_weight, S_I2I), (VIEW, IMG, 0.5 * self.i2i_weight, S_I2I)]
        if self.t2t_weight:
            terms_nosmooth += [(TXT2, TXT, 0.5 * self.t2t_weight, S_T2T), (TXT, TXT2, 0.5 * self.t2t_weight, S_T2T)]
        off = env.world_rank * b
        if ls == 0.0:
            total, slots = InfoNCEFn.apply(logit_scale, terms + terms_nosmooth, off, 0.0, 4, *local, *gathered, labels)
        else:                                   # ICL / TCL never use label smoothing [ref: breast_clip.py:86-100]
            t1, slots = InfoNCEFn.apply(logit_scale, terms, off, ls, 4, *local, *gathered, labels)
            total = t1
            if terms_nosmooth:
                t2, slots2 = InfoNCEFn.apply(logit_scale, terms_nosmooth, off, 0.0, 4, *local, *gathered, labels)
                total, slots = t1 + t2, slots + slots2
        self.last_terms = slots                 # weighted partial sums (device tensor; no host sync)
        _log(is_train, lambda: [
            ("loss/contrastive/steps_i2t", slots[S_I2T] * 2.0), ("loss/contrastive/steps_t2i", slots[S_T2I] * 2.0),
            ("loss/contrastive/steps_i2i", slots[S_I2I] / (self.i2i_weight or 1.0)),
            ("loss/contrastive/steps_t2t", slots[S_T2T] / (self.t2t_weight or 1.0)),
            ("params/logit_scale", logit_scale.detach()), ("params/temperature", 1.0 / logit_scale.detach())])
        return total
