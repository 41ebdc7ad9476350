"""[ref: loss/combined_loss.py:6-29]"""
from typing import List

import torch.nn as nn


class CombinedLoss(nn.Module):
    def __init__(self, loss_list: List[nn.Module]):
        super().__init__()
        self.loss_list = loss_list

    def forward(self, **kwargs):
        loss_dict = dict()
        total_loss = 0.0
        for loss in self.loss_list:
            cur = loss(**kwargs)
            loss_dict[loss.name] = cur
            total_loss = total_loss + cur * loss.loss_ratio
        loss_dict["total"] = total_loss
        return loss_dict
