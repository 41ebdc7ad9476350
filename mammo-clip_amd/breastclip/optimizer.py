"""build_optimizer [ref: optimizer/__init__.py:10-32].  The reference's ``no_decay`` branch is dead code
(``getattr`` on a dict is always ``[]``), so weight decay applies to EVERY parameter incl. logit_scale / BN / LayerNorm;
that behaviour is kept.  The update itself stays torch.optim (host-side "next" row N2 in SURVEY.md section 8f)."""
from typing import Dict

import torch
from torch import nn


def build_optimizer(model: nn.Module, optim_config: Dict):
    name = optim_config["name"].lower()
    params = model.parameters()
    if name == "sgd":
        return torch.optim.SGD(params, **optim_config["config"])
    if name == "adamw":
        return torch.optim.AdamW(params, **optim_config["config"])
    raise NotImplementedError(f"Not implemented optimizer : {name}")
