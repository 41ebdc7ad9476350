"""build_model [ref: model/__init__.py:10-21]; only the contrastive pre-training model is on the hot path."""
from typing import Dict

from torch import nn

from .clip import BreastClip


def build_model(model_config: Dict, loss_config: Dict, tokenizer=None) -> nn.Module:
    if model_config["name"].lower() == "clip_custom":
        return BreastClip(model_config, loss_config, tokenizer)
    raise KeyError(f"Not supported model: {model_config['name']}")
