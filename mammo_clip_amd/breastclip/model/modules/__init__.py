"""Factories name -> encoder [ref: model/modules/__init__.py:11-89].  Only the encoders of the contrastive
pre-training configs are provided: 'tf_efficientnetv2-detect' (= EfficientNet-B2, out_dim 1408),
'tf_efficientnet_b5_ns-detect' (= EfficientNet-B5, out_dim 2048) and the HuggingFace BERT text encoder."""
from typing import Dict

from .efficientnet_custom import EfficientNet
from .projection import LinearProjectionHead, MLPProjectionHead
from .text_encoder import HuggingfaceTextEncoder


def load_image_encoder(config_image_encoder: Dict):
    src, name = config_image_encoder["source"].lower(), config_image_encoder["name"].lower()
    weights = config_image_encoder.get("weights_path")
    if src == "cnn" and name == "tf_efficientnetv2-detect":
        enc = EfficientNet.from_pretrained("efficientnet-b2", weights_path=weights, num_classes=1)
        enc.out_dim = 1408
    elif src == "cnn" and name == "tf_efficientnet_b5_ns-detect":
        enc = EfficientNet.from_pretrained("efficientnet-b5", weights_path=weights, num_classes=1)
        enc.out_dim = 2048
    else:
        raise KeyError(f"Not supported image encoder: {config_image_encoder}")
    # extension key (BASELINE config #5): fp8 (OCP e4m3, per-tensor scaled) operands for the late-stage 1x1 convolutions
    enc.set_fp8(bool(config_image_encoder.get("fp8", False)))
    # extension key: activation recompute mode of the MBConv blocks (memory of a kept graph vs backward work)
    enc.set_recompute(int(config_image_encoder.get("recompute", 0)))
    return enc


def load_text_encoder(config_text_encoder: Dict, vocab_size: int):
    if config_text_encoder["source"].lower() == "huggingface":
        return HuggingfaceTextEncoder(
            name=config_text_encoder["name"], vocab_size=vocab_size, pretrained=config_text_encoder["pretrained"],
            gradient_checkpointing=config_text_encoder.get("gradient_checkpointing", False),
            cache_dir=config_text_encoder.get("cache_dir", ""),
            trust_remote_code=config_text_encoder.get("trust_remote_code", False),
            config=config_text_encoder.get("config"))
    raise KeyError(f"Not supported text encoder: {config_text_encoder}")


def load_projection_head(embedding_dim: int, config_projection_head: Dict):
    name = config_projection_head["name"].lower()
    if name == "linear":
        return LinearProjectionHead(embedding_dim=embedding_dim, projection_dim=config_projection_head["proj_dim"])
    if name == "mlp":
        return MLPProjectionHead(embedding_dim, config_projection_head["proj_dim"], config_projection_head["dropout"])
    raise KeyError(f"Not supported text encoder: {config_projection_head}")
