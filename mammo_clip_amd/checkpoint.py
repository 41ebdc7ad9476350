"""Checkpoint I/O in the reference's ``.tar`` layout [ref: trainer.py:215-237, trainer_ddp.py:234-256]:

    {"model": state_dict, "optimizer": ..., "scheduler": ..., "config": cfg, "epoch": int, "train_loss": float}

The model ``state_dict`` keys / shapes / dtypes are the reference's (fp32, OIHW conv weights, ``transformers.BertModel``
names), so files written here load into the reference with ``strict=True`` and released Mammo-CLIP checkpoints load here.
Consumers downstream read ``ckpt["model"]`` [ref: Classifiers/models/breast_clip_classifier.py:13-17]."""
import os
import shutil
from typing import Dict, Optional

import torch


def save_checkpoint(path: str, model: torch.nn.Module, optimizer=None, scheduler=None, config: Optional[Dict] = None,
                    epoch: int = 0, train_loss: float = 0.0, best: bool = False, scaler=None) -> str:
    """``scaler``: an ``engine.LossScaler`` (f16 storage build) -- its state goes under the extra key ``"scaler"`` like the
    reference's GradScaler would have to be saved for an exact resume (the reference itself resumes model weights only,
    SURVEY.md appendix B 9; readers of the reference layout ignore the key)."""
    sd = {k: v.detach().to("cpu") for k, v in model.state_dict().items()}
    ckpt = {"model": sd,
            "optimizer": optimizer.state_dict() if optimizer is not None else None,
            "scheduler": scheduler.state_dict() if scheduler is not None else None,
            "config": config, "epoch": int(epoch), "train_loss": float(train_loss)}
    if scaler is not None:
        ckpt["scaler"] = scaler.state_dict()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(ckpt, path)
    if best:                                   # "<name>-best.tar" beside it, like the reference
        stem = path[:-len(".tar")] if path.endswith(".tar") else path
        shutil.copyfile(path, stem.rsplit("-epoch-", 1)[0] + "-best.tar")
    return path


def load_checkpoint(path: str, model: torch.nn.Module, optimizer=None, scheduler=None, strict: bool = True, scaler=None) -> Dict:
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt["model"] if "model" in ckpt else ckpt
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}     # DDP-wrapped saves
    model.load_state_dict(sd, strict=strict)
    if optimizer is not None and ckpt.get("optimizer") is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    if scheduler is not None and ckpt.get("scheduler") is not None:
        scheduler.load_state_dict(ckpt["scheduler"])
    if scaler is not None and ckpt.get("scaler") is not None:
        scaler.load_state_dict(ckpt["scaler"])
    return ckpt
