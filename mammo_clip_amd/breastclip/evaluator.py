"""Forward-only entry points of the reference's Evaluator [ref: evaluator.py:126-194]:
``encode_image`` / ``encode_text`` return L2-normalised projected embeddings as numpy arrays, ``zeroshot_scores`` is
the softmax over cosine similarities the zero-shot metrics are computed from (evaluator.py:171).  Dataset handling,
prompt tokenisation and the sklearn metrics around them stay with the caller (out of scope)."""
from typing import Dict, Optional

import numpy as np
import torch

from . import _tokens
from .model import build_model


class Evaluator:
    def __init__(self, model=None, ckpt_path: Optional[str] = None, tokenizer=None, device=None):
        """Either an already built model, or a reference-layout checkpoint (``{"model", "config", ...}``,
        trainer.py:215-237) whose ``config["model"]`` / ``config["loss"]`` rebuild it [ref: evaluator.py:24-27,52-58]."""
        self.device = torch.device("cuda") if device is None else torch.device(device)
        if model is None:
            ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
            self.ckpt_config = ckpt["config"]
            model = build_model(self.ckpt_config["model"], self.ckpt_config["loss"], tokenizer)
            sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in ckpt["model"].items()}
            model.load_state_dict(sd, strict=False)                       # evaluator.py:151 uses strict=False
        self.model = model.to(self.device).eval()

    @torch.no_grad()
    def encode_image(self, image: torch.Tensor) -> np.ndarray:
        self.model.eval()
        return self.model.encode_image_normalized(image.to(self.device)).float().cpu().numpy()

    @torch.no_grad()
    def encode_text(self, text_token: Dict) -> np.ndarray:
        if isinstance(text_token, (str, list)):
            raise TypeError("pass tokenised input ({'input_ids', 'attention_mask'}); tokenisation is the caller's")
        self.model.eval()
        m = self.model
        emb = m.encode_text(_tokens.to_device(text_token, self.device))
        emb = m.text_projection(emb) if m.projection else emb
        from .. import ops
        return ops.l2norm_fwd(emb.float().contiguous())[0].cpu().numpy()      # the HIP normalise kernel, like encode_image

    @staticmethod
    def zeroshot_scores(image_embeddings, text_embeddings) -> np.ndarray:
        """softmax over prompts of the cosine similarity [ref: evaluator.py:171].  numpy inputs (what encode_image /
        encode_text return, and what the reference computes on): host arithmetic like the reference; HIP tensors: the
        normalisation and the similarity matrix run on the device (l2norm + fp32 GEMM kernels of head.hip)."""
        if torch.is_tensor(image_embeddings) and image_embeddings.is_cuda:
            from .. import ops
            a, _ = ops.l2norm_fwd(image_embeddings.float().contiguous())
            b, _ = ops.l2norm_fwd(text_embeddings.float().contiguous().to(a.device))
            n, d = a.shape
            m = b.shape[0]
            st = torch.empty((n, m), dtype=torch.float32, device=a.device)
            ops.sgemm(a, d, 1, b, 1, d, st, m, n, m, d)                       # a @ b.T
            s = st.cpu().numpy()
        else:
            a = image_embeddings / np.linalg.norm(image_embeddings, axis=1, keepdims=True)
            b = text_embeddings / np.linalg.norm(text_embeddings, axis=1, keepdims=True)
            s = a @ b.T
        e = np.exp(s - s.max(axis=1, keepdims=True))
        return e / e.sum(axis=1, keepdims=True)
