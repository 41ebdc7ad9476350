"""BERT encoder (BioClinicalBERT = BERT-base-cased shape), functional torch-fp32 restatement.

The reference delegates to the third-party ``transformers.BertModel`` (model/modules/text_encoder.py:18-49;
pinned transformers==4.41.1 in environment.yml:208, 5.15.0 in this container) and returns
``last_hidden_state``.  This file restates the published BERT algorithm (Devlin et al. 2018; identical
arithmetic in both transformers versions): embeddings word+position+token_type -> LayerNorm(eps 1e-12)
-> dropout; 12 x [self-attention (scale 1/sqrt(64), additive padding mask, softmax, dropout, PV),
dense + dropout + residual + LayerNorm, dense 3072 + erf-GELU, dense + dropout + residual + LayerNorm].
The pooler is not used (text_encoder.py:49).  Dropout is identity here (eval, or p forced to 0).
Keys follow ``BertModel.state_dict()``.
"""
import math
from dataclasses import dataclass
from typing import Dict

import torch
import torch.nn.functional as F


# Storage-rounding emulation, see oracle/efficientnet.py (same switch semantics): "W" linear weights, "X" hidden state
# (embedding / LayerNorm outputs: GEMM operand AND residual), "XR" the residual-path copy of the hidden state only,
# "QKV", "PR" attention probabilities, "CTX", "AO" attention-output dense, "H1" FFN1 pre-activation, "HG" GELU output,
# "O" FFN2 output.
ROUND = None
ROUND_DTYPE = torch.bfloat16


ROUND_SR = None     # torch.Generator: STOCHASTIC rounding to bf16 instead of round-to-nearest (independent draws of the same noise)


def _q(tag, x):
    if ROUND is None or tag not in ROUND:
        return x
    if ROUND_SR is not None and ROUND_DTYPE == torch.bfloat16 and x.dtype == torch.float32:
        # unbiased random rounding: add uniform noise below the bf16 ulp, truncate the low 16 bits
        bits = x.detach().contiguous().view(torch.int32)
        noise = torch.randint(0, 1 << 16, bits.shape, generator=ROUND_SR, device=bits.device, dtype=torch.int32)
        r = ((bits + noise) & -65536).view(torch.float32)
        return x + (r - x).detach()
    return x + (x.to(ROUND_DTYPE).to(x.dtype) - x).detach()


@dataclass
class BertShape:
    vocab: int = 28996
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    inter: int = 3072
    max_pos: int = 512
    type_vocab: int = 2
    ln_eps: float = 1e-12


def embeddings(sd: Dict[str, torch.Tensor], p: str, input_ids, token_type_ids, cfg: BertShape):
    T = input_ids.shape[1]
    pos = torch.arange(T, device=input_ids.device)
    x = (sd[p + "embeddings.word_embeddings.weight"][input_ids]
         + sd[p + "embeddings.token_type_embeddings.weight"][token_type_ids]
         + sd[p + "embeddings.position_embeddings.weight"][pos][None])
    return _q("X", F.layer_norm(x, (cfg.hidden,), sd[p + "embeddings.LayerNorm.weight"],
                                sd[p + "embeddings.LayerNorm.bias"], cfg.ln_eps))


def layer(sd, p: str, x, ext_mask, cfg: BertShape, taps=None):
    """One encoder layer.  x [b,T,H]; ext_mask [b,1,1,T] additive (0 or finfo.min)."""
    b, T, H = x.shape
    nh, hd = cfg.heads, cfg.hidden // cfg.heads

    def lin(name, t):
        return F.linear(t, _q("W", sd[p + name + ".weight"]), sd[p + name + ".bias"])

    def split(t):
        return t.view(b, T, nh, hd).permute(0, 2, 1, 3)

    q, k, v = split(_q("QKV", lin("attention.self.query", x))), split(_q("QKV", lin("attention.self.key", x))), \
        split(_q("QKV", lin("attention.self.value", x)))
    scores = q @ k.transpose(-1, -2) / math.sqrt(hd) + ext_mask
    probs = _q("PR", torch.softmax(scores, dim=-1))
    ctx = _q("CTX", (probs @ v).permute(0, 2, 1, 3).reshape(b, T, H))
    a = _q("X", F.layer_norm(_q("AO", lin("attention.output.dense", ctx)) + _q("XR", x), (H,),
                             sd[p + "attention.output.LayerNorm.weight"],
                             sd[p + "attention.output.LayerNorm.bias"], cfg.ln_eps))
    h = _q("HG", F.gelu(_q("H1", lin("intermediate.dense", a))))          # exact erf GELU
    y = _q("X", F.layer_norm(_q("O", lin("output.dense", h)) + _q("XR", a), (H,), sd[p + "output.LayerNorm.weight"],
                             sd[p + "output.LayerNorm.bias"], cfg.ln_eps))
    if taps is not None:
        taps[p + "probs"] = probs
        taps[p + "attn_out"] = a
    return y


def forward(sd, tokens: Dict[str, torch.Tensor], cfg: BertShape, prefix: str = "", taps=None):
    """HuggingfaceTextEncoder.forward (text_encoder.py:47-49) -> last_hidden_state [b,T,H]."""
    ids, mask = tokens["input_ids"], tokens["attention_mask"]
    tt = tokens.get("token_type_ids")
    if tt is None:
        tt = torch.zeros_like(ids)
    x = embeddings(sd, prefix, ids, tt, cfg)
    ext = (1.0 - mask[:, None, None, :].to(x.dtype)) * torch.finfo(x.dtype).min
    for i in range(cfg.layers):
        x = layer(sd, f"{prefix}encoder.layer.{i}.", x, ext, cfg, taps)
    return x
