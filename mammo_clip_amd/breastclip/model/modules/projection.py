"""Projection heads [ref: model/modules/projection.py:4-29].  The linear head (the one every pre-training config
uses) runs as an fp32 HIP GEMM; its L2 normalisation lives in model/clip.py like in the reference."""
import torch
from torch import nn

from .... import ops


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous()
        m, k = x.shape
        n = w.shape[0]
        y = torch.empty((m, n), dtype=torch.float32, device=x.device)
        ops.sgemm(x, k, 1, w, 1, k, y, n, m, n, k, bias=b)          # y = x @ w.T + b
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        m, k = x.shape
        n = w.shape[0]
        dx = torch.empty_like(x)
        ops.sgemm(dy, n, 1, w, k, 1, dx, k, m, k, n)                 # dx = dy @ w
        dw = torch.empty_like(w)
        ops.sgemm(dy, 1, n, x, k, 1, dw, k, n, k, m)                 # dw = dy.T @ x
        ones = torch.ones((1, m), dtype=torch.float32, device=x.device)
        db = torch.empty((1, n), dtype=torch.float32, device=x.device)
        ops.sgemm(ones, m, 1, dy, n, 1, db, n, 1, n, m)              # db = sum_rows dy
        return dx, dw, db.view(n)


class LinearProjectionHead(nn.Module):
    def __init__(self, embedding_dim, projection_dim):
        super().__init__()
        self.projection = nn.Linear(embedding_dim, projection_dim)   # parameter container

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("mammo_clip_amd.LinearProjectionHead runs only on a HIP device (no CPU fallback)")
        return _LinearFn.apply(x.float(), self.projection.weight, self.projection.bias)


class MLPProjectionHead(nn.Module):
    """Present in the reference (projection.py:4-20) but unused by the pre-training configs; out of the hot path."""

    def __init__(self, embedding_dim, projection_dim, dropout):
        super().__init__()
        raise NotImplementedError("MLPProjectionHead is outside the accelerated hot path (SURVEY.md section 8a row P)")
