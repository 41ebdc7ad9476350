#!/usr/bin/env python3
"""A/B of the 256x256 tile kernel (gemm256.hip) against the 128x128 kernel of gemm.hip on the model's plain NT shapes.
usage: python scripts/gemm256_bench.py            (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mammo_clip_amd  # noqa: F401,E402
from mammo_clip_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
SHAPES = [  # (batch, M, N, K, what)
    (1, 44544, 1824, 304, "expand 304->1824 @48x29 x32"), (1, 44544, 304, 1824, "dgrad/proj 1824->304"),
    (1, 44544, 3072, 512, "expand 512->3072"), (1, 44544, 512, 3072, "dgrad 3072->512"), (1, 44544, 2048, 512, "head"),
    (32, 1392, 304, 1824, "project (gated weights) x32 img"), (32, 5415, 176, 1056, "project 1056->176 x32 img"),
    (1, 173280, 176, 1056, "dgrad 1056->176"), (1, 173280, 128, 768, "dgrad 768->128"),
    (1, 173280, 1056, 176, "expand 176->1056 @95x57 x32"), (1, 173280, 768, 128, "expand 128->768 @95x57 x32"),
    (1, 8192, 2304, 768, "BERT qkv b32"), (1, 16384, 2304, 768, "BERT qkv b64"), (1, 16384, 768, 768, "BERT out b64"),
    (1, 16384, 3072, 768, "BERT ffn1 b64"), (1, 16384, 768, 3072, "BERT ffn2 b64"),
    (1, 4096, 4096, 4096, "4096^3"), (1, 8192, 8192, 8192, "8192^3"),
    (1, 44544, 512, 3072, "dgrad 3072->512 + residual"), (1, 173280, 176, 1056, "dgrad 1056->176 + residual"),
    (1, 173280, 128, 768, "dgrad 768->128 + residual")]


def run(mode, b, M, N, K, reps=20, res=False):
    os.environ["MC_GEMM_256"] = str(mode)
    g = torch.Generator(device=DEV).manual_seed(1)
    x = (torch.randn((b * M, K), generator=g, device=DEV)).to(torch.bfloat16)
    w = (torch.randn((b, N, K), generator=g, device=DEV) * K ** -0.5).to(torch.bfloat16)
    y = torch.empty((b * M, N), device=DEV, dtype=torch.bfloat16)
    r = torch.randn((b * M, N), generator=g, device=DEV).to(torch.bfloat16) if res else None

    def call():
        ops.gemm(x, w, y, M, N, K, K, K, N, batch=b, sA=(M * K, 0), sB=(N * K, 0), sC=(M * N, 0), R=r, ldr=N if res else 0)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, y


if __name__ == "__main__":
    print(f"{'shape':44s} {'128^2 us':>9s} {'TF/s':>7s} {'256^2 us':>9s} {'TF/s':>7s} {'GB/s(alg)':>9s}  maxdiff")
    for (b, M, N, K, what) in SHAPES:
        res = what.endswith("residual")
        t0, y0 = run(0, b, M, N, K, res=res)
        y0 = y0.float().clone()
        t1, y1 = run(2, b, M, N, K, res=res)
        fl = 2.0 * b * M * N * K
        by = 2.0 * b * (M * K + N * K + M * N)
        d = float((y1.float() - y0).abs().max())
        print(f"{what[:30]:30s} {b:2d}x{M:6d}x{N:4d}x{K:4d} {t0:9.1f} {fl/t0/1e6:7.0f} {t1:9.1f} {fl/t1/1e6:7.0f} {by/t1/1e3:9.0f}  {d:.3g}")
