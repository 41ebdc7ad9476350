#!/usr/bin/env python3
"""Split-K sweep of the 256 x 256 TN weight-gradient kernel (gemm256_tn.hip) on the model's weight-gradient shapes: time of
ops.gemm (tile kernel + split-K reduce) against the number of K splits.  The cost model (mc_gemm256_tn_splits) counts rounds and
K tiles; every split also writes and re-reads an fp32 [M, N] partial.   usage: python scripts/tn_split_sweep.py   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mammo_clip_amd  # noqa: F401,E402
from mammo_clip_amd import lib as L, ops  # noqa: E402

DEV = torch.device("cuda:0")
SHAPES = [(176, 1056, 173280), (1056, 176, 173280), (128, 768, 173280), (768, 128, 173280), (304, 1824, 44544), (1824, 304, 44544),
          (512, 3072, 44544), (3072, 512, 44544), (3072, 768, 16384), (768, 3072, 16384), (2304, 768, 16384), (768, 768, 16384),
          (512, 1824, 44544), (176, 768, 173280), (304, 1056, 44544), (2048, 512, 44544)]


def run(n_out, k_in, rows, splits, reps=10):
    g = torch.Generator(device=DEV).manual_seed(1)
    dy = torch.randn((rows, n_out), generator=g, device=DEV).to(ops.BF16)
    x = torch.randn((rows, k_in), generator=g, device=DEV).to(ops.BF16)
    dw = torch.empty((n_out, k_in), dtype=torch.float32, device=DEV)
    ws = torch.empty((splits, n_out, k_in), dtype=torch.float32, device=DEV) if splits > 1 else None

    def call():
        ops.gemm(dy, x, dw, n_out, k_in, rows, n_out, k_in, k_in, a_kmajor=1, b_kmajor=1, c_f32=1, splits=splits, splitk_ws=ws, kind="wgrad")
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if __name__ == "__main__":
    lib = L.load()
    for (m, n, k) in SHAPES:
        s0 = lib.mc_gemm256_tn_splits(m, n, k, 0)
        cands = sorted({s for s in (s0, s0 // 2, s0 // 3, s0 // 4, (s0 * 3) // 4, s0 * 2, 8, 16, 24, 32, 48, 64) if s >= 1 and k // (64 * s) >= 4})
        res = {s: run(m, n, k, s) for s in cands}
        best = min(res, key=res.get)
        by = 2.0 * k * (m + n)
        print(f"dW {m:5d} x {n:5d}  K = {k:6d}: model {s0:3d} splits {res[s0]:7.1f} us ({by / res[s0] / 1e6:5.2f} TB/s) | best {best:3d} {res[best]:7.1f} us ({by / res[best] / 1e6:5.2f} TB/s) | "
              + " ".join(f"{s}:{t:.0f}" for s, t in res.items()), flush=True)
