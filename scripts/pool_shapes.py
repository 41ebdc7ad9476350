#!/usr/bin/env python3
"""Per-shape rate of the squeeze pass (mc_bnact_pool: per-image channel means of silu(bn1(d)); late blocks also store the
activation) on the EfficientNet-B5 depthwise outputs of 64 images (two views of a 32-pair micro-batch) at 1520 x 912.
Developer tool (GPU box): python scripts/pool_shapes.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_amd  # noqa: F401
from mammo_clip_amd import ops

DEV = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SHAPES = [("b0-2", 760 * 456, 48, 0), ("b3", 380 * 228, 144, 0), ("b4-7", 380 * 228, 240, 0), ("b8", 190 * 114, 240, 0), ("b9-12", 190 * 114, 384, 0),
          ("b13", 95 * 57, 384, 0), ("b14-20", 95 * 57, 768, 1), ("b21-26", 95 * 57, 1056, 1), ("b27", 48 * 29, 1056, 1), ("b28-36", 48 * 29, 1824, 1),
          ("b37-38", 48 * 29, 3072, 1)]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, hw, c, keep in SHAPES:
    x = torch.randn(N * hw, c, device=DEV).to(ops.BF16)
    sc, sh = torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.2
    t = timeit(lambda: ops.bnact_pool(x, N, hw, c, sc, sh, 1, keep_act=bool(keep)))
    gb = (4 if keep else 2) * N * hw * c / 1e9
    print(f"{name:8s} hw {hw:7d} c {c:5d} keep {keep}  {t * 1e3:8.1f} us  {gb / t:6.2f} TB/s")
