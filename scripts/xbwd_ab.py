#!/usr/bin/env python3
"""A/B of the fused expand-conv backward (ops.xbwd_rows: weight + data gradient from one pass over the upstream gradient) against
the two row-streaming launches it replaces, on the expand geometries of EfficientNet-B5 at 1520x912, 32 images.
usage: python scripts/xbwd_ab.py   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mammo_clip_amd  # noqa: F401,E402
from mammo_clip_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for (hw, n, k) in ((380 * 228, 144, 24), (380 * 228, 240, 40), (190 * 114, 384, 64)):
    for imgs in (32,):
        M = imgs * hw
        dy = torch.randn(M, n, device=DEV).to(ops.BF16)
        x = torch.randn(M, k, device=DEV).to(ops.BF16)
        w = (torch.randn(n, k, device=DEV) * k ** -0.5).to(ops.BF16)
        w_t = w.t().contiguous()
        r = torch.randn(M, k, device=DEV).to(ops.BF16)
        td = timeit(lambda: ops.linear_dgrad(dy, w, residual=r, w_t=w_t))
        tw = timeit(lambda: ops.linear_wgrad(dy, x))
        tf = timeit(lambda: ops.xbwd_rows(dy, x, w_t, residual=r))
        by_sep = 2.0 * M * (2 * n + 3 * k)
        by_f = 2.0 * M * (n + 3 * k)
        print(f"M={M:8d} N={n:3d} K={k:2d}: dgrad {td:6.3f} + wgrad {tw:6.3f} = {td + tw:6.3f} ms ({by_sep / (td + tw) / 1e9:5.2f} TB/s) -> fused {tf:6.3f} ms "
              f"({by_f / tf / 1e9:5.2f} TB/s)  x{(td + tw) / tf:4.2f}", flush=True)
        del dy, x, r
        torch.cuda.empty_cache()
