"""Tiny driver for PMC passes over the depthwise kernels: one launch per shape (run under rocprofv3 --pmc ...)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mammo_clip_amd  # noqa: F401
from mammo_clip_amd import ops

DEV = torch.device("cuda:0")
b = 32
import mammo_clip_amd.lib as L
lib = L.load()
# (shape, lane mode): 5x5 on the marching kernel (mode 0) AND on the lane = column kernel (mode 1) for the A/B of the counters
CASES = (((768, 3, 1, 95, 57), -1), ((240, 3, 1, 380, 228), -1), ((24, 3, 1, 760, 456), -1),
         ((384, 5, 1, 190, 114), 0), ((384, 5, 1, 190, 114), 1), ((1056, 5, 1, 95, 57), 0), ((1056, 5, 1, 95, 57), 1))
for (c, k, s, h, w), mode in CASES:
    lib.mc_dwconv_set_lane_mode(mode)
    oh, ow = (h + s - 1) // s, (w + s - 1) // s
    x = torch.ones(b * h * w, c, device=DEV, dtype=torch.bfloat16)
    wk = torch.ones(k * k, c, device=DEV)
    dy = torch.ones(b * oh * ow, c, device=DEV, dtype=torch.bfloat16)
    pad = (k - 1) // 2
    for _ in range(2):
        ops.dwconv_fwd(x, wk, b, h, w, c, k, s, pad, pad, oh, ow, pro=(torch.ones(c, device=DEV), torch.zeros(c, device=DEV)), stats=True)
        ops.dwconv_bwd_weight(x, dy, b, h, w, c, k, s, pad, pad, oh, ow)
    torch.cuda.synchronize()
    print("alg bytes", c, k, s, "lane mode", mode, 2 * c * b * (h * w + oh * ow))
