"""Developer timing of the stride-2 depthwise data gradient with the BatchNorm-backward epilogue (the B5 shapes, 32 images).
usage: python scripts/dw_s2_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mammo_clip_amd  # noqa: F401
from mammo_clip_amd import ops

DEV = torch.device("cuda:0")
n = 32
for (c, k, h, w) in ((144, 3, 760, 456), (240, 5, 380, 228), (384, 3, 190, 114), (1056, 5, 95, 57)):
    oh, ow = (h + 1) // 2, (w + 1) // 2
    pl = pt = (k - 2) // 2
    e = torch.randn(n * h * w, c, device=DEV).to(torch.bfloat16)
    dd = torch.randn(n * oh * ow, c, device=DEV).to(torch.bfloat16)
    wk = torch.randn(k * k, c, device=DEV)
    st = ops.BNStats()
    st.mean, st.invstd = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    st.scale, st.shift, st.count = torch.ones(c, device=DEV), torch.zeros(c, device=DEV), float(n * h * w)
    fn = lambda: ops.dwconv_bwd_data(dd, wk, n, h, w, c, k, 2, pl, pt, oh, ow, epi=(e, st))
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    by = 2.0 * c * n * (2 * h * w + oh * ow)
    print(f"s2 dgrad+epi k{k} c={c:5d} {h}x{w}  {ms:7.3f} ms  {by / ms / 1e6:7.1f} GB/s")
    del e, dd
