"""Tiny driver for PMC passes over the round-6 launches (run under rocprofv3 --pmc ..., scripts/pmc_run.sh): the fused expand +
depthwise forward (MODE 4) beside the plain depthwise forward it contains, and the fused backward with its e rows formed from the
block input (MODE 5) beside MODE 3, at B5 block shapes, 32 images."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mammo_clip_amd  # noqa: F401
from mammo_clip_amd import ops
import mammo_clip_amd.lib as L

DEV = torch.device("cuda:0")
N = 32
lib = L.load()
for name, k, s, h, w, cin, c, pad in (("b4-7", 3, 1, 380, 228, 40, 240, 1), ("b9-12", 5, 1, 190, 114, 64, 384, 2), ("b3", 3, 2, 760, 456, 24, 144, 0)):
    oh, ow = (h + s - 1) // s, (w + s - 1) // s
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(N * h * w, cin, device=DEV, generator=g).to(ops.BF16)
    we = (torch.randn(c, cin, device=DEV, generator=g) * cin ** -0.5).to(ops.BF16)
    wk = torch.randn(k * k, c, device=DEV, generator=g) * 0.3
    sc, sh = torch.rand(c, device=DEV, generator=g) * 0.3 + 0.8, torch.randn(c, device=DEV, generator=g) * 0.3
    e = ops.linear_fwd(x, we)
    lib.mc_dwconv_set_lane_mode(1)                 # the lane = column form of the plain forward: the kernel MODE 4 is built on
    for _ in range(2):
        ops.dwconv_fwd(e, wk, N, h, w, c, k, s, pad, pad, oh, ow, pro=(sc, sh), stats=True)
        ops.mbconv_xdw_fwd(x, we, (sc, sh), wk, N, h, w, c, k, s, pad, pad, oh, ow, stats=True)
    lib.mc_dwconv_set_lane_mode(-1)
    if k == 3 and s == 1:
        dd = torch.randn(N * h * w, c, device=DEV, generator=g).to(ops.BF16)
        st = ops.BNStats()
        st.mean, st.invstd = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        st.scale, st.shift, st.count = sc, sh, float(N * h * w)
        wflip = wk.flip(0).contiguous()
        for _ in range(2):
            ops.dwconv_bwd_fused(dd, e, st, wflip, N, h, w, c, k, pad, pad, oh, ow)
            ops.dwconv_bwd_fused(dd, None, st, wflip, N, h, w, c, k, pad, pad, oh, ow, xw=(x, we))
    torch.cuda.synchronize()
    print("shape", name, "algorithmic bytes: dw", 2 * N * c * (h * w + oh * ow), "xdw", 2 * N * (cin * h * w + c * oh * ow))
    del x, e
    torch.cuda.empty_cache()
