#!/usr/bin/env python3
"""A/B of the fused expand + depthwise forward launch (mc_mbconv_xdw_fwd, conv_lane.hip MODE 4) against the two launches it
replaces (expand GEMM with the BatchNorm0 statistics epilogue + depthwise forward), per EfficientNet-B5 block shape at
32 images of 1520 x 912 (developer tool; run on the GPU box: python scripts/xdw_ab.py [n_images])."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_amd  # noqa: F401
from mammo_clip_amd import ops

DEV = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
# (block, k, s, h, w, cin, cexp, pad_l/pad_t)
SHAPES = [("b3", 3, 2, 760, 456, 24, 144, 0), ("b4-7", 3, 1, 380, 228, 40, 240, 1), ("b8", 5, 2, 380, 228, 40, 240, 1),
          ("b9-12", 5, 1, 190, 114, 64, 384, 2), ("b13", 3, 2, 190, 114, 64, 384, 1), ("b14-19", 3, 1, 95, 57, 128, 768, 1),
          ("b20", 5, 1, 95, 57, 128, 768, 2),
          ("B2b3-4@912", 3, 1, 228, 228, 24, 144, 1), ("B2b6-7@912", 5, 1, 114, 114, 48, 288, 2), ("B2b9-11@912", 3, 1, 57, 57, 88, 528, 1)]


def timeit(fn, reps=10):
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


print(f"{'block':12s} {'expand':>8s} {'dw':>8s} {'two':>8s} {'xdw':>8s} {'gram':>8s} {'xdw+gram':>9s}  GB/s(xdw: x + d)   max|d-d2|/max")
for name, k, s, h, w, cin, c, pad in SHAPES:
    oh, ow = (h + s - 1) // s, (w + s - 1) // s
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(N * h * w, cin, device=DEV, generator=g).to(ops.BF16)
    we = (torch.randn(c, cin, device=DEV, generator=g) * cin ** -0.5).to(ops.BF16)
    wk = torch.randn(k * k, c, device=DEV, generator=g) * 0.3
    sc, sh = torch.rand(c, device=DEV, generator=g) * 0.3 + 0.8, torch.randn(c, device=DEV, generator=g) * 0.3
    t_e = timeit(lambda: ops.linear_fwd(x, we, stats=True))
    e = ops.linear_fwd(x, we)
    t_d = timeit(lambda: ops.dwconv_fwd(e, wk, N, h, w, c, k, s, pad, pad, oh, ow, pro=(sc, sh), stats=True))
    d2 = ops.dwconv_fwd(e, wk, N, h, w, c, k, s, pad, pad, oh, ow, pro=(sc, sh))
    del e
    if not ops.mbconv_xdw_ok(N, h, w, cin, c, k, s, pad, pad, oh, ow):
        print(f"{name:12s} {t_e:8.3f} {t_d:8.3f} {t_e + t_d:8.3f}   (fused launch not supported)")
        continue
    t_x = timeit(lambda: ops.mbconv_xdw_fwd(x, we, (sc, sh), wk, N, h, w, c, k, s, pad, pad, oh, ow, stats=True))
    t_g = timeit(lambda: ops.bn_gram_partials(x, we, N * h * w)[0])
    d = ops.mbconv_xdw_fwd(x, we, (sc, sh), wk, N, h, w, c, k, s, pad, pad, oh, ow)
    err = float((d.float() - d2.float()).abs().max() / d2.float().abs().max())
    gbs = 2 * N * (h * w * cin + oh * ow * c) / (t_x * 1e-3) / 1e9
    print(f"{name:12s} {t_e:8.3f} {t_d:8.3f} {t_e + t_d:8.3f} {t_x:8.3f} {t_g:8.3f} {t_x + t_g:9.3f}  {gbs:8.0f}           {err:.2e}")
    del x, d, d2
    torch.cuda.empty_cache()
