"""how much does the last, partly filled 32-column strip of the depthwise kernels cost? (w = 224 = 7 strips vs w = 228 = 7.125)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_amd
from mammo_clip_amd import ops
DEV = torch.device("cuda:0")
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for (c, k, h, ws) in [(240, 3, 380, (224, 228, 256)), (384, 5, 190, (96, 114, 128)), (768, 5, 95, (32, 57, 64)), (1824, 5, 48, (29, 32))]:
    for w in ws:
        n = 32
        x = torch.randn(n * h * w, c, device=DEV).bfloat16()
        wk = torch.randn(k * k, c, device=DEV)
        sc, sh = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
        ms = t(lambda: ops.dwconv_fwd(x, wk, n, h, w, c, k, 1, (k - 1) // 2, (k - 1) // 2, h, w, pro=(sc, sh), stats=True))
        print(f"k{k} c={c} {h}x{w}: {ms:.3f} ms  {ms / (h * w) * 1e6:.3f} ns/pixel-column-of-{c}", flush=True)
        del x
