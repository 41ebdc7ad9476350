"""Developer A/B of the two depthwise kernel forms (marching vs lane = column) on every depthwise shape of EfficientNet-B5 at
1520x912, 32 images: forward with the BatchNorm+SiLU prologue + statistics, stride-1 data gradient with the BatchNorm-backward
epilogue, weight gradient.  Prints ms per launch for both forms -- the table behind conv.hip::use_lane_fwd / use_lane_bww.
usage: python scripts/dw_form_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mammo_clip_amd  # noqa: F401
import mammo_clip_amd.lib as L
from mammo_clip_amd import ops
from oracle import arch as oarch

DEV = torch.device("cuda:0")
lib = L.load()
n = 32


def timeit(fn, iters=4):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


arch = oarch.build_arch("efficientnet-b5")
chain = oarch.spatial_chain(arch, 1520, 912)
seen = set()
print("# k s c  in -> out      |  fwd+pro march / lane  |  dgrad+epi march / lane  |  wgrad march / lane  |  policy: dgrad+epi + wgrad -> fused backward   (ms per launch, 32 images)")
for blk in arch.blocks:
    (h, w), (oh, ow) = chain[blk.idx], chain[blk.idx + 1]
    key = (blk.cexp, blk.k, blk.s, h, w)
    if key in seen:
        continue
    seen.add(key)
    c, k, s = blk.cexp, blk.k, blk.s
    l, r, t, b = blk.pad
    x = torch.randn(n * h * w, c, device=DEV).to(torch.bfloat16)
    dy = torch.randn(n * oh * ow, c, device=DEV).to(torch.bfloat16)
    wk = torch.randn(k * k, c, device=DEV)
    sc, sh = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
    st = ops.BNStats()
    st.mean, st.invstd, st.scale, st.shift, st.count = torch.zeros(c, device=DEV), torch.ones(c, device=DEV), sc, sh, float(n * h * w)
    wflip = wk.flip(0).contiguous()
    res = []
    for mode in (0, 1):
        lib.mc_dwconv_set_lane_mode(mode)
        f = timeit(lambda: ops.dwconv_fwd(x, wk, n, h, w, c, k, s, l, t, oh, ow, pro=(sc, sh), stats=True))
        e = timeit(lambda: ops.dwconv_bwd_data(dy, wk, n, h, w, c, k, 1, l, t, oh, ow, w_kkc_flipped=wflip, epi=(x, st))) if s == 1 else float("nan")
        g = timeit(lambda: ops.dwconv_bwd_weight(x, dy, n, h, w, c, k, s, l, t, oh, ow, pro=(sc, sh)))
        res.append((f, e, g))
    lib.mc_dwconv_set_lane_mode(-1)
    (f0, e0, g0), (f1, e1, g1) = res
    fused = ""
    if s == 1 and ops.dwconv_bwd_fused_ok(n, h, w, c, k, s, l, t, oh, ow, force=True):
        # round 5: the two backward launches under the POLICY's form choice against the one fused launch (conv_lane.hip MODE 3)
        ep = timeit(lambda: ops.dwconv_bwd_data(dy, wk, n, h, w, c, k, 1, l, t, oh, ow, w_kkc_flipped=wflip, epi=(x, st)))
        gp = timeit(lambda: ops.dwconv_bwd_weight(x, dy, n, h, w, c, k, s, l, t, oh, ow, pro=(sc, sh)))
        fu = timeit(lambda: ops.dwconv_bwd_fused(dy, x, st, wflip, n, h, w, c, k, l, t, oh, ow))
        fused = f" | {ep:7.3f} + {gp:7.3f} = {ep + gp:7.3f} -> {fu:7.3f} {'F' if fu < ep + gp else ' '}"
    print(f"k{k} s{s} c={c:5d} {h:3d}x{w:3d}->{oh:3d}x{ow:3d} | {f0:7.3f} {f1:7.3f} {'L' if f1 < f0 else ' '} | {e0:7.3f} {e1:7.3f} {'L' if e1 < e0 else ' '} | {g0:7.3f} {g1:7.3f} {'L' if g1 < g0 else ' '}{fused}", flush=True)
    del x, dy
    torch.cuda.empty_cache()
