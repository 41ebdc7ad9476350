#!/usr/bin/env python3
"""Random-geometry screen of the round-6 fused launches against the launches they replace (developer tool, GPU box):
  * mc_mbconv_xdw_fwd (conv_lane.hip MODE 4) against expand GEMM + depthwise launch and against fp32 torch, with the reference's
    static paddings (symmetric for stride 1; (0,1) / (1,2)-style asymmetric left pads for stride 2), every supported cin / c;
  * mc_dwconv_bwd_fused with xw (MODE 5) against the same launch reading the stored e.
python scripts/xdw_fuzz.py [cases] [seed]"""
import os
import random
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import mammo_clip_amd  # noqa: F401
from mammo_clip_amd import ops

DEV = torch.device("cuda:0")
BF = ops.BF16
CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


bad = 0
done_f = done_b = 0
for case in range(CASES):
    k, s = rng.choice([3, 5]), rng.choice([1, 1, 2])
    n = rng.choice([1, 2, 3, 5, 9, 17])
    h, w = rng.randint(k, 90), rng.choice([rng.randint(k, 40), rng.randint(40, 140), rng.randint(140, 320)])
    cin = 8 * rng.randint(1, 16)
    c = 8 * rng.randint(1, 48)
    if s == 1:
        pl = pt = (k - 1) // 2
    else:
        pl, pt = rng.choice([((k - 2) // 2, (k - 2) // 2), ((k - 1) // 2, (k - 1) // 2)])
    oh, ow = (h + 2 * pt - k + (1 if s == 2 and pt == (k - 2) // 2 else 0)) // s + 1, (w + 2 * pl - k + (1 if s == 2 and pl == (k - 2) // 2 else 0)) // s + 1
    if oh < 1 or ow < 1:
        continue
    g = torch.Generator(device=DEV).manual_seed(case)
    x = torch.randn(n * h * w, cin, device=DEV, generator=g).to(BF)
    we = (torch.randn(c, cin, device=DEV, generator=g) * cin ** -0.5).to(BF)
    wk = torch.randn(k * k, c, device=DEV, generator=g) * 0.3
    pro = (torch.rand(c, device=DEV, generator=g) * 0.6 + 0.7, torch.randn(c, device=DEV, generator=g) * 0.3)
    tag = f"case {case}: k{k} s{s} n{n} {h}x{w} cin{cin} c{c} pad({pl},{pt}) out {oh}x{ow}"
    if ops.mbconv_xdw_ok(n, h, w, cin, c, k, s, pl, pt, oh, ow):
        done_f += 1
        y, part = ops.mbconv_xdw_fwd(x, we, pro, wk, n, h, w, c, k, s, pl, pt, oh, ow, stats=True)
        e16 = ops.linear_fwd(x, we)
        y2 = ops.dwconv_fwd(e16, wk, n, h, w, c, k, s, pl, pt, oh, ow, pro=pro)
        a0 = F.silu((x.float() @ we.float().T) * pro[0] + pro[1]).to(BF).float().view(n, h, w, c).permute(0, 3, 1, 2)
        pr, pb = (ow - 1) * s + k - w - pl, (oh - 1) * s + k - h - pt
        ref = F.conv2d(F.pad(a0, (pl, max(pr, 0), pt, max(pb, 0))), wk.t().contiguous().view(c, 1, k, k), None, s, 0, 1, c)[:, :, :oh, :ow]
        e1, e2 = rel(y.view(n, oh, ow, c).permute(0, 3, 1, 2), ref), rel(y, y2)
        st = part.double().sum(0)
        e3 = rel(st[0].float(), y.float().double().sum(0).float())
        ok = torch.isfinite(y.float()).all() and e1 <= 1e-2 and e2 <= 2e-2 and e3 <= 1e-4
        if not ok:
            bad += 1
            print("FWD MISMATCH", tag, f"vs torch {e1:.3e} vs two launches {e2:.3e} stats {e3:.3e}")
    if k == 3 and s == 1 and cin <= 64 and ops.dwconv_bwd_fused_ok(n, h, w, c, 3, 1, 1, 1, h, w, force=True, cin=cin):
        done_b += 1
        dd = torch.randn(n * h * w, c, device=DEV, generator=g).to(BF)
        e = ops.linear_fwd(x, we)
        ef = e.float()
        st = ops.BNStats()
        mean, var = ef.mean(0), ef.var(0, unbiased=False)
        st.mean, st.invstd = mean.contiguous(), (var + 1e-3).rsqrt().contiguous()
        st.scale = (pro[0] * st.invstd).contiguous()
        st.shift = (pro[1] - mean * st.scale).contiguous()
        st.count = float(n * h * w)
        wflip = wk.flip(0).contiguous()
        dz0, p0, dw0 = ops.dwconv_bwd_fused(dd, e, st, wflip, n, h, w, c, 3, 1, 1, h, w)
        dz1, p1, dw1 = ops.dwconv_bwd_fused(dd, None, st, wflip, n, h, w, c, 3, 1, 1, h, w, xw=(x, we))
        s0, s1 = p0.double().sum(0), p1.double().sum(0)
        ep = float(((s0 - s1).abs() / s0.abs().amax(dim=1, keepdim=True)).max())
        ok = rel(dz1, dz0) <= 1e-2 and float((dz1 == dz0).float().mean()) >= 0.97 and ep <= 3e-3 and rel(dw1, dw0) <= 3e-3
        if not ok:
            bad += 1
            print("BWD MISMATCH", tag, f"dz {rel(dz1, dz0):.3e} equal {float((dz1 == dz0).float().mean()):.4f} partials {ep:.3e} dw {rel(dw1, dw0):.3e}")
    torch.cuda.synchronize()
print(f"{done_f} fused-forward cases, {done_b} fused-backward cases, {bad} mismatches")
sys.exit(1 if bad else 0)
