"""Developer probe: does the late-stage backward chain (projection data gradient -> squeeze-excite sums -> BatchNorm1 + swish
backward) run faster per image when its tensors fit the 256 MiB Infinity Cache?  Runs the chain on n = 32 / 16 / 8 images of the
stage-6 and stage-7 shapes of EfficientNet-B5 at 1520x912; the depthwise output d rotates over several buffers (cold, as in the
model: written a forward ago), the data gradient dA1 is consumed right after its producer.  Prints us per image and launch.
usage: python scripts/mall_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mammo_clip_amd  # noqa: F401
from mammo_clip_amd import ops

DEV = torch.device("cuda:0")
SHAPES = [(1824, 304, 48 * 29, "stage 7"), (1056, 176, 95 * 57, "stage 6"), (768, 128, 95 * 57, "stage 5")]


def stats_for(c):
    st = ops.BNStats()
    st.mean, st.invstd = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    st.scale, st.shift, st.count = torch.ones(c, device=DEV), torch.zeros(c, device=DEV), 1.0
    return st


def probe(c, cout, hw, n, rounds=6):
    nbuf = max(2, int(1.2e9 // (2 * n * hw * c)))
    ds = [torch.randn(n * hw, c, device=DEV).to(torch.bfloat16) for _ in range(nbuf)]
    dp = torch.randn(n * hw, cout, device=DEV).to(torch.bfloat16)
    wp = (torch.randn(cout, c, device=DEV) * cout ** -0.5).to(torch.bfloat16)
    wp_t = wp.t().contiguous()
    st = stats_for(c)
    gamma = torch.ones(c, device=DEV)
    gate = torch.rand(n, c, device=DEV)
    dpooled = torch.randn(n, c, device=DEV)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(rounds)]
    for it in range(rounds + 2):
        d = ds[it % nbuf]
        e = ev[it - 2] if it >= 2 else None
        if e: e[0].record()
        da1 = ops.linear_dgrad(dp, wp, w_t=wp_t)
        if e: e[1].record()
        sums = ops.bnact_se_sums(d, da1, n, hw, c, st, 1)
        part1 = ops.bn_partials_from_se_sums(sums, gate, dpooled, 1.0 / hw)
        if e: e[2].record()
        dd, _, _ = ops.bnact_bwd(d, n, hw, c, st, gamma, 1, g=da1, mul=gate, add=dpooled, add_scale=1.0 / hw, partials=part1)
        if e: e[3].record()
        del da1, dd
    torch.cuda.synchronize()
    t = [sum(e[i].elapsed_time(e[i + 1]) for e in ev) / rounds * 1e3 for i in range(3)]
    return t


if __name__ == "__main__":
    print("# c cout hw | n | MB per expanded tensor | us per image: dgrad, se_sums(+partials), bn1 apply(+finalize) | sum")
    for (c, cout, hw, what) in SHAPES:
        for n in (32, 16, 8, 4):
            t = probe(c, cout, hw, n)
            print(f"{what} c={c:5d} cout={cout:4d} hw={hw:5d} | n={n:2d} | {2 * n * hw * c / 1e6:6.0f} MB | "
                  f"{t[0] / n:7.2f} {t[1] / n:7.2f} {t[2] / n:7.2f} | {sum(t) / n:7.2f}", flush=True)
            torch.cuda.empty_cache()
