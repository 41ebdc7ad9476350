"""Per-shape micro-benchmarks of the hot kernels on the EfficientNet-B5 @1520x912 layer shapes (GPU only).
Prints achieved algorithmic GB/s and TFLOP/s per shape so pathologies are visible."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import mammo_clip_amd  # noqa: F401
from mammo_clip_amd import ops
from oracle import arch as oarch

DEV = torch.device("cuda:0")
BF = torch.bfloat16


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


def main(b=8, which=("gemm", "dw", "bn")):
    arch = oarch.build_arch("efficientnet-b5")
    chain = oarch.spatial_chain(arch, 1520, 912)
    seen = set()
    print(f"# batch {b} images; columns: op shape ms GB/s TFLOP/s")
    for blk in arch.blocks:
        (h, w), (oh, ow) = chain[blk.idx], chain[blk.idx + 1]
        key = (blk.cin, blk.cexp, blk.cout, blk.k, blk.s, h, w)
        if key in seen:
            continue
        seen.add(key)
        M, M2 = b * h * w, b * oh * ow
        if "gemm" in which:
            shapes = []
            if blk.expand != 1:
                shapes.append(("expand", M, blk.cexp, blk.cin))
            shapes.append(("project", M2, blk.cout, blk.cexp))
            for (nm, m, n, k) in shapes:
                x = torch.randn(m, k, device=DEV).to(BF)
                wt = torch.randn(n, k, device=DEV).to(BF)
                dy = torch.randn(m, n, device=DEV).to(BF)
                by = 2 * (m * k + m * n + n * k)
                fl = 2 * m * n * k
                for kind, fn in (("fwd+stats", lambda: ops.linear_fwd(x, wt, stats=True)), ("fwd", lambda: ops.linear_fwd(x, wt)),
                                 ("dgrad", lambda: ops.linear_dgrad(dy, wt)), ("wgrad", lambda: ops.linear_wgrad(dy, x))):
                    ms = timeit(fn)
                    print(f"gemm {nm:7s} {kind:9s} M={m:9d} N={n:5d} K={k:5d} {ms:8.3f} ms {by/ms/1e6:8.1f} GB/s {fl/ms/1e9:8.1f} TF", flush=True)
                del x, wt, dy
        if "dw" in which:
            c, k, s = blk.cexp, blk.k, blk.s
            x = torch.randn(M, c, device=DEV).to(BF)
            wk = torch.randn(k * k, c, device=DEV)
            sc, sh = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
            dy = torch.randn(M2, c, device=DEV).to(BF)
            l, r, t, bb = blk.pad
            by = 2 * c * (M + M2)
            for kind, fn in (("fwd", lambda: ops.dwconv_fwd(x, wk, b, h, w, c, k, s, l, t, oh, ow)),
                             ("fwd+pro+st", lambda: ops.dwconv_fwd(x, wk, b, h, w, c, k, s, l, t, oh, ow, pro=(sc, sh), stats=True)),
                             ("bww+pro", lambda: ops.dwconv_bwd_weight(x, dy, b, h, w, c, k, s, l, t, oh, ow, pro=(sc, sh))),
                             ("bwd_gather", lambda: ops.dwconv_bwd_data(dy, wk, b, h, w, c, k, s, l, t, oh, ow))):
                ms = timeit(fn)
                print(f"dw k{k}s{s} {kind:10s} c={c:5d} {h}x{w} {ms:8.3f} ms {by/ms/1e6:8.1f} GB/s", flush=True)
            del x, dy
        if "bn" in which:
            for (site, m_img, c) in (("dw-site", oh * ow, blk.cexp), ("expand-site", h * w, blk.cexp)):
                if site == "expand-site" and blk.expand == 1:
                    continue
                x = torch.randn(b * m_img, c, device=DEV).to(BF)
                g = torch.randn(b * m_img, c, device=DEV).to(BF)
                gam, bet = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
                _, part = ops.linear_fwd(torch.randn(4096, 64, device=DEV).to(BF), torch.randn(c, 64, device=DEV).to(BF), stats=True)
                st = ops.bn_finalize(part, 4096, gam, bet, torch.zeros(c, device=DEV), torch.ones(c, device=DEV), 0.01, 1e-3, False)
                el = 2.0 * b * m_img * c
                for kind, fn, nt in (("pool", lambda: ops.bnact_pool(x, b, m_img, c, st.scale, st.shift, 1), 1),
                                     ("se_sums", lambda: ops.bnact_se_sums(x, g, b, m_img, c, st, 1), 2),
                                     ("bwd(red+apply)", lambda: ops.bnact_bwd(x, b, m_img, c, st, gam, 1, g=g), 5)):
                    ms = timeit(fn)
                    print(f"bn {site:11s} {kind:15s} c={c:5d} hw={m_img:7d} {ms:8.3f} ms {nt * el / ms / 1e6:8.1f} GB/s", flush=True)
                del x, g
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8, tuple(sys.argv[2:]) or ("gemm", "dw"))
