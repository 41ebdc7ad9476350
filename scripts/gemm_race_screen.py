"""Race screen for the direct-to-LDS GEMM paths: random shapes (ragged M / N / K tails), every call repeated and compared
BIT for bit with its first result (a DMA / barrier ordering bug shows up as run-to-run differences) and against an fp32
torch reference.  Developer tool; run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_amd  # noqa
from mammo_clip_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
BF = torch.bfloat16
bad = 0
shapes = [(44544, 304, 1824), (44544, 512, 3072), (173280, 176, 1056), (8192, 768, 3072), (8192, 3072, 768), (1392 * 3, 1824, 304)]
rs = torch.Generator().manual_seed(7)
for _ in range(40):
    M = int(torch.randint(100, 30000, (1,), generator=rs)); N = int(torch.randint(9, 200, (1,), generator=rs)) * 8
    K = int(torch.randint(7, 260, (1,), generator=rs)) * 8
    shapes.append((M, N, K))
for (M, N, K) in shapes:
    x = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(BF)
    w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(BF)
    dy = (torch.randn(M, N, device=dev, generator=g) * 0.5).to(BF)
    res = {}
    for rep in range(6):
        y, part = ops.linear_fwd(x, w, stats=True)
        dx = ops.linear_dgrad(dy, w)
        dw = ops.linear_wgrad(dy, x)
        torch.cuda.synchronize()
        cur = dict(y=y, part=part, dx=dx, dw=dw)
        if rep == 0:
            res = {k: v.clone() for k, v in cur.items()}
            ry = x.float() @ w.float().T
            e1 = float((y.float() - ry).abs().max() / ry.abs().max())
            rdx = dy.float() @ w.float()
            e2 = float((dx.float() - rdx).abs().max() / rdx.abs().max())
            rdw = dy.float().T @ x.float()
            e3 = float((dw.float() - rdw).abs().max() / rdw.abs().max())
            if max(e1, e2) > 1e-2 or e3 > 3e-3:
                bad += 1; print("ACCURACY", (M, N, K), e1, e2, e3)
        else:
            for k in cur:
                if not torch.equal(cur[k], res[k]):
                    bad += 1; print("NONDETERMINISTIC", (M, N, K), k, rep, float((cur[k].float() - res[k].float()).abs().max()))
    del x, w, dy
print("shapes", len(shapes), "problems", bad)
