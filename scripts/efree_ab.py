import sys, os
sys.path.insert(0, "/root/repo")
import torch
import mammo_clip_amd
from mammo_clip_amd import ops
DEV = torch.device("cuda:0")
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N = 32
for name, h, w, cin, c in (("b4-7", 380, 228, 40, 240), ("B2b3-4@912", 228, 228, 24, 144)):
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(N * h * w, cin, device=DEV, generator=g).to(ops.BF16)
    we = (torch.randn(c, cin, device=DEV, generator=g) * cin ** -0.5).to(ops.BF16)
    dd = torch.randn(N * h * w, c, device=DEV, generator=g).to(ops.BF16)
    wk = torch.randn(9, c, device=DEV, generator=g) * 0.3
    e = ops.linear_fwd(x, we)
    st = ops.BNStats()
    ef = e.float(); mean, var = ef.mean(0), ef.var(0, unbiased=False); del ef
    st.mean, st.invstd = mean.contiguous(), (var + 1e-3).rsqrt().contiguous()
    st.scale, st.shift, st.count = st.invstd.clone(), (-mean * st.invstd).contiguous(), float(N * h * w)
    wflip = wk.flip(0).contiguous()
    t_e = timeit(lambda: ops.dwconv_bwd_fused(dd, e, st, wflip, N, h, w, c, 3, 1, 1, h, w))
    t_x = timeit(lambda: ops.dwconv_bwd_fused(dd, None, st, wflip, N, h, w, c, 3, 1, 1, h, w, xw=(x, we)))
    t_g = timeit(lambda: ops.linear_fwd(x, we))
    print(f"{name}: fused bwd reading e {t_e:.3f} ms | e rows from x {t_x:.3f} ms | (expand GEMM rebuild {t_g:.3f} ms)")
