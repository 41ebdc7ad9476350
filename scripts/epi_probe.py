import sys, os, torch
sys.path.insert(0, "/root/repo")
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
for (n_img, hw, cexp, cout) in [(32, 346560, 24, 24), (1, 32 * 346560, 24, 24), (320, 34656, 24, 24), (32, 346560, 48, 24), (32, 86640, 240, 40), (32, 86640, 144, 40)]:
    M = n_img * hw
    dp = torch.randn(M, cout, device=DEV).bfloat16(); d = torch.randn(M, cexp, device=DEV).bfloat16()
    wp_t = (torch.randn(cexp, cout, device=DEV) * 0.2).bfloat16()
    st = ops.BNStats(); st.mean = torch.zeros(cexp, device=DEV); st.invstd = torch.ones(cexp, device=DEV); st.scale = torch.ones(cexp, device=DEV); st.shift = torch.zeros(cexp, device=DEV); st.count = float(M)
    coef = torch.randn(3, cexp, device=DEV); gate = torch.rand(n_img, cexp, device=DEV); dpool = torch.randn(n_img, cexp, device=DEV)
    wp = wp_t.t().contiguous()
    a = t(lambda: ops.proj_dgrad_se_sums(dp, wp_t, d, st, n_img, hw))
    b = t(lambda: ops.proj_dgrad_bn_apply(dp, wp_t, d, st, coef, gate, dpool, 1.0 / hw, hw))
    da1 = ops.linear_dgrad(dp, wp, w_t=wp_t)
    c = t(lambda: ops.linear_dgrad(dp, wp, w_t=wp_t)); e = t(lambda: ops.bnact_se_sums(d, da1, n_img, hw, cexp, st, 1))
    print(f"n_img={n_img} hw={hw} c={cexp}: epi1 {a:.3f} ms  epi2 {b:.3f} ms | dgrad {c:.3f}  se_sums {e:.3f}", flush=True)
    del dp, d, da1
