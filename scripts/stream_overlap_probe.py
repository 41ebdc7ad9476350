"""probe: do the two image views overlap usefully when their encoders run on two HIP streams? (timing only)"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mammo_clip_amd  # noqa
from mammo_clip_amd.breastclip.model import build_model
import bench

dev = torch.device("cuda:0")
torch.manual_seed(10)
model = build_model(bench.model_cfg("tf_efficientnet_b5_ns-detect"), bench.LOSS_CFG, types.SimpleNamespace(vocab_size=28996)).to(dev)
model.train()
enc = model.image_encoder
b, H, W = 32, 1520, 912
x1 = torch.randn(b, H, W, 3, device=dev).permute(0, 3, 1, 2)
x2 = torch.randn(b, H, W, 3, device=dev).permute(0, 3, 1, 2)

def seq():
    o1 = enc(x1); o2 = enc(x2)
    (o1.float().sum() + o2.float().sum()).backward()

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def par():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        o1 = enc(x1)
    with torch.cuda.stream(s2):
        o2 = enc(x2)
    cur.wait_stream(s1); cur.wait_stream(s2)
    (o1.float().sum() + o2.float().sum()).backward()

for name, fn in (("sequential", seq), ("two streams", par), ("sequential", seq), ("two streams", par)):
    for p in model.parameters(): p.grad = None
    fn(); torch.cuda.synchronize()
    for p in model.parameters(): p.grad = None
    t0 = time.perf_counter()
    for _ in range(2):
        fn()
        for p in model.parameters(): p.grad = None
    torch.cuda.synchronize()
    print(f"{name:12s} {(time.perf_counter() - t0) / 2 * 1e3:8.1f} ms per (2 views fwd+bwd)")
