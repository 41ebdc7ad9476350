"""no-graph train-mode forward of one 32-pair micro-batch (B5 @1520x912 + BERT T=256), N times: what the 25 graph-less first
passes of the N = 1 headline step cost (run under rocprofv3 --kernel-trace --stats / --pmc; MC_XDW=0|1 switches the fused
expand + depthwise forward launch of round 6, MC_STREAMS=0 puts every kernel alone on the GPU for the counter passes)"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import mammo_clip_amd
from mammo_clip_amd.breastclip.model import build_model
dev = torch.device("cuda:0")
model = build_model(bench.model_cfg("tf_efficientnet_b5_ns-detect"), bench.LOSS_CFG, types.SimpleNamespace(vocab_size=28996)).to(dev)
model.train()
batch = bench.synth_batch_gpu(32, 1520, 912, 256, dev, 1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
with torch.no_grad():
    model(batch, dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        model(batch, dev)
    e1.record()
    torch.cuda.synchronize()
print("forward ms", e0.elapsed_time(e1) / n)
