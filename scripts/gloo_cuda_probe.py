"""Developer probe: which tensor collectives does gloo accept for DEVICE tensors (two ranks sharing one GPU)?
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 scripts/gloo_cuda_probe.py"""
import torch
import torch.distributed as dist

dist.init_process_group("gloo")
r, W = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0")
res = {}
for name, fn in (
    ("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(torch.empty(W * 6, device=dev), torch.ones(6, device=dev) * r)),
    ("reduce_scatter_tensor", lambda: dist.reduce_scatter_tensor(torch.empty(6, device=dev), torch.ones(W * 6, device=dev), op=dist.ReduceOp.SUM)),
    ("all_reduce AVG", lambda: dist.all_reduce(torch.ones(4, device=dev) * (r + 1), op=dist.ReduceOp.AVG)),
):
    try:
        fn()
        torch.cuda.synchronize()
        res[name] = "ok"
    except Exception as e:  # noqa: BLE001
        res[name] = f"FAILS: {type(e).__name__}: {str(e)[:120]}"
if r == 0:
    for k, v in res.items():
        print(f"gloo + device tensors: {k}: {v}")
