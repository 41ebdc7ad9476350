"""All-gather with autograd over RCCL/xGMI [ref: util/dist_autograd.py:5-27]: forward all_gather in rank order,
backward reduce_scatter(SUM) of the per-rank gradients.

MI355X note: the 4 embedding tensors of one step are gathered with ONE collective (``all_gather_fused`` below,
[4,b,D] in -> [W,4,b,D] out) instead of four latency-bound ones; xGMI is point-to-point, messages are <= 1 MiB."""
import torch
import torch.distributed as dist

MIN_TORCH = (2, 4)      # all_gather_into_tensor / reduce_scatter_tensor / ReduceOp.AVG on gloo host + device tensors


def require_tensor_collectives():
    """ONE collective code path on every backend needs ``all_gather_into_tensor``, ``reduce_scatter_tensor`` and
    ``all_reduce(op=AVG)`` from the process group (RCCL always has them; gloo since torch 2.4-ish, verified on 2.10 with
    scripts/gloo_cuda_probe.py).  Called once when the process group exists (engine.init_distributed / Trainer): an older
    torch fails HERE with a clear message instead of inside the first training step."""
    ver = tuple(int(v) for v in torch.__version__.split("+")[0].split(".")[:2])
    missing = [n for n in ("all_gather_into_tensor", "reduce_scatter_tensor") if not hasattr(dist, n)]
    if missing or not hasattr(dist.ReduceOp, "AVG") or ver < MIN_TORCH:
        raise RuntimeError(f"mammo_clip_amd needs torch >= {MIN_TORCH[0]}.{MIN_TORCH[1]} for its collectives "
                           f"(all_gather_into_tensor, reduce_scatter_tensor, all_reduce(AVG) on every backend); "
                           f"this is torch {torch.__version__}" + (f", missing {missing}" if missing else ""))


def _gather_ranks(x):
    """x [...] on every rank -> [W, ...] in rank order: ONE ``all_gather_into_tensor``.  The same call on every backend
    (RCCL, and gloo in the CPU / shared-GPU tests -- torch 2.10's gloo implements the tensor collectives for host and
    device tensors, probed with scripts/gloo_cuda_probe.py), so the world-size-2 tests execute exactly what RCCL will."""
    W = dist.get_world_size()
    x = x.contiguous()
    out = torch.empty((W,) + tuple(x.shape), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out.view(-1), x.reshape(-1))
    return out


def _reduce_scatter_ranks(grad):
    """grad [W, ...] on every rank -> sum over ranks of slice [rank]: ONE ``reduce_scatter_tensor`` (same on every backend)"""
    grad = grad.contiguous()
    out = torch.empty(grad.shape[1:], dtype=grad.dtype, device=grad.device)
    dist.reduce_scatter_tensor(out.view(-1), grad.view(-1), op=dist.ReduceOp.SUM)
    return out


class _FusedGather(torch.autograd.Function):
    """stacked [k,b,D] -> [W,k,b,D] with ONE all-gather; backward ONE reduce-scatter(SUM)."""

    @staticmethod
    def forward(ctx, stacked):
        return _gather_ranks(stacked)

    @staticmethod
    def backward(ctx, grad):
        return _reduce_scatter_ranks(grad)


def DistAutogradAllGatherFunction(partial=False):
    """Name-compatible factory of the reference's per-tensor gather [ref: util/dist_autograd.py:5-27]:
    ``F.apply(x)`` returns the tuple of the W ranks' tensors; backward sums every rank's gradient of this rank's slice
    (``partial=True``: only this rank's own gradient, no collective).  The product path uses ``all_gather_fused``."""

    class _PerTensorGather(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return tuple(_gather_ranks(x).unbind(0))

        @staticmethod
        def backward(ctx, *grads):
            if partial:
                return grads[dist.get_rank()].clone()
            return _reduce_scatter_ranks(torch.stack(grads, 0))

    return _PerTensorGather


def all_gather_fused(tensors):
    """list of k local [b,D] tensors -> list of k gathered [W*b,D] tensors (rank order), one collective."""
    stacked = torch.stack(tensors, 0)
    g = _FusedGather.apply(stacked)                       # [W,k,b,D]
    W, k, b, D = g.shape
    g = g.permute(1, 0, 2, 3).reshape(k, W * b, D)
    return [g[i] for i in range(k)]
