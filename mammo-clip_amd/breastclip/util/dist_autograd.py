"""All-gather with autograd over RCCL/xGMI [ref: util/dist_autograd.py:5-27]: forward all_gather in rank order,
backward reduce_scatter(SUM) of the per-rank gradients.

MI355X note: the 4 embedding tensors of one step are gathered with ONE collective (``all_gather_fused`` below,
[4,b,D] in -> [W,4,b,D] out) instead of four latency-bound ones; xGMI is point-to-point, messages are <= 1 MiB."""
import torch
import torch.distributed as dist


def DistAutogradAllGatherFunction(partial=False):
    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, input):
            ctx.save_for_backward(input)
            output = [torch.zeros_like(input) for _ in range(dist.get_world_size())]
            dist.all_gather(output, input.contiguous())
            return tuple(output)

        @staticmethod
        def backward(ctx, *grads):
            (input,) = ctx.saved_tensors
            grad_out = torch.zeros_like(input)
            if partial:
                grad_out[:] = grads[dist.get_rank()]
            else:
                dist.reduce_scatter(grad_out, [g.contiguous() for g in grads], dist.ReduceOp.SUM)
            return grad_out

    return F


def _tensor_collectives_ok():
    """RCCL ("nccl") has native all_gather_into_tensor / reduce_scatter_tensor; the gloo backend used by the CPU
    tests gets the list forms of the same collectives."""
    return dist.get_backend() == "nccl"


class _FusedGather(torch.autograd.Function):
    """stacked [k,b,D] -> [W,k,b,D] with ONE all-gather; backward ONE reduce-scatter(SUM)."""

    @staticmethod
    def forward(ctx, stacked):
        W = dist.get_world_size()
        stacked = stacked.contiguous()
        out = torch.empty((W * stacked.shape[0],) + tuple(stacked.shape[1:]), dtype=stacked.dtype, device=stacked.device)
        if _tensor_collectives_ok():
            dist.all_gather_into_tensor(out, stacked)
        else:
            dist.all_gather(list(out.chunk(W, 0)), stacked)
        return out.view((W,) + tuple(stacked.shape))

    @staticmethod
    def backward(ctx, grad):
        W = grad.shape[0]
        grad = grad.contiguous()
        out = torch.empty(grad.shape[1:], dtype=grad.dtype, device=grad.device)
        if _tensor_collectives_ok():
            dist.reduce_scatter_tensor(out, grad.view((W * grad.shape[1],) + tuple(grad.shape[2:])), op=dist.ReduceOp.SUM)
        elif grad.is_cuda:
            # gloo with device tensors (single-GPU multi-process tests): no reduce_scatter -> all_reduce + own slice
            full = grad.clone()
            dist.all_reduce(full, op=dist.ReduceOp.SUM)
            out.copy_(full[dist.get_rank()])
        else:
            dist.reduce_scatter(out, [g.contiguous() for g in grad.unbind(0)], op=dist.ReduceOp.SUM)
        return out


def all_gather_fused(tensors):
    """list of k local [b,D] tensors -> list of k gathered [W*b,D] tensors (rank order), one collective."""
    stacked = torch.stack(tensors, 0)
    g = _FusedGather.apply(stacked)                       # [W,k,b,D]
    W, k, b, D = g.shape
    g = g.permute(1, 0, 2, 3).reshape(k, W * b, D)
    return [g[i] for i in range(k)]
