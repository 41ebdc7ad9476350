"""build_optimizer [ref: optimizer/__init__.py:10-32].  The reference's ``no_decay`` branch is dead code
(``getattr`` on a dict is always ``[]``), so weight decay applies to EVERY parameter incl. logit_scale / BN / LayerNorm;
that behaviour is kept.

``AdamW`` below is the hot loop's optimizer step: the same hyper-parameters, update rule
and ``state_dict()`` layout as ``torch.optim.AdamW`` (state per parameter: ``step``, ``exp_avg``, ``exp_avg_sq``), so
optimizer checkpoints move both ways [ref: trainer.py:215-237 stores ``optimizer.state_dict()``], with the update done
by one multi-tensor HIP kernel per 48 parameters (``mc_adamw_step``)."""
from typing import Dict

import torch
from torch import nn

from .. import lib as L


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, **unused):
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by the reference configs")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from .. import ops
        for group in self.param_groups:
            by_step = {}
            images = ops.cached_cast_images([p for p in group["params"] if p.grad is not None])
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.is_sparse or not p.is_contiguous():
                    raise L.MammoClipHipError("AdamW: parameters must be dense contiguous fp32 tensors on the GPU "
                                              "(the HIP kernel is the only path)")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                by_step.setdefault(int(st["step"]), []).append((p, g, st["exp_avg"], st["exp_avg_sq"]))
            b1, b2 = group["betas"]
            stream = torch.cuda.current_stream().cuda_stream
            for t, items in by_step.items():
                arr = (L.AdamwTensor * len(items))()
                for i, (p, g, m, v) in enumerate(items):
                    a = arr[i]
                    a.param, a.grad, a.exp_avg, a.exp_avg_sq, a.numel = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
                    im = images.get(id(p))
                    a.bf16_image = im[1].data_ptr() if im is not None else None
                L.call("mc_adamw_step", arr, len(items), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                       float(group["weight_decay"]), t, stream)
                # the kernel wrote through raw pointers: tell autograd (and the derived-weight-image cache in ops.py,
                # which keys on the version counter) that these tensors changed in place
                torch.autograd.graph.increment_version([it[0] for it in items])
            ops.stamp_cast_images(images)      # the bf16 images written by the kernel are current for the new versions
        return loss


def build_optimizer(model: nn.Module, optim_config: Dict):
    name = optim_config["name"].lower()
    params = list(model.parameters())
    if name == "sgd":
        return torch.optim.SGD(params, **optim_config["config"])
    if name == "adamw":
        if all(p.is_cuda for p in params):
            return AdamW(params, **optim_config["config"])
        return torch.optim.AdamW(params, **optim_config["config"])     # host-side tests of the loop on CPU tensors
    raise NotImplementedError(f"Not implemented optimizer : {name}")
