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
        self._plans = {}          # group index -> host-side launch plan (pointer table, step counters)

    # The per-parameter ``state["step"]`` tensors of torch.optim.AdamW are kept, but the hot loop counts in Python ints
    # and writes them back only when the state is looked at (state_dict) -- 700 tiny CPU tensor updates per step are
    # host time the small configurations cannot hide.
    def _sync_steps(self):
        # loss-scaled steps (step_loss_scaled): the host counts ATTEMPTED steps, the device counts the skipped ones
        skipped = int(self._ls_skipped.item()) if getattr(self, "_ls_skipped", None) is not None else 0
        for plan in self._plans.values():
            for p, t in zip(plan["params"], plan["steps"]):
                self.state[p]["step"].fill_(float(max(t - skipped, 0)))

    def _fold_skipped(self):
        """make the host counts the APPLIED counts and clear the device counter of skipped steps (before plans are rebuilt
        from ``state[p]["step"]``: a rebuilt plan must not subtract the same skips again)"""
        self._sync_steps()
        if getattr(self, "_ls_skipped", None) is not None:
            for plan in self._plans.values():
                plan["steps"][:] = [int(self.state[p]["step"]) for p in plan["params"]]
            self._ls_skipped.zero_()

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def _plan(self, gi, group):
        params = [p for p in group["params"] if p.grad is not None]
        plan = self._plans.get(gi)
        if plan is not None and len(plan["params"]) == len(params) and all(a is b for a, b in zip(plan["params"], params)):
            # the plan caches raw device pointers: it is only valid while every parameter / state tensor still lives
            # where it did (model.to(), p.data = ..., a state moved or replaced by external code all change data_ptr)
            if all(p.data_ptr() == ptrs[0] and self.state[p]["exp_avg"].data_ptr() == ptrs[1]
                   and self.state[p]["exp_avg_sq"].data_ptr() == ptrs[2] for p, ptrs in zip(params, plan["ptrs"])):
                return plan
        if getattr(self, "_ls_skipped", None) is not None:
            # a plan is created or rebuilt while the device counts skipped steps: fold them into the host counts FIRST -- the
            # new plan reads ``state[p]["step"]`` (applied counts) and must not have the same skips subtracted again, and a
            # group planned for the first time after some skips must not inherit them
            self._fold_skipped()
        steps = []
        arr = (L.AdamwTensor * max(len(params), 1))()
        for i, p in enumerate(params):
            if not p.is_cuda or p.dtype != torch.float32 or p.grad.is_sparse or not p.is_contiguous():
                raise L.MammoClipHipError("AdamW: parameters must be dense contiguous fp32 tensors on the GPU "
                                          "(the HIP kernel is the only path)")
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            steps.append(int(st["step"]))
            a = arr[i]
            a.param, a.exp_avg, a.exp_avg_sq, a.numel = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
        ptrs = [(a.param, a.exp_avg, a.exp_avg_sq) for a in arr[:len(params)]]
        plan = {"params": params, "steps": steps, "arr": arr, "gen": -1, "images": {}, "keep": [], "ptrs": ptrs}
        self._plans[gi] = plan
        return plan

    @torch.no_grad()
    def step_loss_scaled(self, scaler):
        """``step()`` under an ``engine.LossScaler`` without a host sync: the kernel reads the scaler's non-finite flag and
        does nothing on a bad step (GradScaler.step() would not call optimizer.step()); its bias corrections use the
        number of APPLIED steps = attempted steps (counted on the host) - skipped steps (counted on the device, in a tensor
        this optimizer owns: it survives a change of scaler).  Returns that counter for ``scaler.update``."""
        dev = self.param_groups[0]["params"][0].device
        if getattr(self, "_ls_skipped", None) is None or self._ls_skipped.device != dev:
            self._fold_skipped()                           # (a counter on another device: fold it into the host counts)
            self._ls_skipped = torch.zeros(1, dtype=torch.float32, device=dev)
        st = scaler.state(dev)
        self._ls = (st.data_ptr() + 4 * scaler._FLAG, self._ls_skipped.data_ptr())
        try:
            self.step()
        finally:
            self._ls = None
        return self._ls_skipped

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans = {}
        self._ls_skipped = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from .. import ops
        if getattr(self, "_ls", None) is None and getattr(self, "_ls_skipped", None) is not None:
            # a plain step() after loss-scaled ones: the kernel is handed the APPLIED count (no device counter in this call)
            self._fold_skipped()
            self._ls_skipped = None
        for gi, group in enumerate(self.param_groups):
            plan = self._plan(gi, group)
            params, arr, steps = plan["params"], plan["arr"], plan["steps"]
            if not params:
                continue
            if plan["gen"] != ops.cache_generation():          # bf16 images the kernel keeps current (see ops._cached)
                plan["images"] = ops.cached_cast_images(params)
                plan["gen"] = ops.cache_generation()
                for i, p in enumerate(params):
                    im = plan["images"].get(id(p))
                    arr[i].bf16_image = im[1].data_ptr() if im is not None else None
            keep = plan["keep"] = []
            for i, p in enumerate(params):
                g = p.grad
                if not g.is_contiguous():
                    g = g.contiguous()
                    keep.append(g)
                arr[i].grad = g.data_ptr()
                steps[i] += 1
            b1, b2 = group["betas"]
            stream = torch.cuda.current_stream().cuda_stream
            hyper = (float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]))
            ls = getattr(self, "_ls", None)
            entry, tail = ("mc_adamw_step_ls", (ls[0], ls[1], stream)) if ls else ("mc_adamw_step", (stream,))
            if min(steps) == max(steps):
                L.call(entry, arr, len(params), *hyper, steps[0], *tail)
            else:                                             # parameters that joined later: one launch set per step count
                for t in sorted(set(steps)):
                    idx = [i for i, s_ in enumerate(steps) if s_ == t]
                    sub = (L.AdamwTensor * len(idx))(*[arr[i] for i in idx])
                    L.call(entry, sub, len(idx), *hyper, t, *tail)
            # the kernel wrote through raw pointers: tell autograd (and the derived-weight-image cache in ops.py,
            # which keys on the version counter) that these tensors changed in place
            torch.autograd.graph.increment_version(params)
            ops.stamp_cast_images(plan["images"])              # ... and that the images it rewrote are current
        return loss


def build_optimizer(model: nn.Module, optim_config: Dict):
    name = optim_config["name"].lower()
    params = list(model.parameters())
    if name == "sgd":
        return torch.optim.SGD(params, **optim_config["config"])
    if name == "adamw":
        return AdamW(params, **optim_config["config"])      # HIP kernel only: .step() raises on CPU parameters
    raise NotImplementedError(f"Not implemented optimizer : {name}")
