"""Fused symmetric-InfoNCE evaluation on the HIP side, shared by both loss classes.

For every (local, gathered, weight) term:  logits = s * local[b,D] . gathered[W*b,D]^T  (fp32 GEMM),
mean cross-entropy against ``labels = arange(b) + rank*b`` and, in the same pass, d loss / d logits; the
embedding / scale gradients are accumulated immediately with two more small GEMMs, so autograd's backward is
a scalar multiply.  Each rank only ever forms its own b x W*b slab [ref: loss/breast_clip.py:46-100]."""
import torch

from ... import ops


class InfoNCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale, terms, label_offset, smoothing, k, *tensors):
        """tensors = k local [b,D] then k gathered [n,D] fp32 tensors, optionally followed by ``labels`` (int64 [b], the
        class index of each local row BEFORE the rank offset [ref: loss/breast_clip.py:43-44]; absent = arange(b));
        terms = [(li, gj, weight, slot)].  Returns (total, slots[4]).
        The gradient GEMMs run in the same pass as the loss, but only when something asks for a gradient
        (validation / no_grad calls evaluate the loss alone)."""
        labels = tensors[2 * k] if len(tensors) > 2 * k else None
        local = [t.contiguous() for t in tensors[:k]]
        allv = [t.contiguous() for t in tensors[k:2 * k]]
        if labels is not None:
            labels = labels.to(device=local[0].device, dtype=torch.int64).contiguous()
        b, D = local[0].shape
        n = allv[0].shape[0]
        dev = local[0].device
        need_grad = any(ctx.needs_input_grad)
        scale = scale.detach().reshape(1).contiguous().float()
        slots = torch.zeros(4, dtype=torch.float32, device=dev)
        gl = [None] * k          # unscaled accumulators: sum_terms dlogits . gathered
        ga = [None] * k          #                        sum_terms dlogits^T . local
        logits = torch.empty((b, n), dtype=torch.float32, device=dev)
        for (li, gj, w, slot) in terms:
            ops.sgemm(local[li], D, 1, allv[gj], 1, D, logits, n, b, n, D, alpha_dev=scale)
            ops.ce_fwd_bwd(logits, label_offset, w, slots[slot:slot + 1], smoothing, labels=labels)
            if not need_grad:
                continue
            if gl[li] is None:
                gl[li] = torch.zeros((b, D), dtype=torch.float32, device=dev)
            if ga[gj] is None:
                ga[gj] = torch.zeros((n, D), dtype=torch.float32, device=dev)
            ops.sgemm(logits, n, 1, allv[gj], D, 1, gl[li], D, b, D, n, beta=1.0)        # += dlogits @ all
            ops.sgemm(logits, 1, n, local[li], D, 1, ga[gj], D, n, D, b, beta=1.0)       # += dlogits.T @ local
        total = torch.zeros(1, dtype=torch.float32, device=dev)
        ones = torch.ones(4, dtype=torch.float32, device=dev)
        ops.sgemm(ones, 4, 1, slots, 1, 1, total, 1, 1, 1, 4)
        ctx.k = k
        ctx.has_labels = labels is not None
        if need_grad:
            dscale = torch.zeros(1, dtype=torch.float32, device=dev)
            for i in range(k):
                if gl[i] is not None:
                    ops.sgemm(gl[i], b * D, 1, local[i], 1, b * D, dscale, 1, 1, 1, b * D, beta=1.0)   # <G_i, local_i>
            ctx.grads = ([None if g is None else ops.scale_f32(g, scale) for g in gl],
                         [None if g is None else ops.scale_f32(g, scale) for g in ga], dscale)
        else:
            ctx.grads = None
        ctx.mark_non_differentiable(slots)
        return total.reshape(()), slots

    @staticmethod
    def backward(ctx, gtotal, _gslots):
        gl, ga, dscale = ctx.grads            # kept until the graph is freed (a second backward with retain_graph works)
        g = gtotal.reshape(1).contiguous().float()
        out_l = [None if t is None else ops.scale_f32(t, g) for t in gl]
        out_a = [None if t is None else ops.scale_f32(t, g) for t in ga]
        tail = (None,) if ctx.has_labels else ()
        return (ops.scale_f32(dscale, g).reshape(()), None, None, None, None) + tuple(out_l) + tuple(out_a) + tail
