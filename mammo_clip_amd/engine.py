"""Data-parallel training step for the contrastive pre-training hot path (one process per GPU, RCCL over xGMI).

Reproduces the call order of the reference's hot loop [ref: trainer_ddp.py:279-308]: zero_grad(set_to_none) ->
forward -> loss -> backward -> AdamW step -> scheduler step, with DDP-average gradient semantics (mean over ranks of
the per-rank mean loss) -- but:
  * gradients are reduced in a few large flat fp32 buckets launched from post-accumulate hooks while backward is
    still running (xGMI is point-to-point: few large collectives, not hundreds of small ones),
  * the loss dict stays on the device (no per-step .cpu() sync),
  * BatchNorm running statistics are per-rank like in the reference between its buffer broadcasts; rank 0's buffers
    are what ``state_dict()`` saves, and ``sync_buffers`` (called by ``validate``) puts them on every rank wherever
    they are read -- the per-forward broadcast of DDP itself is not reproduced because training never reads them.
"""
import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import lib as L
from . import ops
from .breastclip.model.modules import efficientnet_custom as _encmod


class GradBuckets:
    """Flat fp32 gradient buckets filled in reverse registration order (the order backward produces them)."""

    def __init__(self, params: List[torch.nn.Parameter], bucket_bytes: int = 256 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.bucket_of, self.buckets, self.pending, self.handles = {}, [], [], []
        cur, cur_n = [], 0
        for p in reversed(self.params):
            cur.append(p)
            cur_n += -(-p.numel() // 64) * 64          # every slot starts 256-byte aligned (vector loads in AdamW)
            if cur_n * 4 >= bucket_bytes:
                self._close(cur, cur_n)
                cur, cur_n = [], 0
        if cur:
            self._close(cur, cur_n)
        self.seen = set()
        self.enabled = True            # False: gradients are collected by reduce_all() after several backward calls
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._hook)

    def _close(self, plist, n):
        dev = plist[0].device
        flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off, views = 0, []
        for p in plist:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += -(-p.numel() // 64) * 64
        idx = len(self.buckets)
        self.buckets.append((flat, plist, views))
        for p, v in zip(plist, views):
            self.bucket_of[p] = (idx, v)

    def begin(self):
        self.pending = [len(plist) for (_, plist, _) in self.buckets]
        self.handles = []
        self.seen = set()

    def _hook(self, p):
        if not self.enabled or not self.pending:        # not armed (backward outside Trainer.step): plain autograd
            return
        idx, v = self.bucket_of[p]
        flat = self.buckets[idx][0]
        v.copy_(p.grad)
        p.grad = v
        self.seen.add(p)
        self.pending[idx] -= 1
        if self.pending[idx] == 0:
            self._reduce(flat)

    def _reduce(self, flat):
        """mean over ranks: ``all_reduce(AVG)`` -- the same call over RCCL and over gloo (torch 2.10's gloo implements AVG
        for host and device tensors), so the world-size-2 tests run the collective the 8-GPU job runs"""
        self.handles.append(dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=True))

    def reduce_all(self):
        """Micro-batched steps run several backward calls per step, so the per-parameter hooks cannot tell when a
        gradient is final: with ``enabled = False`` the accumulated gradients are moved into the buckets here, after the
        last backward, and all buckets are reduced (no overlap with backward in that mode)."""
        for flat, plist, views in self.buckets:
            for p, v in zip(plist, views):
                if p.grad is None:
                    v.zero_()
                else:
                    v.copy_(p.grad)
                    p.grad = v
            self._reduce(flat)
        for h in self.handles:
            h.wait()
        self.pending = []

    def finish(self):
        """Parameters that got no gradient (e.g. the unused BERT pooler) keep grad None, like under the reference's
        DDP(find_unused_parameters=True); their bucket slots are zero-filled so the collective sizes stay static."""
        for idx, (flat, plist, views) in enumerate(self.buckets):
            if self.pending[idx] > 0:
                for p, v in zip(plist, views):
                    if p in self.seen:
                        continue
                    if p.grad is None:
                        v.zero_()
                    else:                   # gradient from an earlier backward of this step (micro-batched step: e.g.
                        v.copy_(p.grad)     # logit_scale, whose gradient comes from the loss backward alone)
                        p.grad = v
                self._reduce(flat)
        for h in self.handles:
            h.wait()
        self.pending = []


class LossScaler:
    """Dynamic loss scale of the f16 storage build: the policy of ``torch.cuda.amp.GradScaler`` with its default constants
    (init 65536, x 0.5 on a non-finite gradient with the optimizer step skipped, x 2 after 2000 clean steps in a row), which
    is what the reference trains under [ref: trainer.py:271-278, trainer_ddp.py:296-303].  f16 keeps 10 mantissa bits but
    only 5 exponent bits: at 32 x 1520 x 912 the activation gradients of the early stages are ~1e-8 and flush to zero
    without a scale (measured: gradient cosine -0.22 against the scaled run on _blocks.2._bn2.bias).

    Round 5: the step has NO host sync.  Scale, clean-step counter, non-finite flag and skip counter are one small device
    tensor (like GradScaler's own ``_scale`` / ``_growth_tracker``): the loss is multiplied by the device scalar, the unscale
    kernel reads it (``mc_grads_unscale_dev``), the AdamW kernel itself looks at the flag and does nothing on a bad step
    (``mc_adamw_step_ls``), ``mc_loss_scale_update`` applies the policy.  ``scale`` / ``skipped`` / ``last_skipped`` READ the
    device (they synchronise: for logging and tests, not for the hot loop); ``Trainer.step`` returns the device scalars."""
    _SCALE, _GOOD, _FLAG, _SKIPPED, _LAST = 0, 1, 2, 3, 4

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, dynamic=True):
        self.growth_factor, self.backoff_factor = float(growth_factor), float(backoff_factor)
        self.growth_interval, self.dynamic = int(growth_interval), bool(dynamic)
        self._host = [float(init_scale), 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]       # until the first use names the device
        self._state = None
        self._warned = False

    # ---- device state
    def state(self, device):
        if self._state is None or self._state.device != torch.device(device):
            host = self._host if self._state is None else self._state.cpu().tolist()
            self._state = torch.tensor(host, dtype=torch.float32, device=device)
        return self._state

    def scale_tensor(self, device):
        """0-dim device view of the current scale (what the loss is multiplied by)"""
        return self.state(device)[self._SCALE]

    def _read(self, idx):
        return float(self._state[idx].item()) if self._state is not None else self._host[idx]

    scale = property(lambda self: self._read(self._SCALE))                 # (synchronises)
    skipped = property(lambda self: int(self._read(self._SKIPPED)))       # steps skipped since construction (synchronises)
    last_skipped = property(lambda self: self._read(self._LAST) != 0.0)   # did the last finished step skip (synchronises)

    def state_dict(self):
        st = self._state.cpu().tolist() if self._state is not None else list(self._host)
        return {"scale": st[0], "growth_tracker": int(st[1]), "skipped": int(st[3]), "growth_factor": self.growth_factor,
                "backoff_factor": self.backoff_factor, "growth_interval": self.growth_interval, "dynamic": self.dynamic}

    def load_state_dict(self, sd):
        self.growth_factor, self.backoff_factor = float(sd["growth_factor"]), float(sd["backoff_factor"])
        self.growth_interval, self.dynamic = int(sd["growth_interval"]), bool(sd["dynamic"])
        dev = self._state.device if self._state is not None else None
        self._host = [float(sd["scale"]), float(sd["growth_tracker"]), 0.0, float(sd.get("skipped", 0)), 0.0, 0.0, 0.0, 0.0]
        self._state = None
        if dev is not None:
            self.state(dev)

    # ---- the three launches of a loss-scaled optimizer step
    def _grad_table(self, params):
        ps = [p for p in params if p.grad is not None]
        arr = (L.AdamwTensor * max(len(ps), 1))()
        keep = []
        for a, p in zip(arr, ps):
            g = p.grad
            if not g.is_contiguous() or g.dtype != torch.float32:
                raise L.MammoClipHipError("LossScaler: parameter gradients must be dense contiguous fp32 tensors")
            keep.append(g)
            a.grad, a.numel = g.data_ptr(), g.numel()
        return ps, arr, keep

    def unscale_(self, params, sync=False):
        """grad *= 1 / scale for every parameter gradient (one multi-tensor launch per 40 tensors), the non-finite flag goes
        to the device state.  ``sync=True`` (optimizers without a device-side skip): returns True if all gradients are finite."""
        ps, arr, _keep = self._grad_table(params)
        if not ps:
            return True
        st = self.state(ps[0].device)
        base = st.data_ptr()
        L.call("mc_grads_unscale_dev", arr, len(ps), base + 4 * self._SCALE, base + 4 * self._FLAG, ops._st())
        return float(st[self._FLAG].item()) == 0.0 if sync else None

    def update(self, opt_skipped=None):
        """GradScaler.update() on the device; ``opt_skipped``: the optimizer's own device counter of skipped steps"""
        if self._state is None:
            return
        L.call("mc_loss_scale_update", self._state.data_ptr(), opt_skipped.data_ptr() if opt_skipped is not None else None,
               self.growth_factor, self.backoff_factor, self.growth_interval, int(self.dynamic), ops._st())

    def check(self, log=print):
        """One host read of the state (call it every few hundred steps, not per step): warns about a scale that has
        collapsed -- a forward that overflows f16 permanently halves the scale at every step and skips every update while
        the LR schedule still advances."""
        sc, skipped = self.scale, self.skipped
        if sc < 1.0 and not self._warned:
            self._warned = True
            log(f"[mammo_clip_amd] loss scale collapsed to {sc:g} after {skipped} skipped steps: the f16 forward overflows; "
                "no optimizer update is being applied")
        return {"scale": sc, "skipped": skipped}


class Trainer:
    def __init__(self, model, loss_func, optimizer, scheduler=None, device=None, bucket_mb: int = 256,
                 overlap_micro: bool = False, keep_graphs: int = 1, grad_sink: bool = True, keep_recompute: Optional[int] = None,
                 stat_tapes: bool = True, loss_scale="auto"):
        self.model, self.loss_func, self.optimizer, self.scheduler = model, loss_func, optimizer, scheduler
        # loss scaling: "auto" = a dynamic LossScaler in the f16 storage build (the reference's GradScaler), none in the
        # bf16 build (fp32's exponent range); a number = that static scale; a LossScaler = yours; None = off
        if loss_scale == "auto" and os.environ.get("MC_LOSS_SCALE"):
            loss_scale = float(os.environ["MC_LOSS_SCALE"])          # a static scale for every "auto" Trainer of the process
        if loss_scale == "auto":
            # MC_LOSS_SCALE_INIT: initial value of the DYNAMIC scale of every "auto" Trainer (tests: low enough not to skip)
            init = float(os.environ.get("MC_LOSS_SCALE_INIT", 65536.0))
            loss_scale = LossScaler(init_scale=init) if L.STORAGE == "f16" else None
        elif isinstance(loss_scale, (int, float)):
            loss_scale = LossScaler(init_scale=float(loss_scale), dynamic=False) if float(loss_scale) != 1.0 else None
        self.scaler = loss_scale
        self.scaler_check_every = 500           # steps between the scaler's one host read (collapse warning)
        self.device = device
        # Gradient reduction: by default the flat buckets are all-reduced AFTER the last backward (reduce_all) -- the whole
        # exchange is 552 MB per step against >= 1.2 s of backward at 128 pairs per GPU, there is nothing worth hiding, and
        # that mode does not depend on autograd's hook order.  overlap_micro = True (needs grad_sink = False) launches each
        # bucket from the post-accumulate hooks of the last backward instead.
        self.grad_sink = bool(grad_sink)        # parameter gradients combined by multi-tensor adds (ops.GradSink)
        # micro-batched step: the re-forward of a micro-batch replays the BatchNorm statistics / pooled means its first
        # (graph-less) forward recorded instead of computing them again (efficientnet_custom.StatTape)
        self.stat_tapes = bool(stat_tapes) and os.environ.get("MC_STAT_TAPES", "1") != "0"
        self.overlap_micro = bool(overlap_micro) and not self.grad_sink
        self.keep_graphs = max(1, int(keep_graphs))   # micro-batched step: micro-batches forwarded once, graph kept
        # MBConv recompute mode for the KEPT graphs only (EfficientNet.set_recompute): a kept graph in mode 2 is 28 GB instead
        # of 100 GB, so more of them fit; the re-forwarded micro-batches (graph alive for one forward + backward only) keep
        # the model's own mode.  None = the model's mode everywhere.
        self.keep_recompute = keep_recompute
        self._sink = None                       # ops.GradSink of the step in flight (installed only around a backward call)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if self.world > 1:
            from .breastclip.util.dist_autograd import require_tensor_collectives
            require_tensor_collectives()        # clear error on a torch without the tensor collectives / AVG (no silent fallback)
        self.buckets = GradBuckets(list(model.parameters()), bucket_mb << 20) if self.world > 1 else None
        if self.world > 1:
            # identical replicas: rank 0's parameters and buffers everywhere, like torch DDP does at construction
            # [ref: trainer_ddp.py:134] -- only gradients are averaged afterwards
            sync_parameters_and_buffers(model)

    def step(self, batch: Dict, micro_batches: int = 1) -> Dict[str, torch.Tensor]:
        try:
            if micro_batches > 1:
                return self._step_micro(batch, micro_batches)
            return self._step_single(batch)
        except BaseException:
            self._abort_step()                  # no partial gradients / disarmed buckets survive into the next step
            raise

    def _abort_step(self):
        """a step that raised (OOM, a data error) leaves nothing behind: the sink with its partial gradients is dropped,
        the module-level hand-over point is cleared and the bucket hooks are re-armed"""
        self._sink = None
        ops.GRAD_SINK = None
        if self.buckets is not None:
            self.buckets.enabled = True
            self.buckets.pending = []
            self.buckets.handles = []

    def _step_single(self, batch: Dict) -> Dict[str, torch.Tensor]:
        self.model.train()
        self.optimizer.zero_grad(set_to_none=True)
        if self.buckets is not None:
            self.buckets.begin()
            self.buckets.enabled = not self.grad_sink
        outputs = self.model(batch, self.device)
        loss_dict = self.loss_func(**outputs, is_train=True)
        self._backward(lambda: self._seed(loss_dict["total"]).backward())
        self._grads_done(hooked=not self.grad_sink)
        self._optimizer_step()
        if self.scheduler is not None:
            self.scheduler.step()
        return self._result(loss_dict)

    def _seed(self, total):
        """the tensor the step's backward starts from: the loss, times the loss scale (a device scalar) when one is in use"""
        return total if self.scaler is None else total * self.scaler.scale_tensor(total.device)

    def _optimizer_step(self):
        """optimizer update; under a loss scale: unscale the (already rank-averaged) gradients first and skip the update
        if any of them is non-finite, like GradScaler.step() [ref: trainer_ddp.py:300-303] -- without a host sync when the
        optimizer is the HIP AdamW (the kernel reads the flag); any other optimizer: one flag read per step"""
        if self.scaler is None:
            self.optimizer.step()
            return
        params = list(self.model.parameters())
        if hasattr(self.optimizer, "step_loss_scaled"):
            self.scaler.unscale_(params)
            opt_skipped = self.optimizer.step_loss_scaled(self.scaler)
            self.scaler.update(opt_skipped)
        else:
            if self.scaler.unscale_(params, sync=True):
                self.optimizer.step()
            self.scaler.update()
        self._steps_done = getattr(self, "_steps_done", 0) + 1
        if self._steps_done % self.scaler_check_every == 0:
            self.scaler.check()

    def _result(self, loss_dict):
        out = {k: v.detach() for k, v in loss_dict.items()}
        if self.scaler is not None and self.scaler._state is not None:
            # device scalars (no sync): the scale the NEXT step will use and the skipped-step count so far (a skip
            # must be visible to the caller like scaler.get_scale() is in the reference's loop)
            # SNAPSHOTS (one small clone launch each, still no sync): a caller that stores them for logging must not see later
            # steps' update kernels overwrite them.  Note for callers that iterate over the dict: these two keys
            # are not loss terms.
            out["loss_scale"] = self.scaler._state[LossScaler._SCALE].clone()
            out["skipped_steps"] = self.scaler._state[LossScaler._SKIPPED].clone()
        return out

    def _backward(self, run):
        """one backward call; with the gradient sink the hand-written functions deliver their parameter gradients to it"""
        # encoder chains on side streams (model/clip.py MC_STREAMS): their backward nodes run on those streams; the gradients
        # they hand to the sink (or leave in .grad) are consumed on the current stream
        sides = ops.side_streams() if torch.cuda.is_available() else ()
        if sides:
            ops.fork_side()
            ops.FORKED += 1
        if not self.grad_sink:
            try:
                return run()
            finally:
                if sides:
                    ops.FORKED -= 1
                    ops.join_side()
        if self._sink is None:
            self._sink = ops.GradSink()
        # the sink is visible to the backward functions only while THIS backward call runs: a backward outside the Trainer
        # (or after a failed step) sees plain autograd semantics
        ops.GRAD_SINK = self._sink
        try:
            run()
        finally:
            ops.GRAD_SINK = None
            if sides:
                ops.FORKED -= 1
                ops.join_side()
        self._sink.flush()

    def _grads_done(self, hooked: bool):
        """after the last backward of a step: sink -> param.grad, then the data-parallel mean.  ``hooked``: the bucket
        hooks were armed during the backward that made the gradients final (they have launched the full buckets)."""
        if self._sink is not None:
            self._sink.finish()
            self._sink = None
        if self.buckets is not None:
            if hooked:
                self.buckets.finish()
            else:
                self.buckets.reduce_all()
            self.buckets.enabled = True


# ---------------------------------------------------------------------------------------- micro-batched step
def _split_batch(batch: Dict, k: int):
    """k equal slices along the batch dimension of every tensor (token dicts one level down)"""
    n = batch["images"].shape[0]
    assert n % k == 0, "the per-GPU batch must be divisible by the number of micro-batches"
    b = n // k

    def cut(v, i):
        if torch.is_tensor(v) or hasattr(v, "data") and hasattr(v, "mean"):          # tensors, ops.RawImages
            if torch.is_tensor(v):
                return v[i * b:(i + 1) * b]
            return type(v)(v.data[i * b:(i + 1) * b], v.mean, v.std)
        if isinstance(v, (list, tuple)):
            return v[i * b:(i + 1) * b]
        return {kk: cut(vv, i) for kk, vv in v.items()} if hasattr(v, "items") else v
    return [{key: cut(val, i) for key, val in batch.items()} for i in range(k)], b


def _rng_counters(model):
    """(image encoder, text model) call counters of the counter-based dropout / drop-connect seeds"""
    te = model.text_encoder.text_encoder if hasattr(model.text_encoder, "text_encoder") else model.text_encoder
    return model.image_encoder.rng, te


def _step_micro(self, batch: Dict, k: int) -> Dict[str, torch.Tensor]:
    """One optimizer step over a per-GPU batch that does not fit as one pass (global batch 1024):
    the batch is cut into k micro-batches and the contrastive loss is still taken over ALL embeddings of the step
    (and of all ranks) -- exactly the reference's semantics at k x as many data-parallel ranks, each with its own
    BatchNorm batch statistics [ref: trainer_ddp.py:134 DDP without SyncBN; loss/breast_clip.py:29-127].
      1. forward micro-batches 0 .. k-2 without a graph, keep only the embeddings (and the dropout seed counters);
         forward the last micro-batch normally (graph kept);
      2. loss over the concatenated embeddings -> d loss / d embeddings, d loss / d logit_scale, and -- in the same
         backward call -- the whole backward of the last micro-batch;
      3. re-run micro-batches 0 .. k-2 with the SAME seeds (counter-based masks: bit-identical forward) and with the
         BatchNorm running-stat update switched off, and back-propagate their slices of the embedding gradients; the
         gradient buckets are all-reduced from the hooks of the LAST backward (overlapped with it).
    Cost: k - 1 extra forwards per step (of k forwards + k backwards); activation memory: one micro-batch."""
    model = self.model
    enc_ = getattr(model, "image_encoder", None)
    # round 6: a graph-less forward runs the narrow-input MBConv blocks through the fused expand + depthwise launch (the
    # expanded tensor never exists); every other forward of a micro-batched step -- re-forwards AND kept graphs -- takes the same
    # launch (recompute mode 1 on those blocks: the expanded tensor is rebuilt by one GEMM in the backward), so all micro-batches
    # of a step see one arithmetic and a re-forward reproduces its first forward bit for bit
    xdw_modes = enc_.xdw_reforward_modes() if hasattr(enc_, "xdw_reforward_modes") else []
    try:
        return self._step_micro_body(batch, k)
    finally:
        for blk, mode_ in xdw_modes:
            blk.recompute = mode_


def _step_micro_body(self, batch: Dict, k: int) -> Dict[str, torch.Tensor]:
    model = self.model
    model.train()
    self.optimizer.zero_grad(set_to_none=True)
    if self.buckets is not None:
        self.buckets.begin()
        self.buckets.enabled = False
    mbs, b = _split_batch(batch, k)
    irng, trng = _rng_counters(model)
    keys = ("image_embeddings", "text_embeddings", "text_embeddings2", "image_view_embeddings")
    keep = min(self.keep_graphs, k)            # micro-batches whose graph is kept (activation memory: `keep` micro-batches)
    counters, parts, tapes = [], [], []
    set_tape = getattr(_encmod, "set_stat_tape", None) if self.stat_tapes else None
    with torch.no_grad():
        for mb in mbs[:k - keep]:
            counters.append((irng.calls, trng._calls))
            if set_tape is not None:               # record this micro-batch's BatchNorm statistics / pooled means (StatTape)
                tapes.append(_encmod.StatTape())
                set_tape(tapes[-1])
            try:
                out = model(mb, self.device)
            finally:
                if set_tape is not None:
                    set_tape(None)
            parts.append({kk: out[kk] for kk in keys if kk in out})
    # the LAST `keep` micro-batches are forwarded with their graph kept: they are back-propagated straight from the loss
    # and never forwarded again (k - keep extra forwards per step instead of k: at 4 micro-batches per GPU and keep = 1
    # that is 6 % of the step; keep = 2 needs the activations of two micro-batches, ~230 GB at 32 pairs each)
    lives = []
    enc = getattr(model, "image_encoder", None)
    base_modes = None
    if self.keep_recompute is not None and parts and hasattr(enc, "set_recompute"):
        base_modes = [blk.recompute for blk in enc._blocks]
        enc.set_recompute(self.keep_recompute)
    try:
        for mb in mbs[k - keep:]:
            out = model(mb, self.device)
            lives.append({kk: out[kk] for kk in keys if kk in out})
    finally:
        if base_modes is not None:                 # (the mode is read at forward time and travels with each graph)
            for blk, m_ in zip(enc._blocks, base_modes):
                blk.recompute = m_
    after = (irng.calls, trng._calls)
    leaf = {kk: torch.cat([p_[kk] for p_ in parts]).detach().requires_grad_(True) for kk in lives[0]} if parts else {}
    full = {kk: torch.cat(([leaf[kk]] if parts else []) + [lv[kk] for lv in lives]) for kk in lives[0]}
    n = next(iter(full.values())).shape[0]
    outputs = dict(full, labels=torch.arange(n, device=self.device), logit_scale=model.logit_scale.exp())
    if self.buckets is not None and self.overlap_micro and not parts:
        self.buckets.enabled = True            # every micro-batch kept: this is the only backward
    loss_dict = self.loss_func(**outputs, is_train=True)
    self._backward(lambda: self._seed(loss_dict["total"]).backward())   # d loss / d embeddings of the re-run micro-batches, full backward of the kept ones
    del out, lives, full, outputs
    bns = [m for m in model.modules() if hasattr(m, "track_update")]
    for m in bns:
        m.track_update = False
    try:
        for i, mb in enumerate(mbs[:k - keep]):
            irng.calls, trng._calls = counters[i]
            if self.buckets is not None and self.overlap_micro and i == k - keep - 1:
                # gradients become final during the LAST backward: each bucket is all-reduced as soon as its parameters
                # have accumulated their last contribution, overlapped with the rest of that backward
                self.buckets.enabled = True
            if set_tape is not None:               # the re-forward replays them: no statistics epilogues / finalize / squeeze passes
                set_tape(tapes[i].replay())
            try:
                out = model(mb, self.device)
            finally:
                if set_tape is not None:
                    set_tape(None)
            if set_tape is not None:               # (success path only: an assert in the finally would mask the real error)
                assert tapes[i].pos == len(tapes[i].items), "statistics tape out of step with the re-forward"
                tapes[i] = None
            ks = list(leaf)
            self._backward(lambda: torch.autograd.backward([out[kk] for kk in ks], [leaf[kk].grad[i * b:(i + 1) * b] for kk in ks]))
    finally:
        for m in bns:
            m.track_update = True
        irng.calls, trng._calls = after
    self._grads_done(hooked=self.overlap_micro)
    self._optimizer_step()
    if self.scheduler is not None:
        self.scheduler.step()
    return self._result(loss_dict)


Trainer._step_micro = _step_micro
Trainer._step_micro_body = _step_micro_body


def init_distributed():
    """torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) -> (rank, local_rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU.  (Developer override MC_DIST_BACKEND=gloo: lets the multi-rank flow -- launcher, parameter
    # broadcast, fused gather, gradient buckets -- be exercised with several ranks SHARING a device on a box with fewer
    # GPUs than ranks, which RCCL refuses; never used by the product path.)
    backend = os.environ.get("MC_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev and world > 1:
        raise RuntimeError(f"{world} ranks but {ndev} visible GPUs: RCCL needs one GPU per rank")
    device = torch.device(f"cuda:{local % max(ndev, 1)}")
    torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", init_method="env://", world_size=world, rank=rank, device_id=device)
        else:
            dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank)
    return rank, local, world, device


@torch.no_grad()
def _broadcast_flat(tensors, src: int):
    """one broadcast per dtype (few large collectives, not one per tensor)"""
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for ts in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.broadcast(flat, src)
        off = 0
        for t in ts:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


@torch.no_grad()
def sync_parameters_and_buffers(model, src: int = 0):
    """rank ``src``'s parameters AND buffers on every rank (what torch DDP does when it wraps a module)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    params = list(model.parameters())
    _broadcast_flat([p.data for p in params] + list(model.buffers()), src)
    torch.autograd.graph.increment_version(params)      # cached bf16 weight images key on the version counter


@torch.no_grad()
def sync_buffers(model, src: int = 0):
    """BatchNorm running statistics (and every other buffer) of rank ``src`` on all ranks, like the buffer broadcast
    torch DDP performs at the start of each forward in the reference [ref: trainer_ddp.py:134, DDP default
    ``broadcast_buffers=True``].  Training itself uses batch statistics, so this only matters where running statistics
    are READ: evaluation (``validate`` calls it) and checkpoints (rank 0 saves).  One collective per dtype."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    _broadcast_flat(list(model.buffers()), src)


@torch.no_grad()
def validate(model, loss_func, dataloader_dict: Dict, device=None, max_batches: int = 11) -> Dict[str, Dict[str, float]]:
    """Validation pass of the hot loop's caller [ref: trainer_ddp.py:346-409]: eval mode, ``is_train=False`` losses,
    mean over ranks per batch (the reference's all_reduce(SUM)/world, C5 -- here ONE collective per batch for the
    whole loss dict), accumulated per loss key and divided by ``len(dataloader)``.  Reference quirk kept: at most 11
    batches (``idx == 10: break``) are evaluated but the average still divides by the full loader length."""
    model.eval()
    sync_buffers(model)
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    out = {}
    for name, loader in dataloader_dict.items():
        keys = ["total"] + [l.name for l in loss_func.loss_list if l.name != "total"]
        acc = None
        for idx, batch in enumerate(loader):
            ld = loss_func(**model(batch, device), is_train=False)
            vec = torch.stack([ld[k].detach().float().reshape(()) for k in keys])
            if world > 1:
                dist.all_reduce(vec, op=dist.ReduceOp.SUM)
                vec = vec / world
            acc = vec if acc is None else acc + vec          # stays on the device: one host sync per dataset
            if idx + 1 >= max_batches:
                break
        n = max(len(loader), 1)
        vals = (acc / n).cpu().tolist() if acc is not None else [0.0] * len(keys)
        out[name] = dict(zip(keys, vals))
    return out
