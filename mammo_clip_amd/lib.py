"""ctypes binding of libmammoclip_hip.so (the C ABI declared in include/mammoclip_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, an exception is
raised.  The product never routes through a CPU / eager-PyTorch path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# 16-bit storage / MFMA operand type of the process: "bf16" (default) or "f16" (MC_STORAGE=f16, the reference's AMP dtype
# [ref: trainer.py:271-278]).  It is a property of the BUILD of the kernel library (csrc/common_hip.h, -DMC_F16): both
# libraries export the same C ABI, the process loads exactly one of them and ops.BF16 names the matching torch dtype.
STORAGE = os.environ.get("MC_STORAGE", "bf16").lower()
if STORAGE not in ("bf16", "f16"):
    raise ValueError(f"MC_STORAGE must be 'bf16' or 'f16', got {STORAGE!r}")
LIB_PATH = os.path.join(_HERE, "lib", "libmammoclip_hip.so" if STORAGE == "bf16" else "libmammoclip_hip_f16.so")

P, LL, I, F, D, U, ULL = C.c_void_p, C.c_longlong, C.c_int, C.c_float, C.c_double, C.c_uint, C.c_ulonglong


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", P), ("B", P), ("C", P),
        ("M", LL), ("K", LL), ("N", I),
        ("lda", LL), ("ldb", LL), ("ldc", LL),
        ("a_kmajor", I), ("b_kmajor", I),
        ("c_f32", I), ("c_atomic", I),
        ("splits", I),
        ("batch", I), ("nb2", I),
        ("sA1", LL), ("sA2", LL), ("sB1", LL), ("sB2", LL), ("sC1", LL), ("sC2", LL),
        ("bias", P), ("bias_stride1", LL),
        ("act", I),
        ("R", P), ("ldr", LL),
        ("alpha", F),
        ("pro_operand", I),
        ("pro_scale", P), ("pro_shift", P), ("pro_gate", P),
        ("pro_rows_per_img", LL), ("pro_nch", I),
        ("stat_partials", P),
        ("max_grid_m", I),
        ("splitk_ws", P),
        ("split_group_rows", LL), ("split_sub", I), ("split_scale", P),
        ("ab_fp8", I), ("alpha_dev", P),
    ]


class GemmRowsArgs(C.Structure):
    _fields_ = [
        ("X", P), ("M", LL), ("K", I), ("ldx", LL),
        ("W", P), ("N", I), ("ldw", LL),
        ("C", P), ("ldc", LL),
        ("R", P), ("ldr", LL),
        ("pro_scale", P), ("pro_shift", P), ("pro_gate", P), ("pro_rows_per_img", LL),
        ("stat_partials", P), ("bias", P),
        ("epi_mode", I), ("epi_x", P), ("epi_ldx", LL), ("epi_rows_per_img", LL),
        ("epi_scale", P), ("epi_shift", P), ("epi_mean", P), ("epi_invstd", P),
        ("epi_coef", P), ("epi_mul", P), ("epi_add", P), ("epi_add_scale", C.c_float),
        ("epi_sums", P), ("epi_ws", P),
    ]


class WgradRowsArgs(C.Structure):
    _fields_ = [
        ("dY", P), ("N", I), ("lddy", LL),
        ("X", P), ("K", I), ("ldx", LL),
        ("M", LL), ("dW", P), ("ws", P), ("accumulate", I),
        ("pro_scale", P), ("pro_shift", P), ("pro_gate", P), ("pro_rows_per_img", LL),
    ]


class DwconvArgs(C.Structure):
    _fields_ = [
        ("x", P), ("dy", P), ("out", P), ("w_kkc", P),
        ("n", I), ("h", I), ("w", I), ("c", I),
        ("k", I), ("stride", I), ("pad_l", I), ("pad_t", I), ("oh", I), ("ow", I),
        ("pro_scale", P), ("pro_shift", P), ("stat_partials", P),
        ("epi_x", P), ("epi_scale", P), ("epi_shift", P), ("epi_mean", P), ("epi_invstd", P),
        ("dw_out", P), ("stat_rows", I),
        ("xw", P), ("cin", I),
    ]


class BnactArgs(C.Structure):
    _fields_ = [
        ("x", P), ("n_img", LL), ("hw", LL), ("c", I),
        ("scale", P), ("shift", P), ("act", I),
        ("rowscale", P), ("res", P), ("out", P),
        ("pooled", P),
        ("g", P), ("mul", P), ("add", P), ("add_scale", F), ("mean", P), ("invstd", P),
        ("partials", P), ("coef", P), ("dx", P), ("dgate", P), ("split_ws", P),
    ]


class AdamwTensor(C.Structure):
    _fields_ = [("param", P), ("grad", P), ("exp_avg", P), ("exp_avg_sq", P), ("numel", LL), ("bf16_image", P)]


_SIGS = {
    "mc_version": ([], I),
    "mc_storage_is_f16": ([], I),
    "mc_adamw_step": ([C.POINTER(AdamwTensor), I, D, D, D, D, D, LL, P], I),
    "mc_grads_unscale": ([C.POINTER(AdamwTensor), I, F, P, P], I),
    "mc_grads_unscale_dev": ([C.POINTER(AdamwTensor), I, P, P, P], I),
    "mc_adamw_step_ls": ([C.POINTER(AdamwTensor), I, D, D, D, D, D, LL, P, P, P], I),
    "mc_loss_scale_update": ([P, P, F, F, I, I, P], I),
    "mc_gemm_bf16": ([C.POINTER(GemmArgs), P], I),
    "mc_gemm_stat_rows": ([C.POINTER(GemmArgs)], I),
    "mc_gemm_tile_config": ([C.POINTER(GemmArgs)], I),
    "mc_gemm256_tn_eligible": ([C.POINTER(GemmArgs)], I),
    "mc_gemm256_tn_splits": ([LL, LL, LL, LL], I),
    "mc_amax_bf16": ([P, LL, P, P], I),
    "mc_quant_fp8_bf16": ([P, LL, P, P, P, P, P], I),
    "mc_gemm_rows_supported": ([I, I], I),
    "mc_gemm_rows_blocks": ([C.POINTER(GemmRowsArgs)], I),
    "mc_gemm_rows_epi_ws_floats": ([C.POINTER(GemmRowsArgs)], LL),
    "mc_gemm_rows_epi_supported": ([LL, I, I, LL, I], I),
    "mc_gemm_rows_bf16": ([C.POINTER(GemmRowsArgs), P], I),
    "mc_cast_transpose_f32_bf16": ([P, P, I, I, P], I),
    "mc_wgrad_rows_supported": ([I, I], I),
    "mc_wgrad_rows_blocks": ([LL], I),
    "mc_wgrad_rows_bf16": ([C.POINTER(WgradRowsArgs), P], I),
    "mc_xbwd_rows_supported": ([I, I], I),
    "mc_xbwd_rows_blocks": ([LL], I),
    "mc_xbwd_rows_bf16": ([C.POINTER(WgradRowsArgs), P, LL, P, LL, P, LL, P], I),
    "mc_cast_f32_bf16": ([P, P, LL, P], I),
    "mc_cast_bf16_f32": ([P, P, LL, P], I),
    "mc_cast_f32_bf16_lo": ([P, P, LL, P], I),
    "mc_transpose_f32": ([P, P, I, I, P], I),
    "mc_stem_weight_prep": ([P, P, I, P], I),
    "mc_stem_im2col": ([P, LL, LL, LL, LL, I, I, I, I, I, I, I, P, P], I),
    "mc_image_minmax_u8": ([P, LL, LL, I, P, P], I),
    "mc_gate_weights_bf16": ([P, P, I, I, I, P, P], I),
    "mc_stem_im2col_u8": ([P, LL, LL, LL, LL, P, F, F, I, I, I, I, I, I, I, P, P], I),
    "mc_dwconv_stat_rows": ([C.POINTER(DwconvArgs)], I),
    "mc_dwconv_bwd_data_stat_rows": ([C.POINTER(DwconvArgs)], I),
    "mc_dwconv_fwd": ([C.POINTER(DwconvArgs), P], I),
    "mc_dwconv_bwd_data": ([C.POINTER(DwconvArgs), P], I),
    "mc_dwconv_bwd_weight": ([C.POINTER(DwconvArgs), P], I),
    "mc_dwconv_set_lane_mode": ([I], I),
    "mc_dwconv_lane_supported": ([C.POINTER(DwconvArgs)], I),
    "mc_dwconv_bwd_fused_supported": ([C.POINTER(DwconvArgs)], I),
    "mc_dwconv_bwd_fused_preferred": ([C.POINTER(DwconvArgs)], I),
    "mc_dwconv_bwd_fused_stat_rows": ([C.POINTER(DwconvArgs)], I),
    "mc_dwconv_bwd_fused": ([C.POINTER(DwconvArgs), P], I),
    "mc_mbconv_xdw_supported": ([C.POINTER(DwconvArgs)], I),
    "mc_mbconv_xdw_stat_rows": ([C.POINTER(DwconvArgs)], I),
    "mc_mbconv_xdw_fwd": ([C.POINTER(DwconvArgs), P], I),
    "mc_bn_gram_partials": ([P, I, P, P, D, I, I, P, P], I),
    "mc_bn_finalize": ([P, I, I, D, P, P, P, P, F, F, I, P, P, P, P, P], I),
    "mc_bn_eval_coeffs": ([P, P, P, P, F, I, P, P, P], I),
    "mc_bnact_rows": ([C.POINTER(BnactArgs)], I),
    "mc_bnact_img_splits": ([C.POINTER(BnactArgs)], I),
    "mc_bnact_apply": ([C.POINTER(BnactArgs), P], I),
    "mc_bnact_pool": ([C.POINTER(BnactArgs), P], I),
    "mc_bnact_bwd_reduce": ([C.POINTER(BnactArgs), P], I),
    "mc_bnact_bwd_apply": ([C.POINTER(BnactArgs), P], I),
    "mc_bnact_se_dgate": ([C.POINTER(BnactArgs), P], I),
    "mc_bnact_se_sums": ([C.POINTER(BnactArgs), P], I),
    "mc_bn_partials_from_se_sums": ([P, P, P, F, LL, I, P, P], I),
    "mc_bn_bwd_finalize": ([P, I, I, D, P, P, P, P, P, P, P], I),
    "mc_colsum_rows": ([LL, I], I),
    "mc_colsum_bf16": ([P, LL, I, LL, P, P, I, P], I),
    "mc_se_fwd": ([P, P, P, P, P, I, I, I, P, P, P], I),
    "mc_se_bwd": ([P, P, P, P, P, P, P, I, I, I, P, P, P, P, P, P, P], I),
    "mc_dropout_f32": ([P, P, LL, F, ULL, U, P], I),
    "mc_bert_embed_fwd": ([P, P, P, P, P, P, P, F, I, I, I, F, ULL, U, P, P, P, P], I),
    "mc_bert_embed_bwd": ([P, P, P, P, P, P, P, P, P, I, I, I, F, ULL, U, P, P, P, P, P, P], I),
    "mc_add_ln_fwd": ([P, P, P, P, F, LL, I, F, ULL, U, P, P, P, P], I),
    "mc_add_ln_bwd": ([P, P, P, P, P, P, LL, I, F, ULL, U, P, P, P, P, P], I),
    "mc_softmax_fwd": ([P, LL, I, F, ULL, U, P, P, P], I),
    "mc_softmax_bwd": ([P, P, LL, I, F, ULL, U, F, P, P], I),
    "mc_bn_fold_prepare": ([P, P, P, P, D, I, I, P, P, P, P], I),
    "mc_bn_fold_cvec": ([P, P, P, P, P, D, I, I, P, P, P], I),
    "mc_bn_fold_wgrad": ([P, P, P, P, P, D, I, I, P, P], I),
    "mc_attn_supported": ([I, I], I),
    "mc_attn_fwd": ([P, P, I, I, I, F, F, ULL, U, P, P, P], I),
    "mc_attn_bwd": ([P, P, P, P, I, I, I, F, F, ULL, U, P, P], I),
    "mc_gelu_fwd": ([P, P, LL, P], I),
    "mc_gelu_bwd": ([P, P, P, LL, P], I),
    "mc_mask_bias": ([P, P, LL, P], I),
    "mc_eos_gather": ([P, P, I, I, I, P, P], I),
    "mc_eos_scatter": ([P, P, I, I, I, P, P], I),
    "mc_sgemm": ([P, LL, LL, P, LL, LL, P, LL, I, I, I, F, F, P, P, P, P], I),
    "mc_sgemm_ws_floats": ([I, I, I], LL),
    "mc_scale_f32": ([P, P, F, P, LL, P], I),
    "mc_l2norm_fwd": ([P, I, I, P, P, P], I),
    "mc_l2norm_bwd": ([P, P, P, I, I, P, P], I),
    "mc_ce_fwd_bwd": ([P, I, I, P, I, F, F, P, P, P], I),
}

EXPORTS = sorted(list(_SIGS.keys()) + ["mc_last_error"])

_lib = None


class MammoClipHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MammoClipHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The HIP path is the only path.")
    # torch first: its wheel bundles its own libamdhip64 -- the kernels launch on torch's streams and touch torch's device
    # memory, so both must live in ONE HIP runtime.  Loaded after torch, this library's libamdhip64.so.7 dependency
    # resolves to the copy torch already mapped; loaded before it, the process ends up with /opt/rocm's runtime AND
    # torch's, and the first launch fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    lib.mc_last_error.argtypes = []
    lib.mc_last_error.restype = C.c_char_p
    for name, (args, res) in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    if bool(lib.mc_storage_is_f16()) != (STORAGE == "f16"):
        raise MammoClipHipError(f"{LIB_PATH} is not the {STORAGE} build of the kernel library")
    _lib = lib
    return lib


def check(status: int, what: str = ""):
    if status != 0:
        msg = load().mc_last_error().decode("utf-8", "replace")
        raise MammoClipHipError(f"{what} failed (status {status}): {msg}")


class OpTimer:
    """Optional per-call HIP-event timing of C-ABI launches on torch's current stream (used by bench.py for the
    live roofline measurement and for per-op breakdowns).  ``only`` restricts timing to a set of entry points."""

    def __init__(self, only=None, kind_contains=None, keys=None):
        self.only = set(only) if only else None
        self.kind_contains = dict(kind_contains or {})     # entry point -> substring its ``kind`` tag must contain
        self.keys = set(keys) if keys is not None else None  # exact record keys ("entry" or "entry:kind") to time
        self.records = []            # (name, start_event, end_event, (bytes, flops))
        self.tag = None              # set by ops.* right before a call: algorithmic (bytes, flops) of that launch

    def summary(self):
        import torch
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, tag in self.records:
            ms = e0.elapsed_time(e1)
            cnt, tot, by, fl = out.get(name, (0, 0.0, 0, 0))
            out[name] = (cnt + 1, tot + ms, by + (tag[0] if tag else 0), fl + (tag[1] if tag else 0))
        return out


TIMER = None


def call(name: str, *args, kind=None):
    lib = load()
    t = TIMER
    if t is not None and t.keys is not None:
        if (name if kind is None else f"{name}:{kind}") not in t.keys:
            t.tag = None
            t = None
    if t is not None and (t.only is None or (name in t.only and t.kind_contains.get(name, "") in (kind or ""))):
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        status = getattr(lib, name)(*args)
        e1.record()
        t.records.append((name if kind is None else f"{name}:{kind}", e0, e1, t.tag))
        t.tag = None
        check(status, name)
        return
    check(getattr(lib, name)(*args), name)
