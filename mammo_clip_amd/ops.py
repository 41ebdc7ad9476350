"""Thin tensor-level wrappers over the C ABI (lib.py).  torch is used only for device memory and streams.

Every function launches hand-written HIP kernels asynchronously on torch's current stream.  Tensors are
bf16 activations in row-major [rows, channels] (NHWC) layout unless noted, fp32 parameters / statistics.
"""
import ctypes as C
import math
import os

import weakref

import torch

from . import lib as L

BF16 = torch.bfloat16 if L.STORAGE == "bf16" else torch.float16     # the 16-bit storage dtype of this process (lib.STORAGE)


def _p(t):
    return None if t is None else t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEV = getattr(torch._C, "_cuda_getDevice", None)


def _st():
    """raw handle of torch's current stream on this process's device (torch.cuda.current_stream() costs ~8 us per call
    in Python -- 6 ms per launch-bound config-#1 step; the private getter ~0.3 us)"""
    if _RAW_STREAM is None or _GET_DEV is None:
        return torch.cuda.current_stream().cuda_stream
    return _RAW_STREAM(_GET_DEV())


# ---------------------------------------------------------------------------------------- side streams
# The three encoder calls of one BreastClip.forward (view 1, view 2, both reports) are independent until the projection heads
# [ref: model/clip.py:83-108]; model/clip.py can issue them on separate HIP streams so the latency-bound launches of one chain
# (squeeze-excite MLPs, BatchNorm finalize, split-K reduces: ~1 900 launches under 25 us per config-#3 step) run beside the
# large launches of another.  Autograd runs each backward node on the stream its forward ran on.  Rules kept here:
#   * fork(): every side stream waits for the current stream; join(): the current stream waits for every side stream;
#   * a tensor that crosses streams is held by its consumer until a later fork/join orders the allocator's reuse, or it is
#     recorded on the consuming stream (record_stream);
#   * weight images shared by two chains are built BEFORE the fork (EfficientNet.warm_weight_images), BatchNorm running
#     statistics of the side chain are applied after the join in call order (efficientnet_custom._BNDefer).
_SIDE = {}


def _dev_index(device):
    """device ordinal of ``device`` (None, "cuda" and index-less torch.device objects mean the current device)"""
    idx = None if device is None else torch.device(device).index
    return torch.cuda.current_device() if idx is None else idx


def side_stream(i, device=None):
    dev = _dev_index(device)
    key = (dev, i)
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = torch.cuda.Stream(device=dev)
        # Parameters are leaves made on the main stream, the chains' backward nodes produce their gradients on the side streams:
        # autograd orders the accumulation itself, and its once-per-backward "AccumulateGrad node's stream does not match"
        # warning describes exactly this intended arrangement.
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:
            quiet(False)
    return st


def side_streams(device=None):
    dev = _dev_index(device)
    return [st for (d, _i), st in _SIDE.items() if d == dev]


def fork_side(device=None):
    """every side stream created so far waits for the work queued on the current stream"""
    cur = torch.cuda.current_stream(device)
    for st in side_streams(device):
        st.wait_stream(cur)


def join_side(device=None):
    """the current stream waits for the work queued on every side stream"""
    cur = torch.cuda.current_stream(device)
    for st in side_streams(device):
        cur.wait_stream(st)


def _chk_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.MammoClipHipError("mammo_clip_amd ops need tensors on a HIP device (there is no CPU fallback)")


def _note(nbytes, flops=0):
    """algorithmic HBM bytes / flops of the next launch (only looked at when bench.py installs a lib.OpTimer)"""
    if L.TIMER is not None:
        L.TIMER.tag = (int(nbytes), int(flops))


def empty(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


# ------------------------------------------------------------------------------------------- GEMM
def gemm(A, B, C_out, M, N, K, lda, ldb, ldc, a_kmajor=0, b_kmajor=0, c_f32=0, c_atomic=0, splits=1,
         batch=1, nb2=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), bias=None, bias_stride1=0, act=0, R=None, ldr=0,
         alpha=1.0, pro=None, stat_partials=None, max_grid_m=0, splitk_ws=None, split_groups=None, kind=None, stats=False,
         ab_fp8=0, alpha_dev=None):
    """pro = (operand, scale, shift, gate or None, rows_per_img, nch); split_groups = (rows per group, sub-splits, scale)
    stats=True: allocates and returns the [rows, 2, N] column sum / sum-of-squares partials of C (the row count depends on
    the tile configuration the library picks for this problem: mc_gemm_stat_rows on the complete argument block)"""
    a = L.GemmArgs()
    a.A, a.B, a.C = _p(A), _p(B), _p(C_out)
    a.M, a.K, a.N = M, K, N
    a.lda, a.ldb, a.ldc = lda, ldb, ldc
    a.a_kmajor, a.b_kmajor, a.c_f32, a.c_atomic, a.splits = a_kmajor, b_kmajor, c_f32, c_atomic, splits
    a.batch, a.nb2 = batch, nb2
    a.sA1, a.sA2, a.sB1, a.sB2, a.sC1, a.sC2 = sA[0], sA[1], sB[0], sB[1], sC[0], sC[1]
    a.bias, a.bias_stride1, a.act = _p(bias), bias_stride1, act
    a.R, a.ldr, a.alpha = _p(R), ldr, alpha
    a.ab_fp8, a.alpha_dev = ab_fp8, _p(alpha_dev)
    if pro is not None:
        a.pro_operand = pro[0]
        a.pro_scale, a.pro_shift, a.pro_gate = _p(pro[1]), _p(pro[2]), _p(pro[3])
        a.pro_rows_per_img, a.pro_nch = pro[4], pro[5]
    a.stat_partials = _p(stat_partials)
    a.max_grid_m = max_grid_m
    a.splitk_ws = _p(splitk_ws)
    if split_groups is not None:
        a.split_group_rows, a.split_sub, a.split_scale = split_groups[0], split_groups[1], _p(split_groups[2])
    if stats:
        stat_partials = empty((L.load().mc_gemm_stat_rows(C.byref(a)), 2, N), torch.float32, C_out)
        a.stat_partials = _p(stat_partials)
    es = 1 if ab_fp8 else 2
    _note(batch * (es * M * K + es * N * K + (4 if c_f32 else 2) * M * N + (2 * M * N if R is not None else 0)),
          2 * batch * M * N * K)
    if L.TIMER is not None and pro is None:
        # timing classes = the kernel that serves the launch: plain NT tiles (256 x 256 gemm8p_kernel or the 128 x 128
        # direct-to-LDS gemm_kernel), TN weight-gradient tiles (gemm256_tn_kernel) -- each split by the roofline that bounds
        # THIS launch's shape: arithmetic intensity against the 2.5 PFLOP/s : 8 TB/s ridge (SURVEY.md section 8d asks for the
        # MFMA fraction of the MFMA-bound GEMMs and the HBM fraction of the bandwidth-bound ones separately)
        nbytes = batch * (es * M * K + es * N * K + (4 if c_f32 else 2) * M * N + (2 * M * N if R is not None else 0))
        side = "mfma" if 2.0 * batch * M * N * K >= 312.5 * nbytes else "hbm"
        if not a_kmajor and not b_kmajor and N > 64 and K > 48 and not c_f32:
            kind = (kind or "") + ("|glnt256" if L.load().mc_gemm_tile_config(C.byref(a)) == 256 else "|glnt") + "|" + side
        elif a_kmajor and b_kmajor and c_f32 and L.load().mc_gemm256_tn_eligible(C.byref(a)):
            kind = "wgrad|tn256|" + side
    L.call("mc_gemm_bf16", C.byref(a), _st(), kind=kind)
    return stat_partials


def gemm_stat_rows(M, N=128, batch=1):
    a = L.GemmArgs()
    a.M, a.N, a.batch = M, N, batch
    return L.load().mc_gemm_stat_rows(C.byref(a))


# ------------------------------------------------------------------------------------------- fp8 (config #5)
def amax_bf16(x, out=None):
    """max |x| of a contiguous bf16 tensor -> float32 [1] on the device"""
    x = x.contiguous()
    if out is None:
        out = torch.zeros(1, dtype=torch.float32, device=x.device)
    L.call("mc_amax_bf16", _p(x), x.numel(), _p(out), _st())
    return out


def quant_fp8(x, amax, amax_next=None):
    """bf16 -> OCP e4m3 bytes with the per-tensor scale 448 / amax.  Returns (q uint8 like x, dequantisation scale [1])."""
    x = x.contiguous()
    q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    scale = torch.empty(1, dtype=torch.float32, device=x.device)
    L.call("mc_quant_fp8_bf16", _p(x), x.numel(), _p(amax), _p(q), _p(scale), _p(amax_next), _st())
    return q, scale


def linear_fwd_fp8(x, w, amax_x=None, amax_w=None, stats=False, batch_w=None):
    """y[M,N] (bf16) = dequant( e4m3(x)[M,K] . e4m3(w)[N,K]^T ) -- fp8 operands on v_mfma_f32_16x16x32_fp8_fp8, fp32
    accumulation.  amax_*: device scalars (None: measured on the tensor now).  batch_w = (n_img, rows per image): ``w`` is
    [n_img, N, K] with one matrix per image (the gated projection weights).  Returns y (, column statistics partials)."""
    M, K = x.shape
    N = w.shape[-2]
    assert K % 16 == 0, "fp8 operands: K must be a multiple of 16"
    ax = amax_x if amax_x is not None else amax_bf16(x)
    aw = amax_w if amax_w is not None else amax_bf16(w)
    xq, sx = quant_fp8(x, ax)
    wq, sw = quant_fp8(w, aw)
    y = empty((M, N), BF16, x)
    a = dict(ab_fp8=1, alpha_dev=sx * sw)
    if batch_w is not None:
        n_img, hw = batch_w
        part = gemm(xq, wq, y, hw, N, K, K, K, N, batch=n_img, sA=(hw * K, 0), sB=(N * K, 0), sC=(hw * N, 0), stats=stats,
                    kind="fwd_fp8", **a)
    else:
        part = gemm(xq, wq, y, M, N, K, K, K, N, stats=stats, kind="fwd_fp8", **a)
    return (y, part) if stats else y


ROWS_MIN_M = 8192       # below this the tiled kernel is as good


def gemm_rows(x, w, y, residual=None, pro=None, stats=False, kind=None, bias=None):
    """row-streaming 1x1 convolution; stats = True returns the [workgroups, 2, N] BatchNorm partials of the output"""
    a = L.GemmRowsArgs()
    M, K = x.shape
    N = w.shape[0]
    a.X, a.M, a.K, a.ldx = _p(x), M, K, x.stride(0)
    a.W, a.N, a.ldw = _p(w), N, w.stride(0)
    a.C, a.ldc = _p(y), y.stride(0)
    if residual is not None:
        a.R, a.ldr = _p(residual), residual.stride(0)
    if pro is not None:
        a.pro_scale, a.pro_shift, a.pro_gate, a.pro_rows_per_img = _p(pro[0]), _p(pro[1]), _p(pro[2]), pro[3]
    part = empty((L.load().mc_gemm_rows_blocks(C.byref(a)), 2, N), torch.float32, x) if stats else None
    a.stat_partials, a.bias = _p(part), _p(bias)
    _note(2 * M * (K + N) + 2 * N * K + (2 * M * N if residual is not None else 0), 2 * M * N * K)
    L.call("mc_gemm_rows_bf16", C.byref(a), _st(), kind=kind)
    return part


# Measured on MI355X (scripts/epi_probe.py, 32 images): the two epilogue launches beat dgrad + se_sums + apply only for wide
# expanded tensors -- c = 240: 1.13 vs 1.47 ms per block and view; c = 144: 0.82 vs 0.90; c = 48: 1.26 vs 1.28; c = 24: 0.91 vs
# 0.75 (narrow rows: the row-streaming kernel's per-16-row overhead carries too little data, and the sums form runs at two
# waves per SIMD with one iteration of d rows in flight)
PROJ_DGRAD_FUSE_MIN_N = int(os.environ.get("MC_PROJ_DGRAD_FUSE_MIN_N", 144))


def proj_dgrad_fusable(M, n_out, k_in, rows_per_img):
    """can the data gradient of a projection conv (dP [M, k_in] -> G [M, n_out]) carry the squeeze-excite / BatchNorm1
    backward in its epilogue (gemm_rows epi_mode 1 / 2)?  Row-streaming shapes with whole 16-row groups per image."""
    if n_out < PROJ_DGRAD_FUSE_MIN_N:
        return False
    lib_ = L.load()
    return (M >= ROWS_MIN_M and n_out <= 256 and k_in <= 128 and rows_per_img % 16 == 0 and rows_per_img >= 16
            and bool(lib_.mc_gemm_rows_supported(n_out, k_in)) and not _prefer_tiles(n_out, k_in)
            and bool(lib_.mc_gemm_rows_epi_supported(M, n_out, k_in, rows_per_img, 1))
            and bool(lib_.mc_gemm_rows_epi_supported(M, n_out, k_in, rows_per_img, 2)))


def _proj_dgrad_epi_args(dp, w_t, d, stats, rows_per_img, mode):
    a = L.GemmRowsArgs()
    M, K = dp.shape
    N = w_t.shape[0]
    assert d.shape == (M, N) and w_t.shape[1] == K
    a.X, a.M, a.K, a.ldx = _p(dp), M, K, dp.stride(0)
    a.W, a.N, a.ldw = _p(w_t), N, w_t.stride(0)
    a.epi_mode, a.epi_x, a.epi_ldx, a.epi_rows_per_img = mode, _p(d), d.stride(0), rows_per_img
    a.epi_scale, a.epi_shift = _p(stats.scale), _p(stats.shift)
    return a, M, N, K


def proj_dgrad_se_sums(dp, w_t, d, stats, n_img, rows_per_img):
    """bnact_se_sums(d, G) for G = dp . w_t^T WITHOUT G in memory: [5, n_img, c] (gemm_rows epi_mode 1)."""
    a, M, N, K = _proj_dgrad_epi_args(dp, w_t, d, stats, rows_per_img, 1)
    sums = empty((5, n_img, N), torch.float32, dp)
    a.epi_mean, a.epi_invstd, a.epi_sums = _p(stats.mean), _p(stats.invstd), _p(sums)
    ws = empty((L.load().mc_gemm_rows_epi_ws_floats(C.byref(a)),), torch.float32, dp)
    a.epi_ws = _p(ws)
    _note(2 * M * (K + N) + 2 * N * K, 2 * M * N * K)
    L.call("mc_gemm_rows_bf16", C.byref(a), _st(), kind="dgrad_se_sums")
    return sums


def proj_dgrad_bn_apply(dp, w_t, d, stats, coef, mul, add, add_scale, rows_per_img):
    """bnact_bwd's apply pass on (d, G) for G = dp . w_t^T without G in memory (gemm_rows epi_mode 2): returns
    dd = coef0*dz + coef1*d + coef2, dz = (G*mul[img] + add[img]*add_scale) * silu'(bn(d))."""
    a, M, N, K = _proj_dgrad_epi_args(dp, w_t, d, stats, rows_per_img, 2)
    dd = empty((M, N), BF16, dp)
    a.C, a.ldc = _p(dd), dd.stride(0)
    a.epi_coef, a.epi_mul, a.epi_add, a.epi_add_scale = _p(coef), _p(mul), _p(add), float(add_scale)
    _note(2 * M * (K + 2 * N) + 2 * N * K, 2 * M * N * K)
    L.call("mc_gemm_rows_bf16", C.byref(a), _st(), kind="dgrad_bn_apply")
    return dd


def _rows_ok(M, N, K, bias, act):
    return act == 0 and M >= ROWS_MIN_M and L.load().mc_gemm_rows_supported(N, K)


def _prefer_tiles(N, K):
    """plain-operand problems both kernels take: wide outputs over a short reduction (B5 stages 5 / 6: 128 -> 768,
    176 -> 1056 and their data gradients) run 20-26 % faster on the 256 x 256 tile kernel than as 6-9 column tiles of the
    row-streaming kernel (173280 x 1056 x 176: 144 vs 181 us, 173280 x 768 x 128: 80 vs 102 us); narrower outputs
    (693120 x 384 x 64: 152 vs 139 us) and every problem with a fused prologue stay on the row-streaming kernel"""
    return N >= 512 and K >= 128


# Derived weight images (bf16 casts / transposes of fp32 master parameters) are reused until the parameter changes:
# within one step both image views, the forward and the backward pass need the same image.  Only (views of) leaf
# tensors are cached; an entry is tied to the owning tensor OBJECT through a weak reference (addresses and ids are
# recycled by the allocator) and to its version counter (bumped by the optimizer's in-place update / load_state_dict).
_WCACHE = {}
FORK_MISSES = None     # tests set a list: (kind, id(parameter)) of every image built between a fork and its join
FORKED = 0             # > 0 while encoder chains may be in flight on several streams (model/clip.py, engine.Trainer._backward)
_WGEN = 0          # bumped whenever an image is (re)built: lets the optimizer reuse its parameter -> image map


def _cached(kind, src, make):
    base = src._base if src._base is not None else src
    if not base.is_leaf:
        return make()
    key = (kind, id(base), src.data_ptr(), tuple(src.shape), tuple(src.stride()))
    hit = _WCACHE.get(key)
    if hit is not None and hit[0]() is base and hit[1] == base._version and hit[2].device == src.device:
        return hit[2]
    val = make()
    if FORK_MISSES is not None and FORKED:
        FORK_MISSES.append((kind, id(base)))   # (tests: image built while encoder chains may be in flight on several streams)
    if len(_WCACHE) > 4096:
        for k in [k for k, v in _WCACHE.items() if v[0]() is None]:
            del _WCACHE[k]
    _WCACHE[key] = (weakref.ref(base), base._version, val)
    global _WGEN
    _WGEN += 1
    return val


def cached_cast_images(params):
    """{id(param): (cache key, bf16 image)} for parameters whose WHOLE tensor has a live plain-cast image in the cache:
    the optimizer kernel rewrites these images in the pass that updates the parameter (see AdamW.step) and re-stamps
    them with ``stamp_cast_images``, so the next forward finds them current."""
    want = {id(p): p for p in params}
    out = {}
    for key, (ref, _ver, val) in _WCACHE.items():
        if key[0] != "c":
            continue
        base = ref()
        if base is None or id(base) not in want or base is not want[id(base)]:
            continue
        if key[2] == base.data_ptr() and val.numel() == base.numel() and val.is_contiguous() and base.is_contiguous():
            out[id(base)] = (key, val)
    return out


def stamp_cast_images(found):
    for key, _val in found.values():
        ref, _ver, val = _WCACHE[key]
        base = ref()
        if base is not None:
            _WCACHE[key] = (ref, base._version, val)


def cache_generation():
    return _WGEN


def cast_transpose_bf16(src2d):
    """fp32 [rows, cols] -> bf16 [cols, rows]"""
    def make():
        rows, cols = src2d.shape
        dst = empty((cols, rows), BF16, src2d)
        L.call("mc_cast_transpose_f32_bf16", _p(src2d.contiguous()), _p(dst), rows, cols, _st())
        return dst
    return _cached("ct", src2d, make)


def _linear_fwd_impl(x, w, bias=None, act=0, residual=None, stats=False, pro=None, out=None, tag=""):
    """y[M,N] = x[M,K] . w[N,K]^T (+bias)(act)(+residual).  stats -> also returns [rows,2,N] partials."""
    M, K = x.shape
    N = w.shape[0]
    y = out if out is not None else empty((M, N), BF16, x)
    if _rows_ok(M, N, K, bias, act) and (pro is None or pro[0] is not None) and not (
            pro is None and bias is None and not tag and _prefer_tiles(N, K)) and not (
            pro is not None and pro[2] is not None and pro[3] < 16):     # (gated rows kernel: >= 16 rows per image; else the tile GEMM)
        part = gemm_rows(x, w, y, residual=residual, pro=pro, stats=stats, kind="fwd_rows" + tag, bias=bias)
        return (y, part) if stats else y
    if pro is not None and pro[0] is None and residual is None and bias is None and M % pro[3] == 0 and pro[3] >= 256:
        # x is already activated and only carries the per-image gate: one GEMM per image (batched) whose weight tile
        # is scaled by that image's gate while it is staged -- no per-row prologue on the big operand
        hw, n_img = pro[3], M // pro[3]
        if N > 64 and K > 48 and w.is_contiguous():
            # gated copies of the weight (n_img x N x K, a few MB) -> plain batched GEMM, direct-to-LDS staging
            wg = empty((n_img, N, K), BF16, x)
            L.call("mc_gate_weights_bf16", _p(w), _p(pro[2]), n_img, N, K, _p(wg), _st())
            part = gemm(x, wg, y, hw, N, K, x.stride(0), K, y.stride(0), batch=n_img, sA=(hw * x.stride(0), 0), sB=(N * K, 0),
                        sC=(hw * y.stride(0), 0), stats=stats, kind="fwd")
        else:
            part = gemm(x, w, y, hw, N, K, x.stride(0), w.stride(0), y.stride(0), batch=n_img, sA=(hw * x.stride(0), 0),
                        sC=(hw * y.stride(0), 0), pro=(3, None, None, pro[2], hw, K), stats=stats, kind="fwd")
        return (y, part) if stats else y
    p = None
    if pro is not None:
        p = (1, pro[0], pro[1], pro[2], pro[3], K)
    part = gemm(x, w, y, M, N, K, x.stride(0), w.stride(0), y.stride(0), bias=bias, act=act, R=residual,
                ldr=(residual.stride(0) if residual is not None else 0), pro=p, stats=stats, kind="fwd" + tag)
    return (y, part) if stats else y


def gate_weights(w, gate):
    """per-image gated copies of a weight matrix: out[i, n, k] = w[n, k] * gate[i, k]  (bf16)"""
    N, K = w.shape
    n_img = gate.shape[0]
    wg = empty((n_img, N, K), BF16, w)
    L.call("mc_gate_weights_bf16", _p(w), _p(gate), n_img, N, K, _p(wg), _st())
    return wg


def _linear_dgrad_impl(dy, w, residual=None, w_t=None):
    """dx[M,K] = dy[M,N] . w[N,K]  (+ residual).  w_t = w^T [K,N] (bf16) enables the row-streaming kernel."""
    M, N = dy.shape
    K = w.shape[1]
    dx = empty((M, K), BF16, dy)
    if w_t is not None and _rows_ok(M, K, N, None, 0) and not _prefer_tiles(K, N):
        gemm_rows(dy, w_t, dx, residual=residual, kind="dgrad_rows")
        return dx
    if w_t is not None:       # NT form through the transposed weight: both operands k-contiguous (16-byte LDS stores)
        gemm(dy, w_t, dx, M, K, N, dy.stride(0), w_t.stride(0), dx.stride(0), R=residual,
             ldr=(residual.stride(0) if residual is not None else 0), kind="dgrad")
        return dx
    gemm(dy, w, dx, M, K, N, dy.stride(0), w.stride(0), dx.stride(0), b_kmajor=1, R=residual,
         ldr=(residual.stride(0) if residual is not None else 0), kind="dgrad")
    return dx


def _wgrad_splits(m, n, k):
    tiles = math.ceil(m / 128) * math.ceil(n / (128 if n > 64 else (64 if n > 32 else 32)))
    ktiles = math.ceil(k / 64)
    s = max(1, min(math.ceil(1024 / tiles), max(1, ktiles // 8)))
    if s >= 8:
        s -= s % 8          # one XCD per split (see gemm_kernel): keep the 8 XCDs evenly loaded (11 splits measured 13 % slower than 8)
    return s


def _tn256_plan(n_out, k_out, rows, lddy, ldx, group_rows=0):
    """K splits (sub-splits per group in the grouped form) the 256 x 256 TN tile kernel (gemm256_tn.hip) wants for
    dW[n_out, k_out] = dY[rows, n_out]^T . X[rows, k_out]; 0 = that kernel does not take the problem"""
    a = L.GemmArgs()
    a.M, a.N, a.K, a.lda, a.ldb, a.ldc = n_out, k_out, rows, lddy, ldx, k_out
    a.a_kmajor, a.b_kmajor, a.c_f32, a.batch, a.nb2, a.splits = 1, 1, 1, 1, 1, 2
    a.splitk_ws = 16                                       # any non-null value: only looked at, never dereferenced
    if not L.load().mc_gemm256_tn_eligible(C.byref(a)):
        return 0
    return L.load().mc_gemm256_tn_splits(n_out, k_out, rows, group_rows)


def _linear_wgrad_impl(dy, x, pro=None, out=None, tag=""):
    """dw[N,K] (fp32) = dy[M,N]^T . x'[M,K]; x' = prologue(x) when pro = (scale, shift, gate, rows_per_img)."""
    M, N = dy.shape
    K = x.shape[1]
    dw = out if out is not None else empty((N, K), torch.float32, dy)
    if M >= ROWS_MIN_M and L.load().mc_wgrad_rows_supported(N, K) and (pro is None or pro[0] is not None):
        a = L.WgradRowsArgs()
        a.dY, a.N, a.lddy = _p(dy), N, dy.stride(0)
        a.X, a.K, a.ldx, a.M = _p(x), K, x.stride(0), M
        ws = empty((L.load().mc_wgrad_rows_blocks(M), N, K), torch.float32, dy)
        a.dW, a.ws, a.accumulate = _p(dw), _p(ws), (1 if out is not None else 0)
        if pro is not None:
            a.pro_scale, a.pro_shift, a.pro_gate, a.pro_rows_per_img = _p(pro[0]), _p(pro[1]), _p(pro[2]), pro[3]
        _note(2 * M * (N + K) + 4 * N * K, 2 * M * N * K)
        L.call("mc_wgrad_rows_bf16", C.byref(a), _st(), kind="wgrad_rows" + tag)
        return dw
    if pro is not None and pro[0] is None and M % pro[3] == 0:
        # x is already activated and only carries the per-image gate: cut the reduction at image boundaries and apply
        # the gate when the partials are combined (dW = sum_img gate_img (.) dW_img) -- a plain TN GEMM, no prologue
        n_img, hw = M // pro[3], pro[3]
        sub = _tn256_plan(N, K, M, dy.stride(0), x.stride(0), group_rows=hw)
        if not sub:
            tiles = math.ceil(N / 128) * math.ceil(K / 128)
            sub = max(1, min(math.ceil(768 / (tiles * n_img)), hw // 512))
            if (n_img * sub) % 8 and n_img * sub >= 16:
                sub = max(1, sub - 1) if (n_img * (sub - 1)) % 8 == 0 and sub > 1 else sub
        splits = n_img * sub
        ws = empty((splits, N, K), torch.float32, dy)
        gemm(dy, x, dw, N, K, M, dy.stride(0), x.stride(0), dw.stride(0), a_kmajor=1, b_kmajor=1, c_f32=1,
             c_atomic=(1 if out is not None else 0), splits=splits, splitk_ws=ws, split_groups=(hw, sub, pro[2]), kind="wgrad" + tag)
        return dw
    p = None
    if pro is not None:
        p = (2, pro[0], pro[1], pro[2], pro[3], K)
    splits = (_tn256_plan(N, K, M, dy.stride(0), x.stride(0)) if pro is None else 0) or _wgrad_splits(N, K, M)
    ws = empty((splits, N, K), torch.float32, dy) if splits > 1 else None
    gemm(dy, x, dw, N, K, M, dy.stride(0), x.stride(0), dw.stride(0), a_kmajor=1, b_kmajor=1, c_f32=1,
         c_atomic=(1 if out is not None else 0), splits=splits, pro=p, splitk_ws=ws, kind="wgrad" + tag)
    return dw


FUSE_XBWD = os.environ.get("MC_FUSE_XBWD", "1") != "0"       # expand conv backward: weight + data gradient from one pass (xbwd_rows)


_XBWD_OK = {}


def xbwd_rows_ok(M, n_out, k_in):
    """does ONE launch give both gradients of the 1x1 conv [M, k_in] -> [M, n_out] (mc_xbwd_rows_bf16)?"""
    if not FUSE_XBWD or M < ROWS_MIN_M:
        return False
    hit = _XBWD_OK.get((n_out, k_in))
    if hit is None:
        hit = _XBWD_OK[(n_out, k_in)] = bool(L.load().mc_xbwd_rows_supported(n_out, k_in))
    return hit


def xbwd_rows(dy, x, w_t, residual=None):
    """(dx [M, K] bf16 = dy . w_t^T (+ residual), dw [N, K] f32 = dy^T x) from ONE pass over dy [M, N]: the data gradient of
    ``linear_dgrad(dy, ., residual, w_t=w_t)`` and the weight gradient of ``linear_wgrad(dy, x)`` of a 1x1 convolution whose
    output gradient is the wide tensor (expand conv; gemm_wgrad_rows.hip xbwd_rows_kernel).  w_t: [K, N] bf16."""
    M, N = dy.shape
    K = x.shape[1]
    assert w_t.shape == (K, N) and x.shape[0] == M
    a = L.WgradRowsArgs()
    a.dY, a.N, a.lddy = _p(dy), N, dy.stride(0)
    a.X, a.K, a.ldx, a.M = _p(x), K, x.stride(0), M
    dw = empty((N, K), torch.float32, dy)
    ws = empty((L.load().mc_xbwd_rows_blocks(M), N, K), torch.float32, dy)
    a.dW, a.ws, a.accumulate = _p(dw), _p(ws), 0
    dx = empty((M, K), BF16, dy)
    _note(2 * M * (N + 2 * K) + (2 * M * K if residual is not None else 0) + 4 * N * K, 4 * M * N * K)
    L.call("mc_xbwd_rows_bf16", C.byref(a), _p(w_t), w_t.stride(0), _p(dx), dx.stride(0), _p(residual),
           residual.stride(0) if residual is not None else 0, _st(), kind="xbwd_rows")
    return dx, dw


def colsum(x, out=None, accumulate=False):
    """out[c] (+)= sum_m x[m,c]  (bf16 in, fp32 out)"""
    M, Cn = x.shape
    rows = L.load().mc_colsum_rows(M, Cn)
    part = empty((rows, Cn), torch.float32, x)
    if out is None:
        out = empty((Cn,), torch.float32, x)
        accumulate = False
    L.call("mc_colsum_bf16", _p(x), M, Cn, x.stride(0), _p(part), _p(out), int(accumulate), _st())
    return out


def cast_bf16(src, out=None):
    def make():
        s_ = src.contiguous()
        dst = out if out is not None else empty(s_.shape, BF16, s_)
        L.call("mc_cast_f32_bf16", _p(s_), _p(dst), s_.numel(), _st())
        return dst
    return make() if out is not None else _cached("c", src, make)


def cast_bf16_lo(src, out=None):
    """low term of the two-term bf16 split: bf16(src - float(bf16(src)))  (cached like ``cast_bf16``)"""
    def make():
        s_ = src.contiguous()
        dst = out if out is not None else empty(s_.shape, BF16, s_)
        L.call("mc_cast_f32_bf16_lo", _p(s_), _p(dst), s_.numel(), _st())
        return dst
    return make() if out is not None else _cached("clo", src, make)


def cast_f32(src):
    src = src.contiguous()
    dst = empty(src.shape, torch.float32, src)
    L.call("mc_cast_bf16_f32", _p(src), _p(dst), src.numel(), _st())
    return dst


def flipped_taps_f32(w_ck):
    """fp32 depthwise weight [c, k*k] -> taps-major, tap order reversed [k*k, c] (= the filter rotated by 180 degrees: the
    stride-1 data gradient is a forward conv with it); cached per parameter version like the other weight images"""
    def make():
        return transpose_f32(w_ck).flip(0).contiguous()
    return _cached("tfl", w_ck, make)


def transpose_f32(src, cache=False):
    def make():
        rows, cols = src.shape
        dst = empty((cols, rows), torch.float32, src)
        L.call("mc_transpose_f32", _p(src.contiguous()), _p(dst), rows, cols, _st())
        return dst
    return _cached("t", src, make) if cache else make()


# ------------------------------------------------------------------------------------------- stem
def stem_weight_prep(w):
    c0 = w.shape[0]
    out = empty((c0, 32), BF16, w)
    L.call("mc_stem_weight_prep", _p(w.contiguous()), _p(out), c0, _st())
    return out


class RawImages:
    """Raw 8-bit image batch + the dataset's (mean, std): the min-max / mean-std normalisation of the reference's
    dataset [ref: data/datasets/imagetext.py:131-135] is applied inside the stem's load instead of on the host
    (4x less data to hand to the GPU than a normalised fp32 batch).
    ``data``: uint8 [b,3,H,W] (any permuted view of a dense per-image block, e.g. the trainer's
    ``[b,1,H,W,3].squeeze(1).permute(0,3,1,2)``, trainer_ddp.py:288-291)."""

    def __init__(self, data, mean, std):
        assert data.dtype == torch.uint8 and data.dim() == 4 and data.shape[1] == 3, "RawImages: uint8 [b,3,H,W]"
        self.data, self.mean, self.std = data, float(mean), float(std)

    def to(self, device, **kw):
        return RawImages(self.data.to(device, **kw), self.mean, self.std)

    shape = property(lambda self: self.data.shape)
    device = property(lambda self: self.data.device)
    is_cuda = property(lambda self: self.data.is_cuda)
    dtype = property(lambda self: self.data.dtype)


def image_minmax_u8(x):
    """per-image (min, max) of a uint8 batch whose images are dense blocks -> uint32 [2, n]"""
    n = x.shape[0]
    per = x[0].numel()
    dense = sorted(zip(x.stride()[1:], x.shape[1:]), reverse=True)
    run = 1
    for st, sz in reversed(dense):
        assert sz == 1 or st == run, "RawImages: every image must be one dense block of bytes"
        run *= sz
    mm = torch.empty((2, n), dtype=torch.int32, device=x.device)
    L.call("mc_image_minmax_u8", _p(x), x.stride(0), per, n, _p(mm), _st())
    return mm


def stem_im2col(x, pad_l, pad_t, oh, ow):
    """x: fp32 [n,3,h,w] with ANY strides (NCHW or a permuted NHWC view), or RawImages -> bf16 patches [n*oh*ow, 32]."""
    n, c, h, w = x.shape
    if isinstance(x, RawImages):
        d = x.data
        out = empty((n * oh * ow, 32), BF16, d)
        sn, sc, sh, sw = d.stride()
        L.call("mc_stem_im2col_u8", _p(d), sn, sc, sh, sw, _p(image_minmax_u8(d)), x.mean, x.std, n, h, w, pad_l, pad_t,
               oh, ow, _p(out), _st())
        return out
    assert c == 3 and x.dtype == torch.float32
    out = empty((n * oh * ow, 32), BF16, x)
    sn, sc, sh, sw = x.stride()
    L.call("mc_stem_im2col", _p(x), sn, sc, sh, sw, n, h, w, pad_l, pad_t, oh, ow, _p(out), _st())
    return out


# ------------------------------------------------------------------------------------------- depthwise
def _dw_args(n, h, w, c, k, stride, pad_l, pad_t, oh, ow):
    a = L.DwconvArgs()
    a.n, a.h, a.w, a.c, a.k, a.stride, a.pad_l, a.pad_t, a.oh, a.ow = n, h, w, c, k, stride, pad_l, pad_t, oh, ow
    return a


def _dwconv_fwd_impl(x, w_kkc, n, h, w, c, k, stride, pad_l, pad_t, oh, ow, pro=None, stats=False, epi=None):
    """epi = (e, BNStats): BatchNorm(+SiLU)-backward epilogue (stride 1; see mc_dwconv_args.epi_x): returns
    (dZ, [rows,2,c] partials for bn_bwd_finalize) instead of (y, BatchNorm statistics partials)."""
    a = _dw_args(n, h, w, c, k, stride, pad_l, pad_t, oh, ow)
    y = empty((n * oh * ow, c), BF16, x)
    a.x, a.out, a.w_kkc = _p(x), _p(y), _p(w_kkc)
    if pro is not None:
        a.pro_scale, a.pro_shift = _p(pro[0]), _p(pro[1])
    if epi is not None:
        e, st = epi
        assert stride == 1 and e.shape == y.shape
        a.epi_x, a.epi_scale, a.epi_shift, a.epi_mean, a.epi_invstd = _p(e), _p(st.scale), _p(st.shift), _p(st.mean), _p(st.invstd)
        stats = True
    part = None
    if stats:
        rows = L.load().mc_dwconv_stat_rows(C.byref(a))
        part = empty((rows, 2, c), torch.float32, x)
        a.stat_partials, a.stat_rows = _p(part), rows
    _note(2 * n * c * (h * w + oh * ow * (2 if epi is not None else 1)), 2 * n * c * oh * ow * k * k)
    L.call("mc_dwconv_fwd", C.byref(a), _st(), kind=f"k{k}s{stride}" + ("|dgrad_bn" if epi is not None else ""))
    return (y, part) if stats else y


def _dwconv_bwd_data_impl(dy, w_kkc, n, h, w, c, k, stride, pad_l, pad_t, oh, ow, w_kkc_flipped=None, epi=None):
    """dx [n*h*w, c].  stride 1 runs the LDS-tiled forward kernel on the flipped filter; stride 2 the marching
    super-pixel kernel.  epi = (e, BNStats): the launch finishes the BatchNorm + SiLU backward of the conv's input
    silu(bn(e)) and returns (dZ, BatchNorm-backward partials) instead of dx, see dwconv_fwd."""
    if stride == 1 and w_kkc_flipped is not None:
        return _dwconv_fwd_impl(dy, w_kkc_flipped, n, oh, ow, c, k, 1, k - 1 - pad_l, k - 1 - pad_t, h, w, epi=epi)
    assert epi is None or stride == 2
    a = _dw_args(n, h, w, c, k, stride, pad_l, pad_t, oh, ow)
    dx = empty((n * h * w, c), BF16, dy)
    a.dy, a.out, a.w_kkc = _p(dy), _p(dx), _p(w_kkc)
    part = None
    if epi is not None:
        e, st = epi
        assert e.shape == dx.shape
        a.epi_x, a.epi_scale, a.epi_shift, a.epi_mean, a.epi_invstd = _p(e), _p(st.scale), _p(st.shift), _p(st.mean), _p(st.invstd)
        part = empty((L.load().mc_dwconv_bwd_data_stat_rows(C.byref(a)), 2, c), torch.float32, dy)
        a.stat_partials = _p(part)
    _note(2 * n * c * (h * w * (2 if epi is not None else 1) + oh * ow), 2 * n * c * oh * ow * k * k)
    L.call("mc_dwconv_bwd_data", C.byref(a), _st(), kind=f"k{k}s{stride}" + ("|dgrad_bn" if epi is not None else ""))
    return dx if epi is None else (dx, part)


def _dw_fused_args(dd, w_kkc_flipped, n, h, w, c, k, pad_l, pad_t, oh, ow, e, st, xw=None):
    """argument block of the fused stride-1 backward = the data-gradient launch on flipped taps: the conv's OUTPUT geometry
    (oh, ow) is this launch's input geometry.  xw = (x, we) (round 6): the e rows are formed from the block input x [n*h*w, cin]
    and the expand weight we [c, cin] inside the launch -- e is not read (and need not exist)"""
    a = _dw_args(n, oh, ow, c, k, 1, k - 1 - pad_l, k - 1 - pad_t, h, w)
    a.x, a.w_kkc = _p(dd), _p(w_kkc_flipped)
    a.epi_x, a.epi_scale, a.epi_shift, a.epi_mean, a.epi_invstd = _p(e), _p(st.scale), _p(st.shift), _p(st.mean), _p(st.invstd)
    if xw is not None:
        x, we = xw
        assert x.is_contiguous() and we.is_contiguous() and we.shape == (c, x.shape[1])
        a.epi_x, a.xw, a.cin = _p(x), _p(we), x.shape[1]
    return a


_FUSED_OK = {}


def dwconv_bwd_fused_ok(n, h, w, c, k, stride, pad_l, pad_t, oh, ow, force=False, cin=0):
    """does the fused backward launch (mc_dwconv_bwd_fused) take / win this stride-1 conv?  (pointers are not looked at; the
    answer is a function of the geometry alone and is cached: the launch-bound configurations notice every ctypes call)"""
    if stride != 1:
        return False
    key = (n, h, w, c, k, pad_l, pad_t, oh, ow, force, cin)
    hit = _FUSED_OK.get(key)
    if hit is None:
        a = _dw_args(n, oh, ow, c, k, 1, k - 1 - pad_l, k - 1 - pad_t, h, w)
        a.epi_x = 16                                    # any non-null value: only looked at
        if cin:                                         # (cin > 0: the form whose e rows are formed from the block input)
            a.xw, a.cin = 16, cin
        lib_ = L.load()
        hit = _FUSED_OK[key] = bool(lib_.mc_dwconv_bwd_fused_supported(C.byref(a)) if force else lib_.mc_dwconv_bwd_fused_preferred(C.byref(a)))
    return hit


def dwconv_bwd_fused(dd, e, st, w_kkc_flipped, n, h, w, c, k, pad_l, pad_t, oh, ow, xw=None):
    """Whole backward of a stride-1 depthwise conv y = dw(silu(bn0(e))) in one launch (conv_lane.hip MODE 3):
    returns (dZ0 [n*h*w, c] bf16 = dL/d bn0(e), BatchNorm0-backward partials [rows, 2, c], dW [k*k, c] f32 in the conv's own
    tap order).  dd = dL/dy [n*oh*ow, c]; e = the expand conv's output [n*h*w, c]; st = its BatchNorm statistics."""
    a = _dw_fused_args(dd, w_kkc_flipped, n, h, w, c, k, pad_l, pad_t, oh, ow, e, st, xw)
    dz = empty((n * h * w, c), BF16, dd)
    dw = torch.zeros((k * k, c), dtype=torch.float32, device=dd.device)
    rows = L.load().mc_dwconv_bwd_fused_stat_rows(C.byref(a))
    part = empty((rows, 2, c), torch.float32, dd)
    a.out, a.dw_out, a.stat_partials, a.stat_rows = _p(dz), _p(dw), _p(part), rows
    if xw is None:
        _note(2 * n * c * (oh * ow + 2 * h * w), 4 * n * c * h * w * k * k)
    else:
        _note(2 * n * (c * (oh * ow + h * w) + xw[0].shape[1] * h * w), 4 * n * c * h * w * k * k + 2 * n * h * w * c * xw[0].shape[1])
    L.call("mc_dwconv_bwd_fused", C.byref(a), _st(), kind=f"k{k}s1" + ("|x" if xw is not None else ""))
    return dz, part, dw


# ---- round 6: expand 1x1 conv + BatchNorm0 + swish inside the depthwise forward launch (conv_lane.hip MODE 4): the expanded
# tensor of an MBConv block never exists in HBM when no backward needs it stored
XDW = int(os.environ.get("MC_XDW", "1"))      # 0 never; 1 wherever the launch is supported (developer A/B switch)
# stride-1 3x3 blocks whose expanded tensor NEVER exists: fused forward above + fused backward with its e rows formed from the
# block input (mc_dwconv_bwd_fused with xw) + the folded BatchNorm0 backward; 0 = off (A/B)
EFREE = int(os.environ.get("MC_EFREE", "1"))
_XDW_OK = {}


def mbconv_xdw_ok(n, h, w, cin, c, k, stride, pad_l, pad_t, oh, ow):
    """does mc_mbconv_xdw_fwd take this block?  (geometry only; cached: the launch-bound configurations notice ctypes calls)"""
    if not XDW:
        return False
    key = (n, h, w, cin, c, k, stride, pad_l, pad_t, oh, ow)
    hit = _XDW_OK.get(key)
    if hit is None:
        a = _dw_args(n, h, w, c, k, stride, pad_l, pad_t, oh, ow)
        a.cin = cin
        hit = _XDW_OK[key] = bool(L.load().mc_mbconv_xdw_supported(C.byref(a)))
    return hit


def mbconv_xdw_fwd(x, we, pro, w_kkc, n, h, w, c, k, stride, pad_l, pad_t, oh, ow, stats=False):
    """d [n*oh*ow, c] = depthwise(silu(bn0(x . we^T))) in ONE launch [ref: efficientnet_custom.py:104-111]; x [n*h*w, cin],
    we [c, cin] (16-bit image), pro = (scale, shift) of BatchNorm0.  stats -> also the BatchNorm1 partials of d."""
    cin = x.shape[1]
    assert we.shape == (c, cin) and we.is_contiguous() and x.is_contiguous()
    a = _dw_args(n, h, w, c, k, stride, pad_l, pad_t, oh, ow)
    y = empty((n * oh * ow, c), BF16, x)
    a.x, a.out, a.w_kkc, a.xw, a.cin = _p(x), _p(y), _p(w_kkc), _p(we), cin
    a.pro_scale, a.pro_shift = _p(pro[0]), _p(pro[1])
    part = None
    if stats:
        rows = L.load().mc_mbconv_xdw_stat_rows(C.byref(a))
        part = empty((rows, 2, c), torch.float32, x)
        a.stat_partials, a.stat_rows = _p(part), rows
    _note(2 * n * (cin * h * w + c * oh * ow) + 2 * c * cin, 2 * n * c * (oh * ow * k * k + h * w * cin))
    L.call("mc_mbconv_xdw_fwd", C.byref(a), _st(), kind=f"k{k}s{stride}")
    return (y, part) if stats else y


def bn_gram_partials(x, we, rows):
    """BatchNorm statistics partials [2, 2, c] of e = x . we^T without e (bnfold.hip gram_partials_k): from the cin x cin Gram
    matrix and the column sums of x -- one pass over the 6 x narrower block input instead of a statistics epilogue over e.
    Returns (partials, (x^T x, colsum(x))): the folded BatchNorm0 backward of the same block needs the same two (bn_fold_expand_bwd)"""
    c, cin = we.shape
    xtx = linear_wgrad(x, x, tag="_xtx")
    cs = colsum(x)
    part = empty((2, 2, c), torch.float32, x)
    L.call("mc_bn_gram_partials", _p(we), we.stride(0), _p(xtx), _p(cs), float(rows), c, cin, _p(part), _st())
    return part, (xtx, cs)


def _dwconv_bwd_weight_impl(x, dy, n, h, w, c, k, stride, pad_l, pad_t, oh, ow, pro=None):
    a = _dw_args(n, h, w, c, k, stride, pad_l, pad_t, oh, ow)
    dw = torch.zeros((k * k, c), dtype=torch.float32, device=x.device)
    a.x, a.dy, a.out = _p(x), _p(dy), _p(dw)
    if pro is not None:
        a.pro_scale, a.pro_shift = _p(pro[0]), _p(pro[1])
    _note(2 * n * c * (h * w + oh * ow), 2 * n * c * oh * ow * k * k)
    L.call("mc_dwconv_bwd_weight", C.byref(a), _st(), kind=f"k{k}s{stride}")
    return dw


# ------------------------------------------------------------------------------------------- BN family
class BNStats:
    """Per-layer saved statistics of one training-mode BatchNorm call."""
    __slots__ = ("mean", "invstd", "scale", "shift", "count")


def bn_finalize(partials, count, gamma, beta, running_mean, running_var, momentum, eps, update_running):
    rows, _, c = partials.shape
    buf = empty((4, c), torch.float32, partials)
    L.call("mc_bn_finalize", _p(partials), rows, c, float(count), _p(gamma), _p(beta), _p(running_mean),
           _p(running_var), momentum, eps, int(update_running), _p(buf[0]), _p(buf[1]), _p(buf[2]), _p(buf[3]), _st())
    s = BNStats()
    s.mean, s.invstd, s.scale, s.shift, s.count = buf[0], buf[1], buf[2], buf[3], float(count)
    return s


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps):
    c = gamma.shape[0]
    buf = empty((4, c), torch.float32, gamma)
    L.call("mc_bn_eval_coeffs", _p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, c, _p(buf[2]), _p(buf[3]), _st())
    s = BNStats()
    s.mean, s.invstd, s.scale, s.shift, s.count = None, None, buf[2], buf[3], 0.0
    return s


def _bnact(x, n_img, hw, c, scale, shift, act):
    a = L.BnactArgs()
    a.x, a.n_img, a.hw, a.c = _p(x), n_img, hw, c
    a.scale, a.shift, a.act = _p(scale), _p(shift), act
    return a


def bnact_apply(x, n_img, hw, c, scale, shift, act, rowscale=None, res=None):
    a = _bnact(x, n_img, hw, c, scale, shift, act)
    out = empty((n_img * hw, c), BF16, x)
    a.rowscale, a.res, a.out = _p(rowscale), _p(res), _p(out)
    _note(2 * n_img * hw * c * (3 if res is not None else 2))
    L.call("mc_bnact_apply", C.byref(a), _st())
    return out


def _split_ws(a, x, n_img, c, planes=1):
    """scratch for the per-image row splits (combined in split order: bit-reproducible, no float atomics)"""
    sp = L.load().mc_bnact_img_splits(C.byref(a))
    if sp <= 1:
        return None
    ws = empty((sp, planes, n_img, c), torch.float32, x)
    a.split_ws = _p(ws)
    return ws


def bnact_pool(x, n_img, hw, c, scale, shift, act, keep_act=False):
    """pooled[n_img, c] = mean over the image of act(x*scale+shift); keep_act -> also returns act(.) as bf16"""
    a = _bnact(x, n_img, hw, c, scale, shift, act)
    pooled = empty((n_img, c), torch.float32, x)
    a.pooled = _p(pooled)
    y = None
    if keep_act:
        y = empty((n_img * hw, c), BF16, x)
        a.out = _p(y)
    ws = _split_ws(a, x, n_img, c)  # noqa: F841  (kept alive until the launch is enqueued)
    _note((4 if keep_act else 2) * n_img * hw * c)
    L.call("mc_bnact_pool", C.byref(a), _st())
    return (pooled, y) if keep_act else pooled


def bnact_se_dgate(x, g, n_img, hw, c, scale, shift, act):
    a = _bnact(x, n_img, hw, c, scale, shift, act)
    dgate = empty((n_img, c), torch.float32, x)
    a.g, a.dgate = _p(g), _p(dgate)
    ws = _split_ws(a, x, n_img, c)  # noqa: F841
    _note(4 * n_img * hw * c)
    L.call("mc_bnact_se_dgate", C.byref(a), _st())
    return dgate


def bnact_se_sums(x, g, n_img, hw, c, stats, act):
    """[5, n_img, c] per-image sums of ONE pass over (x, g): see mc_bnact_se_sums."""
    a = _bnact(x, n_img, hw, c, stats.scale, stats.shift, act)
    sums = empty((5, n_img, c), torch.float32, x)
    a.g, a.dgate, a.mean, a.invstd = _p(g), _p(sums), _p(stats.mean), _p(stats.invstd)
    ws = _split_ws(a, x, n_img, c, planes=5)  # noqa: F841
    _note(4 * n_img * hw * c)
    L.call("mc_bnact_se_sums", C.byref(a), _st())
    return sums


def bn_partials_from_se_sums(sums, gate, dpooled, add_scale):
    _, n_img, c = sums.shape
    part = empty((n_img, 2, c), torch.float32, sums)
    L.call("mc_bn_partials_from_se_sums", _p(sums), _p(gate), _p(dpooled), float(add_scale), n_img, c, _p(part), _st())
    return part


def bnact_bwd(x, n_img, hw, c, stats, gamma, act, g=None, mul=None, add=None, rowscale=None, add_scale=1.0,
              partials=None):
    """Backward through y = act(BN_train(x)) (* rowscale).  Returns (dx bf16, dgamma, dbeta).
    partials: precomputed [rows, 2, c] reduction (skips the reduce pass over the big tensors)."""
    a = _bnact(x, n_img, hw, c, stats.scale, stats.shift, act)
    a.g, a.mul, a.add, a.rowscale = _p(g), _p(mul), _p(add), _p(rowscale)
    a.add_scale = add_scale
    a.mean, a.invstd = _p(stats.mean), _p(stats.invstd)
    if partials is not None:
        part, rows = partials, partials.shape[0]
    else:
        rows = L.load().mc_bnact_rows(C.byref(a))
        part = empty((rows, 2, c), torch.float32, x)
        a.partials = _p(part)
        _note(2 * n_img * hw * c * (2 if g is not None else 1))
        L.call("mc_bnact_bwd_reduce", C.byref(a), _st())
    # dgamma / dbeta are handed to autograd as parameter gradients: own tensors (a row of a shared buffer is a view, which
    # AccumulateGrad clones -- one copy launch per gradient); coefA/B/C stay together
    dgamma, dbeta = empty((c,), torch.float32, x), empty((c,), torch.float32, x)
    coef = empty((3, c), torch.float32, x)
    L.call("mc_bn_bwd_finalize", _p(part), rows, c, float(n_img * hw), _p(gamma), _p(stats.mean), _p(stats.invstd),
           _p(dgamma), _p(dbeta), _p(coef), _st())
    dx = empty((n_img * hw, c), BF16, x)
    a.coef, a.dx = _p(coef), _p(dx)
    _note(2 * n_img * hw * c * (3 if g is not None else 2))
    L.call("mc_bnact_bwd_apply", C.byref(a), _st())
    return dx, dgamma, dbeta


def bnact_bwd_reduce_dz(x, n_img, hw, c, stats, act, g):
    """One pass over (x, g): dz = g * act'(bn(x)) stored (bf16) AND the BatchNorm-backward reductions (sum dz,
    sum dz*xhat) as partials -- the input of bn_bwd_coefs / bn_fold_expand_bwd for blocks whose data-gradient kernel has
    no BatchNorm epilogue (stride 2)."""
    a = _bnact(x, n_img, hw, c, stats.scale, stats.shift, act)
    a.g = _p(g)
    a.mean, a.invstd = _p(stats.mean), _p(stats.invstd)
    rows = L.load().mc_bnact_rows(C.byref(a))
    part = empty((rows, 2, c), torch.float32, x)
    dz = empty((n_img * hw, c), BF16, x)
    a.partials, a.dx = _p(part), _p(dz)
    _note(2 * n_img * hw * c * 3)
    L.call("mc_bnact_bwd_reduce", C.byref(a), _st(), kind="dz")
    return dz, part


def bn_bwd_coefs(partials, count, stats, gamma):
    """Finalize a BatchNorm-backward reduction: partials [rows, 2, c] = (sum dz, sum dz*xhat) -> (coef [3, c], dgamma,
    dbeta) with dx = coef[0]*dz + coef[1]*x + coef[2] (the apply pass, or the folded GEMM operands of bn_fold_*)."""
    c = partials.shape[-1]
    dgamma, dbeta = empty((c,), torch.float32, partials), empty((c,), torch.float32, partials)
    coef = empty((3, c), torch.float32, partials)
    L.call("mc_bn_bwd_finalize", _p(partials), partials.shape[0], c, float(count), _p(gamma), _p(stats.mean),
           _p(stats.invstd), _p(dgamma), _p(dbeta), _p(coef), _st())
    return coef, dgamma, dbeta


def bn_fold_expand_bwd(dz, x, we_f32, we_bf16, coef, dbeta, rows, residual=None, gram=None):
    """Backward of e = x We^T under training-mode BatchNorm, from dz = dL/d bn(e) alone (bnfold.hip): returns
    (dx [rows, cin] bf16 (+ residual), dWe [cexp, cin] fp32).  Neither e nor de is read or written: the BatchNorm
    backward's linear combination lives in the small folded operands (A.We, G = We^T diag(B) We, Sxx)."""
    n, k = we_f32.shape
    fused = xbwd_rows_ok(x.shape[0], n, k)                  # round 5: dz is read ONCE for both of its GEMMs (below)
    t1 = None if fused else linear_wgrad(dz, x)            # dz^T x   [cexp, cin]
    if gram is not None:                                   # (round 6: the forward's Gram statistics pass left both behind)
        xtx, cs = gram
    else:
        xtx = linear_wgrad(x, x, tag="_xtx")               # x^T x    [cin, cin]
        cs = colsum(x)
    w1t, wb, sxx = empty((k, n), BF16, x), empty((n, k), BF16, x), empty((k, k), BF16, x)
    L.call("mc_bn_fold_prepare", _p(we_f32), _p(coef), _p(xtx), _p(cs), float(rows), n, k, _p(w1t), _p(wb), _p(sxx), _st())
    gt = linear_wgrad(we_bf16, wb, tag="_gt")              # (We^T (B.We))^T  [cin, cin] fp32
    gtb, cvec = empty((k, k), BF16, x), empty((k,), torch.float32, x)
    L.call("mc_bn_fold_cvec", _p(gt), _p(we_f32), _p(coef), _p(dbeta), _p(cs), float(rows), n, k, _p(gtb), _p(cvec), _st())
    if BF16 == torch.bfloat16:
        r = linear_fwd(x, gtb, bias=cvec, residual=residual, tag="_fold")   # x G + cvec (+ skip gradient)
    else:
        # f16 storage build: gtb holds rows * G (bnfold.hip keeps the folded operands inside f16's exponent range), the
        # factor goes back in through the tile GEMM's alpha
        r = empty((x.shape[0], k), BF16, x)
        gemm(x, gtb, r, x.shape[0], k, k, x.stride(0), k, k, bias=cvec, R=residual,
             ldr=(residual.stride(0) if residual is not None else 0), alpha=1.0 / float(rows), kind="fwd_fold")
    if fused:
        dx, t1 = xbwd_rows(dz, x, w1t, residual=r)         # dz (A.We) + r   and   dz^T x   from one pass over dz
    else:
        dx = linear_dgrad(dz, we_bf16, residual=r, w_t=w1t)    # + dz (A.We)
    wx = empty((n, k), torch.float32, x)
    gemm(wb, sxx, wx, n, k, k, k, k, k, c_f32=1, kind="fold")      # (B.We) Sxx  (Sxx symmetric)
    dwe = t1
    L.call("mc_bn_fold_wgrad", _p(t1), _p(wx), _p(coef), _p(dbeta), _p(cs), float(rows), n, k, _p(dwe), _st())
    return dx, dwe


# ------------------------------------------------------------------------------------------- SE / dropout
def se_fwd(pooled, w1, b1, w2, b2):
    n, c = pooled.shape
    cs = w1.shape[0]
    gate = empty((n, c), torch.float32, pooled)
    ws = empty((n, cs), torch.float32, pooled)
    L.call("mc_se_fwd", _p(pooled), _p(w1), _p(b1), _p(w2), _p(b2), n, c, cs, _p(gate), _p(ws), _st())
    return gate


def se_bwd(pooled, gate, dgate, w1, b1, w2, b2):
    n, c = pooled.shape
    cs = w1.shape[0]
    dpooled = empty((n, c), torch.float32, pooled)
    # four own tensors (they are handed to autograd as parameter gradients: views of a shared buffer would be cloned by
    # AccumulateGrad); the kernel writes every element, so no zero-fill either
    dw1, dw2 = empty((cs, c), torch.float32, pooled), empty((c, cs), torch.float32, pooled)
    db1, db2 = empty((cs,), torch.float32, pooled), empty((c,), torch.float32, pooled)
    ws = empty((n, c + 2 * cs), torch.float32, pooled)
    L.call("mc_se_bwd", _p(pooled), _p(gate), _p(dgate), _p(w1), _p(b1), _p(w2), _p(b2), n, c, cs, _p(dpooled),
           _p(dw1), _p(db1), _p(dw2), _p(db2), _p(ws), _st())
    return dpooled, dw1, db1, dw2, db2


def dropout_f32(x, p, seed, stream_id):
    if p <= 0.0:
        return x
    y = torch.empty_like(x)
    L.call("mc_dropout_f32", _p(x), _p(y), x.numel(), float(p), int(seed), int(stream_id), _st())
    return y


# ------------------------------------------------------------------------------------------- BERT pieces
def bert_embed_fwd(ids, tt, word, pos, typ, gamma, beta, eps, p, seed, sid):
    b, t = ids.shape
    h = word.shape[1]
    y = empty((b * t, h), BF16, word)
    mean = empty((b * t,), torch.float32, word)
    rstd = empty((b * t,), torch.float32, word)
    L.call("mc_bert_embed_fwd", _p(ids), _p(tt), _p(word), _p(pos), _p(typ), _p(gamma), _p(beta), eps, b, t, h,
           float(p), int(seed), int(sid), _p(y), _p(mean), _p(rstd), _st())
    return y, mean, rstd


def bert_embed_bwd(dy, ids, tt, word, pos, typ, gamma, mean, rstd, p, seed, sid):
    b, t = ids.shape
    h = word.shape[1]
    dword, dpos, dtyp = torch.zeros_like(word), torch.zeros_like(pos), torch.zeros_like(typ)
    dgamma, dbeta = torch.zeros_like(gamma), torch.zeros_like(gamma)
    L.call("mc_bert_embed_bwd", _p(dy), _p(ids), _p(tt), _p(word), _p(pos), _p(typ), _p(gamma), _p(mean), _p(rstd),
           b, t, h, float(p), int(seed), int(sid), _p(dword), _p(dpos), _p(dtyp), _p(dgamma), _p(dbeta), _st())
    return dword, dpos, dtyp, dgamma, dbeta


def add_ln_fwd(x, res, gamma, beta, eps, p, seed, sid):
    rows, h = x.shape
    y = empty((rows, h), BF16, x)
    mean = empty((rows,), torch.float32, x)
    rstd = empty((rows,), torch.float32, x)
    L.call("mc_add_ln_fwd", _p(x), _p(res), _p(gamma), _p(beta), eps, rows, h, float(p), int(seed), int(sid), _p(y),
           _p(mean), _p(rstd), _st())
    return y, mean, rstd


def add_ln_bwd(dy, x, res, gamma, mean, rstd, p, seed, sid):
    rows, h = x.shape
    dx, dres = empty((rows, h), BF16, x), empty((rows, h), BF16, x)
    gb = torch.zeros((2,) + tuple(gamma.shape), dtype=gamma.dtype, device=gamma.device)
    dgamma, dbeta = gb[0], gb[1]
    L.call("mc_add_ln_bwd", _p(dy), _p(x), _p(res), _p(gamma), _p(mean), _p(rstd), rows, h, float(p), int(seed),
           int(sid), _p(dx), _p(dres), _p(dgamma), _p(dbeta), _st())
    return dx, dres, dgamma, dbeta


def softmax_fwd(scores, p, seed, sid):
    t = scores.shape[-1]
    rows = scores.numel() // t
    probs = empty(scores.shape, BF16, scores)
    pd = empty(scores.shape, BF16, scores) if p > 0 else probs
    L.call("mc_softmax_fwd", _p(scores), rows, t, float(p), int(seed), int(sid), _p(probs), _p(pd), _st())
    return probs, pd


def softmax_bwd(probs, dpd, p, seed, sid, alpha):
    t = probs.shape[-1]
    rows = probs.numel() // t
    ds = empty(probs.shape, BF16, probs)
    L.call("mc_softmax_bwd", _p(probs), _p(dpd), rows, t, float(p), int(seed), int(sid), float(alpha), _p(ds), _st())
    return ds


def attn_supported(t, head_dim):
    """Whether the fused attention kernels take this shape (otherwise: batched GEMM + softmax kernels)."""
    import os
    if os.environ.get("MC_FUSED_ATTN", "1") == "0":      # developer switch: A/B against the unfused kernels
        return False
    return bool(L.load().mc_attn_supported(int(t), int(head_dim)))


def attn_fwd(qkv, maskb, b, t, nh, alpha, p, seed, sid):
    """ctx [b*t, nh*64] bf16 and lse [b*nh*t, 2] fp32 from qkv [b*t, 3*nh*64] (fused scores/softmax/dropout/context)."""
    ctx = empty((b * t, nh * 64), BF16, qkv)
    lse = empty((b * nh * t, 2), torch.float32, qkv)
    L.call("mc_attn_fwd", _p(qkv), _p(maskb), b, t, nh, float(alpha), float(p), int(seed), int(sid), _p(ctx), _p(lse), _st())
    return ctx, lse


def attn_bwd(qkv, maskb, dctx, lse, b, t, nh, alpha, p, seed, sid):
    dqkv = empty(qkv.shape, BF16, qkv)
    L.call("mc_attn_bwd", _p(qkv), _p(maskb), _p(dctx), _p(lse), b, t, nh, float(alpha), float(p), int(seed), int(sid),
           _p(dqkv), _st())
    return dqkv


def gelu_fwd(x):
    y = torch.empty_like(x)
    L.call("mc_gelu_fwd", _p(x), _p(y), x.numel(), _st())
    return y


def gelu_bwd(dy, x):
    dx = torch.empty_like(x)
    L.call("mc_gelu_bwd", _p(dy), _p(x), _p(dx), x.numel(), _st())
    return dx


def mask_bias(mask):
    out = empty(mask.shape, torch.float32, mask)
    L.call("mc_mask_bias", _p(mask), _p(out), mask.numel(), _st())
    return out


def eos_gather(hid, mask, b, t, h):
    out = empty((b, h), torch.float32, hid)
    L.call("mc_eos_gather", _p(hid), _p(mask), b, t, h, _p(out), _st())
    return out


def eos_scatter(dout, mask, b, t, h):
    dh = empty((b * t, h), BF16, dout)
    L.call("mc_eos_scatter", _p(dout), _p(mask), b, t, h, _p(dh), _st())
    return dh


# ------------------------------------------------------------------------------------------- heads / loss
def sgemm(a, ars, acs, b, brs, bcs, c, ldc, m, n, k, alpha=1.0, beta=0.0, bias=None, alpha_dev=None):
    nws = L.load().mc_sgemm_ws_floats(m, n, k)
    ws = empty((nws,), torch.float32, c) if nws > 0 else None
    L.call("mc_sgemm", _p(a), ars, acs, _p(b), brs, bcs, _p(c), ldc, m, n, k, float(alpha), float(beta), _p(bias),
           _p(alpha_dev), _p(ws), _st())


def scale_f32(x, scalar_dev=None, alpha=1.0):
    y = torch.empty_like(x)
    L.call("mc_scale_f32", _p(x), _p(scalar_dev), float(alpha), _p(y), x.numel(), _st())
    return y


def l2norm_fwd(x):
    rows, d = x.shape
    y = torch.empty_like(x)
    norm = empty((rows,), torch.float32, x)
    L.call("mc_l2norm_fwd", _p(x), rows, d, _p(y), _p(norm), _st())
    return y, norm


def l2norm_bwd(dy, y, norm):
    rows, d = y.shape
    dx = torch.empty_like(y)
    L.call("mc_l2norm_bwd", _p(dy.contiguous()), _p(y), _p(norm), rows, d, _p(dx), _st())
    return dx


def ce_fwd_bwd(logits, label_offset, w, loss_out, smoothing=0.0, labels=None):
    rows, n = logits.shape
    row_ws = empty((rows,), torch.float32, logits)
    L.call("mc_ce_fwd_bwd", _p(logits), rows, n, _p(labels), int(label_offset), float(w), float(smoothing), _p(loss_out),
           _p(row_ws), _st())


# ------------------------------------------------------------------------------------------- dispatcher-visible operators
# The 1x1-convolution / linear and depthwise-convolution calls of the model go through ``torch.ops.mammoclip.*`` (registered
# in custom_ops.py with torch.library: schema + HIP implementation + meta implementation; ~2.5 us of dispatcher per call):
# they show up in the profiler and in traces under their operator names.  The wrappers below keep the keyword interface the
# autograd functions use; forms the operator schemas do not carry (caller-provided output buffer, developer timing tags,
# the BatchNorm-backward epilogue of the stride-1 depthwise data gradient, GELU epilogue) call the implementation directly.
def _none_if_empty(t):
    return None if t is None or t.numel() == 0 else t


def linear_fwd(x, w, bias=None, act=0, residual=None, stats=False, pro=None, out=None, tag=""):
    if out is not None or tag or act or not (torch.is_tensor(x) and x.is_cuda):
        _chk_dev(x, w)
        return _linear_fwd_impl(x, w, bias, act, residual, stats, pro, out, tag)
    ps, pf, pg, rpi = pro if pro is not None else (None, None, None, 0)
    y, part = _OP_CONV1X1(x, w, bias, residual, ps, pf, pg, int(rpi), bool(stats), pro is not None)
    return (y, part) if stats else y


def linear_dgrad(dy, w, residual=None, w_t=None):
    if not dy.is_cuda:
        _chk_dev(dy, w)
    return _OP_CONV1X1_DGRAD(dy, w, residual, w_t)


def linear_wgrad(dy, x, pro=None, out=None, tag=""):
    if out is not None or tag or not dy.is_cuda:
        _chk_dev(dy, x)
        return _linear_wgrad_impl(dy, x, pro, out, tag)
    ps, pf, pg, rpi = pro if pro is not None else (None, None, None, 0)
    return _OP_CONV1X1_WGRAD(dy, x, ps, pf, pg, int(rpi), pro is not None)


def dwconv_fwd(x, w_kkc, n, h, w, c, k, stride, pad_l, pad_t, oh, ow, pro=None, stats=False, epi=None):
    if epi is not None or not x.is_cuda:
        _chk_dev(x, w_kkc)
        return _dwconv_fwd_impl(x, w_kkc, n, h, w, c, k, stride, pad_l, pad_t, oh, ow, pro, stats, epi)
    ps, pf = pro if pro is not None else (None, None)
    y, part = _OP_DWCONV_BN(x, w_kkc, n, h, w, k, stride, pad_l, pad_t, oh, ow, ps, pf, bool(stats))
    return (y, part) if stats else y


def dwconv_bwd_data(dy, w_kkc, n, h, w, c, k, stride, pad_l, pad_t, oh, ow, w_kkc_flipped=None, epi=None):
    if epi is not None or not dy.is_cuda:
        _chk_dev(dy, w_kkc)
        return _dwconv_bwd_data_impl(dy, w_kkc, n, h, w, c, k, stride, pad_l, pad_t, oh, ow, w_kkc_flipped, epi)
    return _OP_DWCONV_DGRAD(dy, w_kkc, n, h, w, k, stride, pad_l, pad_t, oh, ow, w_kkc_flipped)


def dwconv_bwd_weight(x, dy, n, h, w, c, k, stride, pad_l, pad_t, oh, ow, pro=None):
    if not x.is_cuda:
        _chk_dev(x, dy)
    ps, pf = pro if pro is not None else (None, None)
    return _OP_DWCONV_WGRAD(x, dy, n, h, w, k, stride, pad_l, pad_t, oh, ow, ps, pf)


# ------------------------------------------------------------------------------------------- gradient sink
class GradSink:
    """Parameter gradients of the hand-written backward functions, combined by multi-tensor adds.

    Autograd sums the gradients a parameter receives with one ``at::native add`` launch per parameter and contribution:
    both image views use the image encoder's ~700 parameters (two contributions per backward), a micro-batched step runs
    up to 32 backward calls -- 38 000 tiny launches per 1024-pair step, 2 % of its GPU time.  With a sink installed
    (``ops.GRAD_SINK``, engine.Trainer does it) the stem / MBConv / head / BERT functions hand their parameter gradients
    over here and return None to autograd; ``flush()`` (after every backward call) folds the pending gradients into one
    accumulator per parameter with ``torch._foreach_add_`` (a handful of launches for all tensors), ``finish()`` leaves the
    result in ``param.grad``.  No sink installed: plain autograd semantics (what DDP wrappers and third-party loops see).
    """

    def __init__(self):
        self.acc = {}
        self.pending = []

    def deliver(self, params, grads):
        for p_, g in zip(params, grads):
            if g is not None and p_.requires_grad:
                # contiguous fp32 storage, kept as a FLAT view: every list handed to _foreach_add_ then has identical
                # (1-D, unit) strides and one dtype, which is what its multi-tensor fast path requires -- a single
                # 4-D view with a different size-1 stride sends the whole call down the one-launch-per-tensor path
                if g.dtype != p_.dtype:
                    g = g.to(p_.dtype)
                self.pending.append((p_, g.contiguous().view(-1)))

    @torch.no_grad()
    def flush(self):
        """Same summation order as autograd: the contributions of ONE backward call are summed among themselves first
        (in arrival order, like the engine's input buffer), then added to what earlier calls left (like AccumulateGrad)."""
        if not self.pending:
            return
        groups = {}
        for p_, g in self.pending:
            groups.setdefault(p_, []).append(g)
        self.pending = []
        level = 1
        while True:
            dst = [gs[0] for gs in groups.values() if len(gs) > level]
            if not dst:
                break
            torch._foreach_add_(dst, [gs[level] for gs in groups.values() if len(gs) > level])
            level += 1
        dst, src = [], []
        for p_, gs in groups.items():
            a = self.acc.get(p_)
            if a is None:
                self.acc[p_] = gs[0]                       # the first gradient becomes the accumulator
            else:
                dst.append(a)
                src.append(gs[0])
        if dst:
            torch._foreach_add_(dst, src)

    @torch.no_grad()
    def finish(self):
        self.flush()
        for p_, a in self.acc.items():
            a = a.view_as(p_)
            if p_.grad is None:
                p_.grad = a
            else:
                p_.grad.add_(a)
        self.acc = {}


GRAD_SINK = None


def deliver_param_grads(params, grads):
    """backward functions: hand ``grads`` (same order as ``params``) to the installed sink and return Nones for autograd,
    or return them unchanged when no sink is installed"""
    if GRAD_SINK is None:
        return tuple(grads)
    GRAD_SINK.deliver(params, grads)
    return (None,) * len(grads)


# the overloads the wrappers above call (torch.ops.mammoclip.<name>.default, resolved once: two attribute look-ups and the
# overload resolution per call are host time the launch-bound configurations notice).  custom_ops binds them through
# _bind_model_ops() at the end of its own import, so either module may be imported first.
_OP_CONV1X1 = _OP_CONV1X1_DGRAD = _OP_CONV1X1_WGRAD = _OP_DWCONV_BN = _OP_DWCONV_DGRAD = _OP_DWCONV_WGRAD = None


def _bind_model_ops():
    global _OP_CONV1X1, _OP_CONV1X1_DGRAD, _OP_CONV1X1_WGRAD, _OP_DWCONV_BN, _OP_DWCONV_DGRAD, _OP_DWCONV_WGRAD
    ns = torch.ops.mammoclip
    _OP_CONV1X1, _OP_CONV1X1_DGRAD, _OP_CONV1X1_WGRAD = ns.conv1x1.default, ns.conv1x1_dgrad.default, ns.conv1x1_wgrad.default
    _OP_DWCONV_BN, _OP_DWCONV_DGRAD, _OP_DWCONV_WGRAD = ns.dwconv_bn.default, ns.dwconv_dgrad.default, ns.dwconv_wgrad.default


from . import custom_ops  # noqa: E402,F401  (registers torch.ops.mammoclip.* and calls _bind_model_ops(); imported last)
