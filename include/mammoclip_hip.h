/*
 * mammoclip_hip.h -- C ABI of libmammoclip_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * Mammo-CLIP image<->text contrastive pre-training hot path.
 *
 * The reference (batmanlab/Mammo-CLIP) has no FFI of its own: its hot path bottoms out in torch
 * operators (conv2d / batch_norm / linear / matmul / cross_entropy ...).  Every entry point below
 * replaces one such operator call-site (or a fused group of them) and cites it as
 *   [ref: <file>:<lines>]  relative to  src/codebase/breastclip/ .
 * The host-side mirror of the reference's Python API (build_model / build_loss / BreastClip ...) lives in
 * mammo_clip_amd/breastclip/ and binds these symbols with ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless stated otherwise
 *   - bf16 tensors are raw uint16 bit patterns; activations are row-major [rows, channels] (NHWC),
 *     channels % 8 == 0, base pointers 16-byte aligned
 *   - the 16-bit storage / MFMA operand type is a property of the library BUILD: libmammoclip_hip.so holds bf16
 *     (default), libmammoclip_hip_f16.so (compiled from the same sources with -DMC_F16) holds IEEE f16 -- the
 *     reference's AMP dtype [ref: trainer.py:271-278] -- behind the SAME symbols: wherever this header says "bf16" /
 *     mc_bf16, the f16 build reads and writes f16 bit patterns.  mc_storage_is_f16() tells which one a process loaded.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous
 *   - return 0 on success, non-zero on error; mc_last_error() gives the message (thread-local)
 *   - no hidden allocation: workspaces / partial buffers are caller-provided
 */
#ifndef MAMMOCLIP_HIP_H
#define MAMMOCLIP_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t mc_bf16;

const char* mc_last_error(void);
int mc_version(void);
int mc_storage_is_f16(void);   /* 0: bf16 build, 1: f16 build (see Conventions) */

/* ------------------------------------------------------------------------------------------------
 * bf16 MFMA GEMM:  C[M,N] (+)= alpha * op(A)[M,K] . op(B)[K,N] + bias[N]  -> act -> + R
 * [ref: model/modules/efficientnet_custom.py:104,122,283 (1x1 convs _expand_conv/_project_conv/_conv_head),
 *       model/modules/text_encoder.py:48 (BertModel linears and attention matmuls),
 *       and their autograd backward (dgrad: b_kmajor, wgrad: a_kmajor+b_kmajor+split-K atomics)]
 * a_kmajor = 0: A[m*lda + k]   1: A[k*lda + m]        b_kmajor = 0: B[n*ldb + k]   1: B[k*ldb + n]
 * batch index z -> (b1, b2) = (z / nb2, z % nb2); operand offsets b1*s?1 + b2*s?2 (elements).
 * Prologue (fused BN + SiLU (+ SE gate) on the streamed operand, [ref: efficientnet_custom.py:110-119]):
 *   pro_operand 1 = A (k-contiguous: row = pixel, k = channel), 2 = B (k-major: k = pixel, n = channel)
 *   v' = silu(v*pro_scale[c] + pro_shift[c]) * pro_gate[(pixel / pro_rows_per_img) * pro_nch + c]
 * stat_partials (optional, bf16 output only): float[rows][2][N] with rows = mc_gemm_stat_rows();
 *   row r holds the column sums / sums of squares of the C rows handled by workgroup-row r.
 */
typedef struct mc_gemm_args {
    const mc_bf16* A;
    const mc_bf16* B;
    void* C;
    long long M, K;
    int N;
    long long lda, ldb, ldc;
    int a_kmajor, b_kmajor;
    int c_f32, c_atomic;
    int splits;
    int batch, nb2;
    long long sA1, sA2, sB1, sB2, sC1, sC2;
    const float* bias;
    long long bias_stride1;
    int act;                 /* 0 none, 1 erf-GELU */
    const mc_bf16* R;
    long long ldr;
    float alpha;             /* 0 is read as 1 */
    int pro_operand;
    const float* pro_scale;
    const float* pro_shift;
    const float* pro_gate;
    long long pro_rows_per_img;
    int pro_nch;
    float* stat_partials;
    int max_grid_m;          /* 0 = default */
    float* splitk_ws;        /* optional float[splits*M*N]: split-K partial tiles are combined by a second kernel
                                (C = sum, or C += sum when c_atomic) instead of per-element atomics */
    /* grouped split-K (weight gradient of a conv whose input carries a per-image, per-channel factor, e.g. the SE gate):
     * the reduction index is cut at multiples of split_group_rows (rows per image), split_sub splits per group
     * (splits = groups * split_sub), and partial s is multiplied by split_scale[s / split_sub][column] when the
     * partials are combined:  dW[n][k] = sum_img gate[img][k] * sum_{m in img} dY[m][n] * A[m][k]  -- no prologue needed */
    long long split_group_rows;
    int split_sub;
    const float* split_scale;
    /* fp8 operands (BASELINE config #5: fp8 weights / activations for the pointwise convolutions,
     * efficientnet_custom.py:104,122,283): A and B hold OCP e4m3 bytes (gfx950's fp8, NOT the fnuz form), lda / ldb /
     * strides in elements (= bytes), K, lda, ldb multiples of 16; plain NT layout, bf16 output, fp32 accumulation
     * (v_mfma_f32_16x16x32_fp8_fp8).  The product of the two per-tensor dequantisation scales is passed in alpha and/or
     * alpha_dev (device scalar, multiplied onto alpha; produced by mc_quant_fp8_bf16 without a host round trip). */
    int ab_fp8;
    const float* alpha_dev;
} mc_gemm_args;
int mc_gemm_bf16(const mc_gemm_args* args, void* stream);
/* rows of the stat_partials buffer the launch described by `args` (the COMPLETE argument block) will write */
int mc_gemm_stat_rows(const mc_gemm_args* args);
/* which tile kernel serves `args`: 256 = the 256 x 256 x 64 direct-to-LDS kernel (gemm256.hip: plain NT operands, bf16
 * output, enough well-filled tiles for the 256 CUs), 128 = the 128-row tile family (gemm.hip) */
int mc_gemm_tile_config(const mc_gemm_args* args);
/* Weight-gradient (TN) problems -- a_kmajor = b_kmajor = 1, fp32 output, plain operands, K = the long pixel / token
 * reduction [ref: autograd backward of efficientnet_custom.py:104,122,283, text_encoder.py:47-49] -- run on the
 * 256 x 256 x 64 transpose-read kernel of gemm256_tn.hip when mc_gemm256_tn_eligible() says so (mc_gemm_bf16 routes them
 * itself).  mc_gemm256_tn_splits: the number of K splits that kernel wants (splitk_ws = float[splits*M*N]); with
 * group_rows > 0 (grouped form, split_group_rows) the number of sub-splits per group (split_sub). */
int mc_gemm256_tn_eligible(const mc_gemm_args* args);
int mc_gemm256_tn_splits(long long m, long long n, long long k, long long group_rows);

/* fp8 operand preparation (BASELINE config #5; OCP e4m3 as implemented by gfx950).
 * mc_amax_bf16:       amax[0] = max(amax[0], max |x|)   (caller zero-fills amax; integer atomic, deterministic)
 * mc_quant_fp8_bf16:  y = e4m3(clamp(x * 448 / amax_in[0], +-448));  scale_out[0] = amax_in[0] / 448 (the dequantisation
 *                     scale, optional);  amax_next[0] = max(amax_next[0], max |x|) (optional: delayed scaling measures the
 *                     tensor it converts);  n % 8 == 0, contiguous. */
int mc_amax_bf16(const mc_bf16* x, long long n, float* amax, void* stream);
int mc_quant_fp8_bf16(const mc_bf16* x, long long n, const float* amax_in, unsigned char* y, float* scale_out,
                      float* amax_next, void* stream);

/* Row-streaming variant for the HBM-bound 1x1 convolutions (small weight matrix, millions of pixels):
 *   C[M,N] = pro(X)[M,K] . W[N,K]^T (+ R);  N <= 256, K <= 384 (mc_gemm_rows_supported), bf16 in/out.
 * Weights live in LDS for the whole launch, each workgroup writes complete, contiguous output rows.
 * The data gradient of a 1x1 conv is the same call with the transposed weight (mc_cast_transpose_f32_bf16).
 * stat_partials (optional): float[mc_gemm_rows_blocks(args)][2][N] column sums / sums of squares of C. */
typedef struct mc_gemm_rows_args {
    const mc_bf16* X;
    long long M;
    int K;
    long long ldx;
    const mc_bf16* W;
    int N;
    long long ldw;
    mc_bf16* C;
    long long ldc;
    const mc_bf16* R;
    long long ldr;
    const float* pro_scale;
    const float* pro_shift;
    const float* pro_gate;
    long long pro_rows_per_img;
    float* stat_partials;
    const float* bias;       /* optional float[N], added to every output row (after the statistics, with the residual) */
    /* Fused elementwise epilogues for the DATA GRADIENT of an MBConv projection conv (round 3).  G = X.W^T is d loss / d (gated
     * activation) [ref: efficientnet_custom.py:117-122 backwards]; it is rounded to bf16 like the stored tensor would be but never
     * goes to memory (N <= 256, K <= 128, no residual / bias / statistics / prologue):
     *   epi_mode 1: the five per-image sums of mc_bnact_se_sums over (epi_x, G) -> epi_sums [5][n_img][N]; C is not written;
     *   epi_mode 2: mc_bnact_bwd_apply on (epi_x, G): C = coef0*dz + coef1*epi_x + coef2,
     *               dz = (G * epi_mul[img] + epi_add[img] * epi_add_scale) * silu'(epi_x * epi_scale + epi_shift).
     * epi_x [M,N] (leading dimension epi_ldx) = the depthwise conv output in front of BatchNorm1; epi_rows_per_img % 16 == 0. */
    int epi_mode;
    const mc_bf16* epi_x;
    long long epi_ldx;
    long long epi_rows_per_img;
    const float* epi_scale;
    const float* epi_shift;
    const float* epi_mean;
    const float* epi_invstd;
    const float* epi_coef;   /* [3][N] (mc_bn_bwd_finalize) */
    const float* epi_mul;    /* [n_img][N] */
    const float* epi_add;    /* [n_img][N] */
    float epi_add_scale;
    float* epi_sums;         /* epi_mode 1: [5][n_img][N] */
    float* epi_ws;           /* epi_mode 1: float[mc_gemm_rows_epi_ws_floats(args)] */
} mc_gemm_rows_args;
int mc_gemm_rows_supported(int n, int k);
long long mc_gemm_rows_epi_ws_floats(const mc_gemm_rows_args* args);
/* can mc_gemm_rows_bf16 run the epilogue form epi_mode (1 / 2) of an [M, K] x [N, K]^T data gradient with rows_per_img rows per
 * image?  (N <= 256, K <= 128, whole 16-row groups per image, at most one image boundary per wave range) */
int mc_gemm_rows_epi_supported(long long M, int N, int K, long long rows_per_img, int epi_mode);
int mc_gemm_rows_blocks(const mc_gemm_rows_args* args);     /* persistent workgroups of the launch = rows of stat_partials */
int mc_gemm_rows_bf16(const mc_gemm_rows_args* args, void* stream);

/* Streaming weight gradient of the same layers: dW[N,K] (fp32) (+)= dY[M,N]^T . pro(X)[M,K]; both operands are
 * staged row-major and read with gfx950's LDS transpose-read (ds_read_b64_tr_b16); no atomics.
 * ws: float[mc_wgrad_rows_blocks(M) * N * K] scratch (per-workgroup partials, reduced by a second kernel). */
typedef struct mc_wgrad_rows_args {
    const mc_bf16* dY;
    int N;
    long long lddy;
    const mc_bf16* X;
    int K;
    long long ldx;
    long long M;
    float* dW;               /* [N, K] contiguous */
    float* ws;
    int accumulate;
    const float* pro_scale;  /* prologue on X: silu(x*scale[k] + shift[k]) * gate[(m / rows_per_img) * K + k] */
    const float* pro_shift;
    const float* pro_gate;
    long long pro_rows_per_img;
} mc_wgrad_rows_args;
int mc_wgrad_rows_supported(int n, int k);
int mc_wgrad_rows_blocks(long long m);
int mc_wgrad_rows_bf16(const mc_wgrad_rows_args* args, void* stream);
/* Round 5 -- BOTH gradients of such a 1x1 convolution e = x W^T from ONE pass over the upstream gradient dY [M, N] (the expand
 * conv of an MBConv block: dY is the widest tensor of the block, 6x its input) [ref: autograd backward of
 * efficientnet_custom.py:104]:   dW[N, K] (fp32) (+)= dY^T x   (args as for mc_wgrad_rows_bf16, no prologue; ws with
 * mc_xbwd_rows_blocks(M) * N * K floats)   and   dX[M, K] (bf16) = dY . wt[K, N]^T (+ r[M, K])   -- what
 * mc_gemm_rows_bf16(dY, wt, residual r) returns.  Shapes: mc_xbwd_rows_supported(N, K) (K <= 64 and the expand geometries
 * of EfficientNet-B2 / -B5: (96,16) (144,24) (240,40) (288,48) (384,64)). */
int mc_xbwd_rows_supported(int n, int k);
int mc_xbwd_rows_blocks(long long m);
int mc_xbwd_rows_bf16(const mc_wgrad_rows_args* args, const mc_bf16* wt, long long ldwt, mc_bf16* dx, long long lddx,
                      const mc_bf16* r, long long ldr, void* stream);

/* ------------------------------------------------------------------------------------------------
 * small helpers */
int mc_cast_f32_bf16(const float* src, mc_bf16* dst, long long n, void* stream);
/* low term of the two-term bf16 split of fp32 weights, dst = bf16(src - float(bf16(src))): with mc_cast_f32_bf16's image as
 * the high term, x . (hi + lo)^T carries the weights to 2^-17 instead of 2^-9 (the opt-in "hi + lo" operand mode of the BERT
 * linears -- the eval-mode parity configuration of DESIGN.md (c); the reference's own path is fp32 / fp16 autocast,
 * trainer.py:271-278) */
int mc_cast_f32_bf16_lo(const float* src, mc_bf16* dst, long long n, void* stream);
int mc_cast_bf16_f32(const mc_bf16* src, float* dst, long long n, void* stream);
int mc_transpose_f32(const float* src, float* dst, int rows, int cols, void* stream); /* dst[c][r] = src[r][c] */
int mc_cast_transpose_f32_bf16(const float* src, mc_bf16* dst, int rows, int cols, void* stream); /* dst[c][r] = bf16(src[r][c]) */
/* stem weight [C0,3,3,3] fp32 (OIHW) -> bf16 [C0,32], k = cin*9 + kh*3 + kw, zero padded 27..31 */
int mc_stem_weight_prep(const float* w, mc_bf16* out, int c0, void* stream);

/* ------------------------------------------------------------------------------------------------
 * stem im2col: fp32 image batch with arbitrary element strides (NCHW or the trainer's permuted NHWC
 * view, [ref: trainer_ddp.py:288-291]) -> bf16 patches [n*oh*ow, 32] for the 3x3 stride-2 stem conv
 * with static padding (pad_l, pad_t) [ref: efficientnet_custom.py:273, efficient_net_custom_utils.py:248-276] */
int mc_stem_im2col(const float* x, long long sn, long long sc, long long sh, long long sw,
                   int n, int h, int w, int pad_l, int pad_t, int oh, int ow, mc_bf16* out, void* stream);
/* out[img][n][k] = bf16(w[n][k] * gate[img][k]) (w dense [n,k] bf16, k % 8 == 0): per-image SE-gated copies of the
 * projection weight [ref: efficientnet_custom.py:119,122  x = sigmoid(se) * x; _project_conv(x)] -- scaling the small
 * operand instead of the activation tensor lets the projection run as a plain batched GEMM */
int mc_gate_weights_bf16(const mc_bf16* w, const float* gate, int n_img, int n, int k, mc_bf16* out, void* stream);

/* input pipeline in front of the stem: raw 8-bit pixels instead of a normalised fp32
 * batch.  [ref: data/datasets/imagetext.py:131-135]  x = float32(u8); x -= x.min(); x /= x.max(); (x - mean) / std
 * per image, every step rounded to float32 in that order -- reproduced bit for bit, then the same patches as above.
 * minmax: unsigned[2][n] workspace = per-image min, then per-image max (filled by mc_image_minmax_u8 from a DENSE
 * per-image block of elems_per_image bytes at stride sn; the [b,1,H,W,3] -> [b,3,H,W] permute of
 * trainer_ddp.py:288-291 is again just the (sc, sh, sw) element strides). */
int mc_image_minmax_u8(const unsigned char* x, long long sn, long long elems_per_image, int n,
                       unsigned int* minmax, void* stream);
int mc_stem_im2col_u8(const unsigned char* x, long long sn, long long sc, long long sh, long long sw,
                      const unsigned int* minmax, float mean, float std, int n, int h, int w, int pad_l,
                      int pad_t, int oh, int ow, mc_bf16* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * depthwise k x k convolution, NHWC bf16, static asymmetric zero padding
 * [ref: efficientnet_custom.py:109 (_depthwise_conv) + :110-111 (_bn1 statistics), :105-107 fused as
 *  prologue: x' = silu(x*pro_scale[c] + pro_shift[c]) applied to in-range inputs only]
 * w_kkc: fp32 [k*k][C] (tap-major).  stat_partials: float[mc_dwconv_stat_rows()][2][C] (optional). */
typedef struct mc_dwconv_args {
    const mc_bf16* x;        /* [n,h,w,c] */
    const mc_bf16* dy;       /* [n,oh,ow,c]  (backward only) */
    void* out;               /* fwd: y bf16 [n,oh,ow,c]; bwd_data: dx bf16 [n,h,w,c]; bwd_weight: dw f32 [k*k][c] (+=) */
    const float* w_kkc;
    int n, h, w, c;
    int k, stride, pad_l, pad_t, oh, ow;
    const float* pro_scale;
    const float* pro_shift;
    float* stat_partials;
    /* BatchNorm(+SiLU)-backward epilogue of the DATA GRADIENT launch: mc_dwconv_fwd on flipped taps for a stride-1 conv,
     * mc_dwconv_bwd_data for a stride-2 conv (stat_partials then has mc_dwconv_bwd_data_stat_rows() rows and epi_x is
     * [n,h,w,c], the shape of the gradient).  epi_x = the conv output e [n,oh,ow,c] that fed
     * silu(bn(e)); the kernel then writes dZ = y * silu'(e*epi_scale + epi_shift) instead of y and stat_partials receives
     * [rows][2][c] = (sum dZ, sum dZ * (e - epi_mean) * epi_invstd), the input of mc_bn_bwd_finalize.  NULL = plain conv. */
    const mc_bf16* epi_x;
    const float* epi_scale;
    const float* epi_shift;
    const float* epi_mean;
    const float* epi_invstd;
    /* round 5 -- fused backward of a stride-1 depthwise conv (mc_dwconv_bwd_fused): weight gradient f32 [k*k][c] (+=) */
    float* dw_out;
    /* rows the caller allocated in stat_partials (what mc_dwconv_stat_rows / mc_dwconv_bwd_*_stat_rows returned for THIS
     * argument block); 0 = not checked.  A launch whose kernel form would write a different number of rows is refused --
     * the form can change between the two calls through mc_dwconv_set_lane_mode / MC_DW_LANE. */
    int stat_rows;
    /* round 6 -- mc_mbconv_xdw_fwd: the expand 1x1 conv in front of the depthwise conv, run inside its staging.  xw = the
     * expand weight [c][cin] (16-bit, row-major, c = the depthwise conv's channels); x is then the BLOCK INPUT [n,h,w,cin].
     * mc_dwconv_bwd_fused with xw != NULL: epi_x is the block input [n,oh,ow,cin] (the launch's output geometry = the conv's
     * input geometry) and the e rows the epilogue / weight gradient need are formed from it in the staging (cin <= 64). */
    const mc_bf16* xw;
    int cin;
} mc_dwconv_args;
int mc_dwconv_stat_rows(const mc_dwconv_args* args);
int mc_dwconv_bwd_data_stat_rows(const mc_dwconv_args* args);
int mc_dwconv_fwd(const mc_dwconv_args* args, void* stream);
int mc_dwconv_bwd_data(const mc_dwconv_args* args, void* stream);
int mc_dwconv_bwd_weight(const mc_dwconv_args* args, void* stream);
/* Two device forms of the forward / stride-1 data gradient exist: the "marching" kernels (a lane owns 2-4 channels, taps
 * in VGPRs) and the "lane = column" kernels of round 4 (a wave owns one channel pair, taps in SGPRs, 16-wave workgroups;
 * conv_lane.hip).  mc_dwconv_fwd picks by shape (5x5 from 50 output columns up); this switch overrides the choice for
 * tests and A/B timing: -1 policy, 0 marching only, 1 lane = column wherever supported.  Returns the previous mode.
 * (Environment variable MC_DW_LANE sets the initial mode.)  Both forms produce bit-identical outputs. */
int mc_dwconv_set_lane_mode(int mode);
int mc_dwconv_lane_supported(const mc_dwconv_args* args);
/* Round 5 -- the whole backward of a STRIDE-1 3x3 depthwise conv whose input was silu(bn0(e)) in ONE launch
 * [ref: efficientnet_custom.py:104-111 backwards: _depthwise_conv, _bn0 + swish]: the data gradient with the BatchNorm0
 * epilogue of mc_dwconv_fwd(epi_x) AND the weight gradient of mc_dwconv_bwd_weight from one staging of (dd, e) -- dd and e
 * are read once, dZ0 is written once (3 passes over the expanded tensor instead of 5).  The argument block is the one of
 * the data-gradient launch (x = dd [n, h, w, c] on the conv's OUTPUT geometry, w_kkc = the taps rotated by 180 degrees,
 * pad = k-1-pad, (oh, ow) = the conv's input geometry, epi_* = e and its BatchNorm statistics, out = dZ0, stat_partials
 * with mc_dwconv_bwd_fused_stat_rows() rows) plus dw_out: f32 [k*k][c], accumulated into (+=) in the conv's OWN tap order.
 * _supported: 3x3, stride 1, c % 8 == 0, epi_x given (the 5x5 form does not fit a wave's registers: see conv_lane.hip).
 * _preferred: the shapes on which this launch measured faster than the two it replaces. */
/* Round 6: with args->xw (+ cin <= 64) the launch does not READ e: epi_x is the block input x and e = x . xw^T is formed per
 * 16-pixel group on the MFMA unit while the rows are staged (the staging of mc_mbconv_xdw_fwd, raw e rounded once to 16 bits
 * like the tensor the expand GEMM stores).  Together with mc_mbconv_xdw_fwd and the folded BatchNorm0 backward
 * (mc_bn_fold_*) the expanded tensor of a stride-1 3x3 block then never exists in HBM, forward or backward. */
int mc_dwconv_bwd_fused_supported(const mc_dwconv_args* args);
int mc_dwconv_bwd_fused_preferred(const mc_dwconv_args* args);
int mc_dwconv_bwd_fused_stat_rows(const mc_dwconv_args* args);
int mc_dwconv_bwd_fused(const mc_dwconv_args* args, void* stream);
/* Round 6 -- expand 1x1 conv -> BatchNorm0 + swish -> depthwise k x k conv of an MBConv block in ONE launch
 * [ref: efficientnet_custom.py:104-111: _expand_conv, _bn0, _swish, _depthwise_conv]: the expanded tensor e = x . xw^T never
 * exists in HBM.  The lane = column depthwise kernel (conv_lane.hip) stages the block input x (cin channels per pixel) instead
 * of e: every wave turns 16-pixel groups of the staged rows into its workgroup's 32 expanded channels with
 * v_mfma_f32_16x16x32 (weight slice resident in LDS as operand fragments), applies z = e*pro_scale + pro_shift and
 * silu(z) to the fp32 accumulators (e is NOT rounded to 16 bits on the way, unlike the two-launch form), zeroes the static
 * padding and writes the 16-bit activations into the LDS tile the stencil reads.  Everything behind the staging -- stencil,
 * output tile, BatchNorm1 statistics partials of the output (stat_partials, mc_mbconv_xdw_stat_rows() rows) -- is the plain
 * forward launch.  Arguments: the mc_dwconv_fwd block with x = block input [n,h,w,cin], xw, cin, pro_scale / pro_shift =
 * BatchNorm0's scale / shift over the c expanded channels (required: training mode takes them from mc_bn_gram_partials +
 * mc_bn_finalize or from a statistics tape, eval mode from the running statistics); epi_* must be NULL.
 * _supported: k in {3,5}, stride in {1,2}, c % 8 == 0, cin % 8 == 0, cin <= 128. */
int mc_mbconv_xdw_supported(const mc_dwconv_args* args);
int mc_mbconv_xdw_stat_rows(const mc_dwconv_args* args);
int mc_mbconv_xdw_fwd(const mc_dwconv_args* args, void* stream);

/* ------------------------------------------------------------------------------------------------
 * training-mode BatchNorm pieces [ref: efficientnet_custom.py:64,74,88,177,205; momentum 0.01, eps 1e-3]
 * finalize: partials float[rows][2][C] (sum, sum of squares) -> mean, invstd, scale = gamma*invstd,
 * shift = beta - mean*scale; running stats updated with the unbiased variance when update_running. */
int mc_bn_finalize(const float* partials, int rows, int c, double count, const float* gamma,
                   const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                   int update_running, float* mean, float* invstd, float* scale, float* shift, void* stream);
/* eval mode: scale/shift from running statistics */
int mc_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, int c, float* scale, float* shift, void* stream);

/* elementwise family on x[n_img * hw, c] (bf16):  z = x*scale[c] + shift[c];  y = act(z)  (act 0 none, 1 SiLU)
 * apply:  out = y * rowscale[img] + res            (rowscale = drop_connect keep/keep_prob, res = skip input)
 *         [ref: efficientnet_custom.py:123-131, efficient_net_custom_utils.py:129-154]
 * pool:   pooled[img, c] = mean_hw y               [ref: efficientnet_custom.py:115 (SE squeeze), :307 (_avg_pooling)]
 *         with args->out != NULL the pass also stores y (bf16) there and pools the stored values: the late-stage project
 *         convolutions then read the activated tensor with a gate-only GEMM prologue instead of re-evaluating
 *         BN+SiLU once per column tile (forward) and once per row tile (weight gradient) */
typedef struct mc_bnact_args {
    const mc_bf16* x;
    long long n_img, hw;
    int c;
    const float* scale;
    const float* shift;
    int act;
    /* apply */
    const float* rowscale;   /* [n_img] or NULL */
    const mc_bf16* res;      /* or NULL */
    mc_bf16* out;
    /* pool */
    float* pooled;           /* [n_img, c] */
    /* backward: upstream gradient of y is  g*mul[img,c] + add[img,c]  (any of the three may be NULL;
     * g NULL = 0, mul NULL = 1, add NULL = 0), then dz = that * act'(z) * rowscale[img] */
    const mc_bf16* g;
    const float* mul;
    const float* add;
    float add_scale;         /* add is multiplied by this (0 is read as 1), e.g. 1/hw for pooled gradients */
    const float* mean;
    const float* invstd;
    float* partials;         /* bwd_reduce: float[mc_bnact_rows()][2][c] = (sum dz, sum dz*xhat) */
    const float* coef;       /* bwd_apply: float[3][c] from mc_bn_bwd_finalize: dx = A*dz + B*x + C */
    mc_bf16* dx;
    float* dgate;            /* se_dgate: [n_img, c] = sum_hw g * act(z) */
    /* pool / se_dgate / se_sums split the rows of an image over mc_bnact_img_splits() workgroups; when that is > 1
     * they need float[splits * n_img * c] (se_sums: * 5) of scratch here and combine it in split order, so results are
     * bit-reproducible run to run (no float atomics) */
    float* split_ws;
} mc_bnact_args;
int mc_bnact_rows(const mc_bnact_args* args);
int mc_bnact_img_splits(const mc_bnact_args* args);
int mc_bnact_apply(const mc_bnact_args* args, void* stream);
int mc_bnact_pool(const mc_bnact_args* args, void* stream);
int mc_bnact_bwd_reduce(const mc_bnact_args* args, void* stream);
int mc_bnact_bwd_apply(const mc_bnact_args* args, void* stream);
int mc_bnact_se_dgate(const mc_bnact_args* args, void* stream);
/* one pass over (x, g): args->dgate = float[5][n_img][c]: {sum g*y, sum g*y', sum g*y'*xhat, sum y', sum y'*xhat}
 * (y = act(z)); [0] is d loss / d gate, [1..4] let the BatchNorm-backward sums be formed without another pass: */
int mc_bnact_se_sums(const mc_bnact_args* args, void* stream);
/* partials[n_img][2][c] (the mc_bn_bwd_finalize input) for upstream gradient g*gate + dpooled*add_scale */
int mc_bn_partials_from_se_sums(const float* sums, const float* gate, const float* dpooled, float add_scale,
                                long long n_img, int c, float* partials, void* stream);
/* dgamma = sum dz*xhat, dbeta = sum dz; coef[0..2][c] for bwd_apply */
int mc_bn_bwd_finalize(const float* partials, int rows, int c, double count, const float* gamma,
                       const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef,
                       void* stream);

/* Training-mode BatchNorm backward folded into the 1x1 convolution in front of it (bnfold.hip; MBConv expand conv +
 * _bn0 [ref: efficientnet_custom.py:104-107]): with e = x We^T (We [n, k] fp32), coef = (A, B, C) from
 * mc_bn_bwd_finalize, dbeta = sum dz and rows = pixels,
 *     dx  = dz (A.We) + x G + cvec        dWe = A.(dz^T x - mean(dz) (x) colsum(x)) + (B.We) Sxx
 * so the consumers of de read dz and x only.  prepare: w1t [k, n] = bf16((A.We)^T), wb [n, k] = bf16(B.We),
 * sxx [k, k] = bf16(x^T x - colsum(x) (x) colsum(x) / rows);  cvec: gt [k, k] = (We^T wb)^T fp32 -> gtb bf16 and
 * cvec [k];  wgrad: dwe = A.(t1 - mean(dz) (x) colsum(x)) + wx  with t1 = dz^T x and wx = wb . sxx */
int mc_bn_fold_prepare(const float* we, const float* coef, const float* xtx, const float* colsum_x, double rows, int n,
                       int k, mc_bf16* w1t, mc_bf16* wb, mc_bf16* sxx, void* stream);
int mc_bn_fold_cvec(const float* gt, const float* we, const float* coef, const float* dbeta, const float* colsum_x,
                    double rows, int n, int k, mc_bf16* gtb, float* cvec, void* stream);
int mc_bn_fold_wgrad(const float* t1, const float* wx, const float* coef, const float* dbeta, const float* colsum_x,
                     double rows, int n, int k, float* dwe, void* stream);
/* Round 6 -- training-mode BatchNorm statistics of e = x . W^T (W [n, k] 16-bit, the MFMA operand image) WITHOUT e
 * [ref: efficientnet_custom.py:104-105: _expand_conv -> _bn0]: column sums and sums of squares of e follow from the k x k Gram
 * matrix of the (6 x narrower) input: sum_e[c] = w_c . colsum(x), and with the centred Gram S = x^T x - colsum (x) colsum / rows
 * the centred second moment is w_c^T S w_c (evaluated in fp64).  Writes TWO partials rows [2][2][n] = (sum, sum of squares) rounded to fp32 + the rounding
 * remainders (mc_bn_finalize adds its rows in fp64) in the layout mc_bn_finalize reads, so finalize / running statistics / the statistics tape are the code every other BatchNorm
 * uses.  xtx [k, k] fp32 (mc_wgrad_rows_bf16(x, x) / the tile TN GEMM), colsum_x [k] fp32 (mc_colsum_bf16). */
int mc_bn_gram_partials(const mc_bf16* w, int ldw, const float* xtx, const float* colsum_x, double rows, int n, int k,
                        float* partials, void* stream);

/* column sums of a bf16 matrix: out[c] = sum_m x[m, c]  (bias gradients).  partials: float[rows][c] */
int mc_colsum_rows(long long m, int c);
int mc_colsum_bf16(const mc_bf16* x, long long m, int c, long long ld, float* partials, float* out,
                   int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * squeeze-excite MLP on pooled features [ref: efficientnet_custom.py:114-119]
 *   r = silu(w1 . pooled + b1);  gate = sigmoid(w2 . r + b2);  w1 [cs, c], w2 [c, cs] fp32 */
int mc_se_fwd(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2,
              int n, int c, int cs, float* gate, float* ws /* float[n*cs] scratch */, void* stream);
/* ws: float[n * (c + 2*cs)] scratch; dw*, db* are WRITTEN (every element by exactly one thread: no zero-fill needed) */
int mc_se_bwd(const float* pooled, const float* gate, const float* dgate, const float* w1, const float* b1,
              const float* w2, const float* b2, int n, int c, int cs, float* dpooled, float* dw1, float* db1,
              float* dw2, float* db2, float* ws, void* stream);

/* dropout on fp32 vectors (pooled image features [ref: efficientnet_custom.py:310-312]); y = x * mask/(1-p),
 * mask = philox(seed, stream_id, index): calling it on the gradient with the same ids is the backward. */
int mc_dropout_f32(const float* x, float* y, long long n, float p, unsigned long long seed,
                   unsigned int stream_id, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BERT pieces [ref: model/modules/text_encoder.py:47-49 -> transformers BertModel] */
/* embeddings: y = dropout(LN(word[ids] + pos[t] + type[tt]))  -> bf16 [b*t, h]; saves mean/rstd */
int mc_bert_embed_fwd(const long long* ids, const long long* tt, const float* word, const float* pos,
                      const float* type, const float* gamma, const float* beta, float eps, int b, int t,
                      int h, float p, unsigned long long seed, unsigned int stream_id, mc_bf16* y,
                      float* mean, float* rstd, void* stream);
int mc_bert_embed_bwd(const mc_bf16* dy, const long long* ids, const long long* tt, const float* word,
                      const float* pos, const float* type, const float* gamma, const float* mean,
                      const float* rstd, int b, int t, int h, float p, unsigned long long seed,
                      unsigned int stream_id, float* dword, float* dpos, float* dtype, float* dgamma,
                      float* dbeta, void* stream);    /* all outputs accumulated (+=) */
/* y = LN(dropout(x) + res) ; rows x h */
int mc_add_ln_fwd(const mc_bf16* x, const mc_bf16* res, const float* gamma, const float* beta, float eps,
                  long long rows, int h, float p, unsigned long long seed, unsigned int stream_id,
                  mc_bf16* y, float* mean, float* rstd, void* stream);
/* dx = dropout-masked grad wrt x, dres = grad wrt res (pre-LN sum grad); dgamma/dbeta accumulated (+=) */
int mc_add_ln_bwd(const mc_bf16* dy, const mc_bf16* x, const mc_bf16* res, const float* gamma,
                  const float* mean, const float* rstd, long long rows, int h, float p,
                  unsigned long long seed, unsigned int stream_id, mc_bf16* dx, mc_bf16* dres,
                  float* dgamma, float* dbeta, void* stream);
/* row softmax of fp32 scores [rows, t] -> probs (bf16) and dropped probs (bf16; may alias probs when p == 0) */
int mc_softmax_fwd(const float* scores, long long rows, int t, float p, unsigned long long seed,
                   unsigned int stream_id, mc_bf16* probs, mc_bf16* probs_drop, void* stream);
/* dscores = probs * (dp - sum(probs*dp)) * alpha with dp = dprobs_drop * mask/(1-p); bf16 out */
int mc_softmax_bwd(const mc_bf16* probs, const float* dprobs_drop, long long rows, int t, float p,
                   unsigned long long seed, unsigned int stream_id, float alpha, mc_bf16* dscores, void* stream);
/* Fused self-attention core (attn.hip), head size 64, 32 <= t <= 256, t % 32 == 0 (mc_attn_supported):
 *   ctx[b*t, nh*64] = dropout(softmax(alpha * Q K^T + mask_bias[b, key])) V   per (sequence, head),
 * qkv = [b*t, 3*nh*64] (Q | K | V column blocks).  The [b, nh, t, t] scores / probabilities are never stored: the
 * forward keeps lse[b*nh*t][2] = (row max, 1 / row sum) and the backward recomputes them.  Dropout masks are the
 * same function of (seed, stream_id, element) as mc_softmax_fwd's.
 * Replaces transformers BertSelfAttention.forward as called from the reference's text encoder
 * [ref: model/modules/text_encoder.py:47-49]. */
int mc_attn_supported(int t, int head_dim);
int mc_attn_fwd(const mc_bf16* qkv, const float* mask_bias, int b, int t, int nh, float alpha, float p,
                unsigned long long seed, unsigned int stream_id, mc_bf16* ctx, float* lse, void* stream);
/* dqkv [b*t, 3*nh*64] (every element written) from dctx [b*t, nh*64] */
int mc_attn_bwd(const mc_bf16* qkv, const float* mask_bias, const mc_bf16* dctx, const float* lse, int b, int t,
                int nh, float alpha, float p, unsigned long long seed, unsigned int stream_id, mc_bf16* dqkv,
                void* stream);
int mc_gelu_fwd(const mc_bf16* x, mc_bf16* y, long long n, void* stream);
int mc_gelu_bwd(const mc_bf16* dy, const mc_bf16* x, mc_bf16* dx, long long n, void* stream);
/* mask bias for attention: out[b, t] = (1 - mask[b,t]) * -3.0e38-ish (finfo.min)  */
int mc_mask_bias(const long long* mask, float* out, long long n, void* stream);
/* eos pooling [ref: model/clip.py:65-68]: out[b, :] = h[b, sum(mask[b]) - 1, :]  (bf16 -> fp32) */
int mc_eos_gather(const mc_bf16* hid, const long long* mask, int b, int t, int h, float* out, void* stream);
int mc_eos_scatter(const float* dout, const long long* mask, int b, int t, int h, mc_bf16* dhid, void* stream);

/* ------------------------------------------------------------------------------------------------
 * fp32 small GEMM with arbitrary strides: C[m,n] = alpha * sum_k A[m*ars + k*acs] * B[k*brs + n*bcs] + beta*C + bias[n]
 * [ref: model/modules/projection.py:23-29 (LinearProjectionHead); loss/breast_clip.py:46-100 (logits)] */
/* alpha_dev (optional): device scalar multiplied into alpha (e.g. the learnable logit scale).
 * ws (optional): float[mc_sgemm_ws_floats(m, n, k)] scratch; lets few-tile / long-k products (projection heads, their
 * weight gradients) spread over the chip by split-K, partials combined in split order (deterministic) */
long long mc_sgemm_ws_floats(int m, int n, int k);
int mc_sgemm(const float* a, long long ars, long long acs, const float* b, long long brs, long long bcs,
             float* c, long long ldc, int m, int n, int k, float alpha, float beta, const float* bias,
             const float* alpha_dev, float* ws, void* stream);
/* y[i] = x[i] * (*scalar_dev) * alpha */
int mc_scale_f32(const float* x, const float* scalar_dev, float alpha, float* y, long long n, void* stream);
/* y = x / ||x||_2 per row (no epsilon) [ref: model/clip.py:90-91]; bwd: dx = (dy - y*(y.dy)) / ||x|| */
int mc_l2norm_fwd(const float* x, int rows, int d, float* y, float* norm, void* stream);
int mc_l2norm_bwd(const float* dy, const float* y, const float* norm, int rows, int d, float* dx, void* stream);
/* mean cross-entropy over rows of logits [rows, n], weight w; the target of row r is labels[r] + label_offset
 * (labels: int64 device array or NULL = r):
 * loss_out[0] += w * mean_r CE_r (label smoothing `smoothing`);  dlogits = w/rows * (softmax - target)  (in place)
 * row_ws: `rows` floats of scratch -- per-row terms, summed in a fixed order (bit-reproducible, no float atomics)
 * [ref: loss/breast_clip.py:43-100 (labels = labels + rank*batch; F.cross_entropy)] */
int mc_ce_fwd_bwd(float* logits, int rows, int n, const long long* labels, int label_offset, float w, float smoothing,
                  float* loss_out, float* row_ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * optimizer step of the hot loop [ref: breastclip/optimizer/__init__.py:28-29 ->
 * torch.optim.AdamW(model.parameters(), lr, weight_decay); trainer_ddp.py:300-303].  Multi-tensor, in place, fp32:
 *   p -= lr*wd*p;  m += (1-b1)(g-m);  v = b2*v + (1-b2) g*g;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * `tensors` is a HOST array of device pointers (one entry per parameter); `step` = t >= 1 (after increment).
 * The LR schedule [ref: breastclip/scheduler/warmup_cosine.py:41-50] is a host scalar: pass the scheduled lr.
 * Hyper-parameters are doubles: the derived scalars (1-b2, bias corrections) are formed in double as torch does. */
typedef struct {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    long long numel;
    mc_bf16* bf16_image;   /* optional: bf16 copy of the updated parameter (same element order), or NULL */
} mc_adamw_tensor;
int mc_adamw_step(const mc_adamw_tensor* tensors, int n_tensors, double lr, double beta1, double beta2, double eps,
                  double weight_decay, long long step, void* stream);
/* Loss-scaled step of the f16 storage build [ref: trainer.py:271-278: the reference's backward runs under
 * torch.cuda.amp.GradScaler; scaler.step() unscales the gradients and skips the update when one of them is inf / nan]:
 * grad[i] *= inv_scale in place for every tensor of the list (only .grad and .numel are read), *found_inf (a device
 * float, zeroed by the caller) is set to 1 if any unscaled value is not finite. */
int mc_grads_unscale(const mc_adamw_tensor* tensors, int n_tensors, float inv_scale, float* found_inf, void* stream);
/* The same step with NO host synchronisation (round 5): the dynamic scale, the non-finite flag and the skip decision stay on
 * the device, like torch's own GradScaler keeps them [ref: trainer_ddp.py:296-303 scaler.scale / step / update].
 *   state = float[8] on the device: {scale, clean steps in a row, found_inf of the step in flight, steps skipped, outcome of
 *           the last step (1 = skipped), -, -, -}
 *   mc_grads_unscale_dev : grad *= 1 / *scale_dev, sets *found_inf (= &state[2]) on a non-finite value
 *   mc_adamw_step_ls     : mc_adamw_step that does nothing when *found_inf != 0 and takes its bias corrections from the number
 *                          of APPLIED steps, step - *skipped (skipped = the optimizer's own device counter)
 *   mc_loss_scale_update : GradScaler.update(): consumes and clears the flag; scale *= backoff on a bad step, *= growth after
 *                          growth_interval clean ones (dynamic != 0); counts the skip in state[3] and in *opt_skipped */
int mc_grads_unscale_dev(const mc_adamw_tensor* tensors, int n_tensors, const float* scale_dev, float* found_inf, void* stream);
int mc_adamw_step_ls(const mc_adamw_tensor* tensors, int n_tensors, double lr, double beta1, double beta2, double eps,
                     double weight_decay, long long step, const float* found_inf, const float* skipped, void* stream);
int mc_loss_scale_update(float* state, float* opt_skipped, float growth_factor, float backoff_factor, int growth_interval,
                         int dynamic, void* stream);

#ifdef __cplusplus
}
#endif
#endif
