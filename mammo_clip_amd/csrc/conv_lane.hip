// Depthwise k x k convolution, "lane = column" form (round 4), NHWC bf16, gfx950.
// [ref: efficientnet_custom.py:109-111  _depthwise_conv (+ static ZeroPad2d, efficient_net_custom_utils.py:248-276)]
//
// The marching kernels of conv.hip give a lane 2-4 CHANNELS of a few columns, so the K*K filter taps of those channels
// live in 36-50 VGPRs per lane and the 5x5 instances run at two waves per SIMD: VALU-issue bound at 0.61 VALU activity
// (profiles/r03_*).  Here all 64 lanes of a wave own the SAME channel pair and a lane is a pixel column:
//   * the K*K taps of the pair are wave-uniform and live in SGPRs (v_pk_fma_f32 takes the 64-bit SGPR pair as an operand:
//     measured 4.6 SIMD cycles per instruction at four waves per SIMD against 6.9 for the VGPR form at two);
//   * a 1024-thread workgroup (16 waves = one per channel pair of a 32-channel tile, ONE workgroup per CU with all the
//     LDS) marches down a strip of 64 * NCOL output columns; each wave keeps its A = ceil(K/S) partial output rows in
//     registers (20 VGPRs for 5x5), so the whole kernel needs < 128 VGPRs and runs four waves per SIMD;
//   * staged pixels are [row][position][17 dwords] in LDS (16 channel pairs + 1 pad): a lane reads ONE dword (its wave's
//     pair) per pixel and consecutive lanes are 17 dwords apart -- conflict-free ds_read_b32; columns are stored
//     de-interleaved by their residue mod NCOL*S so that "lane x reads column x*NCOL*S + i" is a unit-stride access;
//   * outputs go back through an LDS tile in the same format and leave as coalesced 16-byte NHWC stores;
//   * software pipeline with ONE barrier per block of RB input rows: in interval b every thread (1) stores the block
//     b+1 it prefetched into registers to LDS (BN+SiLU prologue applied on the way) and copies the finished output rows
//     of block b-1 from LDS to global memory, (2) issues the global loads of block b+2, (3) computes block b.  All
//     waves run the same mix, so VALU, LDS and memory instructions of different waves overlap without role splitting;
//   * work is cut into per-workgroup ranges of "virtual rows" (image, strip, output row) of equal length, so every CU
//     gets the same number of rows whatever the image count; the two 32-channel tiles that share a 128-byte line run on
//     the same XCD (same L2).
// The accumulator rotation is static: a block is RB = 4 input rows, the rotation period P = A*S rows, and the interval
// body is unrolled over the U = P / gcd(RB, P) phases.
#include "common_hip.h"
#include <cstdlib>
#include "../../include/mammoclip_hip.h"

namespace lane {

constexpr int pmod_c(int a, int m) { return ((a % m) + m) % m; }
constexpr int fdiv_c(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
constexpr int gcd_c(int a, int b) { return b == 0 ? a : gcd_c(b, a % b); }

// G > 1 ("several images per wave", narrow maps): the 64 lanes of a wave are G groups of LPI = 64 / G lanes, group g works on
// image g of a group of G consecutive images -- a 57- or 29-column map then still fills the wave with two output columns per
// lane.  A staged row is G segments (one per image, each with its own K-1 column halo); the work unit is (image group, row).
template <int K, int S, int NCOL, int G = 1> struct Cfg {
    static constexpr int K_ = K, S_ = S, NCOL_ = NCOL, G_ = G;
    static constexpr int WAVES = 16;
    static constexpr int NT = WAVES * 64;
    static constexpr int TCH = 2 * WAVES;               // channels per tile: one channel pair per wave
    static constexpr int VPP = TCH / 8;                 // 16-byte vectors per staged pixel
    static constexpr int PXD = TCH / 2 + 1;             // dwords per LDS pixel (odd: lane = column reads are conflict-free)
    static constexpr int NS = NCOL * S;                 // input-pixel distance between neighbouring lanes
    static constexpr int LPI = 64 / G;                  // lanes per image
    static constexpr int IWMAX = 64 * NCOL * S;         // staged input columns per row: RB * IWMAX * VPP = a multiple of NT vectors
    // staged input columns per segment (32-column segments of a 5-tap kernel get one more: the 29-column maps need 33)
    static constexpr int SEGI = IWMAX / G + ((G > 1 && IWMAX / G == 32 && K == 5) ? 1 : 0);
    static constexpr int TOW = (SEGI - K) / S + 1;      // output columns per segment / strip (the last column slots of a group idle)
    static constexpr int IW_T = (TOW - 1) * S + K;      // staged input columns per segment
    static constexpr int NIN = (NCOL - 1) * S + K;      // input pixels a lane reads per row
    static constexpr int HQ = (IW_T + NS - 1) / NS;     // positions per residue class
    static constexpr int SEGP = HQ * NS;                // positions per segment
    static constexpr int IWP = G * SEGP;
    static constexpr int A = (K + S - 1) / S;           // output rows in flight per lane
    static constexpr int P = A * S;                     // accumulator rotation period (input rows)
    static constexpr int RB = 4;                        // input rows per block
    static constexpr int ORB = RB / S;                  // output rows a block completes
    static constexpr int U = P / gcd_c(RB, P);          // phases of the unrolled interval body
    static constexpr int IN_DW = RB * IWP * PXD;
    static constexpr int TOWP = 64 * NCOL;              // output-tile positions per row (lane x, column i -> i * 64 + x)
    static constexpr int OUT_DW = ORB * TOWP * PXD;
    static constexpr int NV = (RB * G * IW_T * VPP + NT - 1) / NT;
    static constexpr int NVO = (ORB * G * TOW * VPP + NT - 1) / NT;
    static constexpr int LDS_BYTES = (2 * IN_DW + 2 * OUT_DW) * 4 + 2 * TCH * 4 + (128 * 16 + 24) * 4;
    static constexpr int LDS_BYTES_FUSED = LDS_BYTES + 2 * OUT_DW * 4;      // MODE 3: the e rows have a tile of their own
    // MODE 4 (expand conv inside the staging): the staged pixels of a block in 16-pixel MFMA column groups, MTW per wave
    static constexpr int PB = RB * G * IW_T;
    static constexpr int NT16 = (PB + 15) / 16;
    static constexpr int MTW = (NT16 + WAVES - 1) / WAVES;
    // MODE 5 (fused backward, e rows from the block input): the ORB x G x TOW pixels of an e tile in 16-pixel groups
    static constexpr int PBE = ORB * G * TOW;
    static constexpr int MTWE = ((PBE + 15) / 16 + WAVES - 1) / WAVES;
    static_assert(RB % S == 0 && (K - 1) % S == 0, "block / tap geometry");
    static_assert(TOW <= LPI * NCOL, "a group's lanes cover its segment");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// Block descriptors live in LDS (a ring of 128, refilled 64 at a time by the lanes of wave 0 from a closed form of
// block index -> (item, block)); everybody reads the few fields a pipeline stage needs when it needs them -- no long-lived
// scalar state besides the filter taps (50 SGPRs for 5x5), no per-block cursor arithmetic on any wave's critical path.
enum { D_FLAGS = 0, D_INB_LO, D_INB_HI, D_RLO, D_RHI, D_CLO, D_CHI, D_OUTB_LO, D_OUTB_HI, D_ORLO, D_ORHI, D_OCHI, D_NSEG, D_EB_LO, D_EB_HI, D_ERHI,
       D_WORDS = 16 };
constexpr int NDESC = 128;

// EPI (stride 1): the launch is the DATA GRADIENT of a depthwise conv whose input was silu(bn0(e)); the kernel reads e at
// the output position, writes dZ0 = dA0 * silu'(e*scale+shift) and leaves the BatchNorm-backward partials (see conv.hip).
// The e rows travel through the OUTPUT tile: they are staged into the slots the lane later overwrites with dZ0.
// MODE 2 (weight gradient, dw[kh,kw,c] += dy[o,x,c] * x'[o*S+kh, x*S+kw, c]): the same staging with the dy rows of a block in
// the "output" tile (dy row o enters when input row o*S is processed); the registers hold the K*K tap accumulators of the
// wave's channel pair and an A-deep window of unpacked dy rows; one butterfly sum per tap + one atomic per (tap, channel)
// and workgroup at the end.
// MODE 3 (round 5, stride 1): the WHOLE backward of a depthwise conv whose input was a0 = silu(bn0(e)) in one launch -- the data
// gradient with the BatchNorm0 epilogue of MODE 1 AND the weight gradient, from ONE staging of (dd with its halo, e):
//   dA0[q] = sum_t dd[q + t - (K-1-pad)] wflip[t]        dW[K*K-1-t] += dd[q + t - (K-1-pad)] a0[q]
// pair every staged dd value with the SAME (output pixel q, tap t), so the weight gradient is one more packed FMA beside each
// FMA of the data gradient.  a0[q] is evaluated once, when output row q.y meets its first dd row (an A-deep register window of
// a0 and of the packed e values, which the epilogue needs again when the row completes); the e rows therefore enter K-1 rows
// EARLIER than in MODE 1 and get an LDS tile of their own.  dd and e are read once, dZ0 is written once: 3 passes over the
// expanded tensor instead of 5 (MODE 1: dd + e read, dZ0 written; MODE 2: dd + e read).
// MODE 4 (round 6, mc_mbconv_xdw_fwd): the forward launch with the EXPAND 1x1 CONV of the MBConv block inside the staging
// [ref: efficientnet_custom.py:104-111: _expand_conv -> _bn0 -> swish -> _depthwise_conv].  The global tensor is the block input x
// (cin = 32 * KC channels at most, 6 x narrower than the expanded tensor e, which never exists in HBM): the staged pixels of a
// block are cut into 16-pixel groups, MTW per wave; a lane prefetches its 16-byte pieces of x straight into MFMA B-operand
// fragments (pixel = lane & 15, k group = lane >> 4), one block ahead like the plain form's vectors; at store time the wave runs
// D = W_frag . X_frag^T on v_mfma_f32_16x16x32 against the workgroup's 32 x cin weight slice (resident in LDS as A-operand
// fragments), which leaves every lane with 2 x 4 consecutive expanded channels of ONE pixel: BatchNorm0 + swish on the fp32
// accumulators, zero for the static padding, 16-bit pairs into the [row][position][17 dwords] tile -- the layout the stencil
// reads, so everything behind the staging (stencil, output tile, BatchNorm1 statistics) is MODE 0 unchanged.
template <int K, int S, int NCOL, int MODE, int G, int KC = 0>
__global__ __launch_bounds__(1024, 4) void dwconv_lane_fwd_kernel(const mc_dwconv_args p, int strips, int nunits, int cpairs,
                                                                  int ctiles, int ymax, int xmap) {
    using C = Cfg<K, S, NCOL, G>;
    // MODE 5 (round 6): MODE 3 whose e rows -- the expand conv's output at the output positions, read for silu'(bn0(e)), the
    // BatchNorm0 reductions and a0 = silu(bn0(e)) of the weight gradient -- are FORMED from the block input x (epi_x = x
    // [n, oh, ow, cin], xw = the expand weight) by the MFMA staging of MODE 4 instead of being read: with MODE 4 in the forward
    // the expanded tensor of a stride-1 3x3 block never exists in HBM, forward or backward.  The tile holds RAW e (16-bit, rounded
    // once from the fp32 accumulators, like the tensor the expand GEMM stores): everything behind the staging is MODE 3.
    constexpr bool XE = MODE == 5;
    constexpr bool FUSED = MODE == 3 || XE;
    constexpr bool XF = MODE == 4;
    static_assert(!(XF || XE) || (KC >= 1 && KC <= 4), "MODE 4 / 5: cin <= 128");
    constexpr bool EPI = MODE == 1 || FUSED, BWW = MODE == 2, ETILE = EPI || BWW;       // ETILE: a second global tensor staged beside the input
    static_assert(!EPI || S == 1, "the BatchNorm-backward epilogue is provided for stride 1");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* const s_in = smem;                                    // [2][RB][IWP][PXD]
    uint32_t* const s_out = smem + 2 * C::IN_DW;                    // [2][ORB][TOW][PXD]
    uint32_t* const s_e = FUSED ? smem + 2 * C::IN_DW + 2 * C::OUT_DW : s_out;   // e / dy rows (MODE 1 / 2: they travel in the output tile)
    float* const pro_lds = reinterpret_cast<float*>(smem + 2 * C::IN_DW + (FUSED ? 4 : 2) * C::OUT_DW);   // [2][TCH]
    int* const s_desc = reinterpret_cast<int*>(pro_lds + 2 * C::TCH);                        // [NDESC][D_WORDS]
    uint4* const s_w = reinterpret_cast<uint4*>(smem + (XE ? C::LDS_BYTES_FUSED : C::LDS_BYTES) / 4);   // MODE 4 / 5: [KC][2][64] A-operand fragments

    // ---- workgroup -> (channel tile, virtual-row range); the two tiles of a 128-byte line share an XCD (block id % 8)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, rr = bid >> 3;
    // xmap (round 5): units are (y slot, tile pair) with the pair index fastest; an XCD runs a CONTIGUOUS range of them, i.e.
    // neighbouring channel tiles of the same rows meet in one L2.  Pixels whose byte pitch is not a multiple of 128 (c = 240:
    // 480 bytes, c = 1056, 1824: odd multiples of 64) put 64-byte tile pieces of up to three tiles into one 128-byte line; with
    // only the two tiles of a pair sharing an L2 (unit % 8 -> XCD) the c = 240 launches fetched 1.8 x their algorithmic bytes
    // from HBM (profiles/r05_cfg3_pmc_by_kernel.csv): forward 0.911 -> 0.722 ms, weight gradient 0.833 -> 0.620 ms
    const int member = rr & 1;
    const int unit = xmap ? xcd * xmap + (rr >> 1) : (rr >> 1) * 8 + xcd;      // xmap = units per XCD: XCD x runs units [x * xmap, (x + 1) * xmap)
    if (unit >= nunits) return;
    const int cpair = unit % cpairs, yslot = unit / cpairs;
    const int ycp = (nunits - cpair + cpairs - 1) / cpairs;          // y slots of this tile pair
    const int ct = 2 * cpair + member;
    if (ct >= ctiles) return;

    const int tid = threadIdx.x, x = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = ct * C::TCH;
    const int cl = c0 + 2 * wv;                                      // the wave's channel pair
    const bool ch_ok = cl < p.c;
    const bool has_pro = p.pro_scale != nullptr;

    // ---- taps of the wave's channel pair: wave-uniform -> SGPRs
    f32x2_t w[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        float a = 0.f, b = 0.f;
        if (ch_ok) { a = p.w_kkc[(long long)t * p.c + cl]; b = p.w_kkc[(long long)t * p.c + cl + 1]; }
        w[t] = f32x2_t{__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(a))),
                       __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(b)))};
    }
    f32x2_t e_sc = {0.f, 0.f}, e_sh = {0.f, 0.f}, e_mu = {0.f, 0.f};
    if constexpr (EPI) {
        if (ch_ok) {
            e_mu = f32x2_t{__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(p.epi_mean[cl]))),
                           __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(p.epi_mean[cl + 1])))};
            e_sc = f32x2_t{__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(p.epi_scale[cl]))),
                           __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(p.epi_scale[cl + 1])))};
            e_sh = f32x2_t{__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(p.epi_shift[cl]))),
                           __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(p.epi_shift[cl + 1])))};
        }
        if constexpr (K == 5) {
            // Round 5: keep the epilogue's BatchNorm parameters in VECTOR registers where the 50 tap SGPRs fill the scalar file.
            // The compiler knows they are wave-uniform and holds them in SGPRs otherwise -- and then spills TAPS: the 5x5 data-
            // gradient instances reloaded spilled taps with 447-585 v_readlane_b32 per kernel body (179 of the 1128 VALU
            // instructions of an interval); with these six values in VGPRs: 116-136 (ISA counts, scripts/lane_isa_mix.py)
            asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1" : "+v"(e_mu.x), "+v"(e_mu.y));
            asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1" : "+v"(e_sc.x), "+v"(e_sc.y));
            asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1" : "+v"(e_sh.x), "+v"(e_sh.y));
        }
    }
    if (has_pro && tid < 2 * C::TCH) {
        const int ch = tid % C::TCH;
        const float* src = tid < C::TCH ? p.pro_scale : p.pro_shift;
        pro_lds[tid] = (c0 + ch < p.c) ? src[c0 + ch] : 0.f;
    }
    if constexpr (XF || XE) {
        // expand weights of the tile's 32 channels as MFMA A-operand fragments: fragment (kc, f), lane (i = l & 15, kg = l >> 4) =
        // xw[c0 + f*16 + i][kc*32 + kg*8 .. +8]; zero rows / columns beyond c / cin
        for (int idx = tid; idx < KC * 2 * 64; idx += C::NT) {
            const int l = idx & 63, f = (idx >> 6) & 1, kc = idx >> 7;
            const int ch = c0 + f * 16 + (l & 15), k = kc * 32 + (l >> 4) * 8;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (ch < p.c && k < p.cin) v = *reinterpret_cast<const uint4*>(p.xw + (long long)ch * p.cin + k);
            s_w[idx] = v;
        }
    }

    // (plain local copies: lambdas that capture the by-value argument struct by reference make the compiler keep a
    // private-memory image of it)
    const int a_h = p.h, a_w = p.w, a_c = p.c, a_oh = p.oh, a_ow = p.ow, a_pad_t = p.pad_t, a_pad_l = p.pad_l, a_n = p.n;
    const int a_cin = XF ? p.cin : p.c;                                  // channels per pixel of the STAGED global tensor
    const bf16_t* const a_x = p.x;
    const bf16_t* const a_epi_x = p.epi_x;
    const bf16_t* const a_dy = p.dy;
    bf16_t* const a_out = reinterpret_cast<bf16_t*>(p.out);
    // ---- per-thread staging geometry (constant for the whole kernel)
    const int in_row_pitch = a_w * a_cin, out_row_pitch = a_ow * a_c;    // elements (< 2^31: one image row)
    const int in_img_pitch = G > 1 ? a_h * in_row_pitch : 0, out_img_pitch = G > 1 ? a_oh * out_row_pitch : 0;   // (G > 1: small maps)
    const int vv = tid % C::VPP;
    const bool st_ch = c0 + vv * 8 < p.c;
    unsigned meta[C::NV];                                  // row | segment << 4 | col << 8 | LDS dword offset << 16
#pragma unroll
    for (int i = 0; i < C::NV; ++i) {
        const int v = tid + i * C::NT;
        const int pidx = v / C::VPP;
        const int row = pidx / (G * C::IW_T), seg = (pidx % (G * C::IW_T)) / C::IW_T, col = pidx % C::IW_T;
        const int pos = seg * C::SEGP + (col % C::NS) * C::HQ + col / C::NS;
        meta[i] = (unsigned)row | ((unsigned)seg << 4) | ((unsigned)col << 8) | ((unsigned)((row * C::IWP + pos) * C::PXD + vv * 4) << 16);
        if (v >= C::RB * G * C::IW_T * C::VPP || !st_ch) meta[i] = 0xffffu;   // row 15, segment 15, col 255: never valid
    }
    unsigned metao[C::NVO];                                // out row | segment << 4 | col << 8 | LDS dword offset << 16
#pragma unroll
    for (int i = 0; i < C::NVO; ++i) {
        const int v = tid + i * C::NT;
        const int pidx = v / C::VPP;
        const int row = pidx / (G * C::TOW), seg = (pidx % (G * C::TOW)) / C::TOW, col = pidx % C::TOW;
        const int pos = (col % NCOL) * 64 + seg * C::LPI + col / NCOL;
        metao[i] = (unsigned)row | ((unsigned)seg << 4) | ((unsigned)col << 8) | ((unsigned)((row * C::TOWP + pos) * C::PXD + vv * 4) << 16);
        if (v >= C::ORB * G * C::TOW * C::VPP || !st_ch) metao[i] = 0xffffu;
    }

    // element offsets of the thread's vectors inside a block: constants for the whole kernel (round 5: they were re-derived from
    // `meta` with two v_mul_lo_u32 per vector and interval -- 19 quarter-rate multiplies beside 208 packed FMAs)
    // (not in the weight-gradient mode: its 50 tap accumulators leave no registers -- measured +4..6 % there, -5..6 % on the
    // 5x5 data gradient together with the epilogue parameters in VGPRs)
    constexpr bool PRE = MODE != 2;
    int goff[PRE ? C::NV : 1], goffo[PRE ? C::NVO : 1];
    auto in_off = [&](int i) {
        if constexpr (PRE) return goff[i];
        else return (int)((meta[i] >> 4) & 0xfu) * in_img_pitch + (int)(meta[i] & 0xfu) * in_row_pitch + (int)((meta[i] >> 8) & 0xffu) * a_c;
    };
    auto out_off = [&](int i) {
        if constexpr (PRE) return goffo[i];
        else return (int)((metao[i] >> 4) & 0xfu) * out_img_pitch + (int)(metao[i] & 0xfu) * out_row_pitch + (int)((metao[i] >> 8) & 0xffu) * a_c;
    };
    if constexpr (PRE) {
#pragma unroll
        for (int i = 0; i < C::NV; ++i)
            goff[i] = (int)((meta[i] >> 4) & 0xfu) * in_img_pitch + (int)(meta[i] & 0xfu) * in_row_pitch + (int)((meta[i] >> 8) & 0xffu) * a_c;
#pragma unroll
        for (int i = 0; i < C::NVO; ++i)
            goffo[i] = (int)((metao[i] >> 4) & 0xfu) * out_img_pitch + (int)(metao[i] & 0xfu) * out_row_pitch + (int)((metao[i] >> 8) & 0xffu) * a_c;
    }

    // ---- MODE 4: the lane's pixel in each of its wave's MTW 16-pixel groups (B-operand column = lane & 15, k group = lane >> 4)
    const int xkg = x >> 4;
    unsigned xmeta[XF ? C::MTW : 1];                       // row | segment << 4 | col << 8 | LDS dword offset of the pixel << 16
    int xgoff[XF ? C::MTW : 1];
    unsigned xkok = 0;                                     // bit kc: this lane's 8 input channels of K chunk kc exist
    if constexpr (XF) {
#pragma unroll
        for (int j = 0; j < C::MTW; ++j) {
            const int pidx = (wv * C::MTW + j) * 16 + (x & 15);
            const int row = pidx / (G * C::IW_T), seg = (pidx % (G * C::IW_T)) / C::IW_T, col = pidx % C::IW_T;
            const int pos = seg * C::SEGP + (col % C::NS) * C::HQ + col / C::NS;
            xmeta[j] = (unsigned)row | ((unsigned)seg << 4) | ((unsigned)col << 8) | ((unsigned)((row * C::IWP + pos) * C::PXD) << 16);
            if (pidx >= C::PB) xmeta[j] = 0xffffu;
            xgoff[j] = seg * in_img_pitch + row * in_row_pitch + col * a_cin + xkg * 8;
        }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) xkok |= (kc * 32 + xkg * 8 < a_cin ? 1u : 0u) << kc;
    }

    // ---- MODE 5: the lane's pixel in each of its wave's MTWE 16-pixel groups of the e tile (output positions)
    const int e_cin = XE ? p.cin : 0;
    unsigned emeta[XE ? C::MTWE : 1];                      // row | segment << 4 | col << 8 | LDS dword offset of the pixel << 16
    int egoff[XE ? C::MTWE : 1];
    unsigned ekok = 0;
    if constexpr (XE) {
        const int xrow = a_ow * e_cin, ximg = G > 1 ? a_oh * xrow : 0;
#pragma unroll
        for (int j = 0; j < C::MTWE; ++j) {
            const int pidx = (wv * C::MTWE + j) * 16 + (x & 15);
            const int row = pidx / (G * C::TOW), seg = (pidx % (G * C::TOW)) / C::TOW, col = pidx % C::TOW;
            const int pos = (col % NCOL) * 64 + seg * C::LPI + col / NCOL;
            emeta[j] = (unsigned)row | ((unsigned)seg << 4) | ((unsigned)col << 8) | ((unsigned)((row * C::TOWP + pos) * C::PXD) << 16);
            if (pidx >= C::PBE) emeta[j] = 0xffffu;
            egoff[j] = seg * ximg + row * xrow + col * e_cin + xkg * 8;
        }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) ekok |= (kc * 32 + xkg * 8 < e_cin ? 1u : 0u) << kc;
    }

    // ---- block descriptors: closed form of (block index -> item, block of the item), 64 blocks at a time by the lanes
    // of wave 0.  A range is: rest of the first (image, strip) unit, whole units, head of the last unit.
    int* const s_gen = s_desc + NDESC * D_WORDS;                      // range parameters (thread 0 computes them once)
    enum { G_U0 = 0, G_U1, G_O0, G_O1, G_NR0, G_NB0, G_NBF, G_NFULL, G_NBTOT, G_H, G_W, G_C, G_OH, G_OW, G_PT, G_PL, G_STRIPS, G_C0, G_N, G_CIN, G_WORDS = 24 };
    auto nblk_of = [&](int nrows) { return ((nrows - 1) * S + K + C::RB - 1) / C::RB; };
    if (tid == 0) {
        const long long vt = (long long)((a_n + G - 1) / G) * strips * a_oh;      // (image groups, strip, row)
        const long long v0 = vt * yslot / ycp, v1 = vt * (yslot + 1) / ycp;         // [v0, v1) of (image, strip, row)
        const int u0 = (int)(v0 / a_oh), u1 = (int)(v1 / a_oh);
        const int o0 = (int)(v0 - (long long)u0 * a_oh), o1 = (int)(v1 - (long long)u1 * a_oh);
        const int nr0 = u0 == u1 ? o1 - o0 : a_oh - o0;
        const int nb0 = nr0 > 0 ? nblk_of(nr0) : 0, nbf = nblk_of(a_oh);
        const int nfull = u1 - u0 - 1 > 0 ? u1 - u0 - 1 : 0;
        const int nbl = (u1 > u0 && o1 > 0) ? nblk_of(o1) : 0;
        s_gen[G_U0] = u0; s_gen[G_U1] = u1; s_gen[G_O0] = o0; s_gen[G_O1] = o1; s_gen[G_NR0] = nr0; s_gen[G_NB0] = nb0;
        s_gen[G_NBF] = nbf; s_gen[G_NFULL] = nfull; s_gen[G_NBTOT] = nb0 + nfull * nbf + nbl;   // blocks of this workgroup
        s_gen[G_H] = a_h; s_gen[G_W] = a_w; s_gen[G_C] = a_c; s_gen[G_OH] = a_oh; s_gen[G_OW] = a_ow; s_gen[G_PT] = a_pad_t;
        s_gen[G_PL] = a_pad_l; s_gen[G_STRIPS] = strips; s_gen[G_C0] = c0; s_gen[G_N] = a_n; s_gen[G_CIN] = XE ? p.cin : a_cin;
    }
    __syncthreads();
    const int nbtot = __builtin_amdgcn_readfirstlane(s_gen[G_NBTOT]);
    auto gen_desc = [&](int base) {                        // lane x of wave 0: descriptor of block base + x
        const int t = base + x;
        int* d = s_desc + (t & (NDESC - 1)) * D_WORDS;
        if (t >= nbtot) { d[D_FLAGS] = 0; return; }
        const int u0 = s_gen[G_U0], u1 = s_gen[G_U1], o0 = s_gen[G_O0], o1 = s_gen[G_O1], nr0 = s_gen[G_NR0], nb0 = s_gen[G_NB0],
                  nbf = s_gen[G_NBF], nfull = s_gen[G_NFULL];
        const int g_h = s_gen[G_H], g_w = s_gen[G_W], g_c = s_gen[G_C], g_oh = s_gen[G_OH], g_ow = s_gen[G_OW], g_pt = s_gen[G_PT],
                  g_pl = s_gen[G_PL], g_strips = s_gen[G_STRIPS], g_c0 = s_gen[G_C0], g_n = s_gen[G_N];
        int u, oy0, nrows, b;
        if (t < nb0) { u = u0; oy0 = o0; nrows = nr0; b = t; }
        else {
            const int t2 = t - nb0, k = t2 / nbf;
            if (k < nfull) { u = u0 + 1 + k; oy0 = 0; nrows = g_oh; b = t2 - k * nbf; }
            else { u = u1; oy0 = 0; nrows = o1; b = t2 - nfull * nbf; }
        }
        const int grp = u / g_strips, strip = u - grp * g_strips;
        const int img = grp * G;                                        // first image of the group
        const int ox0 = strip * C::TOW;
        const int iy0 = oy0 * S - g_pt + b * C::RB, ix0 = ox0 * S - g_pl;
        // (MODE 4 stages the block input: cin channels per pixel, all of them -- no channel-tile offset)
        const long long inb = XF ? (((long long)img * g_h + iy0) * g_w + ix0) * (long long)s_gen[G_CIN]
                                 : (((long long)img * g_h + iy0) * g_w + ix0) * (long long)g_c + g_c0;
        const int need = (nrows - 1) * S + K - b * C::RB;             // input rows of this block the item still needs
        int rhi = g_h - iy0; if (rhi > C::RB) rhi = C::RB; if (rhi > need) rhi = need;
        int chi = g_w - ix0; if (chi > C::IW_T) chi = C::IW_T;
        const int o_first = BWW ? b * C::RB / S : (b * C::RB - (K - 1)) / S;   // (exact: RB and K-1 are multiples of S)
        const long long outb = (((long long)img * g_oh + oy0 + o_first) * g_ow + ox0) * (long long)g_c + g_c0;
        int orhi = nrows - o_first; if (orhi > C::ORB) orhi = C::ORB;
        int ochi = g_ow - ox0; if (ochi > C::TOW) ochi = C::TOW;
        d[D_FLAGS] = 1 | (b == 0 ? 2 : 0);
        d[D_INB_LO] = (int)(unsigned)inb; d[D_INB_HI] = (int)(inb >> 32);
        d[D_RLO] = iy0 < 0 ? -iy0 : 0; d[D_RHI] = rhi;
        d[D_CLO] = ix0 < 0 ? -ix0 : 0; d[D_CHI] = chi;
        d[D_OUTB_LO] = (int)(unsigned)outb; d[D_OUTB_HI] = (int)(outb >> 32);
        d[D_ORLO] = o_first < 0 ? -o_first : 0; d[D_ORHI] = orhi;
        d[D_OCHI] = ochi;
        d[D_NSEG] = g_n - img < G ? g_n - img : G;                     // images of the group that exist
        if constexpr (FUSED) {                                          // the e rows of the output rows that ENTER in this block
            // (MODE 5: the rows are formed from the block input -- its element offset, cin channels per pixel, all of them)
            const long long eb = XE ? (((long long)img * g_oh + oy0 + b * C::RB) * g_ow + ox0) * (long long)s_gen[G_CIN]
                                    : (((long long)img * g_oh + oy0 + b * C::RB) * g_ow + ox0) * (long long)g_c + g_c0;
            int erhi = nrows - b * C::RB; if (erhi > C::ORB) erhi = C::ORB;
            d[D_EB_LO] = (int)(unsigned)eb; d[D_EB_HI] = (int)(eb >> 32); d[D_ERHI] = erhi;
        }
    };
    if (wv == 0) { gen_desc(0); gen_desc(64); }            // blocks 0 .. 127
    // the output tile doubles as the e tile of the epilogue form: slots no stage ever writes (columns >= TOW) must not
    // hold NaN patterns (0 * NaN in the reductions)
    // (the weight-gradient form sums over ALL lanes: the staged positions no stage ever writes must be zero as well)
    for (int i = tid; i < (FUSED ? 4 : 2) * C::OUT_DW + ((BWW || FUSED) ? 2 * C::IN_DW : 0); i += C::NT) ((BWW || FUSED) ? s_in : s_out)[i] = 0u;
    __syncthreads();                                        // pro_lds, descriptors

    // MODE 4, 3x3 instances with cin <= 64 (registers to spare): the lane's A-operand fragments live in registers for the whole
    // kernel -- the per-group weight reads were half of the staging's LDS instructions (profiles/r06_xdw_sq_counters.csv: LDS busy
    // 0.47 against 0.35 for the plain forward)
    constexpr bool WREG = XF && K == 3 && KC <= 2;
    uint4 wreg[WREG ? KC : 1][2];
    if constexpr (WREG) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) { wreg[kc][0] = s_w[(kc * 2 + 0) * 64 + x]; wreg[kc][1] = s_w[(kc * 2 + 1) * 64 + x]; }
    }
    uint4 vals[XF ? 1 : C::NV];
    unsigned inb = 0;
    uint4 xf[XF ? C::MTW : 1][XF ? KC : 1];               // MODE 4: B-operand fragments of the block in flight
    unsigned xinb = 0;
    uint4 evals[(ETILE && !XE) ? C::NVO : 1];
    uint4 xe[XE ? C::MTWE : 1][XE ? KC : 1];              // MODE 5: B-operand fragments of the e rows in flight
    unsigned einb = 0;
    // global -> registers for block q
    // (always executed -- `live` = the block exists: straight-line loads keep the compiler from treating the prefetch registers as
    // conditionally defined, which costs copies of in-flight registers and a forced wait)
    auto stage_load = [&](int q, bool live) {
        const int* d = s_desc + (q & (NDESC - 1)) * D_WORDS;
        const long long base = ((long long)d[D_INB_HI] << 32) | (unsigned)d[D_INB_LO];
        const int rlo = d[D_RLO], rhi = d[D_RHI], clo = d[D_CLO], chi = d[D_CHI], nseg = d[D_NSEG];
        if constexpr (XF) {
            const bf16_t* org = a_x + base;
            xinb = 0;
#pragma unroll
            for (int j = 0; j < C::MTW; ++j) {
                const int row = (int)(xmeta[j] & 0xfu), seg = (int)((xmeta[j] >> 4) & 0xfu), col = (int)((xmeta[j] >> 8) & 0xffu);
                const bool ok = live && (xmeta[j] & 0xffffu) != 0xffffu && row >= rlo && row < rhi && col >= clo && col < chi && seg < nseg;
#pragma unroll
                for (int kc = 0; kc < KC; ++kc)     // unconditional loads, clamped addresses (zeroed at use where they do not exist)
                    xf[j][kc] = *reinterpret_cast<const uint4*>((ok && ((xkok >> kc) & 1u)) ? org + xgoff[j] + kc * 32 : a_x);
                xinb |= (ok ? 1u : 0u) << j;
            }
        } else {
        const bf16_t* org = a_x + base + vv * 8;
        inb = 0;
#pragma unroll
        for (int i = 0; i < C::NV; ++i) {
            const int row = (int)(meta[i] & 0xfu), seg = (int)((meta[i] >> 4) & 0xfu), col = (int)((meta[i] >> 8) & 0xffu);
#ifdef LANE_NO_LOAD
            const bool ok = live && row >= rlo && row < rhi && col >= clo && col < chi && seg < nseg && a_n < 0;
#else
            const bool ok = live && row >= rlo && row < rhi && col >= clo && col < chi && seg < nseg;
#endif
            vals[i] = *reinterpret_cast<const uint4*>(ok ? org + in_off(i) : a_x);   // unconditional load, clamped address
            inb |= (ok ? 1u : 0u) << i;
        }
        }
        if constexpr (XE) {                                // the x pixels under the e rows that ENTER in block q
            const long long obase = ((long long)d[D_EB_HI] << 32) | (unsigned)d[D_EB_LO];
            const int orhi = d[D_ERHI], ochi = d[D_OCHI];
            const bf16_t* eorg = a_epi_x + obase;
            einb = 0;
#pragma unroll
            for (int j = 0; j < C::MTWE; ++j) {
                const int row = (int)(emeta[j] & 0xfu), seg = (int)((emeta[j] >> 4) & 0xfu), col = (int)((emeta[j] >> 8) & 0xffu);
                const bool ok = live && (emeta[j] & 0xffffu) != 0xffffu && row < orhi && col < ochi && seg < nseg;
#pragma unroll
                for (int kc = 0; kc < KC; ++kc)
                    xe[j][kc] = *reinterpret_cast<const uint4*>((ok && ((ekok >> kc) & 1u)) ? eorg + egoff[j] + kc * 32 : a_epi_x);
                einb |= (ok ? 1u : 0u) << j;
            }
        } else if constexpr (ETILE) {                      // e rows of the output rows block q completes (MODE 3: that enter in it) / its dy rows
            const long long obase = FUSED ? (((long long)d[D_EB_HI] << 32) | (unsigned)d[D_EB_LO]) : (((long long)d[D_OUTB_HI] << 32) | (unsigned)d[D_OUTB_LO]);
            const int orlo = FUSED ? 0 : d[D_ORLO], orhi = FUSED ? d[D_ERHI] : d[D_ORHI], ochi = d[D_OCHI];
            const bf16_t* const esrc = BWW ? a_dy : a_epi_x;
            const bf16_t* eorg = esrc + obase + vv * 8;
            einb = 0;
#pragma unroll
            for (int i = 0; i < C::NVO; ++i) {
                const int row = (int)(metao[i] & 0xfu), seg = (int)((metao[i] >> 4) & 0xfu), col = (int)((metao[i] >> 8) & 0xffu);
                const bool ok = live && row >= orlo && row < orhi && col < ochi && seg < nseg;
                evals[i] = *reinterpret_cast<const uint4*>(ok ? eorg + out_off(i) : esrc);
                einb |= (ok ? 1u : 0u) << i;
            }
        }
    };
    // registers -> LDS in-buffer `buf` (BN+SiLU prologue on real pixels; padding stays zero)
    auto stage_store = [&](int buf) {
        uint32_t* dst = s_in + buf * C::IN_DW;
        if constexpr (XF) {
#pragma unroll
            for (int j = 0; j < C::MTW; ++j) {
                f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    uint4 xv = xf[j][kc];
                    if (!((xkok >> kc) & 1u)) xv = make_uint4(0u, 0u, 0u, 0u);      // (K padding: 0 * garbage must not be NaN)
                    const bf16x8_t b = __builtin_bit_cast(bf16x8_t, xv);
                    if constexpr (WREG) {
                        acc0 = MC_MFMA_16x16x32(__builtin_bit_cast(bf16x8_t, wreg[kc][0]), b, acc0, 0, 0, 0);
                        acc1 = MC_MFMA_16x16x32(__builtin_bit_cast(bf16x8_t, wreg[kc][1]), b, acc1, 0, 0, 0);
                    } else {
                        acc0 = MC_MFMA_16x16x32(__builtin_bit_cast(bf16x8_t, s_w[(kc * 2 + 0) * 64 + x]), b, acc0, 0, 0, 0);
                        acc1 = MC_MFMA_16x16x32(__builtin_bit_cast(bf16x8_t, s_w[(kc * 2 + 1) * 64 + x]), b, acc1, 0, 0, 0);
                    }
                }
                if ((xmeta[j] & 0xffffu) != 0xffffu) {
                    // the lane holds channels f*16 + kg*4 .. +3 (f = 0, 1) of pixel lane & 15: BatchNorm0 + swish, zero padding
                    const uint32_t keep = ((xinb >> j) & 1u) ? 0xffffffffu : 0u;
                    uint32_t* dd = dst + (xmeta[j] >> 16) + xkg * 2;
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        const f32x4_t a = f ? acc1 : acc0;
                        const float4 sc = *reinterpret_cast<const float4*>(&pro_lds[f * 16 + xkg * 4]);
                        const float4 sh = *reinterpret_cast<const float4*>(&pro_lds[C::TCH + f * 16 + xkg * 4]);
                        const f32x2_t z0 = silu2_f(__builtin_elementwise_fma(f32x2_t{a[0], a[1]}, f32x2_t{sc.x, sc.y}, f32x2_t{sh.x, sh.y}));
                        const f32x2_t z1 = silu2_f(__builtin_elementwise_fma(f32x2_t{a[2], a[3]}, f32x2_t{sc.z, sc.w}, f32x2_t{sh.z, sh.w}));
                        dd[f * 8] = pack_bf2(z0.x, z0.y) & keep;
                        dd[f * 8 + 1] = pack_bf2(z1.x, z1.y) & keep;
                    }
                }
            }
            return;
        }
        float ps[8], pt[8];
        if (has_pro) { load8f(&pro_lds[vv * 8], ps); load8f(&pro_lds[C::TCH + vv * 8], pt); }
#pragma unroll
        for (int i = 0; i < C::NV; ++i) {
            if ((meta[i] & 0xffffu) != 0xffffu) {
                const bool real = (inb >> i) & 1u;
                uint4 val = real ? vals[i] : make_uint4(0u, 0u, 0u, 0u);
                if (has_pro && real) {
                    float f[8];
                    unpack8(val, f);
                    bn_silu8(f, ps, pt);
                    val = pack8(f);
                }
                uint32_t* dd = dst + (meta[i] >> 16);
                dd[0] = val.x; dd[1] = val.y; dd[2] = val.z; dd[3] = val.w;
            }
        }
    };
    // finished output rows of block q: LDS out-buffer `buf` -> global
    auto flush_out = [&](int q, int buf) {
        const int* d = s_desc + (q & (NDESC - 1)) * D_WORDS;
        const long long obase = ((long long)d[D_OUTB_HI] << 32) | (unsigned)d[D_OUTB_LO];
        const int orlo = d[D_ORLO], orhi = d[D_ORHI], ochi = d[D_OCHI], nseg = d[D_NSEG];
        const uint32_t* src = s_out + buf * C::OUT_DW;
        bf16_t* org = a_out + obase + vv * 8;
#pragma unroll
        for (int i = 0; i < C::NVO; ++i) {
            const int row = (int)(metao[i] & 0xfu), seg = (int)((metao[i] >> 4) & 0xfu), col = (int)((metao[i] >> 8) & 0xffu);
            if (row >= orlo && row < orhi && col < ochi && seg < nseg) {
                const uint32_t* sp = src + (metao[i] >> 16);
                const uint4 val = make_uint4(sp[0], sp[1], sp[2], sp[3]);
#ifdef LANE_NO_STORE
                if (a_n < 0)
#endif
                *reinterpret_cast<uint4*>(org + out_off(i)) = val;
            }
        }
    };
    auto estore = [&](int buf) {                           // EPI / BWW: the e / dy rows of the block just loaded, into its output slots
        if constexpr (XE) {
            uint32_t* dst = s_e + buf * C::OUT_DW;
#pragma unroll
            for (int j = 0; j < C::MTWE; ++j) {
                f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    uint4 xv = xe[j][kc];
                    if (!((ekok >> kc) & 1u)) xv = make_uint4(0u, 0u, 0u, 0u);
                    const bf16x8_t b = __builtin_bit_cast(bf16x8_t, xv);
                    acc0 = MC_MFMA_16x16x32(__builtin_bit_cast(bf16x8_t, s_w[(kc * 2 + 0) * 64 + x]), b, acc0, 0, 0, 0);
                    acc1 = MC_MFMA_16x16x32(__builtin_bit_cast(bf16x8_t, s_w[(kc * 2 + 1) * 64 + x]), b, acc1, 0, 0, 0);
                }
                if ((emeta[j] & 0xffffu) != 0xffffu) {
                    // raw e of channels f*16 + kg*4 .. +3 (f = 0, 1) of pixel lane & 15; rows / columns that do not exist: zero
                    const uint32_t keep = ((einb >> j) & 1u) ? 0xffffffffu : 0u;
                    uint32_t* dd = dst + (emeta[j] >> 16) + xkg * 2;
                    dd[0] = pack_bf2(acc0[0], acc0[1]) & keep; dd[1] = pack_bf2(acc0[2], acc0[3]) & keep;
                    dd[8] = pack_bf2(acc1[0], acc1[1]) & keep; dd[9] = pack_bf2(acc1[2], acc1[3]) & keep;
                }
            }
        } else if constexpr (ETILE) {
            uint32_t* dst = s_e + buf * C::OUT_DW;
#pragma unroll
            for (int i = 0; i < C::NVO; ++i) {
                if ((metao[i] & 0xffffu) != 0xffffu) {
                    const uint4 val = ((einb >> i) & 1u) ? evals[i] : make_uint4(0u, 0u, 0u, 0u);
                    uint32_t* dd = dst + (metao[i] >> 16);
                    dd[0] = val.x; dd[1] = val.y; dd[2] = val.z; dd[3] = val.w;
                }
            }
        }
    };

    f32x2_t acc[NCOL][C::A];                               // BWW: the dy window g[column][slot]
#pragma unroll
    for (int i = 0; i < NCOL; ++i)
#pragma unroll
        for (int a = 0; a < C::A; ++a) acc[i][a] = f32x2_t{0.f, 0.f};
    f32x2_t dwa[(BWW || FUSED) ? K * K : 1];               // BWW / FUSED: tap accumulators of the wave's channel pair
#pragma unroll
    for (int t = 0; t < ((BWW || FUSED) ? K * K : 1); ++t) dwa[t] = f32x2_t{0.f, 0.f};
    f32x2_t a0w[FUSED ? NCOL : 1][FUSED ? C::A : 1];       // FUSED: a0 = silu(bn0(e)) of the output rows in flight (0 where no pixel exists)
    uint32_t ew[FUSED ? NCOL : 1][FUSED ? C::A : 1];       // ... and their packed e values (the epilogue needs them when the row completes)
#pragma unroll
    for (int i = 0; i < (FUSED ? NCOL : 1); ++i)
#pragma unroll
        for (int a = 0; a < (FUSED ? C::A : 1); ++a) { a0w[i][a] = f32x2_t{0.f, 0.f}; ew[i][a] = 0u; }
    f32x2_t ssum = {0.f, 0.f}, ssq = {0.f, 0.f};

    // interval q: store block q (loaded in interval q-1), flush block q-2, load block q+1, compute block q-1.
    // in-buffer of block q: q & 1; out-buffer of block q: q & 1.  nbtot + 2 intervals.
    stage_load(0, nbtot > 0);
    int q = 0;
    while (true) {
#pragma unroll
        for (int ph = 0; ph < C::U; ++ph) {
            if (wv == 0 && q >= 66 && ((q - 2) & 63) == 0) gen_desc(q + 62);   // refill the slots of blocks q-66 .. q-3
            const bool v_st = q < nbtot, v_ld = q + 1 < nbtot, v_cp = q >= 1 && q <= nbtot, v_fl = q >= 2 && !BWW;
            // (1) registers -> LDS for block q; finished rows of block q-2 -> global (EPI: then the e rows of block q)
            __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): the prefetched registers (see conv.hip)
            if (v_st) stage_store(q & 1);
            if (v_fl) flush_out(q - 2, q & 1);
            if (ETILE && v_st) estore(q & 1);
            // (2) prefetch block q+1  (issuing it BEFORE the output stores was measured slower: 0.272 -> 0.318 ms at c = 384)
            stage_load(q + 1, v_ld);
            // (3) compute block q-1
            if (v_cp) {
                if constexpr (BWW) {
                    const int* d = s_desc + ((q - 1) & (NDESC - 1)) * D_WORDS;
                    if (__builtin_amdgcn_readfirstlane(d[D_FLAGS]) & 2) {       // new item: the dy window starts empty
#pragma unroll
                        for (int i = 0; i < NCOL; ++i)
#pragma unroll
                            for (int a = 0; a < C::A; ++a) acc[i][a] = f32x2_t{0.f, 0.f};
                    }
                    const uint32_t* lin = s_in + ((q - 1) & 1) * C::IN_DW + ((x / C::LPI) * C::SEGP + x % C::LPI) * C::PXD + wv;
                    const uint32_t* lg = s_out + ((q - 1) & 1) * C::OUT_DW + x * C::PXD + wv;
#pragma unroll
                    for (int j = 0; j < C::RB; ++j) {
                        const int jr = (ph * C::RB + j) % C::P;
                        f32x2_t in[C::NIN];
#pragma unroll
                        for (int i = 0; i < C::NIN; ++i) {
                            const uint32_t v = lin[(j * C::IWP + (i % C::NS) * C::HQ + i / C::NS) * C::PXD];
                            in[i] = f32x2_t{bf_lo(v), bf_hi(v)};
                        }
                        if (jr % S == 0) {                                  // dy row (block row j / S) enters the window
                            const int sl = (jr / S) % C::A;
#pragma unroll
                            for (int i = 0; i < NCOL; ++i) {
                                const uint32_t v = lg[((j / S) * C::TOWP + i * 64) * C::PXD];
                                acc[i][sl] = f32x2_t{bf_lo(v), bf_hi(v)};
                            }
                        }
#pragma unroll
                        for (int kh = 0; kh < K; ++kh) {
                            if (pmod_c(jr - kh, S) != 0) continue;
                            const int sl = pmod_c(fdiv_c(jr - kh, S), C::A);
#pragma unroll
                            for (int kw = 0; kw < K; ++kw)
#pragma unroll
                                for (int i = 0; i < NCOL; ++i)
                                    dwa[kh * K + kw] = __builtin_elementwise_fma(acc[i][sl], in[i * S + kw], dwa[kh * K + kw]);
                        }
                    }
                } else if constexpr (FUSED) {
                const int* d = s_desc + ((q - 1) & (NDESC - 1)) * D_WORDS;
                const int orlo = __builtin_amdgcn_readfirstlane(d[D_ORLO]), orhi = __builtin_amdgcn_readfirstlane(d[D_ORHI]);
                const int erhi = __builtin_amdgcn_readfirstlane(d[D_ERHI]);
                const int ochi = d[D_OCHI], nseg = d[D_NSEG];
                if (__builtin_amdgcn_readfirstlane(d[D_FLAGS]) & 2) {       // new item: no output row has entered yet
#pragma unroll
                    for (int i = 0; i < NCOL; ++i)
#pragma unroll
                        for (int a = 0; a < C::A; ++a) { a0w[i][a] = f32x2_t{0.f, 0.f}; ew[i][a] = 0u; }
                }
                const uint32_t* lin = s_in + ((q - 1) & 1) * C::IN_DW + ((x / C::LPI) * C::SEGP + x % C::LPI) * C::PXD + wv;
                const uint32_t* le = s_e + ((q - 1) & 1) * C::OUT_DW + x * C::PXD + wv;
                uint32_t* lout = s_out + ((q - 1) & 1) * C::OUT_DW + x * C::PXD + wv;
                uint32_t cm[NCOL];                                     // all ones where the lane's i-th column exists
#pragma unroll
                for (int i = 0; i < NCOL; ++i) cm[i] = (ch_ok && (x % C::LPI) * NCOL + i < ochi && x / C::LPI < nseg) ? 0xffffffffu : 0u;
#pragma unroll
                for (int j = 0; j < C::RB; ++j) {
                    const int jr = (ph * C::RB + j) % C::P;            // rotation index of this dd row (static)
                    f32x2_t in[C::NIN];
#pragma unroll
                    for (int i = 0; i < C::NIN; ++i) {
                        const uint32_t v = lin[(j * C::IWP + (i % C::NS) * C::HQ + i / C::NS) * C::PXD];
                        in[i] = f32x2_t{bf_lo(v), bf_hi(v)};
                    }
                    {   // the output row whose FIRST dd row this is enters the window: e-tile row j, slot jr % A
                        const int s0 = jr % C::A;
                        const bool e_ok = j < erhi;                    // (wave-uniform; rows beyond the item: a0 = 0)
#pragma unroll
                        for (int i = 0; i < NCOL; ++i) {
                            uint32_t ev = 0u;
                            f32x2_t a = {0.f, 0.f};
                            if (e_ok) {
                                ev = le[(j * C::TOWP + i * 64) * C::PXD] & cm[i];
                                const f32x2_t e2 = {bf_lo(ev), bf_hi(ev)};
                                const f32x2_t sa = silu2_f(__builtin_elementwise_fma(e2, e_sc, e_sh));
                                // pixels that do not exist (columns beyond the map / the strip): a0 = 0, whatever silu(shift) is
                                a = f32x2_t{__uint_as_float(__float_as_uint(sa.x) & cm[i]), __uint_as_float(__float_as_uint(sa.y) & cm[i])};
                            }
                            ew[i][s0] = ev;
                            a0w[i][s0] = a;
                        }
                    }
#pragma unroll
                    for (int kh = 0; kh < K; ++kh) {
                        const int sl = pmod_c(jr - kh, C::A);
#pragma unroll
                        for (int kw = 0; kw < K; ++kw)
#pragma unroll
                            for (int i = 0; i < NCOL; ++i) {
                                if (kh == 0 && kw == 0) acc[i][sl] = in[i + kw] * w[kh * K + kw];
                                else acc[i][sl] = __builtin_elementwise_fma(in[i + kw], w[kh * K + kw], acc[i][sl]);
                                dwa[kh * K + kw] = __builtin_elementwise_fma(in[i + kw], a0w[i][sl], dwa[kh * K + kw]);
                            }
                    }
                    {   // output row complete: out-tile row j
                        const int sl = pmod_c(jr - (K - 1), C::A);
                        const bool o_ok = j >= orlo && j < orhi;      // (wave-uniform)
#pragma unroll
                        for (int i = 0; i < NCOL; ++i) {
                            uint32_t* slot = lout + (j * C::TOWP + i * 64) * C::PXD;
                            const uint32_t ewv = ew[i][sl];
                            const f32x2_t e2 = {bf_lo(ewv), bf_hi(ewv)};
                            const f32x2_t z = __builtin_elementwise_fma(e2, e_sc, e_sh);
                            const f32x2_t dz = acc[i][sl] * silu_grad2_f(z);
                            const uint32_t o2 = pack_bf2(dz.x, dz.y);
                            *slot = o2;
                            if (o_ok) {
                                const uint32_t m = o2 & cm[i];
                                const f32x2_t r = {bf_lo(m), bf_hi(m)};          // reductions of the stored (rounded) dZ0
                                ssum += r;
                                ssq = __builtin_elementwise_fma(r, e2 - e_mu, ssq);
                            }
                        }
                    }
                }
                } else {
                const int* d = s_desc + ((q - 1) & (NDESC - 1)) * D_WORDS;
                const int orlo = __builtin_amdgcn_readfirstlane(d[D_ORLO]), orhi = __builtin_amdgcn_readfirstlane(d[D_ORHI]);
                const int ochi = d[D_OCHI], nseg = d[D_NSEG];
                const uint32_t* lin = s_in + ((q - 1) & 1) * C::IN_DW + ((x / C::LPI) * C::SEGP + x % C::LPI) * C::PXD + wv;
                uint32_t* lout = s_out + ((q - 1) & 1) * C::OUT_DW + x * C::PXD + wv;
                uint32_t cm[NCOL];                                     // all ones where the lane's i-th column exists
#pragma unroll
                for (int i = 0; i < NCOL; ++i) cm[i] = (ch_ok && (x % C::LPI) * NCOL + i < ochi && x / C::LPI < nseg) ? 0xffffffffu : 0u;
#pragma unroll
                for (int j = 0; j < C::RB; ++j) {
                    const int jr = (ph * C::RB + j) % C::P;            // rotation index of this input row (static)
                    f32x2_t in[C::NIN];
#pragma unroll
                    for (int i = 0; i < C::NIN; ++i) {
                        const uint32_t v = lin[(j * C::IWP + (i % C::NS) * C::HQ + i / C::NS) * C::PXD];
                        in[i] = f32x2_t{bf_lo(v), bf_hi(v)};
                    }
#pragma unroll
                    for (int kh = 0; kh < K; ++kh) {
                        if (pmod_c(jr - kh, S) != 0) continue;
                        const int sl = pmod_c(fdiv_c(jr - kh, S), C::A);
#pragma unroll
                        for (int kw = 0; kw < K; ++kw)
#pragma unroll
                            for (int i = 0; i < NCOL; ++i) {
                                // the first contribution of an output row (tap row 0) starts its accumulator
                                if (kh == 0 && kw == 0) acc[i][sl] = in[i * S + kw] * w[kh * K + kw];
                                else acc[i][sl] = __builtin_elementwise_fma(in[i * S + kw], w[kh * K + kw], acc[i][sl]);
                            }
                    }
                    if (pmod_c(jr - (K - 1), S) == 0) {                 // an output row is complete: out-tile row j / S
                        const int sl = pmod_c(fdiv_c(jr - (K - 1), S), C::A);
                        const bool o_ok = j / S >= orlo && j / S < orhi;  // (wave-uniform)
#pragma unroll
                        for (int i = 0; i < NCOL; ++i) {
                            uint32_t* slot = lout + ((j / S) * C::TOWP + i * 64) * C::PXD;
                            if constexpr (EPI) {
                                // (masked: the slots of columns that do not exist are never staged -- they hold what this lane
                                // wrote there a block ago, and 0 * NaN must not reach the reductions)
                                const uint32_t ew = *slot & cm[i];
                                const f32x2_t e2 = {bf_lo(ew), bf_hi(ew)};
                                const f32x2_t z = __builtin_elementwise_fma(e2, e_sc, e_sh);
                                const f32x2_t dz = acc[i][sl] * silu_grad2_f(z);
                                const uint32_t o2 = pack_bf2(dz.x, dz.y);
                                *slot = o2;
                                if (o_ok) {
                                    const uint32_t m = o2 & cm[i];
                                    const f32x2_t r = {bf_lo(m), bf_hi(m)};          // reductions of the stored (rounded) dZ0
                                    ssum += r;
                                    ssq = __builtin_elementwise_fma(r, e2 - e_mu, ssq);   // centred: no cancelling difference at the end
                                }
                            } else {
                                const uint32_t o2 = pack_bf2(acc[i][sl].x, acc[i][sl].y);
                                *slot = o2;
                                if (o_ok) {
                                    const uint32_t m = o2 & cm[i];
                                    const f32x2_t r = {bf_lo(m), bf_hi(m)};          // statistics of the stored (rounded) tensor
                                    ssum += r;
                                    ssq = __builtin_elementwise_fma(r, r, ssq);
                                }
                            }
                        }
                    }
                }
            }
                }
            __syncthreads();
            ++q;
            if (q >= nbtot + 2) goto done;
        }
    }
done:
    if constexpr (BWW || FUSED) {
        // FUSED: the launch ran on the FLIPPED taps (data gradient = forward conv with the filter rotated by 180 degrees); the
        // gradient is returned for the conv's own tap order: accumulator t belongs to tap K*K-1-t
        float* dw = FUSED ? p.dw_out : reinterpret_cast<float*>(p.out);
#pragma unroll
        for (int t = 0; t < K * K; ++t) {
            const float a = wave_sum(dwa[t].x), b = wave_sum(dwa[t].y);
            const int tt = FUSED ? K * K - 1 - t : t;
            if (x == 0 && ch_ok) { atomicAdd(dw + (long long)tt * p.c + cl, a); atomicAdd(dw + (long long)tt * p.c + cl + 1, b); }
        }
        if constexpr (BWW) return;
    }
    if (p.stat_partials) {
        // the 64 lanes of a wave hold the same channel pair: butterfly sums, lane 0 writes
        float s0 = wave_sum(ssum.x), s1 = wave_sum(ssum.y), q0 = wave_sum(ssq.x), q1 = wave_sum(ssq.y);
        if (x == 0 && ch_ok) {
            if constexpr (EPI) {
                // sum dZ * xhat = invstd * sum dZ * (e - mean)
                q0 = p.epi_invstd[cl] * q0;
                q1 = p.epi_invstd[cl + 1] * q1;
            }
            float* r0 = p.stat_partials + ((long long)yslot * 2 + 0) * p.c + cl;
            float* r1 = p.stat_partials + ((long long)yslot * 2 + 1) * p.c + cl;
            r0[0] = s0; r0[1] = s1; r1[0] = q0; r1[1] = q1;
            if (yslot == 0)
                for (int yy = ycp; yy < ymax; ++yy) {                 // tile pairs with one y slot fewer: zero rows
                    float* z0 = p.stat_partials + ((long long)yy * 2) * p.c + cl;
                    z0[0] = 0.f; z0[1] = 0.f; z0[p.c] = 0.f; z0[p.c + 1] = 0.f;
                }
        }
    }
}

struct Plan { int strips, nunits, cpairs, ctiles, ymax, grid, xmap; };
static int xmap_mode() { static const int m = [] { const char* e = getenv("MC_LANE_XMAP"); return e ? atoi(e) : 1; }(); return m; }
template <typename C> Plan plan(const mc_dwconv_args& p) {
    Plan m;
    m.strips = C::G_ > 1 ? 1 : mc_div_up(p.ow, C::TOW);             // (image groups: the map fits one segment)
    m.ctiles = mc_div_up(p.c, C::TCH);
    m.cpairs = (m.ctiles + 1) / 2;
    const long long vt = (long long)((p.n + C::G_ - 1) / C::G_) * m.strips * p.oh;
    long long ycap = vt / 8;                                          // at least 8 output rows per workgroup
    if (ycap < 1) ycap = 1;
    long long nunits = ycap * m.cpairs;
    if (nunits > 128) nunits = 128;                                   // 256 CUs, two tiles per unit
    if (nunits < m.cpairs) nunits = m.cpairs;
    m.nunits = (int)nunits;
    m.ymax = (m.nunits + m.cpairs - 1) / m.cpairs;
    m.grid = 16 * ((m.nunits + 7) / 8);
    // contiguous unit ranges per XCD (see the kernel) where the units divide evenly over the 8 XCDs; else unit % 8 -> XCD
    m.xmap = (xmap_mode() && m.nunits % 8 == 0) ? m.nunits / 8 : 0;
    return m;
}

template <typename C, int MODE, int KC = 0> int launch(const mc_dwconv_args& p, hipStream_t st) {
    static unsigned long long attr_done = 0;
    auto kern = dwconv_lane_fwd_kernel<C::K_, C::S_, C::NCOL_, MODE, C::G_, KC>;
    constexpr int lds = MODE == 3 ? C::LDS_BYTES_FUSED : (MODE == 4 ? C::LDS_BYTES + KC * 2048 : (MODE == 5 ? C::LDS_BYTES_FUSED + KC * 2048 : C::LDS_BYTES));
    static_assert(lds <= 160 * 1024, "LDS budget");
    MC_SET_MAX_LDS(attr_done, kern, lds);
    const Plan m = plan<C>(p);
    hipLaunchKernelGGL(kern, dim3(m.grid), dim3(C::NT), lds, st, p, m.strips, m.nunits, m.cpairs, m.ctiles, m.ymax, m.xmap);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

// configuration by map width and image count: two output columns per lane wherever the registers allow it (not: stride 2,
// the 5x5 weight gradient), and as many images per wave (4 / 2 / 1) as fit the map into a lane group
static int gmax1() { static const int g = [] { const char* e = getenv("MC_DW_LANE_G"); return e ? atoi(e) : 4; }(); return g; }
// G > 1 addresses a staged vector as seg * image pitch + row * row pitch + col * c in 32-bit arithmetic (seg < G): the
// image-group forms are only picked while G whole images stay below 2^31 elements (a tall, narrow, wide-channel
// tensor would otherwise wrap silently); larger maps take the one-image forms, whose offsets are row-sized
static bool g_fits(const mc_dwconv_args& p, int g) {
    const long long in_img = (long long)p.h * p.w * p.c, out_img = (long long)p.oh * p.ow * p.c;
    return g * (in_img > out_img ? in_img : out_img) < (1ll << 31);
}
template <int K, int S, int MODE, typename F> auto pick(const mc_dwconv_args& p, F&& f) {
    // one column per lane: stride 2; the 5x5 weight gradient (50 tap accumulators: no registers for two); the fused backward
    // (its third LDS tile -- the e rows -- does not fit beside two-column input / output tiles)
    constexpr bool one_col = (S == 2) || (MODE == 2 && K == 5) || MODE == 3 || MODE == 5;
    if constexpr (one_col) {
        if (gmax1() >= 2 && p.n >= 2 && p.ow <= Cfg<K, S, 1, 2>::TOW && g_fits(p, 2)) return f(Cfg<K, S, 1, 2>{});
        return f(Cfg<K, S, 1, 1>{});
    } else {
        // several images per wave for maps that do not fill 64 lanes x 2 columns (MC_DW_LANE_G=1 switches it off for A/B)
        static const int gmax = gmax1();
        if (gmax >= 4 && p.n >= 4 && p.ow <= Cfg<K, S, 2, 4>::TOW && g_fits(p, 4)) return f(Cfg<K, S, 2, 4>{});
        if (gmax >= 2 && p.n >= 2 && p.ow <= Cfg<K, S, 2, 2>::TOW && g_fits(p, 2)) return f(Cfg<K, S, 2, 2>{});
        if (gmax == 0 && p.ow <= Cfg<K, S, 2, 1>::TOW) return f(Cfg<K, S, 2, 1>{});
        if (p.ow <= Cfg<K, S, 1, 1>::TOW) return f(Cfg<K, S, 1, 1>{});
        return f(Cfg<K, S, 2, 1>{});
    }
}
template <int MODE, typename F> auto pick_ks(const mc_dwconv_args& p, F&& f) {
    if (p.k == 3) return p.stride == 1 ? pick<3, 1, MODE>(p, f) : pick<3, 2, (MODE == 1 ? 0 : MODE)>(p, f);
    return p.stride == 1 ? pick<5, 1, MODE>(p, f) : pick<5, 2, (MODE == 1 ? 0 : MODE)>(p, f);
}

}  // namespace lane

// Shapes the lane = column form takes (the marching kernels of conv.hip keep everything else): at most 128 tile pairs
// (c <= 8192) and enough columns to fill a 64-lane strip reasonably.
extern "C" int mc_dwconv_lane_supported(const mc_dwconv_args* a) {
    const mc_dwconv_args& p = *a;
    if (p.c % 8 != 0 || p.c > 8192) return 0;
    if (!((p.k == 3 || p.k == 5) && (p.stride == 1 || p.stride == 2))) return 0;
    if (p.epi_x && p.stride != 1) return 0;
    return 1;
}

extern "C" int mc_dwconv_lane_stat_rows(const mc_dwconv_args* a) {
    const mc_dwconv_args& p = *a;
    if (p.epi_x) {
        if (p.k == 3) return lane::pick<3, 1, 1>(p, [&](auto cfg) { return lane::plan<decltype(cfg)>(p).ymax; });
        return lane::pick<5, 1, 1>(p, [&](auto cfg) { return lane::plan<decltype(cfg)>(p).ymax; });
    }
    return lane::pick_ks<0>(p, [&](auto cfg) { return lane::plan<decltype(cfg)>(p).ymax; });
}

extern "C" int mc_dwconv_fwd_lane(const mc_dwconv_args* a, void* stream) {
    const mc_dwconv_args& p = *a;
    MC_CHECK(mc_dwconv_lane_supported(a), "dwconv_fwd_lane: unsupported shape");
    MC_CHECK(p.x && p.w_kkc && p.out, "dwconv_fwd_lane: null x / w / out");
    MC_CHECK(!p.epi_x || (p.epi_scale && p.epi_shift && p.epi_mean && p.epi_invstd && p.stat_partials),
             "dwconv_fwd_lane: the BatchNorm-backward epilogue needs scale/shift/mean/invstd and stat_partials");
    hipStream_t st = (hipStream_t)stream;
    if (p.epi_x) {
        if (p.k == 3) return lane::pick<3, 1, 1>(p, [&](auto cfg) { return lane::launch<decltype(cfg), 1>(p, st); });
        return lane::pick<5, 1, 1>(p, [&](auto cfg) { return lane::launch<decltype(cfg), 1>(p, st); });
    }
    return lane::pick_ks<0>(p, [&](auto cfg) { return lane::launch<decltype(cfg), 0>(p, st); });
}

// ---- MODE 3: fused backward (stride 1).  Argument block = the data-gradient launch of MODE 1 (x = dd on the conv's OUTPUT
// geometry as (h, w), flipped taps, pads K-1-pad, (oh, ow) = the conv's input geometry, epi_* = e and its BatchNorm
// statistics) + dw_out [k*k][c] f32, accumulated into (+=, the conv's own tap order).
// 3x3 only: the 5x5 form needs 50 tap accumulators + the two windows + the data gradient's partial rows in one wave --
// ~177 live VGPRs against the 128 of a 16-wave workgroup (compiled and measured in round 5: 177-224 spilled registers); the
// 5x5 backward stays two launches (MODE 1 + MODE 2).
extern "C" int mc_dwconv_bwd_fused_lane_supported(const mc_dwconv_args* a) {
    const mc_dwconv_args& p = *a;
    // (round 6: xw != NULL -- epi_x is the block input x [n, oh, ow, cin] and the e rows are formed from it: cin <= 64)
    if (p.xw && !(p.cin > 0 && p.cin % 8 == 0 && p.cin <= 64)) return 0;
    return mc_dwconv_lane_supported(a) && p.k == 3 && p.stride == 1 && p.epi_x != nullptr;
}

extern "C" int mc_dwconv_bwd_fused_lane_stat_rows(const mc_dwconv_args* a) {
    const mc_dwconv_args& p = *a;
    return lane::pick<3, 1, 3>(p, [&](auto cfg) { return lane::plan<decltype(cfg)>(p).ymax; });
}

extern "C" int mc_dwconv_bwd_fused_lane(const mc_dwconv_args* a, void* stream) {
    const mc_dwconv_args& p = *a;
    MC_CHECK(mc_dwconv_bwd_fused_lane_supported(a), "dwconv_bwd_fused: stride-1 conv with the BatchNorm epilogue operands (epi_x) only");
    MC_CHECK(p.x && p.w_kkc && p.out && p.dw_out, "dwconv_bwd_fused: null dd / w / out / dw_out");
    MC_CHECK(p.epi_scale && p.epi_shift && p.epi_mean && p.epi_invstd && p.stat_partials,
             "dwconv_bwd_fused: the BatchNorm-backward epilogue needs scale/shift/mean/invstd and stat_partials");
    MC_CHECK(!p.pro_scale, "dwconv_bwd_fused: dd carries no prologue");
    MC_CHECK(!(p.stat_rows > 0) || p.stat_rows == mc_dwconv_bwd_fused_lane_stat_rows(a), "dwconv_bwd_fused: stat_partials was sized for another configuration");
    hipStream_t st = (hipStream_t)stream;
    if (p.xw) {
        MC_CHECK(mc_aligned16(p.epi_x) && mc_aligned16(p.xw), "dwconv_bwd_fused (e from the block input): x / xw must be 16-byte aligned");
        return lane::pick<3, 1, 5>(p, [&](auto cfg) {
            using Cf = decltype(cfg);
            if (p.cin <= 32) return lane::launch<Cf, 5, 1>(p, st);
            return lane::launch<Cf, 5, 2>(p, st);
        });
    }
    return lane::pick<3, 1, 3>(p, [&](auto cfg) { return lane::launch<decltype(cfg), 3>(p, st); });
}

// public names (include/mammoclip_hip.h).  _preferred: where the fused launch measured faster than the two launches it
// replaces (scripts/dw_form_ab.py, 32 images; MC_DW_FUSED=0 never / =1 wherever supported, for A/B runs)
extern "C" int mc_dwconv_bwd_fused_supported(const mc_dwconv_args* a) { return mc_dwconv_bwd_fused_lane_supported(a); }
extern "C" int mc_dwconv_bwd_fused_stat_rows(const mc_dwconv_args* a) { return mc_dwconv_bwd_fused_lane_stat_rows(a); }
extern "C" int mc_dwconv_bwd_fused(const mc_dwconv_args* a, void* stream) { return mc_dwconv_bwd_fused_lane(a, stream); }
extern "C" int mc_dwconv_bwd_fused_preferred(const mc_dwconv_args* a) {
    static const int mode = [] { const char* e = getenv("MC_DW_FUSED"); return e ? atoi(e) : -1; }();
    if (mode == 0 || !mc_dwconv_bwd_fused_lane_supported(a)) return 0;
    if (mode == 1) return 1;
    const mc_dwconv_args& p = *a;
    // measured (32 images, two launches under the form policy -> fused, ms): c = 768 at 95x57 0.347 -> 0.286, c = 1824 at 48x29
    // 0.223 -> 0.183, c = 3072 0.460 -> 0.343, c = 240 at 380x228 1.494 -> 1.345 (the whole-pixel marching pair); narrow pixels
    // lose (c = 24 at 760x456: 0.711 -> 1.265 -- 48-byte pixels in 32-channel tiles)
    // (B2 at 912 x 912, c = 144 at 228 x 228: forcing the fused form everywhere it is supported 153.7 -> 152.9 ms per cfg2 step)
    return (p.ow >= 50 && p.c >= 128) || (p.ow <= 30 && p.n >= 2 && p.c >= 128);
}

extern "C" int mc_dwconv_bwd_weight_lane(const mc_dwconv_args* a, void* stream) {
    const mc_dwconv_args& p = *a;
    MC_CHECK(mc_dwconv_lane_supported(a) && !p.epi_x, "dwconv_bwd_weight_lane: unsupported shape");
    MC_CHECK(p.x && p.dy && p.out, "dwconv_bwd_weight_lane: null x / dy / out");
    hipStream_t st = (hipStream_t)stream;
    return lane::pick_ks<2>(p, [&](auto cfg) { return lane::launch<decltype(cfg), 2>(p, st); });
}

// ---- MODE 4: expand 1x1 conv + BatchNorm0 + swish inside the staging of the depthwise forward (include/mammoclip_hip.h)
extern "C" int mc_mbconv_xdw_supported(const mc_dwconv_args* a) {
    const mc_dwconv_args& p = *a;
    if (!mc_dwconv_lane_supported(a) || p.epi_x) return 0;
    return p.cin > 0 && p.cin % 8 == 0 && p.cin <= 128;
}

extern "C" int mc_mbconv_xdw_stat_rows(const mc_dwconv_args* a) {
    const mc_dwconv_args& p = *a;
    return lane::pick_ks<4>(p, [&](auto cfg) { return lane::plan<decltype(cfg)>(p).ymax; });
}

extern "C" int mc_mbconv_xdw_fwd(const mc_dwconv_args* a, void* stream) {
    const mc_dwconv_args& p = *a;
    MC_CHECK(mc_mbconv_xdw_supported(a), "mbconv_xdw_fwd: unsupported shape (k in {3,5}, stride in {1,2}, c % 8 == 0, cin % 8 == 0, cin <= 128, no epilogue)");
    MC_CHECK(p.x && p.xw && p.w_kkc && p.out, "mbconv_xdw_fwd: null x / xw / w / out");
    MC_CHECK(p.pro_scale && p.pro_shift, "mbconv_xdw_fwd: BatchNorm0 scale / shift (pro_scale / pro_shift) are required");
    MC_CHECK(mc_aligned16(p.x) && mc_aligned16(p.xw) && mc_aligned16(p.out), "mbconv_xdw_fwd: operands must be 16-byte aligned");
    MC_CHECK(!(p.stat_partials && p.stat_rows > 0) || p.stat_rows == mc_mbconv_xdw_stat_rows(a),
             "mbconv_xdw_fwd: stat_partials was sized for another configuration");
    hipStream_t st = (hipStream_t)stream;
    return lane::pick_ks<4>(p, [&](auto cfg) {
        using Cf = decltype(cfg);
        if (p.cin <= 32) return lane::launch<Cf, 4, 1>(p, st);
        if (p.cin <= 64) return lane::launch<Cf, 4, 2>(p, st);
        return lane::launch<Cf, 4, 4>(p, st);
    });
}
