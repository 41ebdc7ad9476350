// Depthwise k x k convolution (k in {3,5}, stride in {1,2}) for NHWC bf16, gfx950.
// [ref: efficientnet_custom.py:109-111  _depthwise_conv (+ static ZeroPad2d, efficient_net_custom_utils.py:248-276)]
//
// Forward (and stride-1 data gradient, = forward with flipped taps) and weight gradient are "marching" kernels:
// column strips walked top to bottom, input rows staged through LDS with coalesced 16-byte loads (the BatchNorm+SiLU
// of the producing expand conv is applied once per element while staging, zero padding is inserted after the
// activation), partial output rows / tap accumulators held in registers.  Workgroups are persistent over their work
// items so the per-channel sum / sum-of-squares for the following training-mode BatchNorm leave the forward kernel
// as a small [workgroups][2][C] partial buffer (deterministic, no atomics).  The stride-2 data gradient is a gather.
#include "common_hip.h"
#include <atomic>
#include <type_traits>
#include <cstdlib>
#include "../../include/mammoclip_hip.h"

namespace {

// generic gather form of the data gradient (any stride): dx[ih,iw] = sum_{kh,kw} dy[(ih+pt-kh)/S, (iw+pl-kw)/S] * w[kh,kw]
template <int K>
__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(const mc_dwconv_args p) {
    const int cvn = p.c / 8;
    const long long total = (long long)p.n * p.h * p.w * cvn;
    const int S = p.stride;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int cv = (int)(i % cvn);
        long long pix = i / cvn;
        int ix = (int)(pix % p.w);
        int iy = (int)((pix / p.w) % p.h);
        long long img = pix / ((long long)p.w * p.h);
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            int ty = iy + p.pad_t - kh;
            if (ty < 0 || (ty % S) != 0) continue;
            int oy = ty / S;
            if (oy >= p.oh) continue;
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                int tx = ix + p.pad_l - kw;
                if (tx < 0 || (tx % S) != 0) continue;
                int ox = tx / S;
                if (ox >= p.ow) continue;
                float g[8], wv[8];
                unpack8(*reinterpret_cast<const uint4*>(p.dy + ((img * p.oh + oy) * (long long)p.ow + ox) * p.c + cv * 8), g);
                load8f(p.w_kkc + (long long)(kh * K + kw) * p.c + cv * 8, wv);
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = fmaf(g[q], wv[q], acc[q]);
            }
        }
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + pix * p.c + cv * 8) = pack8(acc);
    }
}

// ------------------------------------------------------------------------------------------------------------
// "Marching" forward kernel.  A lane owns CPL (2 or 4) channels of NCOL adjacent output columns and walks down
// the rows of its segment: every staged input row is read from LDS once per lane (NIN reads of 4*CPL/2 bytes; the
// 64 lanes of a wave cover PXW pixels x LP lanes = contiguous LDS), unpacked once, and scattered with packed fp32
// FMAs (v_pk_fma_f32, two channels per instruction) into the A = ceil(K/S) output rows it touches; the filter taps
// of the lane's channels live in registers.  Rows are staged in blocks of RB = NR*A*S input rows so the
// accumulator rotation is static.  Compared with a 2-D halo tile this removes the vertical halo (each input row is
// loaded and BN+SiLU-activated once per column strip), all LDS weight reads, and most LDS activation reads and
// bf16->fp32 unpacks.  All per-thread staging geometry (global offset, LDS offset, row, column) is constant for
// the whole kernel and kept in registers, so a staged block costs a handful of instructions per 16-byte vector.
constexpr int pmod_c(int a, int m) { return ((a % m) + m) % m; }
constexpr int fdiv_c(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

template <int K, int S, int CPL, int LP, int NCOL_ = 2> struct MarchCfg {
    static constexpr int H2 = CPL / 2;                     // packed fp32 pairs per lane
    static constexpr int PXW = 64 / LP;                    // pixels per wave (LP = lanes per pixel)
    static constexpr int NCOL = NCOL_;                     // adjacent output columns per lane
    static constexpr int NS = NCOL * S;                    // input-pixel distance between neighbouring lanes' bases
    static constexpr int TOW = 4 * PXW * NCOL;             // output columns per strip (4 waves)
    static constexpr int A = (K + S - 1) / S;              // output rows in flight per lane
    static constexpr int P = A * S;                        // input rows per accumulator rotation period
    static constexpr int IW_T = (TOW - 1) * S + K;         // staged input columns
    static constexpr int NIN = (NCOL - 1) * S + K;         // input pixels a lane reads per row
    static constexpr int TCH = LP * CPL;                   // channels per tile
    static constexpr int PXB = TCH * 2;                    // bytes per staged pixel
    static constexpr int VPP = PXB / 16;                   // 16-byte vectors per staged pixel
    // LDS conflict avoidance.  8-byte reads, 128-byte pixels: the two pixels of a half-wave are NS pixels apart and
    // would share banks, so pixel p is stored at position p ^ ((p >> log2(NS)) & 1) (dense, no padding).  4-byte
    // reads, 64-byte pixels: pad the pixel stride so the 4 pixels of a wave land in 4 different 16-bank groups.
    static constexpr bool SWZ = (CPL == 4 && LP == 16);
    static constexpr int SWZ_BIT = (NS == 2) ? 1 : 2;
    static constexpr int PSB = (CPL == 2 && LP == 16) ? (NS == 2 ? 96 : 80) : (CPL == 2 && LP == 32) ? (NS == 4 ? 160 : 144) : PXB;
    static constexpr int IWP = SWZ ? (IW_T + (2 << SWZ_BIT) - 1) / (2 << SWZ_BIT) * (2 << SWZ_BIT) : IW_T;
    static constexpr int NR_ = (24576 + P * IW_T * PXB / 2) / (P * IW_T * PXB);
    static constexpr int NR = (NR_ < 1 || LP == 60) ? 1 : NR_;   // periods per staged block (~24 KB of loads in flight; the whole-pixel form: one period, register budget)
    static constexpr int RB = P * NR;                      // input rows per staged block
    static constexpr int TS = 256 - 256 % VPP;             // staging threads (vector index within a pixel fixed per thread)
    static constexpr int NV = (RB * IW_T * VPP + TS - 1) / TS;
    static constexpr int BUF_BYTES = RB * IWP * PSB;
    static constexpr int OCC = (K == 3 && S == 1) ? 3 : 2;           // workgroups per CU the register budget is set for
};

#ifdef MARCH_PROF
__device__ unsigned long long g_march_prof[8];     // developer phase profile (scripts/dwbench.hip)
#define MPROF(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[i] += t_ - tprof; tprof = t_; } while (0)
#else
#define MPROF(i)
#endif

// EPI (stride 1 only): the launch is the DATA GRADIENT of a depthwise conv whose input was silu(bn0(e)) -- the output of
// this "forward on flipped taps" is dA0, and the kernel finishes the step that follows it in the MBConv backward
// [ref: efficientnet_custom.py:104-107 backwards]: it reads e at the output position, writes dZ0 = dA0 * silu'(e*scale+shift)
// instead of dA0 and leaves the BatchNorm-backward reduction (sum dZ0, sum dZ0 * xhat0) in stat_partials -- the separate
// reduce pass over (e, dA0) disappears and the BN0 apply pass becomes a plain linear combination.
template <int K, int S, int CPL, int LP, int NCOL, bool EPI = false>
__global__ __launch_bounds__(256, (K == 3 && S == 1 && !EPI ? 3 : 2)) void dwconv_march_fwd_kernel(const mc_dwconv_args p, int strips, int segs,
                                                                                int seg_rows, int ctiles, int gy) {
    static_assert(!EPI || S == 1, "the BatchNorm-backward epilogue is provided for stride 1");
    using C = MarchCfg<K, S, CPL, LP, NCOL>;
    typedef typename std::conditional<CPL == 4, uint2, uint32_t>::type ldsv_t;
    // EPI: the e rows of the output rows a staged block completes travel through LDS beside the input block (same
    // prefetch, position-major like the dy tile of the weight-gradient kernel: a wave's reads are contiguous)
    constexpr int XQ = C::TOW / NCOL;                      // lane column groups per strip
    constexpr int E_BYTES = EPI ? C::RB * C::TOW * C::PXB : 0;
    constexpr int NVE = EPI ? (C::RB * C::TOW * C::VPP + C::TS - 1) / C::TS : 1;
    constexpr int MAIN_BYTES = C::BUF_BYTES > 8192 ? C::BUF_BYTES : 8192;
    __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN_BYTES + E_BYTES];
    __shared__ __attribute__((aligned(16))) float pro_lds[2][C::TCH];      // prologue scale / shift of the tile's channels
    unsigned char* const esm = smem + MAIN_BYTES;
    // XCD-aware decomposition: the channel tiles of one (image, strip, segment) share 128-byte lines, so they are
    // given consecutive slots on the SAME XCD (workgroup id % 8) and meet in that XCD's L2.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ct = slot % ctiles, y = (slot / ctiles) * 8 + xcd;
    if (y >= gy) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane % LP, px = lane / LP;
    const bool lane_ok = px < C::PXW;
    const int c0 = ct * C::TCH;
    const int cl = c0 + lq * CPL;                          // the lane's first channel
    const bool ch_ok = lane_ok && cl < p.c;
    const int xl0 = (wave * C::PXW + (lane_ok ? px : 0)) * C::NCOL;     // tile-local first output column
    const bool has_pro = p.pro_scale != nullptr;
    // LDS read bases (see SWZ above): reads whose table offset is even use lb_p, odd ones lb_m
    const int qb = C::SWZ ? ((xl0 * S) >> C::SWZ_BIT) & 1 : 0;
    const int lbase = xl0 * S * C::PSB + lq * (CPL * 2);
    const int lb_p = lbase + qb * C::PSB, lb_m = lbase - qb * C::PSB;

    f32x2_t w[K * K][C::H2];
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
        for (int h = 0; h < C::H2; ++h) {
            w[t][h] = f32x2_t{0.f, 0.f};
            if (ch_ok) w[t][h] = *reinterpret_cast<const f32x2_t*>(p.w_kkc + (long long)t * p.c + cl + 2 * h);
        }

    // EPI: BatchNorm parameters of the lane's channels (z = e*scale + shift, xhat = (e - mean) * invstd)
    // (the loop accumulates sum dZ and the CENTRED sum dZ * (e - mean) -- round 4: the uncentred form ended in a cancelling
    // difference whose round-off depended on the summation order; invstd enters when the partials are written)
    f32x2_t e_sc[C::H2], e_sh[C::H2], e_mu[C::H2];
    if constexpr (EPI) {
#pragma unroll
        for (int h = 0; h < C::H2; ++h) {
            e_sc[h] = e_sh[h] = e_mu[h] = f32x2_t{0.f, 0.f};
            if (ch_ok) {
                e_sc[h] = *reinterpret_cast<const f32x2_t*>(p.epi_scale + cl + 2 * h);
                e_sh[h] = *reinterpret_cast<const f32x2_t*>(p.epi_shift + cl + 2 * h);
                e_mu[h] = *reinterpret_cast<const f32x2_t*>(p.epi_mean + cl + 2 * h);
            }
        }
    }

    // ---- per-thread staging geometry, constant for the whole kernel
    const int vv = tid % C::VPP;
    const int cs = c0 + vv * 8;
    const bool st_ok = tid < C::TS && cs < p.c;
    const long long in_row_pitch = (long long)p.w * p.c;   // elements
    unsigned meta[C::NV];                                  // row | col << 8 | LDS byte offset << 16 (in 16-byte units)
#pragma unroll
    for (int i = 0; i < C::NV; ++i) {
        const int v = tid + i * C::TS;
        const int row = v / (C::IW_T * C::VPP), col = (v / C::VPP) % C::IW_T;
        const int pos = C::SWZ ? (col ^ ((col >> C::SWZ_BIT) & 1)) : col;
        meta[i] = (unsigned)row | ((unsigned)col << 8) | ((unsigned)(((row * C::IWP + pos) * C::PSB + vv * 16) >> 4) << 16);
        if (v >= C::RB * C::IW_T * C::VPP) meta[i] = 0xffu;    // row 255: never valid
    }
    unsigned metae[NVE];                                   // EPI: the same for the e block (RB rows x TOW columns)
    if constexpr (EPI) {
#pragma unroll
        for (int i = 0; i < NVE; ++i) {
            const int v = tid + i * C::TS;
            const int row = v / (C::TOW * C::VPP), col = (v / C::VPP) % C::TOW;
            metae[i] = (unsigned)row | ((unsigned)col << 8) |
                       ((unsigned)((((row * NCOL + col % NCOL) * XQ + col / NCOL) * C::PXB + vv * 16) >> 4) << 16);
            if (v >= C::RB * C::TOW * C::VPP) metae[i] = 0xffu;
        }
    }
    const int ebase = (wave * C::PXW + (lane_ok ? px : 0)) * C::PXB + lq * (CPL * 2);
    if (has_pro) {
        for (int t2 = tid; t2 < 2 * C::TCH; t2 += 256) {       // (the whole-pixel tiles have more than 128 channels)
            const int ch = t2 % C::TCH;
            const float* src = t2 < C::TCH ? p.pro_scale : p.pro_shift;
            pro_lds[t2 / C::TCH][ch] = (c0 + ch < p.c) ? src[c0 + ch] : 0.f;
        }
    }

    f32x2_t ssum[C::H2], ssq[C::H2];
#pragma unroll
    for (int h = 0; h < C::H2; ++h) { ssum[h] = f32x2_t{0.f, 0.f}; ssq[h] = f32x2_t{0.f, 0.f}; }
    f32x2_t acc[C::NCOL][C::A][C::H2];
    bf16_t* optr = nullptr;                                // lane's pixel in the next output row to complete
    int o_next = 0;
    unsigned col_ok = 0;                                   // bit i: the lane's i-th output column exists (and its channels do)
    const long long row_pitch = (long long)p.ow * p.c;
    const int pix_pitch = p.c;

    const int nitems = p.n * strips * segs;
    auto item_geom = [&](int it, int& img, int& ox0, int& oy0, int& nrows, int& nblk) {
        const int strip = it % strips;
        const int seg = (it / strips) % segs;
        img = it / (strips * segs);
        ox0 = strip * C::TOW;
        oy0 = seg * seg_rows;
        nrows = p.oh - oy0 < seg_rows ? p.oh - oy0 : seg_rows;
        nblk = ((nrows - 1) * S + K + C::RB - 1) / C::RB;
    };

    uint4 vals[C::NV];
    uint4 evals[NVE];
    unsigned einb = 0;
    int nrows_l = 0;                                       // rows of the item being LOADED (for the e block's row mask)
    unsigned inb = 0;                                      // vectors of the staged block that hold real pixels
    unsigned colmask = 0;                                  // per item: vectors whose column lies inside the image
    // global -> registers for block b of item (img, ox0, oy0)
    auto stage_load = [&](int img, int ox0, int oy0, int b, bool new_item) {
        const int iy0 = oy0 * S - p.pad_t + b * C::RB, ix0 = ox0 * S - p.pad_l;
        if (new_item) {
            colmask = 0;
#pragma unroll
            for (int i = 0; i < C::NV; ++i) {
                const int ix = ix0 + (int)((meta[i] >> 8) & 0xffu);
                if (st_ok && ix >= 0 && ix < p.w) colmask |= 1u << i;
            }
        }
        const bf16_t* org = p.x + ((long long)img * p.h + iy0) * in_row_pitch + (long long)ix0 * p.c + c0;
        inb = 0;
#pragma unroll
        for (int i = 0; i < C::NV; ++i) {
            // unconditional load (padding / out-of-tile vectors read a harmless valid address and are zeroed when
            // stored): straight-line code keeps all loads of the block in flight together
            const int iy = iy0 + (int)(meta[i] & 0xffu);
            const bool ok = ((colmask >> i) & 1u) && iy >= 0 && iy < p.h;
            const unsigned goff = (meta[i] & 0xffu) * (unsigned)in_row_pitch + ((meta[i] >> 8) & 0xffu) * (unsigned)p.c + vv * 8;
            const bf16_t* a = ok ? org + goff : p.x;
            vals[i] = *reinterpret_cast<const uint4*>(a);
            inb |= (ok ? 1u : 0u) << i;
        }
        if constexpr (EPI) {
            // e rows of the output rows block b completes: o = b*RB - (K-1) + row  (only rows / columns that exist)
            const int o0 = b * C::RB - (K - 1);
            const bf16_t* eorg = p.epi_x + ((long long)img * p.oh + oy0 + o0) * ((long long)p.ow * p.c) + (long long)ox0 * p.c + c0;
            einb = 0;
#pragma unroll
            for (int i = 0; i < NVE; ++i) {
                const int o = o0 + (int)(metae[i] & 0xffu);
                const bool ok = st_ok && (metae[i] & 0xffu) != 0xffu && o >= 0 && o < nrows_l && ox0 + (int)((metae[i] >> 8) & 0xffu) < p.ow;
                const long long goff = (long long)(metae[i] & 0xffu) * ((long long)p.ow * p.c) + (long long)((metae[i] >> 8) & 0xffu) * p.c + vv * 8;
                evals[i] = *reinterpret_cast<const uint4*>(ok ? eorg + goff : p.epi_x);
                einb |= (ok ? 1u : 0u) << i;
            }
        }
    };
    // registers -> LDS, with the fused BN+SiLU prologue on real pixels (zero padding stays zero)
    auto stage_store = [&]() {
        // explicit vmcnt(0): the uses below are conditional, and without an unconditional wait the compiler's waitcnt
        // bookkeeping treats the prefetched registers as possibly-pending on later paths and drains the NEXT block's
        // loads (vmcnt(0)) right before the compute loop, which serialises load latency with compute
        __builtin_amdgcn_s_waitcnt(0x0F70);
        float ps[8], pt[8];
        if (has_pro) { load8f(&pro_lds[0][vv * 8], ps); load8f(&pro_lds[1][vv * 8], pt); }
#pragma unroll
        for (int i = 0; i < C::NV; ++i) {
            if (tid < C::TS && (meta[i] & 0xffu) != 0xffu) {
                const bool real = (inb >> i) & 1u;
                uint4 val = real ? vals[i] : make_uint4(0u, 0u, 0u, 0u);
                if (has_pro && real) {
                    float f[8];
                    unpack8(val, f);
                    bn_silu8(f, ps, pt);
                    val = pack8(f);
                }
                *reinterpret_cast<uint4*>(smem + ((meta[i] >> 16) << 4)) = val;
            }
        }
        if constexpr (EPI) {
#pragma unroll
            for (int i = 0; i < NVE; ++i)
                if (tid < C::TS && (metae[i] & 0xffu) != 0xffu)
                    *reinterpret_cast<uint4*>(esm + ((metae[i] >> 16) << 4)) = ((einb >> i) & 1u) ? evals[i] : make_uint4(0u, 0u, 0u, 0u);
        }
    };

    int it = y, img = 0;
    int ox0 = 0, oy0 = 0, nrows = 0, nblk = 0, b = 0;
    if (it >= nitems) return;                 // (never: gy <= nitems)
    item_geom(it, img, ox0, oy0, nrows, nblk);
    nrows_l = nrows;
    stage_load(img, ox0, oy0, 0, true);
#ifdef MARCH_PROF
    unsigned long long pacc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tprof = __builtin_amdgcn_s_memtime();
#endif
    while (true) {
        __syncthreads();                       // previous block fully consumed
        MPROF(0);
        stage_store();
        MPROF(1);
        __syncthreads();
        MPROF(2);
        // prefetch the block after (it, b) while this one computes
        int it2 = it, img2 = img;
        int ox2 = ox0, oy2 = oy0, nrows2 = nrows, nblk2 = nblk, b2 = b + 1;
        if (b2 >= nblk) {
            it2 = it + gy; b2 = 0;
            if (it2 < nitems) item_geom(it2, img2, ox2, oy2, nrows2, nblk2);
        }
        const bool more = it2 < nitems;
        nrows_l = nrows2;
        if (more) stage_load(img2, ox2, oy2, b2, b2 == 0);
        MPROF(3);

        if (b == 0) {                          // new item: clear the accumulators, aim the running output row pointer
#pragma unroll
            for (int i = 0; i < C::NCOL; ++i)
#pragma unroll
                for (int a = 0; a < C::A; ++a)
#pragma unroll
                    for (int h = 0; h < C::H2; ++h) acc[i][a][h] = f32x2_t{0.f, 0.f};
            o_next = fdiv_c(-(K - 1), S);
            optr = reinterpret_cast<bf16_t*>(p.out) + (((long long)img * p.oh + oy0 + o_next) * p.ow + ox0 + xl0) * p.c + cl;

            col_ok = 0;
#pragma unroll
            for (int i = 0; i < C::NCOL; ++i) col_ok |= (ch_ok && ox0 + xl0 + i < p.ow ? 1u : 0u) << i;
        }
#pragma unroll 1
        for (int sb = 0; sb < C::NR; ++sb) {
            const unsigned char* lp_p = smem + lb_p + sb * (C::P * C::IWP * C::PSB);
            const unsigned char* lp_m = smem + lb_m + sb * (C::P * C::IWP * C::PSB);
#pragma unroll
            for (int j = 0; j < C::P; ++j) {
                // EPI: the e values of the output row this step completes (row sb*P + j of the staged e block)
                ldsv_t ev[C::NCOL];
                if constexpr (EPI) {
#pragma unroll
                    for (int i = 0; i < C::NCOL; ++i)
                        ev[i] = *reinterpret_cast<const ldsv_t*>(esm + ebase + ((sb * C::P + j) * NCOL + i) * XQ * C::PXB);
                }
                f32x2_t in[C::NIN][C::H2];
#pragma unroll
                for (int i = 0; i < C::NIN; ++i) {
                    const int t0 = C::SWZ ? (i ^ ((i >> C::SWZ_BIT) & 1)) : i;          // pixel position for qb == 0
                    const unsigned char* a = ((t0 & 1) ? lp_m : lp_p) + (j * C::IWP + t0) * C::PSB;
                    const ldsv_t v = *reinterpret_cast<const ldsv_t*>(a);
                    if constexpr (CPL == 4) {
                        in[i][0] = f32x2_t{bf_lo(v.x), bf_hi(v.x)};
                        in[i][1] = f32x2_t{bf_lo(v.y), bf_hi(v.y)};
                    } else {
                        in[i][0] = f32x2_t{bf_lo(v), bf_hi(v)};
                    }
                }
#pragma unroll
                for (int kh = 0; kh < K; ++kh) {
                    if (pmod_c(j - kh, S) != 0) continue;
                    const int sl = pmod_c(fdiv_c(j - kh, S), C::A);
#pragma unroll
                    for (int kw = 0; kw < K; ++kw)
#pragma unroll
                        for (int i = 0; i < C::NCOL; ++i)
#pragma unroll
                            for (int h = 0; h < C::H2; ++h)
                                acc[i][sl][h] = __builtin_elementwise_fma(w[kh * K + kw][h], in[i * S + kw][h], acc[i][sl][h]);
                }
                if (pmod_c(j - (K - 1), S) == 0) {             // output row o_next is complete
                    const int sl = pmod_c(fdiv_c(j - (K - 1), S), C::A);
                    if (o_next >= 0 && o_next < nrows) {
#pragma unroll
                        for (int i = 0; i < C::NCOL; ++i) {
                            if ((col_ok >> i) & 1u) {
                                uint32_t o2[C::H2];
#pragma unroll
                                for (int h = 0; h < C::H2; ++h) {
                                    if constexpr (EPI) {
                                        uint32_t ew;
                                        if constexpr (CPL == 4) ew = h == 0 ? ev[i].x : ev[i].y; else ew = ev[i];
                                        const f32x2_t e2 = {bf_lo(ew), bf_hi(ew)};
                                        const f32x2_t z = __builtin_elementwise_fma(e2, e_sc[h], e_sh[h]);
                                        const f32x2_t dz = acc[i][sl][h] * silu_grad2_f(z);
                                        o2[h] = pack_bf2(dz.x, dz.y);
                                        const f32x2_t r = {bf_lo(o2[h]), bf_hi(o2[h])};  // reductions of the stored (rounded) dZ0
                                        ssum[h] += r;
                                        ssq[h] = __builtin_elementwise_fma(r, e2 - e_mu[h], ssq[h]);
                                    } else {
                                        o2[h] = pack_bf2(acc[i][sl][h].x, acc[i][sl][h].y);
                                        const f32x2_t r = {bf_lo(o2[h]), bf_hi(o2[h])};   // statistics of the stored (rounded) tensor
                                        ssum[h] += r;
                                        ssq[h] = __builtin_elementwise_fma(r, r, ssq[h]);
                                    }
                                }
                                if constexpr (CPL == 4) *reinterpret_cast<uint2*>(optr + i * pix_pitch) = make_uint2(o2[0], o2[1]);
                                else *reinterpret_cast<uint32_t*>(optr + i * pix_pitch) = o2[0];
                            }
                        }
                    }
                    ++o_next;
                    optr += row_pitch;
#pragma unroll
                    for (int i = 0; i < C::NCOL; ++i)
#pragma unroll
                        for (int h = 0; h < C::H2; ++h) acc[i][sl][h] = f32x2_t{0.f, 0.f};
                }
            }
        }
        MPROF(4);
#ifdef MARCH_PROF
        pacc[5] += 1;
#endif
        if (!more) break;
        it = it2; img = img2; ox0 = ox2; oy0 = oy2; nrows = nrows2; nblk = nblk2; b = b2;
    }

#ifdef MARCH_PROF
    if (tid == 0)
        for (int i = 0; i < 6; ++i) atomicAdd(&g_march_prof[i], pacc[i]);
#endif
    if (p.stat_partials) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);          // [256 threads][2][CPL]
#pragma unroll
        for (int h = 0; h < C::H2; ++h) {
            red[tid * 2 * CPL + 2 * h] = ssum[h].x; red[tid * 2 * CPL + 2 * h + 1] = ssum[h].y;
            red[tid * 2 * CPL + CPL + 2 * h] = ssq[h].x; red[tid * 2 * CPL + CPL + 2 * h + 1] = ssq[h].y;
        }
        __syncthreads();
        for (int t2 = tid; t2 < 2 * C::TCH; t2 += 256) {
            const int ch = t2 % C::TCH, which = t2 / C::TCH;          // 0 = sum, 1 = sum of squares
            float s = 0.f;
            for (int wv = 0; wv < 4; ++wv)
                for (int q = 0; q < C::PXW; ++q) s += red[(wv * 64 + q * LP + ch / CPL) * 2 * CPL + which * CPL + ch % CPL];
            if constexpr (EPI) {
                // which == 1 holds sum dZ * (e - mean): sum dZ * xhat = invstd * that
                if (which == 1 && c0 + ch < p.c) s = p.epi_invstd[c0 + ch] * s;
            }
            if (c0 + ch < p.c) p.stat_partials[((long long)y * 2 + which) * p.c + c0 + ch] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Marching weight-gradient kernel: same strip / staging structure as the forward kernel, but the registers hold the
// K*K tap accumulators of the lane's channels instead of the taps, and the rotating A-deep register window holds the
// unpacked dy rows (each dy row is read from LDS and unpacked once, then meets the K input rows it overlaps):
//   dw[kh,kw,c] += dy[o, x, c] * x'[o*S + kh, x*S + kw, c]        (x' = BN+SiLU prologue applied while staging)
// dy is staged through LDS beside x (position-major so a wave's reads are contiguous).  Accumulators run across all
// items of the persistent workgroup; one LDS reduction + one atomic per (tap, channel) per workgroup at the end.
template <int K, int S, int CPL, int LP, int NCOL>
__global__ __launch_bounds__(256, 2) void dwconv_march_bww_kernel(const mc_dwconv_args p, int strips, int segs, int seg_rows,
                                                                  int ctiles, int gy) {
    using C = MarchCfg<K, S, CPL, LP, NCOL>;
    typedef typename std::conditional<CPL == 4, uint2, uint32_t>::type ldsv_t;
    constexpr int ORB = C::A * C::NR;                      // dy rows per staged block
    constexpr int XQ = C::TOW / NCOL;                      // lane column groups per strip
    constexpr int G_BYTES = ORB * C::TOW * C::PXB;
    constexpr int NVG = (ORB * C::TOW * C::VPP + C::TS - 1) / C::TS;
    constexpr int RED_BYTES = 256 * CPL * 4;
    constexpr int SM_BYTES = (C::BUF_BYTES + G_BYTES > RED_BYTES) ? C::BUF_BYTES + G_BYTES : RED_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SM_BYTES];
    __shared__ __attribute__((aligned(16))) float pro_lds[2][C::TCH];
    unsigned char* gsm = smem + C::BUF_BYTES;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ct = slot % ctiles, y = (slot / ctiles) * 8 + xcd;
    if (y >= gy) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane % LP, px = lane / LP;
    const bool lane_ok = px < C::PXW;
    const int c0 = ct * C::TCH;
    const int cl = c0 + lq * CPL;
    const int xq = wave * C::PXW + (lane_ok ? px : 0);
    const int xl0 = xq * NCOL;
    const bool has_pro = p.pro_scale != nullptr;
    const int qb = C::SWZ ? ((xl0 * S) >> C::SWZ_BIT) & 1 : 0;
    const int lbase = xl0 * S * C::PSB + lq * (CPL * 2);
    const int lb_p = lbase + qb * C::PSB, lb_m = lbase - qb * C::PSB;
    const int gbase = xq * C::PXB + lq * (CPL * 2);        // dy tile: [row][col % NCOL][col / NCOL][channels]

    const int vv = tid % C::VPP;
    const int cs = c0 + vv * 8;
    const bool st_ok = tid < C::TS && cs < p.c;
    const long long in_row_pitch = (long long)p.w * p.c, g_row_pitch = (long long)p.ow * p.c;
    unsigned meta[C::NV], metag[NVG];                      // row | col << 8 | (LDS byte offset / 16) << 16
#pragma unroll
    for (int i = 0; i < C::NV; ++i) {
        const int v = tid + i * C::TS;
        const int row = v / (C::IW_T * C::VPP), col = (v / C::VPP) % C::IW_T;
        const int pos = C::SWZ ? (col ^ ((col >> C::SWZ_BIT) & 1)) : col;
        meta[i] = (unsigned)row | ((unsigned)col << 8) | ((unsigned)(((row * C::IWP + pos) * C::PSB + vv * 16) >> 4) << 16);
        if (v >= C::RB * C::IW_T * C::VPP) meta[i] = 0xffu;
    }
#pragma unroll
    for (int i = 0; i < NVG; ++i) {
        const int v = tid + i * C::TS;
        const int row = v / (C::TOW * C::VPP), col = (v / C::VPP) % C::TOW;
        metag[i] = (unsigned)row | ((unsigned)col << 8) |
                   ((unsigned)((((row * NCOL + col % NCOL) * XQ + col / NCOL) * C::PXB + vv * 16) >> 4) << 16);
        if (v >= ORB * C::TOW * C::VPP) metag[i] = 0xffu;
    }
    if (has_pro) {
        for (int t2 = tid; t2 < 2 * C::TCH; t2 += 256) {       // (the whole-pixel tiles have more than 128 channels)
            const int ch = t2 % C::TCH;
            const float* src = t2 < C::TCH ? p.pro_scale : p.pro_shift;
            pro_lds[t2 / C::TCH][ch] = (c0 + ch < p.c) ? src[c0 + ch] : 0.f;
        }
    }

    f32x2_t acc[K * K][C::H2];
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
        for (int h = 0; h < C::H2; ++h) acc[t][h] = f32x2_t{0.f, 0.f};
    f32x2_t g[C::A][NCOL][C::H2];

    const int nitems = p.n * strips * segs;
    auto item_geom = [&](int it, int& img, int& ox0, int& oy0, int& nrows, int& nblk) {
        const int strip = it % strips;
        const int seg = (it / strips) % segs;
        img = it / (strips * segs);
        ox0 = strip * C::TOW;
        oy0 = seg * seg_rows;
        nrows = p.oh - oy0 < seg_rows ? p.oh - oy0 : seg_rows;
        nblk = ((nrows - 1) * S + K + C::RB - 1) / C::RB;
    };

    uint4 vals[C::NV], gvals[NVG];
    unsigned inb = 0, colmask = 0, ginb = 0, gcolmask = 0;
    auto stage_load = [&](int img, int ox0, int oy0, int nrows, int b, bool new_item) {
        const int iy0 = oy0 * S - p.pad_t + b * C::RB, ix0 = ox0 * S - p.pad_l;
        if (new_item) {
            colmask = 0; gcolmask = 0;
#pragma unroll
            for (int i = 0; i < C::NV; ++i) {
                const int ix = ix0 + (int)((meta[i] >> 8) & 0xffu);
                if (st_ok && ix >= 0 && ix < p.w) colmask |= 1u << i;
            }
#pragma unroll
            for (int i = 0; i < NVG; ++i)
                if (st_ok && ox0 + (int)((metag[i] >> 8) & 0xffu) < p.ow) gcolmask |= 1u << i;
        }
        const bf16_t* org = p.x + ((long long)img * p.h + iy0) * in_row_pitch + (long long)ix0 * p.c + c0;
        inb = 0;
#pragma unroll
        for (int i = 0; i < C::NV; ++i) {
            const int iy = iy0 + (int)(meta[i] & 0xffu);
            const bool ok = ((colmask >> i) & 1u) && iy >= 0 && iy < p.h;
            const unsigned goff = (meta[i] & 0xffu) * (unsigned)in_row_pitch + ((meta[i] >> 8) & 0xffu) * (unsigned)p.c + vv * 8;
            const bf16_t* a = ok ? org + goff : p.x;
            vals[i] = *reinterpret_cast<const uint4*>(a);
            inb |= (ok ? 1u : 0u) << i;
        }
        const int o0 = b * ORB;
        const bf16_t* gorg = p.dy + ((long long)img * p.oh + oy0 + o0) * g_row_pitch + (long long)ox0 * p.c + c0;
        ginb = 0;
#pragma unroll
        for (int i = 0; i < NVG; ++i) {
            const int o = o0 + (int)(metag[i] & 0xffu);
            const bool ok = ((gcolmask >> i) & 1u) && o < nrows;
            const unsigned goff = (metag[i] & 0xffu) * (unsigned)g_row_pitch + ((metag[i] >> 8) & 0xffu) * (unsigned)p.c + vv * 8;
            const bf16_t* a = ok ? gorg + goff : p.dy;
            gvals[i] = *reinterpret_cast<const uint4*>(a);
            ginb |= (ok ? 1u : 0u) << i;
        }
    };
    auto stage_store = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0), see the forward kernel
        float ps[8], pt[8];
        if (has_pro) { load8f(&pro_lds[0][vv * 8], ps); load8f(&pro_lds[1][vv * 8], pt); }
#pragma unroll
        for (int i = 0; i < C::NV; ++i) {
            if (tid < C::TS && (meta[i] & 0xffu) != 0xffu) {
                const bool real = (inb >> i) & 1u;
                uint4 val = real ? vals[i] : make_uint4(0u, 0u, 0u, 0u);
                if (has_pro && real) {
                    float f[8];
                    unpack8(val, f);
                    bn_silu8(f, ps, pt);
                    val = pack8(f);
                }
                *reinterpret_cast<uint4*>(smem + ((meta[i] >> 16) << 4)) = val;
            }
        }
#pragma unroll
        for (int i = 0; i < NVG; ++i) {
            if (tid < C::TS && (metag[i] & 0xffu) != 0xffu) {
                const uint4 val = ((ginb >> i) & 1u) ? gvals[i] : make_uint4(0u, 0u, 0u, 0u);
                *reinterpret_cast<uint4*>(gsm + ((metag[i] >> 16) << 4)) = val;
            }
        }
    };

    int it = y, img = 0;
    int ox0 = 0, oy0 = 0, nrows = 0, nblk = 0, b = 0;
    if (it >= nitems) return;
    item_geom(it, img, ox0, oy0, nrows, nblk);
    stage_load(img, ox0, oy0, nrows, 0, true);
    while (true) {
        __syncthreads();
        stage_store();
        __syncthreads();
        int it2 = it, img2 = img;
        int ox2 = ox0, oy2 = oy0, nrows2 = nrows, nblk2 = nblk, b2 = b + 1;
        if (b2 >= nblk) {
            it2 = it + gy; b2 = 0;
            if (it2 < nitems) item_geom(it2, img2, ox2, oy2, nrows2, nblk2);
        }
        const bool more = it2 < nitems;
        if (more) stage_load(img2, ox2, oy2, nrows2, b2, b2 == 0);

        if (b == 0) {                          // new item: the dy window starts empty
#pragma unroll
            for (int a = 0; a < C::A; ++a)
#pragma unroll
                for (int i = 0; i < NCOL; ++i)
#pragma unroll
                    for (int h = 0; h < C::H2; ++h) g[a][i][h] = f32x2_t{0.f, 0.f};
        }
#pragma unroll
        for (int sb = 0; sb < C::NR; ++sb) {
            const unsigned char* lp_p = smem + lb_p + sb * (C::P * C::IWP * C::PSB);
            const unsigned char* lp_m = smem + lb_m + sb * (C::P * C::IWP * C::PSB);
#pragma unroll
            for (int j = 0; j < C::P; ++j) {
                f32x2_t in[C::NIN][C::H2];
#pragma unroll
                for (int i = 0; i < C::NIN; ++i) {
                    const int t0 = C::SWZ ? (i ^ ((i >> C::SWZ_BIT) & 1)) : i;
                    const unsigned char* a = ((t0 & 1) ? lp_m : lp_p) + (j * C::IWP + t0) * C::PSB;
                    const ldsv_t v = *reinterpret_cast<const ldsv_t*>(a);
                    if constexpr (CPL == 4) {
                        in[i][0] = f32x2_t{bf_lo(v.x), bf_hi(v.x)};
                        in[i][1] = f32x2_t{bf_lo(v.y), bf_hi(v.y)};
                    } else {
                        in[i][0] = f32x2_t{bf_lo(v), bf_hi(v)};
                    }
                }
                if (j % S == 0) {                              // dy row (sb*A + j/S) of this block enters the window
                    const int sl = pmod_c(j / S, C::A);
                    const int d = sb * C::A + j / S;
#pragma unroll
                    for (int i = 0; i < NCOL; ++i) {
                        const ldsv_t v = *reinterpret_cast<const ldsv_t*>(gsm + gbase + (d * NCOL + i) * XQ * C::PXB);
                        if constexpr (CPL == 4) {
                            g[sl][i][0] = f32x2_t{bf_lo(v.x), bf_hi(v.x)};
                            g[sl][i][1] = f32x2_t{bf_lo(v.y), bf_hi(v.y)};
                        } else {
                            g[sl][i][0] = f32x2_t{bf_lo(v), bf_hi(v)};
                        }
                    }
                }
#pragma unroll
                for (int kh = 0; kh < K; ++kh) {
                    if (pmod_c(j - kh, S) != 0) continue;
                    const int sl = pmod_c(fdiv_c(j - kh, S), C::A);
#pragma unroll
                    for (int kw = 0; kw < K; ++kw)
#pragma unroll
                        for (int i = 0; i < NCOL; ++i)
#pragma unroll
                            for (int h = 0; h < C::H2; ++h)
                                acc[kh * K + kw][h] = __builtin_elementwise_fma(g[sl][i][h], in[i * S + kw][h], acc[kh * K + kw][h]);
                }
            }
        }
        if (!more) break;
        it = it2; img = img2; ox0 = ox2; oy0 = oy2; nrows = nrows2; nblk = nblk2; b = b2;
    }

    // reduce the lanes that share a channel (PXW pixels x 4 waves) through LDS, one atomic per (tap, channel) per workgroup
    float* red = reinterpret_cast<float*>(smem);          // [256 threads][CPL]
    for (int t = 0; t < K * K; ++t) {
        __syncthreads();
#pragma unroll
        for (int h = 0; h < C::H2; ++h) {
            f32x2_t a = f32x2_t{0.f, 0.f};
#pragma unroll
            for (int tt = 0; tt < K * K; ++tt)
                if (tt == t) a = acc[tt][h];              // static register indexing
            red[tid * CPL + 2 * h] = lane_ok ? a.x : 0.f;
            red[tid * CPL + 2 * h + 1] = lane_ok ? a.y : 0.f;
        }
        __syncthreads();
        if (tid < C::TCH && c0 + tid < p.c) {
            float s = 0.f;
            for (int wv = 0; wv < 4; ++wv)
                for (int q = 0; q < C::PXW; ++q) s += red[(wv * 64 + q * LP + tid / CPL) * CPL + tid % CPL];
            atomicAdd(reinterpret_cast<float*>(p.out) + (long long)t * p.c + c0 + tid, s);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Stride-2 data gradient, marching form.  With (iy + pad_t) = 2*o' + f and (ix + pad_l) = 2*j + e the transposed
// convolution is a STRIDE-1 stencil over "super pixels" (o', j) of 2x2 outputs:
//     dx[2o'-pt+f, 2j-pl+e] = sum_{d,d'} dy[o'-d, j-d'] * w[f+2d, e+2d']          (taps with f+2d, e+2d' < K)
// so the forward kernel's structure applies with D = ceil(K/2) live super-rows per lane: a lane owns NJ adjacent
// super-columns of its channels, each staged dy row is read once (NJ+D-1 reads), scattered into the D super-rows it
// touches (all K*K taps of the lane's channels, held in registers, are used once per dy row), and super-row o' is
// complete - four output pixels per super-column - as soon as dy row o' has been processed.  The kernel is bound by
// the 4x larger dx write stream.
template <int K, int CPL, int LP, int NJ, bool EPI = false> struct MarchBwdCfg {
    static constexpr int H2 = CPL / 2;
    static constexpr int PXW = 64 / LP;
    static constexpr int D = (K + 1) / 2;                  // super-taps per dimension = live super-rows
    static constexpr int TOWJ = 4 * PXW * NJ;              // super-columns per strip
    static constexpr int IW_T = TOWJ + D - 1;              // staged dy columns
    static constexpr int NIN = NJ + D - 1;
    static constexpr int TCH = LP * CPL;
    static constexpr int PXB = TCH * 2;
    static constexpr int VPP = PXB / 16;
    static constexpr int PSB = (CPL == 2 && LP == 32 && NJ == 2) ? 192 : PXB;      // conflict-free 4-byte reads
    static constexpr int NR_ = (16384 + D * IW_T * PXB / 2) / (D * IW_T * PXB);
    static constexpr int NR0 = NR_ < 1 ? 1 : NR_;
    // epilogue variant: the e values of a block's output pixels are prefetched into REGISTERS beside the dy block (two
    // generations live: 2 * RB * 4 * NJ * CPL/2 registers), so its blocks are shorter
    static constexpr int NR = EPI ? (K == 3 ? (NR0 < 2 ? NR0 : 2) : 1) : NR0;
    static constexpr int RB = D * NR;                      // dy rows per staged block
    static constexpr int TS = 256 - 256 % VPP;
    static constexpr int NV = (RB * IW_T * VPP + TS - 1) / TS;
    static constexpr int BUF_BYTES = RB * IW_T * PSB;
};

// EPI: the launch also finishes the BatchNorm0 + SiLU backward of the expand conv output e = p.epi_x [n,h,w,c] (see the
// stride-1 form above): it writes dZ0 = dA0 * silu'(e*scale + shift) and leaves (sum dZ0, sum dZ0 * xhat0) in stat_partials.
// The e values of the pixels a block completes are loaded straight into registers with the block's dy prefetch (every
// lane reads exactly the pixels it writes).
template <int K, int CPL, int LP, int NJ, bool EPI = false>
__global__ __launch_bounds__(256, 2) void dwconv_march_bwd_s2_kernel(const mc_dwconv_args p, int strips, int segs, int seg_rows,
                                                                     int ctiles, int gy) {
    using C = MarchBwdCfg<K, CPL, LP, NJ, EPI>;
    typedef typename std::conditional<CPL == 4, uint2, uint32_t>::type ldsv_t;
    constexpr int SM_BYTES = (EPI && C::BUF_BYTES < 256 * 2 * CPL * 4) ? 256 * 2 * CPL * 4 : C::BUF_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SM_BYTES];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ct = slot % ctiles, y = (slot / ctiles) * 8 + xcd;
    if (y >= gy) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane % LP, px = lane / LP;
    const bool lane_ok = px < C::PXW;
    const int c0 = ct * C::TCH;
    const int cl = c0 + lq * CPL;
    const bool ch_ok = lane_ok && cl < p.c;
    const int jl0 = (wave * C::PXW + (lane_ok ? px : 0)) * NJ;       // strip-local first super-column
    const int lbase = jl0 * C::PSB + lq * (CPL * 2);
    const int ohv = (p.h + p.pad_t + 1) >> 1;              // super-rows / super-columns that cover the image
    const int owv = (p.w + p.pad_l + 1) >> 1;

    f32x2_t w[K * K][C::H2];
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
        for (int h = 0; h < C::H2; ++h) {
            w[t][h] = f32x2_t{0.f, 0.f};
            if (ch_ok) w[t][h] = *reinterpret_cast<const f32x2_t*>(p.w_kkc + (long long)t * p.c + cl + 2 * h);
        }
    const int vv = tid % C::VPP;
    const int cs = c0 + vv * 8;
    const bool st_ok = tid < C::TS && cs < p.c;
    const long long g_row_pitch = (long long)p.ow * p.c;
    unsigned meta[C::NV];                                  // row | col << 8 | (LDS byte offset / 16) << 16
#pragma unroll
    for (int i = 0; i < C::NV; ++i) {
        const int v = tid + i * C::TS;
        const int row = v / (C::IW_T * C::VPP), col = (v / C::VPP) % C::IW_T;
        meta[i] = (unsigned)row | ((unsigned)col << 8) | ((unsigned)(((row * C::IW_T + col) * C::PSB + vv * 16) >> 4) << 16);
        if (v >= C::RB * C::IW_T * C::VPP) meta[i] = 0xffu;
    }
    f32x2_t acc[C::D][NJ][4][C::H2];                       // [super-row slot][super-column][f*2+e][channel pair]
    constexpr int NE = EPI ? C::RB * 4 * NJ : 1;           // e pixels a lane completes per block: [dy row][f][i][e]
    ldsv_t enext[NE], ecur[NE];
    f32x2_t e_sc[C::H2], e_sh[C::H2], e_mu[C::H2], ssum[C::H2], ssq[C::H2];
#pragma unroll
    for (int h = 0; h < C::H2; ++h) {
        e_sc[h] = e_sh[h] = e_mu[h] = ssum[h] = ssq[h] = f32x2_t{0.f, 0.f};
        if (EPI && ch_ok) {
            e_sc[h] = *reinterpret_cast<const f32x2_t*>(p.epi_scale + cl + 2 * h);
            e_sh[h] = *reinterpret_cast<const f32x2_t*>(p.epi_shift + cl + 2 * h);
            e_mu[h] = *reinterpret_cast<const f32x2_t*>(p.epi_mean + cl + 2 * h);
        }
    }

    const int nitems = p.n * strips * segs;
    auto item_geom = [&](int it, int& img, int& j0, int& s0, int& nrows, int& nblk) {
        const int strip = it % strips;
        const int seg = (it / strips) % segs;
        img = it / (strips * segs);
        j0 = strip * C::TOWJ;
        s0 = seg * seg_rows;
        nrows = ohv - s0 < seg_rows ? ohv - s0 : seg_rows;             // super-rows owned by the item
        nblk = (nrows + C::D - 1 + C::RB - 1) / C::RB;                 // dy rows s0-(D-1) .. s0+nrows-1
    };
    uint4 vals[C::NV];
    unsigned inb = 0, colmask = 0;
    // EPI: e of the pixels the dy rows of block b complete (dy row k of the block completes super-row b*RB + k - (D-1))
    auto epi_load = [&](int img, int j0, int s0, int nrows_i, int b) {
        if constexpr (EPI) {
            const long long ix = 2LL * (j0 + jl0) - p.pad_l;
#pragma unroll
            for (int k = 0; k < C::RB; ++k) {
                const int sr = b * C::RB + k - (C::D - 1);
                const long long iy0 = 2LL * (s0 + sr) - p.pad_t;
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int i = 0; i < NJ; ++i)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const long long x = ix + 2 * i + e, yy = iy0 + f;
                            const bool ok = ch_ok && sr >= 0 && sr < nrows_i && yy >= 0 && yy < p.h && x >= 0 && x < p.w;
                            const bf16_t* a = ok ? p.epi_x + (((long long)img * p.h + yy) * p.w + x) * p.c + cl : p.epi_x;
                            enext[((k * 2 + f) * NJ + i) * 2 + e] = *reinterpret_cast<const ldsv_t*>(a);   // unconditional (see stage_load)
                        }
            }
        }
    };
    auto stage_load = [&](int img, int j0, int s0, int b, bool new_item) {
        const int oy0 = s0 - (C::D - 1) + b * C::RB, ox0 = j0 - (C::D - 1);
        if (new_item) {
            colmask = 0;
#pragma unroll
            for (int i = 0; i < C::NV; ++i) {
                const int ox = ox0 + (int)((meta[i] >> 8) & 0xffu);
                if (st_ok && ox >= 0 && ox < p.ow) colmask |= 1u << i;
            }
        }
        const bf16_t* org = p.dy + ((long long)img * p.oh + oy0) * g_row_pitch + (long long)ox0 * p.c + c0;
        inb = 0;
#pragma unroll
        for (int i = 0; i < C::NV; ++i) {
            const int oy = oy0 + (int)(meta[i] & 0xffu);
            const bool ok = ((colmask >> i) & 1u) && oy >= 0 && oy < p.oh;
            const unsigned goff = (meta[i] & 0xffu) * (unsigned)g_row_pitch + ((meta[i] >> 8) & 0xffu) * (unsigned)p.c + vv * 8;
            const bf16_t* a = ok ? org + goff : p.dy;
            vals[i] = *reinterpret_cast<const uint4*>(a);
            inb |= (ok ? 1u : 0u) << i;
        }
    };
    auto stage_store = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
        for (int i = 0; i < C::NV; ++i) {
            if (tid < C::TS && (meta[i] & 0xffu) != 0xffu) {
                const bool real = (inb >> i) & 1u;
                const uint4 val = make_uint4(real ? vals[i].x : 0u, real ? vals[i].y : 0u, real ? vals[i].z : 0u, real ? vals[i].w : 0u);
                *reinterpret_cast<uint4*>(smem + ((meta[i] >> 16) << 4)) = val;
            }
        }
        if constexpr (EPI) {
#pragma unroll
            for (int i = 0; i < NE; ++i) ecur[i] = enext[i];
        }
    };

    int it = y, img = 0, j0 = 0, s0 = 0, nrows = 0, nblk = 0, b = 0;
    if (it >= nitems) return;
    item_geom(it, img, j0, s0, nrows, nblk);
    stage_load(img, j0, s0, 0, true);
    epi_load(img, j0, s0, nrows, 0);
    int sr_next = 0;                                       // super-row (item-relative) completed by the next dy row
    bf16_t* optr = nullptr;                                // lane's pixel (f = 0, e = 0 of its first super-column) in that super-row
    unsigned ok_mask = 0;                                  // bit (i*2+e): output column exists
    const long long row_pitch = (long long)p.w * p.c;
    while (true) {
        __syncthreads();
        stage_store();
        __syncthreads();
        int it2 = it, img2 = img, j2 = j0, s2 = s0, nrows2 = nrows, nblk2 = nblk, b2 = b + 1;
        if (b2 >= nblk) {
            it2 = it + gy; b2 = 0;
            if (it2 < nitems) item_geom(it2, img2, j2, s2, nrows2, nblk2);
        }
        const bool more = it2 < nitems;
        if (more) { stage_load(img2, j2, s2, b2, b2 == 0); epi_load(img2, j2, s2, nrows2, b2); }

        if (b == 0) {
#pragma unroll
            for (int a = 0; a < C::D; ++a)
#pragma unroll
                for (int i = 0; i < NJ; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int h = 0; h < C::H2; ++h) acc[a][i][q][h] = f32x2_t{0.f, 0.f};
            sr_next = -(C::D - 1);
            const long long iy = 2LL * (s0 + sr_next) - p.pad_t, ix = 2LL * (j0 + jl0) - p.pad_l;
            optr = reinterpret_cast<bf16_t*>(p.out) + (((long long)img * p.h + iy) * p.w + ix) * p.c + cl;
            ok_mask = 0;
#pragma unroll
            for (int i = 0; i < NJ; ++i)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const long long x = ix + 2 * i + e;
                    if (ch_ok && x >= 0 && x < p.w) ok_mask |= 1u << (i * 2 + e);
                }
        }
        constexpr int SB_UNROLL = EPI ? C::NR : 1;            // (the epilogue form indexes its e registers by sb)
#pragma unroll SB_UNROLL
        for (int sb = 0; sb < C::NR; ++sb) {
            const unsigned char* lp = smem + lbase + sb * (C::D * C::IW_T * C::PSB);
#pragma unroll
            for (int jr = 0; jr < C::D; ++jr) {            // dy row jr of the period: enters super-rows jr .. jr+D-1 (mod D)
                f32x2_t in[C::NIN][C::H2];
#pragma unroll
                for (int i = 0; i < C::NIN; ++i) {
                    const ldsv_t v = *reinterpret_cast<const ldsv_t*>(lp + (jr * C::IW_T + i) * C::PSB);
                    if constexpr (CPL == 4) {
                        in[i][0] = f32x2_t{bf_lo(v.x), bf_hi(v.x)};
                        in[i][1] = f32x2_t{bf_lo(v.y), bf_hi(v.y)};
                    } else {
                        in[i][0] = f32x2_t{bf_lo(v), bf_hi(v)};
                    }
                }
                // staged column index of dy column (j - d') for the lane's super-column i:  (D-1) + i - d'
#pragma unroll
                for (int d = 0; d < C::D; ++d) {
                    const int sl = (jr + d) % C::D;
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        if (f + 2 * d >= K) continue;
#pragma unroll
                        for (int dd = 0; dd < C::D; ++dd)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                if (e + 2 * dd >= K) continue;
#pragma unroll
                                for (int i = 0; i < NJ; ++i)
#pragma unroll
                                    for (int h = 0; h < C::H2; ++h)
                                        acc[sl][i][f * 2 + e][h] = __builtin_elementwise_fma(
                                            w[(f + 2 * d) * K + e + 2 * dd][h], in[C::D - 1 + i - dd][h], acc[sl][i][f * 2 + e][h]);
                            }
                    }
                }
                {   // super-row sr_next is complete
                    const int sl = jr % C::D;
                    if (sr_next >= 0 && sr_next < nrows) {
                        const long long iy0 = 2LL * (s0 + sr_next) - p.pad_t;
#pragma unroll
                        for (int f = 0; f < 2; ++f) {
                            if (iy0 + f >= 0 && iy0 + f < p.h) {
#pragma unroll
                                for (int i = 0; i < NJ; ++i)
#pragma unroll
                                    for (int e = 0; e < 2; ++e) {
                                        if ((ok_mask >> (i * 2 + e)) & 1u) {
                                            bf16_t* o = optr + f * row_pitch + (long long)(2 * i + e) * p.c;
                                            uint32_t o2[C::H2];
#pragma unroll
                                            for (int h = 0; h < C::H2; ++h) {
                                                if constexpr (EPI) {
                                                    const ldsv_t evv = ecur[(((sb * C::D + jr) * 2 + f) * NJ + i) * 2 + e];
                                                    uint32_t ew;
                                                    if constexpr (CPL == 4) ew = h == 0 ? evv.x : evv.y; else ew = evv;
                                                    const f32x2_t e2 = {bf_lo(ew), bf_hi(ew)};
                                                    const f32x2_t z = __builtin_elementwise_fma(e2, e_sc[h], e_sh[h]);
                                                    const f32x2_t a2 = acc[sl][i][f * 2 + e][h];
                                                    const f32x2_t dz2 = a2 * silu_grad2_f(z);
                                                    o2[h] = pack_bf2(dz2.x, dz2.y);
                                                    const f32x2_t r = {bf_lo(o2[h]), bf_hi(o2[h])};     // reductions of the stored (rounded) dZ0
                                                    ssum[h] += r;
                                                    ssq[h] = __builtin_elementwise_fma(r, e2 - e_mu[h], ssq[h]);   // centred (see the stride-1 form)
                                                } else {
                                                    o2[h] = pack_bf2(acc[sl][i][f * 2 + e][h].x, acc[sl][i][f * 2 + e][h].y);
                                                }
                                            }
                                            if constexpr (CPL == 4) *reinterpret_cast<uint2*>(o) = make_uint2(o2[0], o2[1]);
                                            else *reinterpret_cast<uint32_t*>(o) = o2[0];
                                        }
                                    }
                            }
                        }
                    }
                    ++sr_next;
                    optr += 2 * row_pitch;
#pragma unroll
                    for (int i = 0; i < NJ; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int h = 0; h < C::H2; ++h) acc[sl][i][q][h] = f32x2_t{0.f, 0.f};
                }
            }
        }
        if (!more) break;
        it = it2; img = img2; j0 = j2; s0 = s2; nrows = nrows2; nblk = nblk2; b = b2;
    }
    if constexpr (EPI) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);          // [256 threads][2][CPL]
#pragma unroll
        for (int h = 0; h < C::H2; ++h) {
            red[tid * 2 * CPL + 2 * h] = ssum[h].x; red[tid * 2 * CPL + 2 * h + 1] = ssum[h].y;
            red[tid * 2 * CPL + CPL + 2 * h] = ssq[h].x; red[tid * 2 * CPL + CPL + 2 * h + 1] = ssq[h].y;
        }
        __syncthreads();
        if (tid < 2 * C::TCH) {
            const int ch = tid % C::TCH, which = tid / C::TCH;          // 0 = sum dZ, 1 = sum dZ * (e - mean)
            float sv = 0.f;
            for (int wv = 0; wv < 4; ++wv)
                for (int q = 0; q < C::PXW; ++q) sv += red[(wv * 64 + q * LP + ch / CPL) * 2 * CPL + which * CPL + ch % CPL];
            if (c0 + ch < p.c) {
                // sum dZ * xhat = invstd * sum dZ * (e - mean)
                if (which == 1) sv = p.epi_invstd[c0 + ch] * sv;
                p.stat_partials[((long long)y * 2 + which) * p.c + c0 + ch] = sv;
            }
        }
    }
}

template <int K, int CPL, int LP, int NJ, bool EPI = false> int march_bwd_s2_plan(const mc_dwconv_args& p, int* strips_, int* segs_, int* seg_rows_, int* ctiles_) {
    using C = MarchBwdCfg<K, CPL, LP, NJ, EPI>;
    const int ohv = (p.h + p.pad_t + 1) >> 1, owv = (p.w + p.pad_l + 1) >> 1;
    const int strips = mc_div_up(owv, C::TOWJ), ctiles = mc_div_up(p.c, C::TCH);
    long long base = (long long)p.n * strips * ctiles;
    int segs = (int)((2048 + base - 1) / base);
    int max_segs = ohv / 16 > 0 ? ohv / 16 : 1;
    if (segs > max_segs) segs = max_segs;
    if (segs < 1) segs = 1;
    const int seg_rows = mc_div_up(mc_div_up(ohv, segs), C::RB) * C::RB;
    segs = mc_div_up(ohv, seg_rows);
    long long nitems = (long long)p.n * strips * segs;
    long long per_xcd = 64 / ctiles;                            // 2 workgroups per CU, capped per XCD (see march_plan)
    if (per_xcd < 1) per_xcd = 1;
    long long cap = 8 * per_xcd;
    long long per = (nitems + cap - 1) / cap;
    *strips_ = strips; *segs_ = segs; *seg_rows_ = seg_rows; *ctiles_ = ctiles;
    return (int)((nitems + per - 1) / per);
}
template <int K, int CPL, int LP, int NJ> int launch_march_bwd_s2_epi(const mc_dwconv_args& p, hipStream_t st) {
    int strips, segs, seg_rows, ctiles;
    const int gy = march_bwd_s2_plan<K, CPL, LP, NJ, true>(p, &strips, &segs, &seg_rows, &ctiles);
    const int gy8 = (gy + 7) / 8 * 8;
    hipLaunchKernelGGL((dwconv_march_bwd_s2_kernel<K, CPL, LP, NJ, true>), dim3(gy8 * ctiles), dim3(256), 0, st, p, strips, segs,
                       seg_rows, ctiles, gy);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

template <int K, int CPL, int LP, int NJ> int launch_march_bwd_s2(const mc_dwconv_args& p, hipStream_t st) {
    using C = MarchBwdCfg<K, CPL, LP, NJ>;
    const int ohv = (p.h + p.pad_t + 1) >> 1, owv = (p.w + p.pad_l + 1) >> 1;
    const int strips = mc_div_up(owv, C::TOWJ), ctiles = mc_div_up(p.c, C::TCH);
    long long base = (long long)p.n * strips * ctiles;
    int segs = (int)((2048 + base - 1) / base);
    int max_segs = ohv / 16 > 0 ? ohv / 16 : 1;
    if (segs > max_segs) segs = max_segs;
    if (segs < 1) segs = 1;
    const int seg_rows = mc_div_up(mc_div_up(ohv, segs), C::RB) * C::RB;
    segs = mc_div_up(ohv, seg_rows);
    long long nitems = (long long)p.n * strips * segs;
    long long per_xcd = 64 / ctiles;                            // 2 workgroups per CU, capped per XCD (see march_plan)
    if (per_xcd < 1) per_xcd = 1;
    long long cap = 8 * per_xcd;
    long long per = (nitems + cap - 1) / cap;
    const int gy = (int)((nitems + per - 1) / per);
    const int gy8 = (gy + 7) / 8 * 8;
    hipLaunchKernelGGL((dwconv_march_bwd_s2_kernel<K, CPL, LP, NJ>), dim3(gy8 * ctiles), dim3(256), 0, st, p, strips, segs, seg_rows,
                       ctiles, gy);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

struct MarchPlan { int strips, segs, seg_rows, ctiles, gy; };
template <typename C> MarchPlan march_plan(const mc_dwconv_args& p) {
    MarchPlan m;
    m.strips = mc_div_up(p.ow, C::TOW);
    m.ctiles = mc_div_up(p.c, C::TCH);
    long long base = (long long)p.n * m.strips * m.ctiles;
    int segs = (int)((3072 + base - 1) / base);
    int max_segs = p.oh / 24 > 0 ? p.oh / 24 : 1;
    if (segs > max_segs) segs = max_segs;
    if (segs < 1) segs = 1;
    m.seg_rows = mc_div_up(mc_div_up(p.oh, segs), C::A * C::NR) * (C::A * C::NR);
    m.segs = mc_div_up(p.oh, m.seg_rows);
    // persistent workgroups: as many as are resident at once, every one with the same item count
    long long nitems = (long long)p.n * m.strips * m.segs;
    // Block b runs on XCD b % 8 and the kernel maps (b >> 3) -> (channel tile, y / 8): an XCD receives ceil(gy / 8) * ctiles
    // workgroups and holds 32 CUs x OCC of them.  The y count is capped PER XCD (round 3): the chip-wide cap 256 * OCC /
    // ctiles put 68 workgroups on the 64 slots of XCDs 0-3 at c = 1056 (17 channel tiles, gy = 28): four stragglers ran a
    // second round and the launch took 1.9x the time of the c = 768 launch for 1.4x the work.
    const long long occ = p.epi_x ? 2 : C::OCC;                   // (the epilogue variant is built for 2 workgroups per CU)
    long long per_xcd = 32 * occ / m.ctiles;
    if (per_xcd < 1) per_xcd = 1;
    long long cap = 8 * per_xcd;
    long long per = (nitems + cap - 1) / cap;
    m.gy = (int)((nitems + per - 1) / per);
    return m;
}

// tile shape by kernel size and channel count: k = 3 runs 4 channels per lane (64 / 48 / 24-channel tiles),
// k = 5 (25 taps per channel in registers) 2 channels per lane (32 / 24-channel tiles)
template <int K, int S, typename F> auto march_dispatch(const mc_dwconv_args& p, F&& f, bool whole_pixels = false) {
    if constexpr (K == 3) {
        if (p.c == 24) return f(MarchCfg<K, S, 4, 6>{});
        // c = 240 (B5 stage 2, 380x228: the largest 3x3 tensors of the network): 64-channel tiles are 128-byte pieces of
        // 480-byte pixels, misaligned with the 128-byte lines on 3 pixels of 4 -- whole pixels (60 lanes x 4 channels, one
        // pixel per wave, 8-column strips whose rows are contiguous 3.8 KB runs) run the forward 1.2-1.35x faster in spite
        // of the 2-in-10 column halo; the weight gradient gains 1.2x once it runs three workgroups per CU (0.725 -> 0.597 ms)
        if (p.c == 240 && S == 1 && whole_pixels) return f(MarchCfg<K, S, 4, 60>{});
        // (c = 144, 288-byte pixels: whole-pixel tiles of 36 lanes leave 28 lanes of a wave idle -- 0.85 vs 0.96 ms without the
        // prologue, 1.18 vs 1.07 ms with it; 72-channel tiles of 18 lanes: 0.90 / 1.13 ms: not used)
        if (p.c % 48 == 0 && p.c < 192) return f(MarchCfg<K, S, 4, 12>{});      // 48, 144: exact 48-channel tiles
        return f(MarchCfg<K, S, 4, 16>{});
    } else {
        // 64-channel tiles (full 128-byte lines; 32-channel tiles measured 1.5x HBM over-fetch), 2 channels per lane
        return f(MarchCfg<K, S, 2, 32, (S == 1 ? 4 : 2)>{});
    }
}
template <int K, int S, typename C> int launch_march(const mc_dwconv_args& p, hipStream_t st) {
    MarchPlan m = march_plan<C>(p);
    int gy8 = (m.gy + 7) / 8 * 8;
    if constexpr (S == 1) {
        if (p.epi_x) {
            hipLaunchKernelGGL((dwconv_march_fwd_kernel<K, S, C::H2 * 2, C::TCH / (C::H2 * 2), C::NCOL, true>), dim3(gy8 * m.ctiles), dim3(256),
                               0, st, p, m.strips, m.segs, m.seg_rows, m.ctiles, m.gy);
            MC_LAUNCH_CHECK();
            return MC_OK;
        }
    }
    hipLaunchKernelGGL((dwconv_march_fwd_kernel<K, S, C::H2 * 2, C::TCH / (C::H2 * 2), C::NCOL>), dim3(gy8 * m.ctiles), dim3(256), 0, st, p,
                       m.strips, m.segs, m.seg_rows, m.ctiles, m.gy);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
template <int K, int S> int launch_march_cp(const mc_dwconv_args& p, hipStream_t st) {
    return march_dispatch<K, S>(p, [&](auto cfg) { return launch_march<K, S, decltype(cfg)>(p, st); }, true);
}
template <int K, int S, typename C> int launch_march_bww(const mc_dwconv_args& p, hipStream_t st) {
    MarchPlan m = march_plan<C>(p);
    long long nitems = (long long)p.n * m.strips * m.segs;      // 2 workgroups per CU here; capped per XCD (see march_plan)
    long long per_xcd = (C::TCH == 240 ? 96 : 64) / m.ctiles;    // (the whole-pixel form fits three workgroups per CU: 156 VGPRs)
    if (per_xcd < 1) per_xcd = 1;
    long long cap = 8 * per_xcd;
    long long per = (nitems + cap - 1) / cap;
    m.gy = (int)((nitems + per - 1) / per);
    int gy8 = (m.gy + 7) / 8 * 8;
    hipLaunchKernelGGL((dwconv_march_bww_kernel<K, S, C::H2 * 2, C::TCH / (C::H2 * 2), C::NCOL>), dim3(gy8 * m.ctiles), dim3(256), 0, st,
                       p, m.strips, m.segs, m.seg_rows, m.ctiles, m.gy);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
template <int K, int S> int launch_march_bww_cp(const mc_dwconv_args& p, hipStream_t st) {
    return march_dispatch<K, S>(p, [&](auto cfg) { return launch_march_bww<K, S, decltype(cfg)>(p, st); }, true);
}
template <int K, int S> int march_rows(const mc_dwconv_args& p) {
    return march_dispatch<K, S>(p, [&](auto cfg) { return march_plan<decltype(cfg)>(p).gy; }, true);
}

int check_common(const mc_dwconv_args& p) {
    MC_CHECK(p.x || p.dy, "dwconv: null input");
    MC_CHECK(p.out && p.n > 0 && p.h > 0 && p.w > 0 && p.c > 0, "dwconv: bad shape");
    MC_CHECK(p.c % 8 == 0, "dwconv: channels must be a multiple of 8");
    MC_CHECK((p.k == 3 || p.k == 5) && (p.stride == 1 || p.stride == 2), "dwconv: k in {3,5}, stride in {1,2}");
    MC_CHECK(p.oh > 0 && p.ow > 0, "dwconv: bad output shape");
    MC_CHECK((p.pro_scale == nullptr) == (p.pro_shift == nullptr), "dwconv: prologue needs scale and shift");
    return MC_OK;
}

// lane = column form (conv_lane.hip): MC_DW_LANE=0 never, =1 wherever it is supported, unset: the policy below
extern "C" int mc_dwconv_lane_supported(const mc_dwconv_args* a);
extern "C" int mc_dwconv_lane_stat_rows(const mc_dwconv_args* a);
extern "C" int mc_dwconv_fwd_lane(const mc_dwconv_args* a, void* stream);
extern "C" int mc_dwconv_bwd_weight_lane(const mc_dwconv_args* a, void* stream);
std::atomic<int> g_lane_mode{-2};     // -2: not read yet; -1 policy; 0 never; 1 wherever supported
int lane_mode() {
    int m = g_lane_mode.load(std::memory_order_relaxed);
    if (m == -2) {
        const char* e = getenv("MC_DW_LANE");
        int want = e ? atoi(e) : -1, expect = -2;
        g_lane_mode.compare_exchange_strong(expect, want);          // (a concurrent mc_dwconv_set_lane_mode wins)
        m = g_lane_mode.load(std::memory_order_relaxed);
    }
    return m;
}
bool use_lane_fwd(const mc_dwconv_args& p) {
    const int mode = lane_mode();
    if (mode == 0 || !mc_dwconv_lane_supported(&p)) return false;
    if (mode == 1) return true;
    // measured per shape on one box (scripts/dw_form_ab.py, 32 images): the lane form wins for every 5x5 map (114 / 57 columns:
    // 1.2-1.45x; 29 columns with four images per wave: 1.1x), for the 3x3 forms at 57 columns, and for the 3x3 29-column maps
    // whose pixels are not a multiple of 128 bytes (c = 1824: the marching kernel's 64-channel tiles are misaligned there;
    // at c = 3072 the marching form stays ahead).  Narrow maps need enough images to fill the wave's lane groups.
    const bool narrow = p.ow <= 30;
    if (p.k == 5) {
        if (p.stride == 1) return p.ow >= 50 || (p.ow <= 29 && p.n >= 4);
        return p.ow >= 100 || (narrow && p.n >= 2);
    }
    if (p.stride == 2) return !p.epi_x && p.ow >= 50 && p.ow < 100;
    return (p.ow >= 50 && p.ow < 100) || (narrow && p.n >= 4 && (p.c * 2) % 128 != 0);
}

bool use_lane_bww(const mc_dwconv_args& p) {
    const int mode = lane_mode();
    if (mode == 0 || p.epi_x || !mc_dwconv_lane_supported(&p)) return false;
    if (mode == 1) return true;
    // measured like the forward forms (scripts/dw_form_ab.py): 5x5 everywhere (29 columns: two images per wave), 3x3 at 57 and
    // at 29 columns (1.1-1.3x)
    const bool narrow = p.ow <= 29;
    if (p.k == 5) return p.stride == 1 ? (p.ow >= 50 || (narrow && p.n >= 2)) : (p.ow >= 100 || (p.ow <= 30 && p.n >= 2));
    return (p.ow >= 50 && p.ow < 100) || (p.stride == 1 && p.ow <= 30 && p.n >= 4);
}

}  // namespace

// developer / test switch of the depthwise forward form: -1 = the built-in policy, 0 = marching kernels only, 1 = the
// lane = column kernels wherever they support the shape.  Returns the previous mode.
extern "C" int mc_dwconv_set_lane_mode(int mode) {
    const int old = g_lane_mode.exchange(mode);
    return old == -2 ? -1 : old;
}

extern "C" int mc_dwconv_stat_rows(const mc_dwconv_args* a) {
    if (use_lane_fwd(*a)) return mc_dwconv_lane_stat_rows(a);
    if (a->k == 3) return a->stride == 1 ? march_rows<3, 1>(*a) : march_rows<3, 2>(*a);
    return a->stride == 1 ? march_rows<5, 1>(*a) : march_rows<5, 2>(*a);
}

// rows of stat_partials written by mc_dwconv_bwd_data with the stride-2 BatchNorm-backward epilogue (epi_x set)
extern "C" int mc_dwconv_bwd_data_stat_rows(const mc_dwconv_args* a) {
    const mc_dwconv_args& p = *a;
    int s1, s2, s3, s4;
    if (p.stride != 2) return 0;
    if (p.k == 3) {
        if (p.c % 48 == 0 && p.c < 192) return march_bwd_s2_plan<3, 4, 12, 1, true>(p, &s1, &s2, &s3, &s4);
        return march_bwd_s2_plan<3, 4, 16, 1, true>(p, &s1, &s2, &s3, &s4);
    }
    return march_bwd_s2_plan<5, 2, 32, 2, true>(p, &s1, &s2, &s3, &s4);
}

extern "C" int mc_dwconv_fwd(const mc_dwconv_args* a, void* stream) {
    const mc_dwconv_args& p = *a;
    if (int e = check_common(p)) return e;
    MC_CHECK(p.x && p.w_kkc, "dwconv_fwd: null x / w");
    MC_CHECK(!p.epi_x || (p.stride == 1 && p.epi_scale && p.epi_shift && p.epi_mean && p.epi_invstd && p.stat_partials),
             "dwconv_fwd: the BatchNorm-backward epilogue needs stride 1, scale/shift/mean/invstd and stat_partials");
    hipStream_t st = (hipStream_t)stream;
    // the kernel form decides how many rows of stat_partials are written: a caller that says how many it allocated is held
    // to the form this launch picks (the form can change between mc_dwconv_stat_rows and here: mc_dwconv_set_lane_mode)
    MC_CHECK(!(p.stat_partials && p.stat_rows > 0) || p.stat_rows == mc_dwconv_stat_rows(a),
             "dwconv_fwd: stat_partials was sized for another kernel form (stat_rows != mc_dwconv_stat_rows now)");
    if (use_lane_fwd(p)) return mc_dwconv_fwd_lane(a, stream);
    if (p.k == 3 && p.stride == 1) return launch_march_cp<3, 1>(p, st);
    if (p.k == 3 && p.stride == 2) return launch_march_cp<3, 2>(p, st);
    if (p.k == 5 && p.stride == 1) return launch_march_cp<5, 1>(p, st);
    return launch_march_cp<5, 2>(p, st);
}

extern "C" int mc_dwconv_bwd_data(const mc_dwconv_args* a, void* stream) {
    const mc_dwconv_args& p = *a;
    if (int e = check_common(p)) return e;
    MC_CHECK(p.dy && p.w_kkc, "dwconv_bwd_data: null dy / w");
    MC_CHECK(!p.epi_x || (p.stride == 2 && p.epi_scale && p.epi_shift && p.epi_mean && p.epi_invstd && p.stat_partials),
             "dwconv_bwd_data: the BatchNorm-backward epilogue is provided for stride 2 here (stride 1: mc_dwconv_fwd on flipped taps) and needs scale/shift/mean/invstd and stat_partials");
    if (p.stride == 2 && p.epi_x) {
        hipStream_t st = (hipStream_t)stream;
        if (p.k == 3) {
            if (p.c % 48 == 0 && p.c < 192) return launch_march_bwd_s2_epi<3, 4, 12, 1>(p, st);
            return launch_march_bwd_s2_epi<3, 4, 16, 1>(p, st);
        }
        return launch_march_bwd_s2_epi<5, 2, 32, 2>(p, st);
    }
    if (p.stride == 2) {                                               // marching form
        hipStream_t st = (hipStream_t)stream;
        if (p.k == 3) {
            if (p.c % 48 == 0 && p.c < 192) return launch_march_bwd_s2<3, 4, 12, 1>(p, st);
            return launch_march_bwd_s2<3, 4, 16, 1>(p, st);
        }
        return launch_march_bwd_s2<5, 2, 32, 2>(p, st);
    }
    long long total = (long long)p.n * p.h * p.w * (p.c / 8);
    int blocks = mc_div_up(total, 256);
    if (blocks > 16384) blocks = 16384;
    if (p.k == 3) hipLaunchKernelGGL((dwconv_bwd_data_kernel<3>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((dwconv_bwd_data_kernel<5>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

extern "C" int mc_dwconv_bwd_weight(const mc_dwconv_args* a, void* stream) {
    const mc_dwconv_args& p = *a;
    if (int e = check_common(p)) return e;
    MC_CHECK(p.x && p.dy, "dwconv_bwd_weight: null x / dy");
    hipStream_t st = (hipStream_t)stream;
    if (use_lane_bww(p)) return mc_dwconv_bwd_weight_lane(a, stream);
    if (p.k == 3 && p.stride == 1) return launch_march_bww_cp<3, 1>(p, st);
    if (p.k == 3 && p.stride == 2) return launch_march_bww_cp<3, 2>(p, st);
    if (p.k == 5 && p.stride == 1) return launch_march_bww_cp<5, 1>(p, st);
    return launch_march_bww_cp<5, 2>(p, st);
}
