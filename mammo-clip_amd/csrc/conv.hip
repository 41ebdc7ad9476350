// Depthwise k x k convolution (k in {3,5}, stride in {1,2}) for NHWC bf16, gfx950.
// [ref: efficientnet_custom.py:109-111  _depthwise_conv (+ static ZeroPad2d, efficient_net_custom_utils.py:248-276)]
//
// HBM-bound stencil: a workgroup stages a (TOH*S+K-1) x (TOW*S+K-1) x 64-channel input halo tile in LDS with
// coalesced 16-byte loads (the BatchNorm+SiLU of the producing expand conv is applied once per element while
// staging, zero padding is inserted after the activation), then every thread computes a strip of R output
// pixels for one 8-channel vector with a sliding window out of LDS.  Workgroups are persistent over spatial
// tiles so the per-channel sum / sum-of-squares for the following training-mode BatchNorm leave the kernel
// as a small [workgroups][2][C] partial buffer (deterministic, no atomics).
#include "common.cuh"
#include "../../include/mammoclip_hip.h"

namespace {

constexpr int TC = 64;              // channels per workgroup tile
constexpr int PIXB = TC * 2 + 16;   // LDS bytes per staged pixel (padded)
constexpr int TOW = 16;

template <int K, int S> struct DwCfg {
    static constexpr int TOH = (S == 1) ? 8 : 4;
    static constexpr int R = 2;                           // output pixels per strip (along W)
    static constexpr int NSTRIP = TOW / R;
    static constexpr int PASSES = TOH * NSTRIP * 8 / 256; // strips per thread
    static constexpr int IH_T = (TOH - 1) * S + K;
    static constexpr int IW_T = (TOW - 1) * S + K;
    static constexpr int NIN = (R - 1) * S + K;           // input vectors per filter row per strip
    static constexpr int TILE_BYTES = IH_T * IW_T * PIXB;
    static constexpr int W_BYTES = K * K * TC * 4;
};

// halo tile staging, split so the global loads of the NEXT tile can be in flight while the current tile computes:
// load_halo issues all 16-byte loads of the (IH_T x IW_T x 64-channel) tile back to back into registers,
// store_halo applies the optional BN+SiLU prologue (in-range pixels only: zero padding stays zero) and writes LDS.
template <int K, int S> struct Halo {
    static constexpr int NV = (DwCfg<K, S>::IH_T * DwCfg<K, S>::IW_T + 31) / 32;
    uint4 vals[NV];
    unsigned inb;
};

template <int K, int S>
__device__ __forceinline__ void load_halo(const mc_dwconv_args& p, Halo<K, S>& hl, long long img, int oy0, int ox0, int c0) {
    using C = DwCfg<K, S>;
    const int tid = threadIdx.x;
    const int c = c0 + (tid & 7) * 8;
    const int iy0 = oy0 * S - p.pad_t, ix0 = ox0 * S - p.pad_l;
    hl.inb = 0;
#pragma unroll
    for (int i = 0; i < Halo<K, S>::NV; ++i) {
        int v = (tid >> 3) + i * 32;
        int ty = v / C::IW_T, tx = v % C::IW_T;
        int iy = iy0 + ty, ix = ix0 + tx;
        hl.vals[i] = make_uint4(0u, 0u, 0u, 0u);
        if (v < C::IH_T * C::IW_T && c < p.c && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w) {
            hl.vals[i] = *reinterpret_cast<const uint4*>(p.x + ((img * p.h + iy) * (long long)p.w + ix) * p.c + c);
            hl.inb |= 1u << i;
        }
    }
}

template <int K, int S>
__device__ __forceinline__ void store_halo(unsigned char* tile, const Halo<K, S>& hl, const float* ps, const float* pt,
                                           bool has_pro) {
    using C = DwCfg<K, S>;
    const int tid = threadIdx.x;
    const int cv = tid & 7;
#pragma unroll
    for (int i = 0; i < Halo<K, S>::NV; ++i) {
        int v = (tid >> 3) + i * 32;
        if (v < C::IH_T * C::IW_T) {
            uint4 val = hl.vals[i];
            if (has_pro && ((hl.inb >> i) & 1u)) {
                float f[8];
                unpack8(val, f);
#pragma unroll
                for (int q = 0; q < 8; ++q) f[q] = silu_f(f[q] * ps[q] + pt[q]);
                val = pack8(f);
            }
            *reinterpret_cast<uint4*>(tile + v * PIXB + cv * 16) = val;
        }
    }
}

template <int K, int S>
__device__ __forceinline__ void stage_input(const mc_dwconv_args& p, unsigned char* tile, long long img, int oy0,
                                            int ox0, int c0, const float* ps, const float* pt, bool has_pro) {
    Halo<K, S> hl;
    load_halo<K, S>(p, hl, img, oy0, ox0, c0);
    store_halo<K, S>(tile, hl, ps, pt, has_pro);
}

template <int K, int S>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const mc_dwconv_args p, int tiles_h, int tiles_w) {
    using C = DwCfg<K, S>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* tile = smem;
    float* wl = reinterpret_cast<float*>(smem + C::TILE_BYTES);       // [K*K][TC]
    const int tid = threadIdx.x;
    const int cv = tid & 7;
    const int c0 = blockIdx.x * TC;
    const int c = c0 + cv * 8;
    const bool cvalid = c < p.c;
    const bool has_pro = p.pro_scale != nullptr;

    for (int i = tid; i < K * K * TC; i += 256) {
        int tap = i / TC, ch = i % TC;
        wl[i] = (c0 + ch < p.c) ? p.w_kkc[(long long)tap * p.c + c0 + ch] : 0.f;
    }
    float ps[8], pt[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { ps[q] = 1.f; pt[q] = 0.f; }
    if (has_pro && cvalid) { load8f(p.pro_scale + c, ps); load8f(p.pro_shift + c, pt); }

    float ssum[8], ssq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { ssum[q] = 0.f; ssq[q] = 0.f; }

    const long long ntiles = (long long)p.n * tiles_h * tiles_w;
    auto tile_pos = [&](long long t, long long& img, int& oy0, int& ox0) {
        ox0 = (int)(t % tiles_w) * TOW;
        oy0 = (int)((t / tiles_w) % tiles_h) * C::TOH;
        img = t / ((long long)tiles_w * tiles_h);
    };
    Halo<K, S> hl;
    long long t = blockIdx.y;
    {
        long long img; int oy0, ox0;
        if (t < ntiles) { tile_pos(t, img, oy0, ox0); load_halo<K, S>(p, hl, img, oy0, ox0, c0); }
    }
    for (; t < ntiles; t += gridDim.y) {
        long long img; int oy0, ox0;
        tile_pos(t, img, oy0, ox0);
        __syncthreads();                       // previous tile fully consumed (also orders the weight staging)
        store_halo<K, S>(tile, hl, ps, pt, has_pro);
        __syncthreads();
        if (t + gridDim.y < ntiles) {          // prefetch the next tile while this one computes
            long long img2; int oy2, ox2;
            tile_pos(t + gridDim.y, img2, oy2, ox2);
            load_halo<K, S>(p, hl, img2, oy2, ox2, c0);
        }
#pragma unroll 1
        for (int pass = 0; pass < C::PASSES; ++pass) {
            const int item = tid + pass * 256;
            const int strip = (item >> 3) % C::NSTRIP;
            const int orow = item / (8 * C::NSTRIP);
            float acc[C::R][8];
#pragma unroll
            for (int r = 0; r < C::R; ++r)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[r][q] = 0.f;
#pragma unroll 1
            for (int kh = 0; kh < K; ++kh) {          // k = 5: keep one filter row live at a time (VGPR budget)
                float in[C::NIN][8];
                const unsigned char* rowp = tile + ((orow * S + kh) * C::IW_T + strip * C::R * S) * PIXB + cv * 16;
#pragma unroll
                for (int i = 0; i < C::NIN; ++i) unpack8(*reinterpret_cast<const uint4*>(rowp + i * PIXB), in[i]);
#pragma unroll
                for (int kw = 0; kw < K; ++kw) {
                    float wv[8];
                    load8f(wl + (kh * K + kw) * TC + cv * 8, wv);
#pragma unroll
                    for (int r = 0; r < C::R; ++r)
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc[r][q] = fmaf(in[r * S + kw][q], wv[q], acc[r][q]);
                }
            }
            const int oy = oy0 + orow;
            if (cvalid && oy < p.oh) {
                bf16_t* yrow = reinterpret_cast<bf16_t*>(p.out) + ((img * p.oh + oy) * (long long)p.ow) * p.c + c;
#pragma unroll
                for (int r = 0; r < C::R; ++r) {
                    int ox = ox0 + strip * C::R + r;
                    if (ox < p.ow) {
                        uint4 o = pack8(acc[r]);
                        *reinterpret_cast<uint4*>(yrow + (long long)ox * p.c) = o;
                        if (p.stat_partials) {
                            float f[8];
                            unpack8(o, f);          // statistics of the stored (bf16-rounded) tensor
#pragma unroll
                            for (int q = 0; q < 8; ++q) { ssum[q] += f[q]; ssq[q] += f[q] * f[q]; }
                        }
                    }
                }
            }
        }
    }

    if (p.stat_partials) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);      // [32 groups][8 cv][16]
        const int grp = tid >> 3;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            red[(grp * 8 + cv) * 16 + q] = ssum[q];
            red[(grp * 8 + cv) * 16 + 8 + q] = ssq[q];
        }
        __syncthreads();
        if (tid < 128) {
            int ch = tid & 63, which = tid >> 6;          // 0 = sum, 1 = sumsq
            float s = 0.f;
            for (int g = 0; g < 32; ++g) s += red[(g * 8 + (ch >> 3)) * 16 + which * 8 + (ch & 7)];
            if (c0 + ch < p.c) p.stat_partials[((long long)blockIdx.y * 2 + which) * p.c + c0 + ch] = s;
        }
    }
}

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

// weight gradient: dw[kh,kw,c] += sum_{n,oy,ox} dy[n,oy,ox,c] * x'[n, oy*S+kh-pt, ox*S+kw-pl, c]
// thread = (4-channel vector, pixel group); all K*K taps accumulate in registers across the persistent tile loop
template <int K, int S>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const mc_dwconv_args p, int tiles_h, int tiles_w) {
    using C = DwCfg<K, S>;
    constexpr int RW = (S == 1) ? 8 : 4;                 // outputs per thread along W
    constexpr int NSTR = TOW / RW;
    constexpr int NINW = (RW - 1) * S + K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* tile = smem;
    const int tid = threadIdx.x;
    const int cq = tid & 15;                              // 4-channel vector index inside the 64-channel tile
    const int pg = tid >> 4;                              // 16 pixel groups
    const int strip = pg % NSTR, orow = pg / NSTR;
    const int c0 = blockIdx.x * TC;
    const int c = c0 + cq * 4;
    const bool cvalid = c < p.c;
    const bool has_pro = p.pro_scale != nullptr;
    const int cv = tid & 7;
    float ps[8], pt[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { ps[q] = 1.f; pt[q] = 0.f; }
    if (has_pro && c0 + cv * 8 < p.c) { load8f(p.pro_scale + c0 + cv * 8, ps); load8f(p.pro_shift + c0 + cv * 8, pt); }

    float acc[K * K][4];
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = 0.f;

    const long long ntiles = (long long)p.n * tiles_h * tiles_w;
    auto tile_pos = [&](long long t, long long& img, int& oy0, int& ox0) {
        ox0 = (int)(t % tiles_w) * TOW;
        oy0 = (int)((t / tiles_w) % tiles_h) * C::TOH;
        img = t / ((long long)tiles_w * tiles_h);
    };
    Halo<K, S> hl;
    long long t = blockIdx.y;
    {
        long long img; int oy0, ox0;
        if (t < ntiles) { tile_pos(t, img, oy0, ox0); load_halo<K, S>(p, hl, img, oy0, ox0, c0); }
    }
    for (; t < ntiles; t += gridDim.y) {
        long long img; int oy0, ox0;
        tile_pos(t, img, oy0, ox0);
        __syncthreads();
        store_halo<K, S>(tile, hl, ps, pt, has_pro);
        __syncthreads();
        if (t + gridDim.y < ntiles) {          // prefetch the next halo tile while this one is consumed
            long long img2; int oy2, ox2;
            tile_pos(t + gridDim.y, img2, oy2, ox2);
            load_halo<K, S>(p, hl, img2, oy2, ox2, c0);
        }
        const int oy = oy0 + orow;
        float g[RW][4];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            int ox = ox0 + strip * RW + r;
            uint2 gv = make_uint2(0u, 0u);
            if (cvalid && oy < p.oh && ox < p.ow)
                gv = *reinterpret_cast<const uint2*>(p.dy + ((img * p.oh + oy) * (long long)p.ow + ox) * p.c + c);
            g[r][0] = bf_lo(gv.x); g[r][1] = bf_hi(gv.x); g[r][2] = bf_lo(gv.y); g[r][3] = bf_hi(gv.y);
        }
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {              // must stay unrolled: acc[] is indexed by kh (registers, not scratch)
            float in[NINW][4];
            const unsigned char* rowp = tile + ((orow * S + kh) * C::IW_T + strip * RW * S) * PIXB + cq * 8;
#pragma unroll
            for (int i = 0; i < NINW; ++i) {
                uint2 v = *reinterpret_cast<const uint2*>(rowp + i * PIXB);
                in[i][0] = bf_lo(v.x); in[i][1] = bf_hi(v.x); in[i][2] = bf_lo(v.y); in[i][3] = bf_hi(v.y);
            }
#pragma unroll
            for (int kw = 0; kw < K; ++kw)
#pragma unroll
                for (int r = 0; r < RW; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[kh * K + kw][q] = fmaf(g[r][q], in[r * S + kw][q], acc[kh * K + kw][q]);
        }
    }
    // reduce the 16 pixel groups per tap through LDS, one atomic per (tap, channel) per workgroup
    float* red = reinterpret_cast<float*>(smem);          // [16 pg][64 ch]
    for (int tap = 0; tap < K * K; ++tap) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) red[pg * 64 + cq * 4 + q] = acc[tap][q];
        __syncthreads();
        if (tid < 64 && c0 + tid < p.c) {
            float s = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < 16; ++g2) s += red[g2 * 64 + tid];
            atomicAdd(reinterpret_cast<float*>(p.out) + (long long)tap * p.c + c0 + tid, s);
        }
    }
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

template <int K, int S> int tiles_of(const mc_dwconv_args& p, int* th, int* tw) {
    *th = mc_div_up(p.oh, DwCfg<K, S>::TOH);
    *tw = mc_div_up(p.ow, TOW);
    return 0;
}
int grid_y_for(const mc_dwconv_args& p, long long ntiles) {
    int ctiles = mc_div_up(p.c, TC);
    long long want = 2048 / ctiles;
    if (want < 64) want = 64;
    if (want > 1024) want = 1024;
    return (int)(ntiles < want ? ntiles : want);
}

template <int K, int S> int launch_fwd(const mc_dwconv_args& p, hipStream_t st) {
    using C = DwCfg<K, S>;
    int th, tw;
    tiles_of<K, S>(p, &th, &tw);
    long long ntiles = (long long)p.n * th * tw;
    dim3 grid(mc_div_up(p.c, TC), grid_y_for(p, ntiles));
    size_t lds = C::TILE_BYTES + C::W_BYTES;
    if (lds < 32 * 8 * 16 * 4) lds = 32 * 8 * 16 * 4;
    hipLaunchKernelGGL((dwconv_fwd_kernel<K, S>), grid, dim3(256), lds, st, p, th, tw);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
template <int K, int S> int launch_bww(const mc_dwconv_args& p, hipStream_t st) {
    using C = DwCfg<K, S>;
    int th, tw;
    tiles_of<K, S>(p, &th, &tw);
    long long ntiles = (long long)p.n * th * tw;
    dim3 grid(mc_div_up(p.c, TC), grid_y_for(p, ntiles));
    size_t lds = C::TILE_BYTES;
    if (lds < 16 * 64 * 4) lds = 16 * 64 * 4;
    hipLaunchKernelGGL((dwconv_bwd_weight_kernel<K, S>), grid, dim3(256), lds, st, p, th, tw);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

}  // namespace

extern "C" int mc_dwconv_stat_rows(const mc_dwconv_args* a) {
    int th = mc_div_up(a->oh, a->stride == 1 ? 8 : 4), tw = mc_div_up(a->ow, TOW);
    return grid_y_for(*a, (long long)a->n * th * tw);
}

extern "C" int mc_dwconv_fwd(const mc_dwconv_args* a, void* stream) {
    const mc_dwconv_args& p = *a;
    if (int e = check_common(p)) return e;
    MC_CHECK(p.x && p.w_kkc, "dwconv_fwd: null x / w");
    hipStream_t st = (hipStream_t)stream;
    if (p.k == 3 && p.stride == 1) return launch_fwd<3, 1>(p, st);
    if (p.k == 3 && p.stride == 2) return launch_fwd<3, 2>(p, st);
    if (p.k == 5 && p.stride == 1) return launch_fwd<5, 1>(p, st);
    return launch_fwd<5, 2>(p, st);
}

extern "C" int mc_dwconv_bwd_data(const mc_dwconv_args* a, void* stream) {
    const mc_dwconv_args& p = *a;
    if (int e = check_common(p)) return e;
    MC_CHECK(p.dy && p.w_kkc, "dwconv_bwd_data: null dy / w");
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
    if (p.k == 3 && p.stride == 1) return launch_bww<3, 1>(p, st);
    if (p.k == 3 && p.stride == 2) return launch_bww<3, 2>(p, st);
    if (p.k == 5 && p.stride == 1) return launch_bww<5, 1>(p, st);
    return launch_bww<5, 2>(p, st);
}
