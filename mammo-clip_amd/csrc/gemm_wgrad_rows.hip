// Streaming weight-gradient GEMM for the HBM-bound pointwise convolutions:
//     dW[N,K] (fp32) (+)= sum_m dY[m,N]^T . pro(X)[m,K],   N,K <= 384 with N+K <= 448,  m = pixels (millions)
// [ref: autograd backward of efficientnet_custom.py:104 (_expand_conv) / :122 (_project_conv)]
//
// The reduction runs over the pixel index, i.e. over the ROWS of both operands, so both MFMA operand fragments are
// "transposed" reads.  gfx950 has exactly the instruction for that: ds_read_b64_tr_b16 (verified on hardware: within
// a 16-lane group lane i supplies the address of row i/4, cols (i%4)*4..+3 of a 4x16 block; lane c receives column c,
// rows 0..3).  So dY and X tiles are staged ROW-MAJOR in LDS with plain coalesced 16-byte loads/stores (the fused
// BN+SiLU(+SE gate) prologue is applied to X on the way in) and every fragment is two transpose-reads -- no scalar
// LDS transposes, no atomics: each persistent workgroup keeps its N x K partial in MFMA accumulators (the 16x16
// output tiles are split over the 4 waves) and writes it once to a workspace that a tiny second kernel reduces.
#include "common.cuh"
#include "../../include/mammoclip_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s4_t;
typedef __attribute__((address_space(3))) s4_t lds_s4_t;

constexpr int MAXCH = 14;      // 16-byte chunks per thread per step: RB * (N + K) / 8 / 256 <= 14
// RB = pixel rows per step (64 .. 512, runtime): narrow layers take more rows per step so that every step moves
// ~50 KB -- enough bytes in flight per workgroup to cover HBM latency with a one-step prefetch.
static int pick_rb(int n, int k) {
    int rb = 512;
    while (rb > 64 && (long long)rb * (n + k) > 28672) rb >>= 1;
    return rb;
}

__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* tile, int rs, int row0, int col0, int lane) {
    // 16x16x32 MFMA operand fragment for reduction rows row0 .. row0+31 and 16 columns col0 .. col0+15:
    // lane (i = l&15, g = l>>4) needs rows row0 + g*8 + 0..7 of column col0 + i
    const int g = lane >> 4, i = lane & 15;
    const unsigned char* a = tile + (size_t)(row0 + g * 8 + (i >> 2)) * rs + (col0 + (i & 3) * 4) * 2;
    s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(a));
    s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(a + 4 * rs));
    typedef __attribute__((ext_vector_type(8))) short s8_t;
    s8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

template <int AFM, int BFM>
__global__ __launch_bounds__(256) void wgrad_rows_kernel(const mc_wgrad_rows_args p, int WB, int af, int bfn, int RB) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int rsy = p.N * 2, rsx = p.K * 2;
    unsigned char* sY = smem;
    unsigned char* sX = smem + RB * rsy;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave / WB, wb = wave % WB;
    const int ny8 = p.N >> 3, nx8 = p.K >> 3;
    const int chY = RB * ny8, chT = chY + RB * nx8;
    const bool has_pro = p.pro_scale != nullptr;
    const int fna = (p.N + 15) >> 4, fkb = (p.K + 15) >> 4;

    f32x4_t acc[AFM][BFM];
#pragma unroll
    for (int i = 0; i < AFM; ++i)
#pragma unroll
        for (int j = 0; j < BFM; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    uint4 regs[MAXCH];
    auto load_step = [&](long long s) {
        const long long m0 = s * RB;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            int c = tid + i * 256;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (c < chY) {
                int row = c / ny8, cc = c - row * ny8;
                if (m0 + row < p.M) v = *reinterpret_cast<const uint4*>(p.dY + (m0 + row) * p.lddy + cc * 8);
            } else if (c < chT) {
                int c2 = c - chY;
                int row = c2 / nx8, cc = c2 - row * nx8;
                if (m0 + row < p.M) v = *reinterpret_cast<const uint4*>(p.X + (m0 + row) * p.ldx + cc * 8);
            }
            regs[i] = v;
        }
    };
    auto store_step = [&](long long s) {
        const long long m0 = s * RB;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            int c = tid + i * 256;
            if (c < chY) {
                int row = c / ny8, cc = c - row * ny8;
                *reinterpret_cast<uint4*>(sY + row * rsy + cc * 16) = regs[i];
            } else if (c < chT) {
                int c2 = c - chY;
                int row = c2 / nx8, cc = c2 - row * nx8;
                uint4 v = regs[i];
                if (has_pro && m0 + row < p.M) {
                    float f[8], sc[8], sh[8];
                    unpack8(v, f);
                    load8f(p.pro_scale + cc * 8, sc);
                    load8f(p.pro_shift + cc * 8, sh);
#pragma unroll
                    for (int q = 0; q < 8; ++q) f[q] = silu_f(f[q] * sc[q] + sh[q]);
                    if (p.pro_gate) {
                        float gv[8];
                        load8f(p.pro_gate + ((m0 + row) / p.pro_rows_per_img) * p.K + cc * 8, gv);
#pragma unroll
                        for (int q = 0; q < 8; ++q) f[q] *= gv[q];
                    }
                    v = pack8(f);
                }
                *reinterpret_cast<uint4*>(sX + row * rsx + cc * 16) = v;
            }
        }
    };

    const long long nsteps = (p.M + RB - 1) / RB;
    long long s = blockIdx.x;
    if (s < nsteps) load_step(s);
    for (; s < nsteps; s += gridDim.x) {
        __syncthreads();                       // previous step's fragment reads are done
        store_step(s);
        __syncthreads();
        if (s + gridDim.x < nsteps) load_step(s + gridDim.x);
        for (int ks = 0; ks < RB / 32; ++ks) {
            bf16x8_t a[AFM], b[BFM];
#pragma unroll
            for (int i = 0; i < AFM; ++i) {
                int fa = wa * af + i;
                if (i < af && fa < fna) a[i] = tr_frag(sY, rsy, ks * 32, fa * 16, lane);
            }
#pragma unroll
            for (int j = 0; j < BFM; ++j) {
                int fb = wb * bfn + j;
                if (j < bfn && fb < fkb) b[j] = tr_frag(sX, rsx, ks * 32, fb * 16, lane);
            }
#pragma unroll
            for (int i = 0; i < AFM; ++i)
#pragma unroll
                for (int j = 0; j < BFM; ++j)
                    if (i < af && j < bfn && wa * af + i < fna && wb * bfn + j < fkb)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // partial [N][K] of this workgroup -> workspace.  D[i][j]: i = n (A operand), j = k (B operand)
    float* W = p.ws + (long long)blockIdx.x * p.N * p.K;
#pragma unroll
    for (int i = 0; i < AFM; ++i)
#pragma unroll
        for (int j = 0; j < BFM; ++j) {
            int fa = wa * af + i, fb = wb * bfn + j;
            if (i < af && j < bfn && fa < fna && fb < fkb) {
                int k = fb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int n = fa * 16 + (lane >> 4) * 4 + r;
                    if (n < p.N && k < p.K) W[(long long)n * p.K + k] = acc[i][j][r];
                }
            }
        }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, int parts, long long nk, float* __restrict__ out,
                                    int accumulate) {
    // 256 threads = 16 part-lanes x 16 elements (64-byte coalesced reads), part-lanes combined through LDS
    __shared__ float sh[16][17];
    const int el = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const long long i = (long long)blockIdx.x * 16 + el;
    float s = 0.f;
    if (i < nk)
        for (int k = pl; k < parts; k += 16) s += ws[(long long)k * nk + i];
    sh[pl][el] = s;
    __syncthreads();
    if (pl != 0 || i >= nk) return;
    s = 0.f;
    for (int k = 0; k < 16; ++k) s += sh[k][el];
    out[i] = accumulate ? out[i] + s : s;
}

}  // namespace

static void wave_split(int n, int k, int* WA, int* WB, int* af, int* bfn) {
    const int fna = (n + 15) / 16, fkb = (k + 15) / 16;
    if (fna >= fkb) { *WA = fna >= 4 ? 4 : (fna >= 2 ? 2 : 1); *WB = 4 / *WA; }
    else { *WB = fkb >= 4 ? 4 : (fkb >= 2 ? 2 : 1); *WA = 4 / *WB; }
    *af = (fna + *WA - 1) / *WA;
    *bfn = (fkb + *WB - 1) / *WB;
}
extern "C" int mc_wgrad_rows_supported(int n, int k) {
    if (n <= 0 || k <= 0 || n % 8 || k % 8 || n > 384 || k > 384 || n + k > 448) return 0;
    int WA, WB, af, bfn;
    wave_split(n, k, &WA, &WB, &af, &bfn);
    return (af <= 4 && bfn <= 4) || (af <= 6 && bfn <= 4) || (af <= 4 && bfn <= 6);
}
extern "C" int mc_wgrad_rows_blocks(long long m) {
    long long steps = (m + 63) / 64;
    long long b = steps < 512 ? steps : 512;
    return (int)(b < 1 ? 1 : b);
}

extern "C" int mc_wgrad_rows_bf16(const mc_wgrad_rows_args* a, void* stream) {
    const mc_wgrad_rows_args& p = *a;
    MC_CHECK(p.dY && p.X && p.dW && p.ws && p.M > 0, "wgrad_rows: bad args");
    MC_CHECK(mc_wgrad_rows_supported(p.N, p.K), "wgrad_rows: unsupported shape (see mc_wgrad_rows_supported)");
    MC_CHECK(p.lddy % 8 == 0 && p.ldx % 8 == 0 && mc_aligned16(p.dY) && mc_aligned16(p.X), "wgrad_rows: alignment");
    MC_CHECK((p.pro_scale == nullptr) == (p.pro_shift == nullptr), "wgrad_rows: prologue needs scale and shift");
    MC_CHECK(!p.pro_gate || (p.pro_scale && p.pro_rows_per_img > 0), "wgrad_rows: gate needs the BN prologue");
    hipStream_t st = (hipStream_t)stream;
    int WA, WB, af, bfn;
    wave_split(p.N, p.K, &WA, &WB, &af, &bfn);
    const int RB = pick_rb(p.N, p.K);
    long long steps = (p.M + RB - 1) / RB;
    const int blocks = (int)(steps < 512 ? steps : 512);      // <= mc_wgrad_rows_blocks(M): ws is large enough
    const size_t lds = (size_t)RB * (p.N + p.K) * 2 + 64;     // + slack: the last fragment may over-read 16 B
    if (af <= 4 && bfn <= 4) hipLaunchKernelGGL((wgrad_rows_kernel<4, 4>), dim3(blocks), dim3(256), lds, st, p, WB, af, bfn, RB);
    else if (af <= 6 && bfn <= 4) hipLaunchKernelGGL((wgrad_rows_kernel<6, 4>), dim3(blocks), dim3(256), lds, st, p, WB, af, bfn, RB);
    else hipLaunchKernelGGL((wgrad_rows_kernel<4, 6>), dim3(blocks), dim3(256), lds, st, p, WB, af, bfn, RB);
    MC_LAUNCH_CHECK();
    long long nk = (long long)p.N * p.K;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(mc_div_up(nk, 16)), dim3(256), 0, st, p.ws, blocks, nk, p.dW, p.accumulate);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
