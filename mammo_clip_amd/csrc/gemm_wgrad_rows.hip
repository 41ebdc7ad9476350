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
#include "common_hip.h"
#include "../../include/mammoclip_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s4_t;
typedef __attribute__((address_space(3))) s4_t lds_s4_t;

// RB = pixel rows per step (multiple of 32, runtime).  A step is prefetched ONE step ahead through registers, so the
// bytes in flight per workgroup = the step size: NCH 16-byte chunks per thread.  Shapes with few output fragments
// per wave have the registers for 16 chunks (64 KB steps, 128 KB in flight per CU), the others take 10 (40 KB).
static int pick_nch(int af, int bfn) { return af * bfn <= 8 ? 16 : 10; }
static int pick_rb(int n, int k, int nch) {
    int rb = (nch * 256 * 8 / (n + k)) / 32 * 32;
    if (rb > 1024) rb = 1024;
    while (rb > 32 && (rb * (n / 8) + 255) / 256 + (rb * (k / 8) + 255) / 256 > nch) rb -= 32;   // whole slots per tensor
    if (rb < 32) rb = 32;
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

// AF x BF = 16x16 output fragments per wave (exact: the host picks the instantiation), WB = waves along K.
// Per-thread chunk geometry (global offset, LDS offset, row) is constant over the steps and lives in registers: a
// step costs a few instructions per 16-byte chunk instead of two runtime divisions.
// NCHY_T > 0 (round 5, the hot shapes): the number of dY slots per thread as a compile-time constant -- "which tensor does slot i
// belong to" then folds away (with the runtime count the per-slot scalar base / pitch selections cost ~120 SGPR spill
// instructions per step, see xbwd_rows_kernel below)
template <int AF, int BF, int MAXCH, int NCHY_T = 0>
__global__ __launch_bounds__(256, 2) void wgrad_rows_kernel(const mc_wgrad_rows_args p, int WB, int RB, int nch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int rsy = p.N * 2, rsx = p.K * 2;
    unsigned char* sY = smem;
    unsigned char* sX = smem + RB * rsy;
    float* spro = reinterpret_cast<float*>(smem + RB * (rsy + rsx) + 32 * 16 * 2 * 6 + 64);   // [2][K] prologue scale / shift
    float* sgate = spro + 2 * p.K;                         // [2][K] SE gate row of the current / next step's image
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave / WB, wb = wave % WB;
    const int ny8 = p.N >> 3, nx8 = p.K >> 3;
    const int chY = RB * ny8, chT = chY + RB * nx8;
    const bool has_pro = p.pro_scale != nullptr;

    f32x4_t acc[AF][BF];
#pragma unroll
    for (int i = 0; i < AF; ++i)
#pragma unroll
        for (int j = 0; j < BF; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (has_pro)                                           // staged in LDS: re-read per step, not held in 160 registers
        for (int i = tid; i < 2 * p.K; i += 256) spro[i] = i < p.K ? p.pro_scale[i] : p.pro_shift[i - p.K];

    // chunk slots: the first nchy slots of every thread belong to dY, the rest to X (uniform per slot, so the base
    // pointer of a slot is scalar and the per-lane part of an address is one 32-bit offset).  meta = row | chunk << 12.
    const int nchy = NCHY_T > 0 ? NCHY_T : (chY + 255) >> 8;
    unsigned meta[MAXCH];
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        meta[i] = 0xfffu;                                  // row 4095: never valid
        if (i < nchy) {
            const int c = tid + i * 256;
            if (c < chY) { const int row = c / ny8; meta[i] = (unsigned)row | ((unsigned)(c - row * ny8) << 12); }
        } else if (i < nch) {
            const int c = tid + (i - nchy) * 256;
            if (c < RB * nx8) { const int row = c / nx8; meta[i] = (unsigned)row | ((unsigned)(c - row * nx8) << 12); }
        }
    }
    // steps never straddle an image when there is a gate (one gate row per step, staged through LDS one step ahead)
    const long long rpi = p.pro_gate ? p.pro_rows_per_img : p.M;
    const long long spi = (rpi + RB - 1) / RB;
    const long long rem = p.M % rpi;                       // a partial last image is allowed
    const long long nsteps = (p.M / rpi) * spi + (rem + RB - 1) / RB;
    auto step_geom = [&](long long s, long long& m0, int& rows, long long& img) {
        img = s / spi;
        const long long ls = s - img * spi;
        m0 = img * rpi + ls * RB;
        const long long left = (p.M - img * rpi < rpi ? p.M - img * rpi : rpi) - ls * RB;
        rows = (int)(left < RB ? left : RB);
    };
    uint4 regs[MAXCH];
    float4 greg = make_float4(1.f, 1.f, 1.f, 1.f);
    auto load_step = [&](long long s) {
        long long m0, img; int rows;
        step_geom(s, m0, rows, img);
        if (p.pro_gate && tid * 4 < p.K) greg = *reinterpret_cast<const float4*>(p.pro_gate + img * p.K + tid * 4);
        const char* yb = reinterpret_cast<const char*>(p.dY + m0 * p.lddy);
        const char* xb = reinterpret_cast<const char*>(p.X + m0 * p.ldx);
        const unsigned ldyb = (unsigned)p.lddy * 2u, ldxb = (unsigned)p.ldx * 2u;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            if (i < nch) {                                 // unconditional loads (rows past the end re-read row 0, zeroed at the store)
                const unsigned row = meta[i] & 0xfffu, cc = meta[i] >> 12;
                const unsigned r = (int)row < rows ? row : 0u;
                const unsigned off = r * (i < nchy ? ldyb : ldxb) + cc * 16u;
                regs[i] = *reinterpret_cast<const uint4*>((i < nchy ? yb : xb) + off);
            }
        }
    };
    auto store_step = [&](long long s, int par) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0) (keeps the compiler from draining the NEXT step's loads early)
        long long m0, img; int rows;
        step_geom(s, m0, rows, img);
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            if (i < nch && (meta[i] & 0xfffu) != 0xfffu) {
                const int row = (int)(meta[i] & 0xfffu), cc8 = (int)(meta[i] >> 12) * 8;
                const bool isx = i >= nchy;
                uint4 v = row < rows ? regs[i] : make_uint4(0u, 0u, 0u, 0u);
                if (has_pro && isx && row < rows) {
                    float f[8], sc[8], sh[8];
                    unpack8(v, f);
                    load8f(spro + cc8, sc);
                    load8f(spro + p.K + cc8, sh);
                    bn_silu8(f, sc, sh);
                    if (p.pro_gate) {
                        float gv[8];
                        load8f(sgate + par * p.K + cc8, gv);
#pragma unroll
                        for (int q = 0; q < 8; ++q) f[q] *= gv[q];
                    }
                    v = pack8(f);
                }
                *reinterpret_cast<uint4*>((isx ? sX + row * rsx : sY + row * rsy) + cc8 * 2) = v;
            }
        }
    };

    long long s = blockIdx.x;
    int par = 0;
    if (s < nsteps) {
        load_step(s);
        if (p.pro_gate && tid * 4 < p.K) *reinterpret_cast<float4*>(sgate + tid * 4) = greg;
    }
    for (; s < nsteps; s += gridDim.x, par ^= 1) {
        __syncthreads();                       // previous step's fragment reads are done; gate row of this step is in LDS
        store_step(s, par);
        __syncthreads();
        if (s + gridDim.x < nsteps) load_step(s + gridDim.x);
        // fragments past N / K read stale LDS (inside the allocation) into accumulators that are never written out
        for (int ks = 0; ks < RB / 32; ++ks) {
            bf16x8_t a[AF], b[BF];
#pragma unroll
            for (int i = 0; i < AF; ++i) a[i] = tr_frag(sY, rsy, ks * 32, (wa * AF + i) * 16, lane);
#pragma unroll
            for (int j = 0; j < BF; ++j) b[j] = tr_frag(sX, rsx, ks * 32, (wb * BF + j) * 16, lane);
#pragma unroll
            for (int i = 0; i < AF; ++i)
#pragma unroll
                for (int j = 0; j < BF; ++j) acc[i][j] = MC_MFMA_16x16x32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (p.pro_gate && s + gridDim.x < nsteps && tid * 4 < p.K)      // next step's gate row -> the other LDS slot
            *reinterpret_cast<float4*>(sgate + (par ^ 1) * p.K + tid * 4) = greg;
    }
    // partial [N][K] of this workgroup -> workspace.  D[i][j]: i = n (A operand), j = k (B operand)
    float* W = p.ws + (long long)blockIdx.x * p.N * p.K;
#pragma unroll
    for (int i = 0; i < AF; ++i)
#pragma unroll
        for (int j = 0; j < BF; ++j) {
            const int k = (wb * BF + j) * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = (wa * AF + i) * 16 + (lane >> 4) * 4 + r;
                if (n < p.N && k < p.K) W[(long long)n * p.K + k] = acc[i][j][r];
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
    if (i < nk) {
        int k = pl;
        for (; k + 48 < parts; k += 64) {                     // four independent loads in flight, summed in part order
            const float a = ws[(long long)k * nk + i], b = ws[(long long)(k + 16) * nk + i];
            const float c = ws[(long long)(k + 32) * nk + i], d = ws[(long long)(k + 48) * nk + i];
            s += a; s += b; s += c; s += d;
        }
        for (; k < parts; k += 16) s += ws[(long long)k * nk + i];
    }
    sh[pl][el] = s;
    __syncthreads();
    if (pl != 0 || i >= nk) return;
    s = 0.f;
    for (int k = 0; k < 16; ++k) s += sh[k][el];
    out[i] = accumulate ? out[i] + s : s;
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 5: BOTH gradients of a 1x1 convolution e = x W^T from ONE pass over the upstream gradient (the expand conv of an MBConv
// block; dY = dL/de resp. the folded BatchNorm form's dZ0, the widest tensor of the block, 6x the block input)
//     dW[N,K] (fp32) (+)= sum_m dY[m,:]^T x[m,:]                 -- exactly wgrad_rows_kernel (no prologue on x)
//     dX[m,K] (bf16)    = dY[m,:] . Wt[K,N]^T (+ R[m,:])         -- the data gradient of mc_gemm_rows_bf16(dY, Wt, R)
// [ref: autograd backward of efficientnet_custom.py:104 (_expand_conv)].  dY is staged once per step for the transpose-reads of
// the weight gradient; the data gradient reads the SAME rows as plain 16-byte MFMA A fragments.  Wave w owns the 16-column
// fragment w of dX (K <= 64: at most four) and keeps its [N x 16] slice of Wt as NKS B fragments in REGISTERS for the whole
// kernel (48 registers at N = 384) -- no weight tile in LDS, two workgroups per CU as before.  The dX tile leaves through a
// small LDS tile as 16-byte rows (+ the residual, staged one step ahead like the operands).
// RF = 16-row fragments per step (RB = 16 RF rows: 64, or 32 where the registers are short), NKS = ceil(N / 32).
struct xbwd_extra {
    const bf16_t* Wt; long long ldwt;       // [K][N] (row = dX column, contiguous along the reduction), the operand of the data gradient
    bf16_t* dX; long long lddx;
    const bf16_t* R; long long ldr;         // optional residual added to dX
};

// NCHY / NCHX = 16-byte chunk slots per thread and step for dY / for x (and as many again for the residual): template constants, so
// that "which tensor does slot i belong to" is decided at compile time -- with runtime slot counts the compiler kept a scalar base /
// pitch selection per slot alive and a third of the step loop's VALU instructions were SGPR spill traffic (236 v_readlane /
// v_writelane of 710, scripts/lane_isa_mix.py-style count)
template <int AF, int BF, int NCHY, int NCHX, int NKS, int RF>
__global__ __launch_bounds__(256, 2) void xbwd_rows_kernel(const mc_wgrad_rows_args p, const xbwd_extra q, int WB, int nchr) {
    constexpr int RB = 16 * RF;
    constexpr int nchy = NCHY, nchx = NCHX, MAXCH = NCHY + 2 * NCHX;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int rsy = p.N * 2, rsx = p.K * 2;
    const int KF = (p.K + 15) >> 4, KO = KF * 16, rso = KO * 2;     // dX tile: KF column fragments, row stride rso bytes
    unsigned char* sY = smem;
    unsigned char* sX = smem + RB * rsy;
    unsigned char* sO = smem + RB * (rsy + rsx) + 32 * 16 * 2 * 6 + 64;             // [RB][KO] bf16 (after the over-read slack)
    unsigned char* sR = sO + RB * rso;                                               // [2][RB][K] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave / WB, wb = wave % WB;
    const int ny8 = p.N >> 3, nx8 = p.K >> 3;
    const int nch = nchy + nchx + nchr;
    const bool has_r = q.R != nullptr;

    f32x4_t acc[AF][BF];
#pragma unroll
    for (int i = 0; i < AF; ++i)
#pragma unroll
        for (int j = 0; j < BF; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // this wave's slice of Wt as MFMA B fragments: lane (i = l & 15, g = l >> 4) holds Wt[wave * 16 + i][ks * 32 + g * 8 .. + 7]
    // (zero beyond K rows / N columns: the A fragments over-read the next LDS row there, finite values)
    bf16x8_t bw[NKS];
    {
        const int col = wave * 16 + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int n0 = ks * 32 + (lane >> 4) * 8;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (wave < KF && col < p.K && n0 < p.N) v = *reinterpret_cast<const uint4*>(q.Wt + (long long)col * q.ldwt + n0);   // (N % 8 == 0: whole chunks)
            bw[ks] = __builtin_bit_cast(bf16x8_t, v);
        }
    }

    // chunk slots: dY slots, then X slots, then residual slots (uniform per slot).  meta = row | chunk << 12
    unsigned meta[MAXCH];
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        meta[i] = 0xfffu;
        int c = -1, per = 1;
        if (i < nchy) { c = tid + i * 256; per = ny8; if (c >= RB * ny8) c = -1; }
        else if (i < nchy + nchx) { c = tid + (i - nchy) * 256; per = nx8; if (c >= RB * nx8) c = -1; }
        else if (i < nch) { c = tid + (i - nchy - nchx) * 256; per = nx8; if (c >= RB * nx8 || !has_r) c = -1; }
        if (c >= 0) { const int row = c / per; meta[i] = (unsigned)row | ((unsigned)(c - row * per) << 12); }
    }
    const long long nsteps = (p.M + RB - 1) / RB;
    uint4 regs[MAXCH];
    auto load_step = [&](long long s) {
        const long long m0 = s * RB;
        const int rows = (int)(p.M - m0 < RB ? p.M - m0 : RB);
        const char* yb = reinterpret_cast<const char*>(p.dY + m0 * p.lddy);
        const char* xb = reinterpret_cast<const char*>(p.X + m0 * p.ldx);
        const char* rb_ = has_r ? reinterpret_cast<const char*>(q.R + m0 * q.ldr) : xb;
        const unsigned ldyb = (unsigned)p.lddy * 2u, ldxb = (unsigned)p.ldx * 2u, ldrb = has_r ? (unsigned)q.ldr * 2u : ldxb;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            if (i < nch) {                                 // unconditional loads (rows past the end re-read row 0, zeroed at the store)
                const unsigned row = meta[i] & 0xfffu, cc = meta[i] >> 12;
                const unsigned r = (int)row < rows ? row : 0u;
                const bool isy = i < nchy, isx = !isy && i < nchy + nchx;
                const unsigned off = r * (isy ? ldyb : (isx ? ldxb : ldrb)) + (row == 0xfffu ? 0u : cc * 16u);
                regs[i] = *reinterpret_cast<const uint4*>((isy ? yb : (isx ? xb : rb_)) + off);
            }
        }
    };
    auto store_step = [&](long long s, int par) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0) (keeps the compiler from draining the NEXT step's loads early)
        const long long m0 = s * RB;
        const int rows = (int)(p.M - m0 < RB ? p.M - m0 : RB);
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            if (i < nch && (meta[i] & 0xfffu) != 0xfffu) {
                const int row = (int)(meta[i] & 0xfffu), cc8 = (int)(meta[i] >> 12) * 8;
                const uint4 v = row < rows ? regs[i] : make_uint4(0u, 0u, 0u, 0u);
                unsigned char* dst = i < nchy ? sY + row * rsy : (i < nchy + nchx ? sX + row * rsx : sR + (par * RB + row) * rsx);
                *reinterpret_cast<uint4*>(dst + cc8 * 2) = v;
            }
        }
    };
    // dX rows of step s (computed into sO during iteration s): + residual, 16-byte rows to global
    auto flush_out = [&](long long s, int par) {
        const long long m0 = s * RB;
        const int rows = (int)(p.M - m0 < RB ? p.M - m0 : RB);
        for (int c = tid; c < RB * nx8; c += 256) {
            const int row = c / nx8, cc8 = (c - row * nx8) * 8;
            if (row >= rows) continue;
            uint4 v = *reinterpret_cast<const uint4*>(sO + row * rso + cc8 * 2);
            if (has_r) {
                const uint4 rv = *reinterpret_cast<const uint4*>(sR + (par * RB + row) * rsx + cc8 * 2);
                float a[8], b[8];
                unpack8(v, a); unpack8(rv, b);
#pragma unroll
                for (int t = 0; t < 8; ++t) a[t] += b[t];
                v = pack8(a);
            }
            *reinterpret_cast<uint4*>(q.dX + (m0 + row) * q.lddx + cc8) = v;
        }
    };

    long long s = blockIdx.x;
    int par = 0;
    long long s_prev = -1;
    if (s < nsteps) load_step(s);
    for (; s < nsteps; s += gridDim.x, par ^= 1) {
        __syncthreads();                       // previous step's fragment reads and its dX tile are complete
        if (s_prev >= 0) flush_out(s_prev, par ^ 1);
        store_step(s, par);
        __syncthreads();
        if (s + gridDim.x < nsteps) load_step(s + gridDim.x);
        // ---- weight gradient: dW += dY^T x  (as wgrad_rows_kernel)
        for (int ks = 0; ks < RB / 32; ++ks) {
            bf16x8_t a[AF], b[BF];
#pragma unroll
            for (int i = 0; i < AF; ++i) a[i] = tr_frag(sY, rsy, ks * 32, (wa * AF + i) * 16, lane);
#pragma unroll
            for (int j = 0; j < BF; ++j) b[j] = tr_frag(sX, rsx, ks * 32, (wb * BF + j) * 16, lane);
#pragma unroll
            for (int i = 0; i < AF; ++i)
#pragma unroll
                for (int j = 0; j < BF; ++j) acc[i][j] = MC_MFMA_16x16x32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        // ---- data gradient: dX[:, wave * 16 .. + 15] = dY . Wt^T
        if (wave < KF) {
            f32x4_t o[RF];
#pragma unroll
            for (int rf = 0; rf < RF; ++rf) o[rf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            const unsigned char* arow = sY + (lane & 15) * rsy + (lane >> 4) * 16;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
                for (int rf = 0; rf < RF; ++rf) {
                    const bf16x8_t af_ = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(arow + rf * 16 * rsy + ks * 64));
                    o[rf] = MC_MFMA_16x16x32(af_, bw[ks], o[rf], 0, 0, 0);
                }
            }
            // C layout: o[rf][r] = dX[rf * 16 + (lane >> 4) * 4 + r][wave * 16 + (lane & 15)]
            bf16_t* so = reinterpret_cast<bf16_t*>(sO);
#pragma unroll
            for (int rf = 0; rf < RF; ++rf)
#pragma unroll
                for (int r = 0; r < 4; ++r) so[(rf * 16 + (lane >> 4) * 4 + r) * KO + wave * 16 + (lane & 15)] = f2bf(o[rf][r]);
        }
        s_prev = s;
    }
    __syncthreads();
    if (s_prev >= 0) flush_out(s_prev, par ^ 1);
    float* W = p.ws + (long long)blockIdx.x * p.N * p.K;
#pragma unroll
    for (int i = 0; i < AF; ++i)
#pragma unroll
        for (int j = 0; j < BF; ++j) {
            const int k = (wb * BF + j) * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = (wa * AF + i) * 16 + (lane >> 4) * 4 + r;
                if (n < p.N && k < p.K) W[(long long)n * p.K + k] = acc[i][j][r];
            }
        }
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
    const int NCH = pick_nch(af, bfn);
    const int RB = pick_rb(p.N, p.K, NCH);
    const long long rpi = p.pro_gate ? p.pro_rows_per_img : p.M;
    long long steps = (p.M / rpi) * ((rpi + RB - 1) / RB) + (p.M % rpi + RB - 1) / RB;
    const long long cap = mc_wgrad_rows_blocks(p.M);          // rows of the caller's workspace
    const int blocks = (int)(steps < cap ? steps : cap);
    // LDS: both tiles + slack so that fragments past N / K (and the last fragment's 16-byte over-read) stay inside
    const size_t lds = (size_t)RB * (p.N + p.K) * 2 + 32 * 16 * 2 * 6 + 64 + (size_t)p.K * 16;
    const int nch = (RB * (p.N / 8) + 255) / 256 + (RB * (p.K / 8) + 255) / 256;   // dY slots + X slots per thread
    MC_CHECK(nch <= NCH, "wgrad_rows: internal: step too large");
    const int nchy_h = (RB * (p.N / 8) + 255) / 256;
    // hot shapes of EfficientNet-B5's projection convs, (c_out, c_exp) = (40, 240) and (64, 384): dY slot count as a template constant
    if (af == 3 && bfn == 4 && nchy_h == 2) { hipLaunchKernelGGL((wgrad_rows_kernel<3, 4, 10, 2>), dim3(blocks), dim3(256), lds, st, p, WB, RB, nch); } else
    if (af == 4 && bfn == 6 && nchy_h == 1) { hipLaunchKernelGGL((wgrad_rows_kernel<4, 6, 10, 1>), dim3(blocks), dim3(256), lds, st, p, WB, RB, nch); } else
#define WG_CASE(A_, B_) if (af == A_ && bfn == B_) { hipLaunchKernelGGL((wgrad_rows_kernel<A_, B_, (A_ * B_ <= 8 ? 16 : 10)>), dim3(blocks), dim3(256), lds, st, p, WB, RB, nch); } else
    WG_CASE(1, 1) WG_CASE(1, 2) WG_CASE(1, 3) WG_CASE(1, 4) WG_CASE(1, 5) WG_CASE(1, 6)
    WG_CASE(2, 1) WG_CASE(2, 2) WG_CASE(2, 3) WG_CASE(2, 4) WG_CASE(2, 5) WG_CASE(2, 6)
    WG_CASE(3, 1) WG_CASE(3, 2) WG_CASE(3, 3) WG_CASE(3, 4) WG_CASE(3, 5) WG_CASE(3, 6)
    WG_CASE(4, 1) WG_CASE(4, 2) WG_CASE(4, 3) WG_CASE(4, 4) WG_CASE(4, 5) WG_CASE(4, 6)
    WG_CASE(5, 1) WG_CASE(5, 2) WG_CASE(5, 3) WG_CASE(5, 4)
    WG_CASE(6, 1) WG_CASE(6, 2) WG_CASE(6, 3) WG_CASE(6, 4)
    { MC_CHECK(false, "wgrad_rows: internal: no instantiation"); }
#undef WG_CASE
    MC_LAUNCH_CHECK();
    long long nk = (long long)p.N * p.K;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(mc_div_up(nk, 16)), dim3(256), 0, st, p.ws, blocks, nk, p.dW, p.accumulate);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

// ---- fused backward (weight + data gradient) of the row-streaming 1x1 convolutions, see xbwd_rows_kernel
struct XbwdCfg { int af, bf, nks, rf; };
static bool xbwd_cfg(int n, int k, XbwdCfg* c) {
    if (!mc_wgrad_rows_supported(n, k) || k > 64) return false;
    int WA, WB;
    wave_split(n, k, &WA, &WB, &c->af, &c->bf);
    c->nks = (n + 31) / 32;
    // rows per step = 16 rf: 32 where the registers are short (N = 384: 96 accumulators + 48 of Wt), 128 for the narrow
    // tensors (a 64-row step of N = 144 is 21 KB: too few bytes in flight per CU -- measured 3.8 TB/s)
    c->rf = n + k > 320 ? 2 : (n + k <= 176 ? 8 : 4);
    // the instantiations below: EfficientNet-B5 (144,24) (240,40) (384,64), -B2 (96,16) (144,24) (288,48) -- fragment counts AND the
    // chunk slots per thread and step (template constants of the kernel) must be the ones of an instantiation
    const int RB = 16 * c->rf, cy = (RB * (n / 8) + 255) / 256, cx = (RB * (k / 8) + 255) / 256;
    return (c->af == 3 && c->bf == 2 && c->nks == 5 && cy == 9 && cx == 2) || (c->af == 4 && c->bf == 3 && c->nks == 8 && cy == 8 && cx == 2) ||
           (c->af == 6 && c->bf == 4 && c->nks == 12 && cy == 6 && cx == 1) || (c->af == 2 && c->bf == 1 && c->nks == 3 && cy == 6 && cx == 1) ||
           (c->af == 5 && c->bf == 3 && c->nks == 9 && cy == 5 && cx == 1);
}
extern "C" int mc_xbwd_rows_supported(int n, int k) { XbwdCfg c; return xbwd_cfg(n, k, &c) ? 1 : 0; }
extern "C" int mc_xbwd_rows_blocks(long long m) {
    long long steps = (m + 31) / 32;
    long long b = steps < 512 ? steps : 512;
    return (int)(b < 1 ? 1 : b);
}

extern "C" int mc_xbwd_rows_bf16(const mc_wgrad_rows_args* a, const mc_bf16* wt, long long ldwt, mc_bf16* dx, long long lddx,
                                 const mc_bf16* r, long long ldr, void* stream) {
    const mc_wgrad_rows_args& p = *a;
    MC_CHECK(p.dY && p.X && p.dW && p.ws && p.M > 0 && wt && dx, "xbwd_rows: bad args");
    XbwdCfg c;
    MC_CHECK(xbwd_cfg(p.N, p.K, &c), "xbwd_rows: unsupported shape (see mc_xbwd_rows_supported)");
    MC_CHECK(!p.pro_scale && !p.pro_shift && !p.pro_gate, "xbwd_rows: no prologue on x (the expand conv's input is stored activated)");
    MC_CHECK(p.lddy % 8 == 0 && p.ldx % 8 == 0 && ldwt % 8 == 0 && lddx % 8 == 0 && (!r || ldr % 8 == 0) && mc_aligned16(p.dY) && mc_aligned16(p.X) &&
             mc_aligned16(wt) && mc_aligned16(dx) && (!r || mc_aligned16(r)), "xbwd_rows: alignment");
    hipStream_t st = (hipStream_t)stream;
    int WA, WB, af, bfn;
    wave_split(p.N, p.K, &WA, &WB, &af, &bfn);
    const int RB = 16 * c.rf;
    const long long steps = (p.M + RB - 1) / RB;
    const long long cap = mc_xbwd_rows_blocks(p.M);
    const int blocks = (int)(steps < cap ? steps : cap);
    const int nchy = (RB * (p.N / 8) + 255) / 256, nchx = (RB * (p.K / 8) + 255) / 256, nchr = r ? nchx : 0;
    const int KO = ((p.K + 15) / 16) * 16;
    const size_t lds = (size_t)RB * (p.N + p.K) * 2 + 32 * 16 * 2 * 6 + 64 + (size_t)RB * KO * 2 + (size_t)2 * RB * p.K * 2;
    xbwd_extra q;
    q.Wt = wt; q.ldwt = ldwt; q.dX = dx; q.lddx = lddx; q.R = r; q.ldr = ldr;
    // (chunk slots per thread and step are template constants of the instantiation: see the kernel)
#define XB_CASE(A_, B_, N_, R_, CY_, CX_) if (c.af == A_ && c.bf == B_ && c.nks == N_ && c.rf == R_ && nchy == CY_ && nchx == CX_) { \
        hipLaunchKernelGGL((xbwd_rows_kernel<A_, B_, CY_, CX_, N_, R_>), dim3(blocks), dim3(256), lds, st, p, q, WB, nchr); } else
    XB_CASE(3, 2, 5, 8, 9, 2) XB_CASE(4, 3, 8, 4, 8, 2) XB_CASE(6, 4, 12, 2, 6, 1) XB_CASE(2, 1, 3, 8, 6, 1) XB_CASE(5, 3, 9, 2, 5, 1)
    { MC_CHECK(false, "xbwd_rows: internal: no instantiation"); }
#undef XB_CASE
    MC_LAUNCH_CHECK();
    long long nk = (long long)p.N * p.K;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(mc_div_up(nk, 16)), dim3(256), 0, st, p.ws, blocks, nk, p.dW, p.accumulate);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
