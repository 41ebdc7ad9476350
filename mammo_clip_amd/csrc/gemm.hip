// Tiled bf16 MFMA GEMM family for gfx950: C[M,N] (+)= alpha * op(A)[M,K] . op(B)[K,N]  (+bias, +residual)
//
// Serves every matmul-shaped op of the hot path that is NOT covered by the row-streaming kernels
// (gemm_rows.hip / gemm_wgrad_rows.hip), i.e. the compute-heavier layers:
//   * pointwise (1x1) convolutions of the late EfficientNet stages, NHWC = GEMM over pixels
//     (efficientnet_custom.py:104,122,283), their dgrad and wgrad (reduction over pixels, split-K combined
//     through a workspace, no atomics)
//   * BERT linears and the attention matmuls (batched over (batch, head) with strides; no transposes)
// Layout / prologue / output type are TEMPLATE parameters, so each instantiation's hot loop carries no dead
// code (a "one kernel, runtime flags" version measured 22k instructions and was instruction-fetch bound):
//   LAY 0 = NT  A[m*lda+k]  B[n*ldb+k]      (forward linears, dgrad through a transposed weight, Q.K^T)
//   LAY 1 = NN  A[m*lda+k]  B[k*ldb+n]      (P.V, dS.K, dgrad through the plain weight)
//   LAY 2 = TN  A[k*lda+m]  B[k*ldb+n]      (weight gradients, dV, dK)
// Operands are staged global -> registers -> LDS (k-major operands are transposed on the way in), fragments
// are read with ds_read_b128 and fed to v_mfma_f32_16x16x32_bf16; 4 waves per workgroup, double-buffered LDS,
// one barrier per K tile, register-staged tiles prefetched across K tiles AND across row blocks.
// The bf16 epilogue goes back through LDS so global stores are 16-byte row segments, and can emit per-column
// sum / sum-of-squares partials (training-mode BatchNorm statistics of the conv output).
#include "common_hip.h"
#include <cstdlib>
#include <type_traits>
#include "../../include/mammoclip_hip.h"

namespace {

__device__ __forceinline__ uint4 zero4() { return make_uint4(0u, 0u, 0u, 0u); }
// component-wise (a select between two uint4 STRUCTS is lowered through scratch memory)
__device__ __forceinline__ uint4 keep4(bool ok, const uint4& a) {
    return make_uint4(ok ? a.x : 0u, ok ? a.y : 0u, ok ? a.z : 0u, ok ? a.w : 0u);
}

// fused prologue on a vector of 8 channels [ch0, ch0+8) of pixel `pix`:
//   v' = silu(v * scale[c] + shift[c]) * gate[(pix / rows_per_img) * nch + c]
__device__ __forceinline__ uint4 apply_prologue(uint4 v, const mc_gemm_args& p, long long pix, int ch0) {
    float f[8];
    unpack8(v, f);
    if (p.pro_scale) {                 // (null: the operand is already activated, only the gate is applied)
        float s[8], t[8];
        load8f(p.pro_scale + ch0, s);
        load8f(p.pro_shift + ch0, t);
        bn_silu8(f, s, t);
    }
    if (p.pro_gate) {
        long long img = pix / p.pro_rows_per_img;
        float g[8];
        load8f(p.pro_gate + img * p.pro_nch + ch0, g);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] *= g[i];
    }
    return pack8(f);
}

// k-major operands (reduction index = row of the stored matrix) stay ROW-MAJOR in LDS ([k][x], 16-byte stores) and
// are turned into MFMA fragments by gfx950's LDS transpose-read: within a 16-lane group lane i supplies the address
// of row i/4, cols (i%4)*4..+3 of a 4x16 block and lane c receives column c, rows 0..3 (verified on hardware).
typedef __attribute__((ext_vector_type(4))) short s4_t;
typedef __attribute__((address_space(3))) s4_t lds_s4_t;
__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* tile, int rs, int row0, int col0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const unsigned char* a = tile + (size_t)(row0 + g * 8 + (i >> 2)) * rs + (col0 + (i & 3) * 4) * 2;
    s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(a));
    s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(a + 4 * rs));
    typedef __attribute__((ext_vector_type(8))) short s8_t;
    s8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}
constexpr int tr_pad_bytes(int bx) { return bx >= 128 ? 48 : (bx >= 64 ? 16 : 32); }   // conflict-free tr-read strides

#ifdef GEMM_PROF
__device__ unsigned long long g_gemm_prof[8];      // developer phase profile (scripts/gemmbench.hip)
#define GPROF(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[i] += t_ - tprof; tprof = t_; } while (0)
#else
#define GPROF(i)
#endif

// GL = operands go global -> LDS directly (global_load_lds_dwordx4, no register staging, no ds_write pass).  The DMA
// writes a wave-instruction's 64 x 16 B lane-linearly, so the LDS image is plain [row][8 x 16 B] (BK = 64) and the
// bank-conflict swizzle is applied on the SOURCE side: the lane that fills slot s of row r fetches chunk
// s ^ ((r >> 1) & 7) (still the same 128-byte line per row); ds_read_b128 of a fragment then spreads 16 rows x one
// chunk over all 64 banks.  Chunks outside the matrices are fetched from a 16-byte zero block.  Plain NT only.
__device__ __attribute__((aligned(16))) unsigned int g_gemm_zero16[4];
typedef __attribute__((address_space(3))) unsigned int lds_u32_t;
typedef __attribute__((address_space(1))) const unsigned int glb_u32_t;
// Issued as inline assembly: the compiler then keeps no book on the DMA, so the kernel's own counted s_waitcnt vmcnt(N)
// decides when a stage is ready and the transfers of LATER stages stay in flight across barriers and LDS reads (with
// the builtin every LDS access after a DMA waits for vmcnt(0)).  dst_wave_base must be wave-uniform; lane l's 16 bytes
// land at dst_wave_base + 16 l.  M0 (the LDS destination) is saved and restored around the instruction.
__device__ __forceinline__ void glds16(const void* src, unsigned dst_wave_base) {     // dst: LDS byte address
    const unsigned lds = __builtin_amdgcn_readfirstlane(dst_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds) : "memory");
}

template <int BM, int BN, int BK, int WGM, int WGN, int LAY, int PRO, bool CF32, bool GL = false>
__global__ __launch_bounds__(WGM * WGN * 64, (WGM * WGN == 8 && !GL) ? 4 : 2) void gemm_kernel(const mc_gemm_args p, const int gm) {
    static_assert(!GL || ((LAY == 0 || LAY == 2) && PRO == 0 && BK == 64 && BM % 64 == 0 && BN % 64 == 0 &&
                          (WGM * WGN == 4 || (WGM * WGN == 8 && LAY == 0))),
                  "direct-to-LDS staging: plain operands (NT or both k-major), 64-wide K tiles, 4 waves (NT: or 8)");
    // 8-wave tiles are held to <= 128 VGPRs so two workgroups (16 waves) fit a CU
    constexpr int NT = WGM * WGN * 64;                // threads per workgroup (4 or 8 waves)
    constexpr bool AKM = (LAY == 2), BKM = (LAY >= 1);
    constexpr int ROWB = BK * 2 + 16;                 // padded LDS row: conflict-light ds_read_b128
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int KCH = BK / 8;                       // 16-byte chunks per row (k-contiguous operands)
    constexpr int RSA = BM * 2 + tr_pad_bytes(BM);    // row strides of the row-major [k][x] tiles of k-major operands
    constexpr int RSB = BN * 2 + tr_pad_bytes(BN);
    constexpr int A_REGS = AKM ? (BK * (BM / 8) + (NT - 1)) / NT : (BM * KCH + (NT - 1)) / NT;
    constexpr int B_REGS = BKM ? (BK * (BN / 8) + (NT - 1)) / NT : (BN * KCH + (NT - 1)) / NT;
    constexpr int A_BYTES = GL ? BM * 128 : (AKM ? BK * RSA : BM * ROWB);
    constexpr int B_BYTES = GL ? BN * 128 : (BKM ? BK * RSB : BN * ROWB);
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES + (GL ? 0 : 64);
    // epilogue tile row bytes (bf16).  GL: the tile must fit the stage that was just consumed (the other one is being
    // filled), so rows are unpadded and the 16-byte chunk index is XOR-swizzled with the row instead
    constexpr int CROW = GL ? BN * 2 : (BN + 8) * 2;
    constexpr int EH = (GL && !CF32 && BM * CROW > STAGE_BYTES) ? 2 : 1;      // epilogue passes (row halves of the tile)
    constexpr int EROWS = BM / EH;
    constexpr int EPI_BYTES = CF32 ? 0 : EROWS * CROW;
    static_assert(!GL || EPI_BYTES <= STAGE_BYTES, "epilogue tile must fit one stage");
    // GL stages: as many as fit beside nothing else in 160 KB at one (8-wave) or two (4-wave) workgroups per CU
    constexpr int NS = GL ? ((WGM * WGN == 8) ? 3 : 2) : 2;
    constexpr int LDS_BYTES = (NS * STAGE_BYTES > EPI_BYTES) ? NS * STAGE_BYTES : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    // XCD-aware tile placement: workgroups are dealt round-robin to the 8 XCDs by linear id, each with its own L2.
    // All column tiles of one row block get consecutive slots on the SAME XCD, so the activation rows are fetched
    // from HBM once per row block and the other column tiles hit that XCD's L2 (grid.y is padded to a multiple of 8).
    // Only when there are enough row workgroups to keep the XCDs balanced; the launch pads grid.y to a multiple of 8
    // (workgroups past gm exit), which keeps the remap a bijection for every gm.
    int bx = blockIdx.x, by = blockIdx.y;
    if (gm >= 16) {
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        bx = (lin >> 3) % gridDim.x;
        by = ((lin >> 3) / gridDim.x) * 8 + (lin & 7);
    }
    int bzz = blockIdx.z;
    if (gm < 16 && p.batch == 1 && (gridDim.z & 7) == 0 && gridDim.z >= 16) {
        // split-K (weight gradients): all output tiles of one K split read the same slab of both operands -> one
        // XCD per split (splits dealt round-robin to the XCDs)
        const int tiles = gridDim.x * gridDim.y;
        const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const int t = (lin >> 3) % tiles;
        bx = t % gridDim.x;
        by = t / gridDim.x;
        bzz = ((lin >> 3) / tiles) * 8 + (lin & 7);
    }
    if (by >= gm) return;
    const int n0 = bx * BN;

    // batch / split decomposition of the z index
    const int split = bzz % p.splits;
    const int bz = bzz / p.splits;
    const int b1 = bz / p.nb2, b2 = bz % p.nb2;
    const bf16_t* __restrict__ A = p.A + b1 * p.sA1 + b2 * p.sA2;
    const bf16_t* __restrict__ B = p.B + b1 * p.sB1 + b2 * p.sB2;
    const long long coff = b1 * p.sC1 + b2 * p.sC2;
    const float* bias = p.bias ? p.bias + (long long)b1 * p.bias_stride1 : nullptr;

    // K range of this split (multiples of BK)
    const long long ktiles = (p.K + BK - 1) / BK;
    const long long tps = (ktiles + p.splits - 1) / p.splits;
    long long kbeg = (long long)split * tps * BK;
    long long kend = kbeg + tps * BK;
    if (p.split_group_rows > 0) {
        // grouped split-K: the reduction index is cut at group (= image) boundaries, split_sub splits per group, so
        // that a per-(group, column) factor can be applied when the partials are combined (see splitk_reduce_kernel)
        const long long grp = split / p.split_sub, j = split % p.split_sub;
        const long long chunk = (p.split_group_rows + p.split_sub - 1) / p.split_sub;
        kbeg = grp * p.split_group_rows + j * chunk;
        kend = kbeg + chunk;
        if (kend > (grp + 1) * p.split_group_rows) kend = (grp + 1) * p.split_group_rows;
    }
    if (kend > p.K) kend = p.K;
    const long long mtiles = (p.M + BM - 1) / BM;
    const bool n_full = n0 + BN <= p.N;

    float colsum[8], colsq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { colsum[i] = 0.f; colsq[i] = 0.f; }

    uint4 ra0[A_REGS], rb0[B_REGS], ra1[A_REGS], rb1[B_REGS];   // two named register stages (stay in VGPRs)

    // ---- global -> registers.  Every load is unconditional (chunks outside the matrices read a clamped, valid address
    // and are zeroed when stored), so the loads of a tile issue back to back with nothing to wait for in between.
    // Per-thread chunk offsets are constant over the tiles: interior tiles add them to a scalar tile base.
    unsigned offA[A_REGS], offB[B_REGS];
#pragma unroll
    for (int i = 0; i < A_REGS; ++i) {
        const int c = tid + i * NT;
        offA[i] = AKM ? (unsigned)((c / (BM / 8)) * p.lda + (c % (BM / 8)) * 8) : (unsigned)((c / KCH) * p.lda + (c % KCH) * 8);
    }
#pragma unroll
    for (int i = 0; i < B_REGS; ++i) {
        const int c = tid + i * NT;
        offB[i] = BKM ? (unsigned)((c / (BN / 8)) * p.ldb + (c % (BN / 8)) * 8) : (unsigned)((c / KCH) * p.ldb + (c % KCH) * 8);
    }
    auto tile_full = [&](long long m0, long long k0) __attribute__((always_inline)) { return n_full && (m0 + BM <= p.M) && (k0 + BK <= kend); };
    auto load_tiles = [&](uint4 (&ra)[A_REGS], uint4 (&rb)[B_REGS], long long m0, long long k0) __attribute__((always_inline)) {
        constexpr bool A_ALL = (AKM ? BK * (BM / 8) : BM * KCH) % NT == 0;     // every thread's chunks lie inside the tile
        constexpr bool B_ALL = (BKM ? BK * (BN / 8) : BN * KCH) % NT == 0;
        if (tile_full(m0, k0) && A_ALL && B_ALL) {
            const bf16_t* ab = AKM ? A + k0 * p.lda + m0 : A + m0 * p.lda + k0;
            const bf16_t* bb = BKM ? B + k0 * p.ldb + n0 : B + (long long)n0 * p.ldb + k0;
#pragma unroll
            for (int i = 0; i < A_REGS; ++i) ra[i] = *reinterpret_cast<const uint4*>(ab + offA[i]);
#pragma unroll
            for (int i = 0; i < B_REGS; ++i) rb[i] = *reinterpret_cast<const uint4*>(bb + offB[i]);
            return;
        }
#pragma unroll
        for (int i = 0; i < A_REGS; ++i) {
            const int c = tid + i * NT;
            long long m, k;
            if (AKM) { m = m0 + (c % (BM / 8)) * 8; k = k0 + c / (BM / 8); }
            else { m = m0 + c / KCH; k = k0 + (c % KCH) * 8; }
            if (m >= p.M) m = AKM ? 0 : p.M - 1;
            if (k >= kend) k = k0;
            ra[i] = *reinterpret_cast<const uint4*>(AKM ? A + k * p.lda + m : A + m * p.lda + k);
        }
#pragma unroll
        for (int i = 0; i < B_REGS; ++i) {
            const int c = tid + i * NT;
            long long n, k;
            if (BKM) { n = n0 + (c % (BN / 8)) * 8; k = k0 + c / (BN / 8); }
            else { n = n0 + c / KCH; k = k0 + (c % KCH) * 8; }
            if (n >= p.N) n = BKM ? 0 : p.N - 1;
            if (k >= kend) k = k0;
            rb[i] = *reinterpret_cast<const uint4*>(BKM ? B + k * p.ldb + n : B + n * p.ldb + k);
        }
    };
    // validity of chunk i of a tile (only consulted for tiles that are not interior)
    auto a_valid = [&](int i, long long m0, long long k0) __attribute__((always_inline)) {
        const int c = tid + i * NT;
        const long long m = AKM ? m0 + (c % (BM / 8)) * 8 : m0 + c / KCH;
        const long long k = AKM ? k0 + c / (BM / 8) : k0 + (c % KCH) * 8;
        return m < p.M && k < kend;
    };
    auto b_valid = [&](int i, long long k0) __attribute__((always_inline)) {
        const int c = tid + i * NT;
        const long long n = BKM ? n0 + (c % (BN / 8)) * 8 : n0 + c / KCH;
        const long long k = BKM ? k0 + c / (BN / 8) : k0 + (c % KCH) * 8;
        return n < p.N && k < kend;
    };

    // ---- registers -> LDS (the fused BN+SiLU(+gate) prologue is applied here, after the loads have landed).
    // k-contiguous operands: [x][BK] rows (padded), read with ds_read_b128.  k-major operands: [BK][x] rows exactly
    // as loaded (16-byte stores), read with the transpose-read.
    auto store_tiles = [&](const uint4 (&ra)[A_REGS], const uint4 (&rb)[B_REGS], int buf, long long m0, long long k0) __attribute__((always_inline)) {
        unsigned char* sA = smem + buf * STAGE_BYTES;
        unsigned char* sB = sA + A_BYTES;
        const bool full = tile_full(m0, k0);
        // chunk index as a compile-time constant (the prologue makes the body large enough that a plain unrolled loop
        // is not always unrolled, and a runtime index would push the register stage into scratch)
        auto store_a = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i < A_REGS) {
                const int c = tid + i * NT;
                if (c < (AKM ? BK * (BM / 8) : BM * KCH)) {
                    uint4 v = keep4(full || a_valid(i, m0, k0), ra[i]);
                    if (!AKM) {
                        const int row = c / KCH, kc = c % KCH;
                        if (PRO == 1) {
                            const long long m = m0 + row, k = k0 + kc * 8;
                            if (m < p.M && k < kend) v = apply_prologue(v, p, m, (int)k);
                        }
                        *reinterpret_cast<uint4*>(sA + row * ROWB + kc * 16) = v;
                    } else {
                        const int xc = c % (BM / 8), kr = c / (BM / 8);
                        *reinterpret_cast<uint4*>(sA + kr * RSA + xc * 16) = v;
                    }
                }
            }
        };
        auto store_b = [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i < B_REGS) {
                const int c = tid + i * NT;
                if (c < (BKM ? BK * (BN / 8) : BN * KCH)) {
                    uint4 v = keep4(full || b_valid(i, k0), rb[i]);
                    if (!BKM) {
                        const int row = c / KCH, kc = c % KCH;
                        if (PRO == 3) {          // batch = image: the SE gate of this image scales the weight columns
                            const long long k = k0 + kc * 8;
                            if (k < kend) {
                                float f[8], g[8];
                                unpack8(v, f);
                                load8f(p.pro_gate + (long long)b1 * p.pro_nch + k, g);
#pragma unroll
                                for (int q = 0; q < 8; ++q) f[q] *= g[q];
                                v = pack8(f);
                            }
                        }
                        *reinterpret_cast<uint4*>(sB + row * ROWB + kc * 16) = v;
                    } else {
                        const int xc = c % (BN / 8), kr = c / (BN / 8);
                        if (PRO == 2) {
                            const long long n = n0 + xc * 8, k = k0 + kr;
                            if (n < p.N && k < kend) v = apply_prologue(v, p, k, (int)n);
                        }
                        *reinterpret_cast<uint4*>(sB + kr * RSB + xc * 16) = v;
                    }
                }
            }
        };
        static_assert(A_REGS <= 8 && B_REGS <= 8, "chunk lists below cover 8 per operand");
        store_a(std::integral_constant<int, 0>{}); store_a(std::integral_constant<int, 1>{});
        store_a(std::integral_constant<int, 2>{}); store_a(std::integral_constant<int, 3>{});
        store_a(std::integral_constant<int, 4>{}); store_a(std::integral_constant<int, 5>{});
        store_a(std::integral_constant<int, 6>{}); store_a(std::integral_constant<int, 7>{});
        store_b(std::integral_constant<int, 0>{}); store_b(std::integral_constant<int, 1>{});
        store_b(std::integral_constant<int, 2>{}); store_b(std::integral_constant<int, 3>{});
        store_b(std::integral_constant<int, 4>{}); store_b(std::integral_constant<int, 5>{});
        store_b(std::integral_constant<int, 6>{}); store_b(std::integral_constant<int, 7>{});
    };

    f32x4_t acc[FM][FN];

    // operands are SWAPPED in the MFMA (D = Bfrag . Afrag^T) so that a lane ends up with 4 consecutive
    // output COLUMNS of one output row: 8-byte LDS / 16-byte global stores in the epilogue instead of scalars.
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* sA = smem + buf * STAGE_BYTES;
        const unsigned char* sB = sA + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8_t af[FM], bfr[FN];
            const int slot = kk * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                if (AKM) af[i] = tr_frag(sA, RSA, kk * 32, wm * WM + i * 16, lane);
                else af[i] = *reinterpret_cast<const bf16x8_t*>(sA + (wm * WM + i * 16 + (lane & 15)) * ROWB + slot * 16);
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if (BKM) bfr[j] = tr_frag(sB, RSB, kk * 32, wn * WN + j * 16, lane);
                else bfr[j] = *reinterpret_cast<const bf16x8_t*>(sB + (wn * WN + j * 16 + (lane & 15)) * ROWB + slot * 16);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = MC_MFMA_16x16x32(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };
    // acc[i][j][r]: row m = wm*WM + i*16 + (lane & 15), col n = wn*WN + j*16 + (lane >> 4)*4 + r

    int ebuf = 0;          // GL: the stage the epilogue may reuse (the one just consumed)
    auto epilogue = [&](const long long m0) __attribute__((always_inline)) {
        __syncthreads();   // all fragment reads done before smem is reused by the epilogue
        unsigned char* const etile = GL ? smem + ebuf * STAGE_BYTES : smem;
        // byte offset of 16-byte chunk `ch` (+ `in` bytes) in tile row `row`
        auto eoff = [&](int row, int ch, int in) __attribute__((always_inline)) {
            return row * CROW + ((GL ? (ch ^ (row & (BN / 8 - 1))) : ch) << 4) + in;
        };
        const float alpha = p.alpha;
        const int mrow = wm * WM + (lane & 15);
        const int ncol = wn * WN + (lane >> 4) * 4;
        if (CF32) {
            const bool to_ws = p.splits > 1 && p.splitk_ws;
            float* C = to_ws ? p.splitk_ws + (long long)split * p.M * p.N : reinterpret_cast<float*>(p.C) + coff;
            const long long ldc = to_ws ? p.N : p.ldc;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                long long m = m0 + mrow + i * 16;
                if (m >= p.M) continue;
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    int n = n0 + ncol + j * 16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (n + r < p.N) {
                            float v = acc[i][j][r] * alpha;
                            if (bias && split == 0) v += bias[n + r];
                            if (p.c_atomic && !to_ws) atomicAdd(C + m * ldc + n + r, v);
                            else C[m * ldc + n + r] = v;
                        }
                    }
                }
            }
        } else {
            bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + coff;
            constexpr int CPR = BN / 8;                 // 16-byte chunks per tile row
            constexpr int RPP = NT / CPR;              // rows per pass
            const int cc = tid % CPR, r0 = tid / CPR;
            const int n = n0 + cc * 8;
#pragma unroll
            for (int eh = 0; eh < EH; ++eh) {          // the tile goes out in EH row slabs of EROWS rows
                if (eh > 0) __syncthreads();
                if (EH == 1 || (wm * WM) / EROWS == eh) {
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        int nn = n0 + ncol + j * 16;
                        float bv[4] = {0.f, 0.f, 0.f, 0.f};
                        if (bias) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) bv[r] = (nn + r < p.N) ? bias[nn + r] : 0.f;
                        }
#pragma unroll
                        for (int i = 0; i < FM; ++i) {
                            uint2 pk = make_uint2(pack_bf2(acc[i][j][0] * alpha + bv[0], acc[i][j][1] * alpha + bv[1]),
                                                  pack_bf2(acc[i][j][2] * alpha + bv[2], acc[i][j][3] * alpha + bv[3]));
                            *reinterpret_cast<uint2*>(etile + eoff(mrow + i * 16 - eh * EROWS, (ncol + j * 16) >> 3, ((ncol + j * 16) & 7) * 2)) = pk;
                        }
                    }
                }
                __syncthreads();
                if (n < p.N) {
                    for (int row = r0; row < EROWS; row += RPP) {
                        long long m = m0 + eh * EROWS + row;
                        if (m >= p.M) break;
                        uint4 v = *reinterpret_cast<const uint4*>(etile + eoff(row, cc, 0));
                        if (p.R) {
                            float f[8], g[8];
                            unpack8(v, f);
                            uint4 rv = *reinterpret_cast<const uint4*>(p.R + coff + m * p.ldr + n);
                            unpack8(rv, g);
#pragma unroll
                            for (int q = 0; q < 8; ++q) f[q] += g[q];
                            v = pack8(f);
                        }
                        if (p.stat_partials) {
                            float f[8];
                            unpack8(v, f);
#pragma unroll
                            for (int q = 0; q < 8; ++q) { colsum[q] += f[q]; colsq[q] += f[q] * f[q]; }
                        }
                        *reinterpret_cast<uint4*>(C + m * p.ldc + n) = v;
                    }
                }
            }
            __syncthreads();
        }
    };

    // ---------------- direct-to-LDS main loop (GL) ----------------
    // Per flat step: wait for this tile's DMA (the only one in flight), barrier (every wave's part has landed and every
    // wave is done with the other stage), read ALL fragments of the tile into registers, issue the next tile's DMA into
    // the other stage, then run the 2 x FM x FN MFMAs from registers while it lands.  No LDS read is ever issued while
    // a DMA into LDS is outstanding, so the waits the compiler places for LDS-DMA coincide with the explicit one.
    if constexpr (GL) {
        constexpr int NW = WGM * WGN;
        constexpr int NI_A = BM / (8 * NW), NI_B = BN / (8 * NW);   // wave-instructions (1 KiB each) per thread and operand
        constexpr int NL = NI_A + NI_B;                        // DMA instructions per thread and tile
        // k-contiguous operand ([x][64 k], 128-byte rows, 8 rows per wave-instruction): slot s of row r holds chunk
        //   s ^ ((r >> 1) & 7).  k-major operand ([64 k][x], 256-byte rows for x = 128, 4 rows per wave-instruction):
        //   slot s of k-row r holds chunk s ^ (((r & 3) | ((r >> 1) & 4)) << 1), which keeps the 32-byte pairs the
        //   transpose-read fetches together and spreads its 8 rows per 32-lane group over all banks.
        // gx = offset along the operand's own rows (m or n), gk = offset along the reduction, in elements
        unsigned goffA[NI_A], goffB[NI_B];
        int gxA[NI_A], gxB[NI_B], gkA[NI_A], gkB[NI_B];
#pragma unroll
        for (int i = 0; i < NI_A; ++i) {
            if (!AKM) {
                const int row = (i * NW + wave) * 8 + (lane >> 3);
                const int ch = (lane & 7) ^ ((row >> 1) & 7);
                gxA[i] = row; gkA[i] = ch * 8; goffA[i] = (unsigned)(row * p.lda + ch * 8);
            } else {
                static_assert(!AKM || BM == 128, "k-major direct staging is laid out for 128-wide tiles");
                const int row = (i * NW + wave) * 4 + (lane >> 4);
                const int ch = (lane & 15) ^ (((row & 3) | ((row >> 1) & 4)) << 1);
                gxA[i] = ch * 8; gkA[i] = row; goffA[i] = (unsigned)(row * p.lda + ch * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < NI_B; ++i) {
            if (!BKM) {
                const int row = (i * NW + wave) * 8 + (lane >> 3);
                const int ch = (lane & 7) ^ ((row >> 1) & 7);
                gxB[i] = row; gkB[i] = ch * 8; goffB[i] = (unsigned)(row * p.ldb + ch * 8);
            } else {
                static_assert(!BKM || BN == 128, "k-major direct staging is laid out for 128-wide tiles");
                const int row = (i * NW + wave) * 4 + (lane >> 4);
                const int ch = (lane & 15) ^ (((row & 3) | ((row >> 1) & 4)) << 1);
                gxB[i] = ch * 8; gkB[i] = row; goffB[i] = (unsigned)(row * p.ldb + ch * 8);
            }
        }
        const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_gemm_zero16);
        const unsigned smem_lds = (unsigned)(uintptr_t)(lds_u32_t*)smem;   // LDS address of the staging buffer
        auto issue = [&](int stage, long long m0, long long k0) __attribute__((always_inline)) {
            const unsigned sA = smem_lds + stage * STAGE_BYTES;       // LDS byte addresses
            const unsigned sB = sA + A_BYTES;
            if (n_full && (m0 + BM <= p.M) && (k0 + BK <= kend)) {
                const bf16_t* ab = AKM ? A + k0 * p.lda + m0 : A + m0 * p.lda + k0;
                const bf16_t* bb = BKM ? B + k0 * p.ldb + n0 : B + (long long)n0 * p.ldb + k0;
#pragma unroll
                for (int i = 0; i < NI_A; ++i) glds16(ab + goffA[i], sA + (i * NW + wave) * 1024);
#pragma unroll
                for (int i = 0; i < NI_B; ++i) glds16(bb + goffB[i], sB + (i * NW + wave) * 1024);
            } else {
#pragma unroll
                for (int i = 0; i < NI_A; ++i) {
                    const long long m = m0 + gxA[i], k = k0 + gkA[i];
                    glds16((m < p.M && k < kend) ? (AKM ? A + k * p.lda + m : A + m * p.lda + k) : zero, sA + (i * NW + wave) * 1024);
                }
#pragma unroll
                for (int i = 0; i < NI_B; ++i) {
                    const long long n = n0 + gxB[i], k = k0 + gkB[i];
                    glds16((n < p.N && k < kend) ? (BKM ? B + k * p.ldb + n : B + n * p.ldb + k) : zero, sB + (i * NW + wave) * 1024);
                }
            }
        };
        const long long ktn = kbeg < kend ? (kend - kbeg + BK - 1) / BK : 0;
        const long long my_mt = by < mtiles ? (mtiles - by + gm - 1) / gm : 0;
        const long long total = my_mt * ktn;
        if (ktn == 0) {
            for (long long mt = by; mt < mtiles; mt += gm) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                epilogue(mt * BM);
            }
        } else {
        long long pm0 = (long long)by * BM, pk0 = kbeg;                  // position of the NEXT tile to fetch
        auto advance = [&]() __attribute__((always_inline)) {
            pk0 += BK;
            if (pk0 >= kend) { pk0 = kbeg; pm0 += (long long)gm * BM; }
        };
        long long cm0 = pm0, ck0 = pk0;                                  // position of the tile being consumed
        constexpr int D = NS - 1;                                        // tiles in flight ahead of the one consumed
        static_assert((D - 1) * NL < 16, "counted vmcnt wait is encoded in the low 4 bits");
#pragma unroll
        for (int t = 0; t < D; ++t)
            if (t < total) { issue(t, pm0, pk0); advance(); }
        // fragment addresses.  k-contiguous: row r, chunk c -> r * 128 + ((c ^ ((r >> 1) & 7)) << 4); the rows of a
        // fragment are (lane & 15) + multiples of 16, so the swizzle term depends on the lane only.
        // k-major (transpose-read, see tr_frag): lane (g, i) addresses k-row kk*32 + g*8 + (i >> 2) [+4 for the upper
        // half], columns col0 + (i & 3) * 4 .. +3 -> 8 bytes inside chunk (col0 >> 3) + ((i & 3) >> 1).
        const int frow = lane & 15, fsw = (frow >> 1) & 7, fkg = lane >> 4;
        const int ti = lane & 15, tg = lane >> 4;
        auto kc_off = [&](int x0, int kk) __attribute__((always_inline)) {      // k-contiguous operand, tile row base x0
            return (x0 + frow) * 128 + (((kk * 4 + fkg) ^ fsw) << 4);
        };
        auto km_off = [&](int col0, int kk, int hi) __attribute__((always_inline)) {   // k-major operand, 256-byte rows
            const int r = kk * 32 + tg * 8 + (ti >> 2) + hi * 4;
            const int sw = ((r & 3) | ((r >> 1) & 4)) << 1;
            return r * 256 + ((((col0 >> 3) + ((ti & 3) >> 1)) ^ sw) << 4) + ((ti & 1) << 3);
        };
        typedef __attribute__((ext_vector_type(8))) short s8_t;
        auto km_frag = [&](const unsigned char* tile, int col0, int kk) __attribute__((always_inline)) {
            s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(tile + km_off(col0, kk, 0)));
            s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(tile + km_off(col0, kk, 1)));
            s8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bf16x8_t, v);
        };
#ifdef GEMM_PROF
        unsigned long long pacc[6] = {0, 0, 0, 0, 0, 0};
        unsigned long long tprof = __builtin_amdgcn_s_memtime();
#endif
        int buf = 0, nbuf = D % NS;             // stage being consumed / stage the next issue goes to
        bool drain = false;                     // stores of an epilogue are in the queue: count nothing, wait for all
        for (long long i = 0; i < total; ++i) {
            if (ck0 == kbeg) {
#pragma unroll
                for (int a = 0; a < FM; ++a)
#pragma unroll
                    for (int b = 0; b < FN; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            }
            // tile i has landed once at most the D-1 tiles issued after it are still outstanding (loads return in order)
            if (D > 1 && !drain && i + D - 1 < total) __builtin_amdgcn_s_waitcnt(0x0F70 | ((D - 1) * NL));
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            drain = false;
            GPROF(0);
            __builtin_amdgcn_s_barrier();       // every wave's part of tile i is in LDS; everyone is done with stage nbuf
            GPROF(1);
            const unsigned char* sA = smem + buf * STAGE_BYTES;
            const unsigned char* sB = sA + A_BYTES;
            bf16x8_t af[2][FM], bfr[2][FN];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int a = 0; a < FM; ++a) {
                    if (AKM) af[kk][a] = km_frag(sA, wm * WM + a * 16, kk);
                    else af[kk][a] = *reinterpret_cast<const bf16x8_t*>(sA + kc_off(wm * WM + a * 16, kk));
                }
#pragma unroll
                for (int b = 0; b < FN; ++b) {
                    if (BKM) bfr[kk][b] = km_frag(sB, wn * WN + b * 16, kk);
                    else bfr[kk][b] = *reinterpret_cast<const bf16x8_t*>(sB + kc_off(wn * WN + b * 16, kk));
                }
            }
#ifdef GL_WAIT_FRAGS
            __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): fragments are in registers
#endif
            if (i + D < total) { issue(nbuf, pm0, pk0); advance(); }
            GPROF(2);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int a = 0; a < FM; ++a)
#pragma unroll
                    for (int b = 0; b < FN; ++b)
                        acc[a][b] = MC_MFMA_16x16x32(bfr[kk][b], af[kk][a], acc[a][b], 0, 0, 0);
            GPROF(3);
            const bool last_k = ck0 + BK >= kend;
            if (last_k) { ebuf = buf; epilogue(cm0); drain = true; }
            buf = buf + 1 == NS ? 0 : buf + 1;
            nbuf = nbuf + 1 == NS ? 0 : nbuf + 1;
            GPROF(4);
#ifdef GEMM_PROF
            pacc[5] += 1;
#endif
            ck0 += BK;
            if (last_k) { ck0 = kbeg; cm0 += (long long)gm * BM; }
        }
#ifdef GEMM_PROF
        if (tid == 0)
            for (int q = 0; q < 6; ++q) atomicAdd(&g_gemm_prof[q], pacc[q]);
#endif
        }
    }

    // ---------------- software-pipelined main loop ----------------
    if constexpr (!GL) {
    // The (row block, K tile) pairs this workgroup owns form one flat sequence; two tiles are always in flight in
    // registers (global loads are issued two steps ahead of the LDS store that consumes them), across K tiles AND
    // across row blocks, so HBM/L2 latency is covered even when a row block has only one or two K tiles.
    const long long ktn = kbeg < kend ? (kend - kbeg + BK - 1) / BK : 0;
    const long long my_mt = by < mtiles ? (mtiles - by + gm - 1) / gm : 0;
    const long long total = my_mt * ktn;
    if (ktn == 0) {
        for (long long mt = by; mt < mtiles; mt += gm) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            epilogue(mt * BM);
        }
    } else {
        // (m0, k0) of flat step i, advanced incrementally (no divisions in the loop)
        long long pm0 = (long long)by * BM, pk0 = kbeg;                  // position of the NEXT tile to load
        auto advance = [&]() __attribute__((always_inline)) {
            pk0 += BK;
            if (pk0 >= kend) { pk0 = kbeg; pm0 += (long long)gm * BM; }
        };
        long long cm0 = pm0, ck0 = pk0;                                  // position of the tile being consumed
        if (0 < total) { load_tiles(ra0, rb0, pm0, pk0); advance(); }
        if (1 < total) { load_tiles(ra1, rb1, pm0, pk0); advance(); }
        int buf = 0;
#ifdef GEMM_PROF
        unsigned long long pacc[6] = {0, 0, 0, 0, 0, 0};
        unsigned long long tprof = __builtin_amdgcn_s_memtime();
#endif
        // one flat step; the loop below is unrolled by two so that each register stage is named statically (a runtime
        // stage selector makes the compiler rotate the stages with register copies, which forces it to wait for the
        // loads that are still in flight)
        auto step = [&](uint4 (&ra)[A_REGS], uint4 (&rb)[B_REGS], long long i) __attribute__((always_inline)) {
            if (ck0 == kbeg) {
#pragma unroll
                for (int a = 0; a < FM; ++a)
#pragma unroll
                    for (int b = 0; b < FN; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            }
            store_tiles(ra, rb, buf, cm0, ck0);
            GPROF(0);
            __syncthreads();
            GPROF(1);
            {   // unconditional (the last two steps re-load their own tile, never consumed): with exactly one tile of
                // loads per step on every path the compiler can count them, and waits vmcnt(N) instead of vmcnt(0)
                const bool more = i + 2 < total;
                load_tiles(ra, rb, more ? pm0 : cm0, more ? pk0 : ck0);
                if (more) advance();
            }
            GPROF(2);
            compute(buf);
            GPROF(3);
            buf ^= 1;
            const bool last_k = ck0 + BK >= kend;
            if (last_k) epilogue(cm0);
            GPROF(4);
#ifdef GEMM_PROF
            pacc[5] += 1;
#endif
            ck0 += BK;
            if (last_k) { ck0 = kbeg; cm0 += (long long)gm * BM; }
        };
        for (long long i = 0; i < total; i += 2) {
            step(ra0, rb0, i);
            if (i + 1 < total) step(ra1, rb1, i + 1);
        }
#ifdef GEMM_PROF
        if (tid == 0)
            for (int q = 0; q < 6; ++q) atomicAdd(&g_gemm_prof[q], pacc[q]);
#endif
    }

    }   // !GL
    // ---------------- column statistics partials ----------------
    if (!CF32 && p.stat_partials) {
        constexpr int CPR = BN / 8;
        constexpr int RPP = NT / CPR;
        float* red = reinterpret_cast<float*>(smem);     // [RPP][BN][2]
        const int cc = tid % CPR, r0 = tid / CPR;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            red[(r0 * BN + cc * 8 + q) * 2 + 0] = colsum[q];
            red[(r0 * BN + cc * 8 + q) * 2 + 1] = colsq[q];
        }
        __syncthreads();
        for (int c = tid; c < BN; c += NT) {
            float s = 0.f, s2 = 0.f;
            for (int r = 0; r < RPP; ++r) { s += red[(r * BN + c) * 2]; s2 += red[(r * BN + c) * 2 + 1]; }
            int n = n0 + c;
            if (n < p.N) {
                float* dst = p.stat_partials + ((long long)bz * gm + by) * 2 * p.N;
                dst[n] = s;
                dst[p.N + n] = s2;
            }
        }
    }
}

// C[m,n] (+)= sum_s ws[s][m][n]  (deterministic split-K combine; replaces per-element atomics)
// scale (optional): float[splits / sub][n], partial k is multiplied by scale[k / sub][column] (grouped split-K)
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int splits, long long mn, int n, float* __restrict__ C,
                                     long long ldc, int accumulate, const float* __restrict__ scale, int sub) {
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= mn) return;
    if (i + 4 <= mn && (n & 3) == 0) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        const int col = (int)(i % n);
        int k = 0;
        if (!scale)
            for (; k + 3 < splits; k += 4) {           // four independent 16-byte loads in flight, summed in split order
                const float4 v0 = *reinterpret_cast<const float4*>(ws + (long long)k * mn + i);
                const float4 v1 = *reinterpret_cast<const float4*>(ws + (long long)(k + 1) * mn + i);
                const float4 v2 = *reinterpret_cast<const float4*>(ws + (long long)(k + 2) * mn + i);
                const float4 v3 = *reinterpret_cast<const float4*>(ws + (long long)(k + 3) * mn + i);
                s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
                s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
                s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
                s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
            }
        for (; k < splits; ++k) {
            float4 v = *reinterpret_cast<const float4*>(ws + (long long)k * mn + i);
            if (scale) {
                const float4 g = *reinterpret_cast<const float4*>(scale + (long long)(k / sub) * n + col);
                v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
            }
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        long long m = i / n;
        int c = (int)(i % n);
        float* d = C + m * ldc + c;
        if (accumulate) { d[0] += s.x; d[1] += s.y; d[2] += s.z; d[3] += s.w; }
        else { d[0] = s.x; d[1] = s.y; d[2] = s.z; d[3] = s.w; }
    } else {
        for (long long j = i; j < mn && j < i + 4; ++j) {
            float s = 0.f;
            for (int k = 0; k < splits; ++k)
                s += ws[(long long)k * mn + j] * (scale ? scale[(long long)(k / sub) * n + j % n] : 1.f);
            float* d = C + (j / n) * ldc + (j % n);
            *d = accumulate ? *d + s : s;
        }
    }
}

template <int BM, int BN, int BK, int WGM, int WGN, int LAY, int PRO, bool CF32, bool GL = false>
int launch(const mc_gemm_args& p, int grid_m, hipStream_t st) {
    constexpr int NT = WGM * WGN * 64;
    dim3 grid(mc_div_up(p.N, BN), grid_m >= 16 ? (grid_m + 7) / 8 * 8 : grid_m, p.batch * p.splits);   // see the XCD remap
    hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, WGM, WGN, LAY, PRO, CF32, GL>), grid, dim3(NT), 0, st, p, grid_m);
    MC_LAUNCH_CHECK();
    if (p.splits > 1 && p.splitk_ws) {
        long long mn = p.M * p.N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(mc_div_up(mc_div_up(mn, 4), NT)), dim3(NT), 0, st, p.splitk_ws,
                           p.splits, mn, p.N, reinterpret_cast<float*>(p.C), p.ldc, p.c_atomic, p.split_scale,
                           p.split_sub > 0 ? p.split_sub : 1);
        MC_LAUNCH_CHECK();
    }
    return MC_OK;
}

static int mc_gemm_glds_enabled() {
    static int on = -1;
    // developer A/B switch: 0 = register staging, 1 = direct-to-LDS 128x128 tiles (default), 2 = also 256x128 tiles
    // (8 waves, 3 stages; measured 0.9-1.07x of the 128x128 configuration on the model's shapes -- not the default)
    if (on < 0) { const char* e = getenv("MC_GEMM_GLDS"); on = e ? atoi(e) : 1; }
    return on;
}

template <int LAY, int PRO, bool CF32>
int dispatch_tile(const mc_gemm_args& p, int grid_m, hipStream_t st) {
    const bool small_k = p.K <= 48;
    // 128x128 tiles run with 8 waves (wave tile 64x32): half the accumulator / staging registers per thread,
    // twice the waves per CU to overlap global->LDS staging with MFMA issue
    if (p.N > 64) {
        if constexpr ((LAY == 0 || LAY == 2) && PRO == 0) {
            // plain operands: direct-to-LDS staging
            if (!small_k && mc_gemm_glds_enabled()) {
                if constexpr (LAY == 0) {
                    // 256 x 128 tiles (8 waves, 3 stages, one workgroup per CU) once there are >= 2 tiles per CU
                    const long long t256 = ((p.M + 255) / 256) * mc_div_up(p.N, 128) * p.batch * p.splits;
                    if (t256 >= 512 && mc_gemm_glds_enabled() == 2) return launch<256, 128, 64, 4, 2, LAY, PRO, CF32, true>(p, grid_m, st);
                }
                return launch<128, 128, 64, 2, 2, LAY, PRO, CF32, true>(p, grid_m, st);
            }
        }
        return small_k ? launch<128, 128, 32, 2, 2, LAY, PRO, CF32>(p, grid_m, st)
                       : launch<128, 128, 64, 2, 2, LAY, PRO, CF32>(p, grid_m, st);
    }
    if (p.N > 32)
        return small_k ? launch<128, 64, 32, 2, 2, LAY, PRO, CF32>(p, grid_m, st)
                       : launch<128, 64, 64, 2, 2, LAY, PRO, CF32>(p, grid_m, st);
    return small_k ? launch<128, 32, 32, 4, 1, LAY, PRO, CF32>(p, grid_m, st)
                   : launch<128, 32, 64, 4, 1, LAY, PRO, CF32>(p, grid_m, st);
}

}  // namespace

// persistent row-block workgroups per (column tile, batch/split)
static int pick_grid_m(const mc_gemm_args& p) {
    long long mtiles = (p.M + 127) / 128;
    long long cap = p.max_grid_m > 0 ? p.max_grid_m : 512;
    long long gm = mtiles < cap ? mtiles : cap;
    if (!p.stat_partials && p.max_grid_m <= 0) {
        // persistent over row blocks only as far as it keeps >= ~4 workgroups per CU in the grid
        long long nt = mc_div_up(p.N, 128) * (long long)(p.batch > 0 ? p.batch : 1) * (p.splits > 0 ? p.splits : 1);
        long long want = 2048 / (nt > 0 ? nt : 1);
        if (want < 1) want = 1;
        gm = mtiles < want ? mtiles : want;
    }
    // XCD-aware placement (see gemm_kernel) wants a multiple of 8; only where every workgroup has several row blocks
    if (gm >= 16 && mtiles >= 2 * gm) gm &= ~7LL;
    return (int)gm;
}
extern "C" int mc_gemm256_eligible(const mc_gemm_args* a);      // gemm256.hip: 256 x 256 tiles for plain NT problems
extern "C" int mc_gemm256_stat_rows(const mc_gemm_args* a);
extern "C" int mc_gemm256_launch(const mc_gemm_args* a, void* stream);
extern "C" int mc_gemm256_tn_eligible(const mc_gemm_args* a);   // gemm256_tn.hip: 256 x 256 tiles for plain TN (weight gradient) problems
extern "C" int mc_gemm256_tn_launch(const mc_gemm_args* a, void* stream);

extern "C" int mc_gemm_tile_config(const mc_gemm_args* a) {
    // 256: the 256 x 256 x 64 kernel of gemm256.hip will run this problem; 128: the tile family of this file
    return (a->K > 0 && mc_gemm256_eligible(a)) ? 256 : 128;
}

extern "C" int mc_gemm_stat_rows(const mc_gemm_args* a) {
    // number of partial rows the launch will write ( = row-block workgroups x batch ); `a` must be the arguments of the
    // launch itself (the tile configuration depends on the whole problem)
    if (a->K > 0 && mc_gemm256_eligible(a)) return mc_gemm256_stat_rows(a);
    mc_gemm_args q = *a;
    static float dummy;
    q.stat_partials = &dummy;                      // the row count of a launch WITH statistics
    return pick_grid_m(q) * (a->batch > 0 ? a->batch : 1);
}

extern "C" int mc_gemm_bf16(const mc_gemm_args* a, void* stream) {
    mc_gemm_args p = *a;
    MC_CHECK(p.A && p.B && p.C, "gemm: null operand");
    MC_CHECK(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem");
    MC_CHECK(mc_aligned16(p.A) && mc_aligned16(p.B) && mc_aligned16(p.C), "gemm: operands must be 16-byte aligned");
    MC_CHECK(p.lda % 8 == 0 && p.ldb % 8 == 0, "gemm: lda/ldb must be multiples of 8 elements");
    MC_CHECK(p.c_f32 || p.ldc % 8 == 0, "gemm: ldc must be a multiple of 8 for bf16 output");
    MC_CHECK((p.a_kmajor ? p.M : p.K) % 8 == 0, "gemm: contiguous extent of A must be a multiple of 8");
    MC_CHECK((p.b_kmajor ? p.N : p.K) % 8 == 0, "gemm: contiguous extent of B must be a multiple of 8");
    MC_CHECK(p.c_f32 || p.N % 8 == 0, "gemm: N must be a multiple of 8 for bf16 output");
    MC_CHECK(!(p.a_kmajor && !p.b_kmajor), "gemm: layout A k-major with B k-contiguous is not provided");
    MC_CHECK(p.act == 0, "gemm: fused activation is not provided (use mc_gelu_fwd)");
    if (p.batch <= 0) p.batch = 1;
    if (p.nb2 <= 0) p.nb2 = 1;
    if (p.splits <= 0) p.splits = 1;
    MC_CHECK(p.splits == 1 || (p.c_f32 && (p.c_atomic || p.splitk_ws)), "gemm: split-K needs fp32 output + workspace or atomics");
    MC_CHECK(!(p.splits > 1 && p.splitk_ws) || (p.batch == 1 && !p.bias), "gemm: workspace split-K is unbatched, no bias");
    MC_CHECK(p.split_group_rows == 0 || (p.splitk_ws && p.split_sub > 0 && p.splits % p.split_sub == 0 &&
                                         (long long)(p.splits / p.split_sub) * p.split_group_rows >= p.K),
             "gemm: grouped split-K needs the workspace, splits = groups * split_sub and groups * split_group_rows >= K");
    MC_CHECK(!p.split_scale || p.split_group_rows > 0, "gemm: split_scale needs grouped split-K");
    MC_CHECK(!(p.stat_partials && (p.c_f32 || p.splits != 1 || p.nb2 != 1)), "gemm: stats need plain bf16 output");
    MC_CHECK(p.pro_operand == 0 || (p.pro_scale && p.pro_shift) || (!p.pro_scale && !p.pro_shift && p.pro_gate),
             "gemm: prologue needs scale+shift (BN+SiLU) and/or a gate");
    MC_CHECK(p.pro_operand != 1 || (!p.a_kmajor && !p.b_kmajor && !p.c_f32), "gemm: A prologue is provided for NT, bf16 output");
    MC_CHECK(p.pro_operand != 2 || (p.a_kmajor && p.b_kmajor && p.c_f32), "gemm: B prologue is provided for TN, fp32 output");
    MC_CHECK(p.pro_operand != 3 || (!p.a_kmajor && !p.b_kmajor && !p.c_f32 && p.pro_gate && !p.pro_scale && p.nb2 == 1),
             "gemm: the per-batch weight gate is provided for NT, bf16 output, gate only");
    MC_CHECK(!p.R || (!p.c_f32 && p.ldr % 8 == 0), "gemm: residual needs bf16 output and ldr % 8 == 0");
    if (p.alpha == 0.f) p.alpha = 1.f;
    if (mc_gemm256_eligible(&p)) return mc_gemm256_launch(&p, stream);
    if (mc_gemm256_tn_eligible(&p)) {
        const int rc = mc_gemm256_tn_launch(&p, stream);
        if (rc != MC_OK || p.splits <= 1) return rc;
        const long long mn = p.M * p.N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(mc_div_up(mc_div_up(mn, 4), 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           p.splitk_ws, p.splits, mn, p.N, reinterpret_cast<float*>(p.C), p.ldc, p.c_atomic, p.split_scale,
                           p.split_sub > 0 ? p.split_sub : 1);
        MC_LAUNCH_CHECK();
        return MC_OK;
    }
    MC_CHECK(!p.ab_fp8 && !p.alpha_dev, "gemm: fp8 operands / alpha_dev need the plain NT bf16-output form (gemm256)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int grid_m = pick_grid_m(p);
    const int lay = p.a_kmajor ? 2 : (p.b_kmajor ? 1 : 0);
    if (lay == 0) {
        if (p.c_f32) return dispatch_tile<0, 0, true>(p, grid_m, st);
        if (p.pro_operand == 1) return dispatch_tile<0, 1, false>(p, grid_m, st);
        if (p.pro_operand == 3) return dispatch_tile<0, 3, false>(p, grid_m, st);
        return dispatch_tile<0, 0, false>(p, grid_m, st);
    }
    if (lay == 1) {
        if (p.c_f32) return dispatch_tile<1, 0, true>(p, grid_m, st);
        return dispatch_tile<1, 0, false>(p, grid_m, st);
    }
    if (p.c_f32) {
        if (p.pro_operand == 2) return dispatch_tile<2, 2, true>(p, grid_m, st);
        return dispatch_tile<2, 0, true>(p, grid_m, st);
    }
    return dispatch_tile<2, 0, false>(p, grid_m, st);
}
