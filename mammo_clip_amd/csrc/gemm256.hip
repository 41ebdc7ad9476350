// 256 x 256 x 64 bf16 (or fp8) MFMA GEMM for gfx950, plain NT operands, bf16 output (round-3 rewrite):
//     C[z][M,N] = alpha * A[z][M,K] . B[z][N,K]^T (+ bias[n]) (+ R[m,n]),   optional per-column sum / sum-of-squares partials
// [ref: model/modules/efficientnet_custom.py:104,122,283 (late-stage 1x1 convolutions, forward and data gradient),
//       model/modules/text_encoder.py:47-49 -> BertModel linears]
//
// Structure (one workgroup of 8 waves per CU, all 160 KB of LDS):
//   * waves 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 fragments of v_mfma_f32_16x16x32_bf16 (128 accumulator registers)
//   * a K tile (128 bytes per row: 64 bf16 / 128 fp8) of A and of B is staged as FOUR 16 KB half-tiles [A0 | A1 | B0 | B1]
//     x 2 stage buffers (+ a 32 KB epilogue slab); half h of A holds, for BOTH wave rows, the 64 rows of C-quadrant row h,
//     half h of B the 32 columns of quadrant column h of all four wave columns
//   * LDS-direct DMA; it writes lane-linearly, so the bank-conflict swizzle sits on the SOURCE address: slot s of LDS row r
//     holds 16-byte chunk s ^ ((r >> 1) & 7) of that row's 128 bytes (conflict-free for ds_read_b128's 16-lane groups)
//   * 4 phases per K tile, one C quadrant (16 MFMAs per wave) and ONE half-tile DMA issue (2 instructions per wave) per
//     phase; the DMA stream runs 3 half-tiles ahead behind a counted s_waitcnt vmcnt(6) once per K tile -- it never drains
//     in the main loop -- and crosses output-tile boundaries (persistent workgroups, one flat K-tile stream):
//         phase 1: read B0 (4) + A0 (8) fragments, stage A1(t+1);  quadrant (0,0)
//         phase 2: read B1 (4),                    stage B0(t+2);  quadrant (0,1)
//         phase 3: read A1 (8),                    stage A0(t+2);  quadrant (1,1)
//         phase 4: (B0 still in registers)         stage B1(t+2);  vmcnt(6): all of tile t+1 has landed;  quadrant (1,0)
//     Hazards: a half-tile is re-staged >= 2 phases after its last ds_read, or 1 phase after (B0) with the reads retired
//     by an explicit lgkmcnt before the reading phase's first barrier; a staged buffer is read one phase after the wait
//     that retires it.  The two wave rows run half a phase apart (one extra barrier for wave row 1 up front): while one
//     wave of a SIMD issues MFMAs its partner reads LDS / issues DMA; s_setprio favours the MFMA wave.
//   * epilogue through the free 32 KB in four 64-row slabs (16-byte coalesced stores, optional bias / residual / BatchNorm
//     column statistics), while the DMAs of the next output tile are already in flight.
// Measured (scripts/g8bench.hip, uniform random [-1,1) operands, MI355X): 8192^3 1374 TFLOP/s (best 1417), 4096^3 1315
// (1331) -- the figures the CDNA guide quotes for this structure; round 2's kernel of the same geometry: 1024 / 993.
// What changed against round 2's kernel of the same geometry (which stayed at ~1.0 PFLOP/s):
//   * the K loop carries NO address arithmetic beyond three scalar adds: the staging DMA is `buffer_load_dwordx4 ... lds`
//     through a buffer descriptor whose base sits on the K tile and whose num_records ends at the last valid byte of the
//     tile's last valid row -- rows beyond M / N come back as zeros from the hardware range check, the K tail is one
//     per-lane constant (lanes whose 16-byte chunk lies beyond K get bit 31 set in their offset), so full and partial
//     tiles run the same branch-free code (round 2: per-K-tile 64-bit divisions to re-derive the stream position, an
//     EXEC-masked partial-tile path inlined into every phase, 15 spilled SGPRs)
//   * work order: the 32 workgroups of an XCD walk the tile grid in panels of 4 row blocks x all column tiles, column-major
//     inside a panel, so the tiles in flight on one L2 form a ~4 x 8 block (12 operand panels per K step instead of 33 for
//     a 1 x 32 strip: the 8192^3 case was HBM/MALL-bound on re-fetched B panels)
#include "common_hip.h"
#include "../../include/mammoclip_hip.h"

namespace g8 {

constexpr int BM = 256, BN = 256, NTHR = 512;
constexpr int HALF_BYTES = 128 * 128;            // 128 LDS rows of 128 bytes
constexpr int STAGE_BYTES = 4 * HALF_BYTES;      // 64 KB: [A0 | A1 | B0 | B1]
constexpr int EPI_OFF = 2 * STAGE_BYTES;         // 128 KB
constexpr int EPI_BYTES = 64 * BN * 2;           // one slab: 64 rows x 256 columns bf16 = 32 KB
constexpr int LDS_BYTES = EPI_OFF + EPI_BYTES;   // 160 KB
constexpr int PANEL_ROWS = 4;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// LDS-direct buffer load of 16 bytes per lane: lane l's bytes land at M0 + 16 l.  voff = per-lane byte offset from the
// descriptor's base (range-checked against num_records: out-of-range lanes deliver zeros).  Inline assembly: the compiler
// keeps no book on it (the kernel counts vmcnt itself); M0 is written in the statement that uses it.
__device__ __forceinline__ void dma16(unsigned voff, u32x4 srd, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                 :: "v"(voff), "s"(srd), "s"(lds_dst) : "memory");
}

#define G8_BAR()                                     \
    do {                                             \
        asm volatile("" ::: "memory");               \
        __builtin_amdgcn_s_barrier();                \
        asm volatile("" ::: "memory");               \
        __builtin_amdgcn_sched_barrier(0);           \
    } while (0)

struct Coords { int z, mt, nt; };

// item `it` (0-based, this workgroup's own count) -> (batch, row block, column tile).  Wave-uniform 32-bit arithmetic,
// evaluated once per OUTPUT tile by the producer and once by the consumer.
//   units (batch, row block) u = xcd + 8 q (q = the XCD's own unit index); the XCD's items are ordered panel by panel
//   (PANEL_ROWS of its units x all NTl column tiles), column-major inside the panel; item index on the XCD = slot + it * S
__device__ __forceinline__ Coords item_coords(int xcd, int slot, int S, int NTl, int MT, int ux, int pr, unsigned it) {
    const unsigned i = (unsigned)slot + it * (unsigned)S;
    const unsigned per = (unsigned)(pr * NTl);
    const unsigned panel = i / per, rem = i - panel * per;
    const unsigned left = (unsigned)ux - panel * (unsigned)pr;
    const unsigned rows = left < (unsigned)pr ? left : (unsigned)pr;
    const unsigned col = rem / rows, row = panel * (unsigned)pr + (rem - col * rows);
    const unsigned unit = (unsigned)xcd + 8u * row;
    Coords c;
    c.z = (int)(unit / (unsigned)MT);
    c.mt = (int)(unit - (unsigned)c.z * (unsigned)MT);
    c.nt = (int)col;
    return c;
}

template <bool STATS, bool FP8>
__global__ __launch_bounds__(NTHR, 2) void gemm8p_kernel(const mc_gemm_args p, const int MT, const int NTl, const int ktn, const int nt_out) {
    constexpr int ES = FP8 ? 1 : 2;             // bytes per operand element
    constexpr int KT = 128 / ES;                // K tile in elements (128 bytes per LDS row)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // ---- this workgroup's share of the tile grid
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, S = gridDim.x >> 3;
    const int RB = p.batch * MT;                                               // (batch, row block) units
    const int ux = RB > xcd ? (RB - xcd + 7) >> 3 : 0;                         // units of this XCD
    const int pr = NTl >= 8 ? PANEL_ROWS : 1;
    const int nloc = ux * NTl;
    const int n_items = nloc > slot ? (nloc - slot + S - 1) / S : 0;
    if (n_items == 0) return;
    const int total = n_items * ktn;                                           // K tiles in this workgroup's stream (< 2^31: host check)

    // ---- DMA source geometry of this thread: 2 wave-instructions per half-tile, each fills 8 LDS rows
    //   instruction i of wave w fills LDS rows lr = (i*8 + w)*8 + (lane >> 3); slot s = lane & 7 holds chunk c = s ^ ((lr >> 1) & 7)
    //   A half h: tile row i*128 + h*64 + w*8 + (lane >> 3);   B half h: tile column i*128 + h*32 + (w >> 2)*64 + (w & 3)*8 + (lane >> 3)
    const int lsub = lane >> 3;
    const int ck = (lane & 7) ^ (((wave * 8 + lsub) >> 1) & 7);
    const unsigned voffA = (unsigned)(((wave * 8 + lsub) * (int)p.lda) * ES + ck * 16);
    const unsigned voffB = (unsigned)((((wave >> 2) * 64 + (wave & 3) * 8 + lsub) * (int)p.ldb) * ES + ck * 16);
    const unsigned rowA = (unsigned)(64 * (int)p.lda * ES), rowB = (unsigned)(32 * (int)p.ldb * ES);   // per h; per i: 128 rows
    typedef __attribute__((address_space(3))) unsigned int lds_u32_t;
    const unsigned smem_lds = (unsigned)(uintptr_t)(lds_u32_t*)smem;
    const unsigned dma_dst = smem_lds + (unsigned)(wave * 1024);               // + buf*STAGE + which*HALF + i*8192

    // ---- producer state: the K tile the DMA stream is at (wave-uniform scalars + one per-lane K-tail constant)
    struct Prod {
        unsigned long long a, b;        // byte address of the tile origin in A / B
        unsigned nra, nrb;              // num_records: bytes from the origin to the end of the last valid row's valid K range
        unsigned oob;                   // per lane: 0x80000000 if this lane's chunk lies beyond K in this tile, else 0
        int kt;                         // K tile index inside the output tile
        unsigned it;                    // item index
        bool valid;
    } pd;
    auto prod_tile = [&](unsigned it) __attribute__((always_inline)) {
        pd.it = it; pd.kt = 0;
        pd.valid = it < (unsigned)n_items;
        pd.oob = (ktn == 1 && ck * (16 / ES) >= (int)p.K) ? 0x80000000u : 0u;
        if (!pd.valid) return;
        const Coords c = item_coords(xcd, slot, S, NTl, MT, ux, pr, it);
        const long long m0 = (long long)c.mt * BM, n0 = (long long)c.nt * BN;
        pd.a = (unsigned long long)(uintptr_t)p.A + (unsigned long long)(((long long)c.z * p.sA1 + m0 * p.lda) * ES);
        pd.b = (unsigned long long)(uintptr_t)p.B + (unsigned long long)(((long long)c.z * p.sB1 + n0 * p.ldb) * ES);
        const long long mr = p.M - m0 > BM ? BM : p.M - m0, nr = (long long)p.N - n0 > BN ? BN : (long long)p.N - n0;
        pd.nra = (unsigned)(((mr - 1) * p.lda + p.K) * ES);
        pd.nrb = (unsigned)(((nr - 1) * p.ldb + p.K) * ES);
    };
    auto prod_next = [&]() __attribute__((always_inline)) {
        if (pd.kt + 1 < ktn) {
            ++pd.kt;
            pd.a += 128; pd.b += 128; pd.nra -= 128; pd.nrb -= 128;
            if (pd.kt + 1 == ktn) pd.oob = (pd.kt * KT + ck * (16 / ES) >= (int)p.K) ? 0x80000000u : 0u;
        } else {
            prod_tile(pd.it + 1);
        }
    };
    // stage half-tile `which` (0 = A0, 1 = A1, 2 = B0, 3 = B1) of the producer's tile into stage buffer `buf`
    auto stage = [&](int buf, auto which_c) __attribute__((always_inline)) {
        constexpr int which = decltype(which_c)::value;
        constexpr bool isA = which < 2;
        constexpr int h = which & 1;
        if (!pd.valid) return;
        const unsigned long long base = isA ? pd.a : pd.b;
        const u32x4 srd = {(unsigned)base, (unsigned)(base >> 32) & 0xffffu, isA ? pd.nra : pd.nrb, 0x00020000u};
        const unsigned v0 = (isA ? voffA + h * rowA : voffB + h * rowB) + pd.oob;
        const unsigned dst = dma_dst + (unsigned)(buf * STAGE_BYTES + which * HALF_BYTES);
        dma16(v0, srd, dst);
        dma16(v0 + 2 * (isA ? rowA : 2 * rowB), srd, dst + 8192u);
    };
    using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;

    // ---- fragment read addresses: LDS row r = base + (lane & 15), chunk (kk*4 + (lane >> 4)) ^ ((r >> 1) & 7); the row
    // bases are multiples of 16, so the swizzle term depends on the lane only; kk = 1 flips chunk bit 2 (byte 64)
    const int frow = lane & 15;
    const unsigned fsw0 = (unsigned)((((lane >> 4) ^ ((frow >> 1) & 7)) << 4));
    const unsigned fA0 = (unsigned)((wm * 64 + frow) * 128) + fsw0;     // + i*2048 (16 rows); kk = 1: byte bit 6 flipped
    const unsigned fB0 = (unsigned)((wn * 32 + frow) * 128) + fsw0;     // + j*2048
    const unsigned fA1 = fA0 ^ 64u, fB1 = fB0 ^ 64u;

    f32x4_t acc[8][4];
    bf16x8_t a0[4][2], a1[4][2], b0[2][2], b1[2][2];

    auto read_a = [&](bf16x8_t (&af)[4][2], int buf, int h) __attribute__((always_inline)) {
        const unsigned char* base = smem + buf * STAGE_BYTES + h * HALF_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *reinterpret_cast<const bf16x8_t*>(base + fA0 + i * 2048);
            af[i][1] = *reinterpret_cast<const bf16x8_t*>(base + fA1 + i * 2048);
        }
    };
    auto read_b = [&](bf16x8_t (&bf)[2][2], int buf, int h) __attribute__((always_inline)) {
        const unsigned char* base = smem + buf * STAGE_BYTES + (2 + h) * HALF_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bf[j][0] = *reinterpret_cast<const bf16x8_t*>(base + fB0 + j * 2048);
            bf[j][1] = *reinterpret_cast<const bf16x8_t*>(base + fB1 + j * 2048);
        }
    };
    // operands swapped (D = Bfrag . Afrag^T): a lane holds 4 consecutive output COLUMNS of one output row
    //   acc[i8][j4][r]: row = wm*128 + i8*16 + (lane & 15), column = wn*64 + j4*16 + (lane >> 4)*4 + r
    auto mma_quad = [&](const bf16x8_t (&af)[4][2], const bf16x8_t (&bf)[2][2], auto ih_c, auto jh_c) __attribute__((always_inline)) {
        constexpr int ih = decltype(ih_c)::value, jh = decltype(jh_c)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (FP8) {
                        typedef __attribute__((ext_vector_type(2))) long l2_t;
                        const l2_t a2 = __builtin_bit_cast(l2_t, af[i][kk]), b2 = __builtin_bit_cast(l2_t, bf[j][kk]);
                        f32x4_t c = acc[ih * 4 + i][jh * 2 + j];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b2[0], a2[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b2[1], a2[1], c, 0, 0, 0);
                        acc[ih * 4 + i][jh * 2 + j] = c;
                    } else {
                        acc[ih * 4 + i][jh * 2 + j] =
                            MC_MFMA_16x16x32(bf[j][kk], af[i][kk], acc[ih * 4 + i][jh * 2 + j], 0, 0, 0);
                    }
                }
        __builtin_amdgcn_s_setprio(0);
    };

    // alpha_dev: a device-resident factor (the product of the fp8 dequantisation scales); pinned before the loop
    float alpha = p.alpha;
    if (p.alpha_dev) {
        alpha *= __builtin_nontemporal_load(p.alpha_dev);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(alpha));
    }

    // consumer position
    unsigned cit = 0;
    int ckt = 0;
    Coords cc = item_coords(xcd, slot, S, NTl, MT, ux, pr, 0);

    // ---- pipeline prologue: tile 0 completely, tile 1 without its last half (stream order per tile: B0, A0, B1, A1)
    prod_tile(0);
    stage(0, C2{}); stage(0, C0{}); stage(0, C3{}); stage(0, C1{});
    prod_next();                                                    // tile 1
    stage(1, C2{}); stage(1, C0{}); stage(1, C3{});
    if (pd.valid) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G8_BAR();
    if (wm == 1) G8_BAR();                      // wave row 1 runs half a phase behind wave row 0 from here on

#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    bool drain = false;                         // stores of an epilogue are in the queue: the next tile wait drains it
    int buf = 0;
    // invariant at the loop head: pd = tile t+1 (its A1 half not issued yet); everything of tile t has landed
    for (int t = 0; t < total; ++t) {
        // ------------------------------------------------ phase 1: quadrant (0,0)
        read_b(b0, buf, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(a0, buf, 0);
        stage(buf ^ 1, C1{});                                    // A1 of tile t+1
        prod_next();                                             // pd = tile t+2
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");       // the 4 B0 reads have retired (B0 is re-staged next phase)
        G8_BAR();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma_quad(a0, b0, C0{}, C0{});
        G8_BAR();
        // ------------------------------------------------ phase 2: quadrant (0,1)
        read_b(b1, buf, 1);
        stage(buf, C2{});                                        // B0 of tile t+2
        G8_BAR();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma_quad(a0, b1, C0{}, C1{});
        G8_BAR();
        // ------------------------------------------------ phase 3: quadrant (1,1)
        read_a(a1, buf, 1);
        stage(buf, C0{});                                        // A0 of tile t+2
        G8_BAR();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma_quad(a1, b1, C1{}, C1{});
        G8_BAR();
        // ------------------------------------------------ phase 4: quadrant (1,0)
        if (drain) {
            // first tile after an epilogue: its global stores share the counter with the DMAs and may complete out of
            // order with them, so nothing can be counted -- wait for everything issued so far, THEN issue B1(t+2): the
            // next tile's vmcnt(6) is exact again
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stage(buf, C3{});
            drain = false;
        } else {
            stage(buf, C3{});                                    // B1 of tile t+2
            if (pd.valid) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");           // everything up to A1(t+1) has landed
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        G8_BAR();
        mma_quad(a1, b0, C1{}, C0{});
        G8_BAR();

        buf ^= 1;
        ++ckt;
        if (ckt == ktn) {
            // ============================================ epilogue of item cc
            if (wm == 0) G8_BAR();              // un-stagger: wave row 0 waits for wave row 1 to finish its last quadrant
            unsigned char* const etile = smem + EPI_OFF;
            const long long m0 = (long long)cc.mt * BM;
            const int n0 = cc.nt * BN;
            bf16_t* const Cb = reinterpret_cast<bf16_t*>(p.C) + (long long)cc.z * p.sC1;
            const bf16_t* const Rb = p.R ? p.R + (long long)cc.z * p.sC1 : nullptr;
            const float* bias = p.bias ? p.bias + (long long)cc.z * p.bias_stride1 : nullptr;
            const int c16 = tid & 31, r0 = tid >> 5;             // copy-out role: 16-byte chunk c16 of rows r0 + 16*pass
            const int ncol = n0 + c16 * 8;
            float csum[8], csq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { csum[q] = 0.f; csq[q] = 0.f; }
            // bias of the lane's 4 x 4 columns.  (Loaded by inline assembly with its own wait: ANY vector-memory load the
            // compiler tracks inside this loop makes it drain vmcnt at the loop head -- and with it the DMA stream -- on
            // every K tile.)  Column groups start at multiples of 4 and N % 8 == 0: a group is inside or outside as a whole.
            float bv[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[j][r] = 0.f;
            if (bias) {
                f32x4_t tb[4];
                const float* bp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
                    bp[j] = bias + (n < p.N ? n : 0);
                }
                asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"
                             "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(tb[0]), "=&v"(tb[1]), "=&v"(tb[2]), "=&v"(tb[3])
                             : "v"(bp[0]), "v"(bp[1]), "v"(bp[2]), "v"(bp[3]) : "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[j][r] = n < p.N ? tb[j][r] : 0.f;
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {                        // slab s = tile rows s*64 .. s*64+63 (wave row s >> 1, quadrant row s & 1)
                if (wm == (s >> 1)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x4_t v = acc[(s & 1) * 4 + i][j];
                            const uint2 pk = make_uint2(pack_bf2(v[0] * alpha + bv[j][0], v[1] * alpha + bv[j][1]),
                                                        pack_bf2(v[2] * alpha + bv[j][2], v[3] * alpha + bv[j][3]));
                            const int row = i * 16 + (lane & 15);
                            const int col = wn * 64 + j * 16 + (lane >> 4) * 4;          // multiple of 4
                            const int ch = col >> 3;
                            *reinterpret_cast<uint2*>(etile + row * (BN * 2) + ((ch ^ (row & 31)) << 4) + (col & 7) * 2) = pk;
                        }
                }
                G8_BAR();
                if (ncol < p.N) {
                    // residual: the 4 vectors of this thread's slab rows in flight together, ONE wait (round 5; the first
                    // form paid 4 dependent round trips per slab).  Rows beyond M re-read row m0 (valid) and are not stored.
                    uint4 rv[4];
                    if (Rb) {
                        const bf16_t* rp[4];
#pragma unroll
                        for (int ps = 0; ps < 4; ++ps) {
                            const long long m = m0 + s * 64 + r0 + ps * 16;
                            rp[ps] = Rb + (m < p.M ? m : m0) * p.ldr + ncol;
                        }
                        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"
                                     "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off\n\ts_waitcnt vmcnt(0)"
                                     : "=&v"(rv[0]), "=&v"(rv[1]), "=&v"(rv[2]), "=&v"(rv[3])
                                     : "v"(rp[0]), "v"(rp[1]), "v"(rp[2]), "v"(rp[3]) : "memory");
                    }
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const int row = r0 + ps * 16;
                        const long long m = m0 + s * 64 + row;
                        if (m < p.M) {
                            uint4 v = *reinterpret_cast<const uint4*>(etile + row * (BN * 2) + ((c16 ^ (row & 31)) << 4));
                            if (Rb) {
                                float f[8], g[8];
                                unpack8(v, f);
                                unpack8(rv[ps], g);
#pragma unroll
                                for (int q = 0; q < 8; ++q) f[q] += g[q];
                                v = pack8(f);
                            }
                            if (STATS) {
                                float f[8];
                                unpack8(v, f);
#pragma unroll
                                for (int q = 0; q < 8; ++q) { csum[q] += f[q]; csq[q] += f[q] * f[q]; }
                            }
                            // nt_out (outputs of 128 MB and more): non-temporal stores -- the lines leave the L2 early instead
                            // of queueing behind each other's evictions; small outputs stay cached for their consumer
                            if (nt_out) __builtin_nontemporal_store((u32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(Cb + m * p.ldc + ncol));
                            else *reinterpret_cast<uint4*>(Cb + m * p.ldc + ncol) = v;
                        }
                    }
                }
                G8_BAR();
            }
            if (STATS) {
                // column statistics of this output tile: 16 row groups -> one value per column, fixed order
                float* red = reinterpret_cast<float*>(etile);            // [16][256][2] floats = 32 KB
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    red[(r0 * BN + c16 * 8 + q) * 2 + 0] = csum[q];
                    red[(r0 * BN + c16 * 8 + q) * 2 + 1] = csq[q];
                }
                G8_BAR();
                if (tid < BN) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { s1 += red[(r * BN + tid) * 2]; s2 += red[(r * BN + tid) * 2 + 1]; }
                    const int n = n0 + tid;
                    if (n < p.N) {
                        float* dst = p.stat_partials + ((long long)cc.z * MT + cc.mt) * 2 * p.N;
                        dst[n] = s1;
                        dst[p.N + n] = s2;
                    }
                }
                G8_BAR();
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            drain = true;
            ckt = 0;
            ++cit;
            if (cit < (unsigned)n_items) cc = item_coords(xcd, slot, S, NTl, MT, ux, pr, cit);
            if (wm == 1) G8_BAR();              // restore the half-phase stagger
        }
    }
    if (wm == 0) G8_BAR();                      // balance the extra barrier of wave row 1
}

}  // namespace g8

// ---- host side ------------------------------------------------------------------------------------------------------
// eligibility + launch, called from mc_gemm_bf16 (gemm.hip) for plain NT bf16-output problems
#include <cstdlib>
// MC_GEMM_256: 0 = never, 1 = by the size rule below (default), 2 = whenever the layout allows (tests / A-B runs)
static int g256_mode() { const char* e = getenv("MC_GEMM_256"); return e ? atoi(e) : 1; }
static int g8_layout_ok(const mc_gemm_args& p) {
    // limits of the descriptor form: 31-bit byte offsets inside a tile, 32-bit item / K-tile counts, rows at least K long
    const int es = p.ab_fp8 ? 1 : 2;
    if (p.lda * 256 * es + 256 >= (1LL << 31) || p.ldb * 256 * es + 256 >= (1LL << 31)) return 0;
    if (p.lda < p.K || p.ldb < p.K) return 0;
    const long long MT = (p.M + 255) / 256, NTl = (p.N + 255) / 256, kt = (p.K + (p.ab_fp8 ? 127 : 63)) / (p.ab_fp8 ? 128 : 64);
    const long long items = (long long)(p.batch > 0 ? p.batch : 1) * MT * NTl;
    return items * kt < (1LL << 30) && items < (1LL << 28);
}
extern "C" int mc_gemm256_eligible(const mc_gemm_args* a) {
    const mc_gemm_args& p = *a;
    const int mode = g256_mode();
    if (mode == 0 && !p.ab_fp8) return 0;
    if (p.a_kmajor || p.b_kmajor || p.c_f32 || p.pro_operand != 0 || p.splits > 1 || p.nb2 > 1) return 0;
    if (!g8_layout_ok(p)) return 0;
    if (p.ab_fp8) return 1;                     // the fp8 operand path exists only here
    if (mode == 2) return 1;
    if (p.N <= 128 || p.K < 64 || p.M < 256) return 0;
    const long long MT = (p.M + 255) / 256, NTl = (p.N + 255) / 256;
    const long long items = (long long)(p.batch > 0 ? p.batch : 1) * MT * NTl;
    if (items < 96) return 0;
    // One 256 x 256 tile per CU and round.  Against the 128 x 128 kernel of gemm.hip (two workgroups per CU, four times
    // finer work granularity) the big tile wins wherever its rounds are reasonably filled (A/B on MI355X, TFLOP/s,
    // scripts/gemm256_bench.py): 44544 x 3072 x 512  884 vs 613, 16384 x 3072 x 768  1056 vs 759, 16384 x 768 x 3072 (192
    // tiles, one 75 % round) 1028 vs 832, 44544 x 1824 x 304  576 vs 510, 44544 x 304 x 1824  512 vs 481 (348 tiles, 41 %
    // column padding), 173280 x 176 x 1056  595 vs 429 (one column tile: the activation operand streams exactly once),
    // 8192^3  1377 vs 887.  It loses at N = 128 (half of every tile is padding: 455 vs 470, both HBM-bound).
    const long long rounds = (items + 255) / 256;
    const double occ = (double)items / (double)(rounds * 256), useful = (double)p.N / (double)(NTl * 256);
    return occ * useful >= 0.38;
}
extern "C" int mc_gemm256_stat_rows(const mc_gemm_args* a) {
    return (int)((a->batch > 0 ? a->batch : 1) * ((a->M + 255) / 256));
}
extern "C" int mc_gemm256_launch(const mc_gemm_args* a, void* stream) {
    mc_gemm_args p = *a;
    if (p.batch <= 0) p.batch = 1;
    if (p.alpha == 0.f) p.alpha = 1.f;
    const int kt_el = p.ab_fp8 ? 128 : 64;
    const int MT = (int)((p.M + 255) / 256), NTl = (int)((p.N + 255) / 256), ktn = (int)((p.K + kt_el - 1) / kt_el);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(256), block(g8::NTHR);
    MC_CHECK(g8_layout_ok(p), "gemm256: leading dimension / problem size beyond the descriptor form's limits");
    static const long long nt_min = getenv("MC_GEMM_NT_BYTES") ? atoll(getenv("MC_GEMM_NT_BYTES")) : (128LL << 20);
    const int nt = (long long)p.batch * p.M * p.N * 2 >= nt_min ? 1 : 0;
    MC_CHECK(p.K % 8 == 0 && p.lda % 8 == 0 && p.ldb % 8 == 0, "gemm256: K, lda, ldb must be multiples of 8");
    if (p.ab_fp8) {
        MC_CHECK(p.K % 16 == 0 && p.lda % 16 == 0 && p.ldb % 16 == 0, "gemm (fp8 operands): K, lda, ldb must be multiples of 16");
        if (p.stat_partials) hipLaunchKernelGGL((g8::gemm8p_kernel<true, true>), grid, block, 0, st, p, MT, NTl, ktn, nt);
        else hipLaunchKernelGGL((g8::gemm8p_kernel<false, true>), grid, block, 0, st, p, MT, NTl, ktn, nt);
    } else if (p.stat_partials) hipLaunchKernelGGL((g8::gemm8p_kernel<true, false>), grid, block, 0, st, p, MT, NTl, ktn, nt);
    else hipLaunchKernelGGL((g8::gemm8p_kernel<false, false>), grid, block, 0, st, p, MT, NTl, ktn, nt);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
