// 256 x 256 x 64 bf16 MFMA GEMM for gfx950, plain NT operands, bf16 output:
//     C[z][M,N] = alpha * A[z][M,K] . B[z][N,K]^T (+ bias[n]) (+ R[m,n]),   optional per-column sum / sum-of-squares partials
// -- the late-stage 1x1 convolutions of EfficientNet (efficientnet_custom.py:104,122,283: forward and data gradient) and
// the BERT linears (text_encoder.py:47-49 -> BertModel), i.e. every launch that used to land on the 128 x 128 tile kernel
// of gemm.hip with plain k-contiguous operands.
//
// Structure (one workgroup of 8 waves per CU, all 160 KB of LDS):
//   * waves 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 fragments of v_mfma_f32_16x16x32_bf16 (128 accumulator registers)
//   * a K tile (64 deep) of A and of B is staged as FOUR 16 KB half-tiles [A0 | A1 | B0 | B1]; half h of A holds, for
//     BOTH wave rows, the 64 rows of C-quadrant row h (tile rows wm*128 + h*64 ..), half h of B the 32 columns of quadrant
//     column h of all four wave columns: a whole half-tile is finished being read after one phase (see below)
//   * direct-to-LDS DMA (global_load_lds_dwordx4, inline asm so the compiler keeps no book on it); the DMA writes
//     lane-linearly, so the bank-conflict swizzle sits on the SOURCE address: slot s of LDS row r holds 16-byte chunk
//     s ^ ((r >> 1) & 7) of that row's 128 bytes (conflict-free for ds_read_b128's 16-lane groups)
//   * 4 phases per K tile, one C quadrant (16 MFMAs per wave) and ONE half-tile DMA issue per phase, two LDS stage
//     buffers; the DMA stream runs 3 half-tiles ahead with a counted s_waitcnt vmcnt(6) once per K tile -- it never
//     drains in the main loop -- and crosses output-tile boundaries (the K tiles of all output tiles of a workgroup form
//     one flat stream):
//         phase 1: read B0 (4) + A0 (8) fragments, stage A1(t+1);  quadrant (0,0)
//         phase 2: read B1 (4),                    stage B0(t+2);  quadrant (0,1)
//         phase 3: read A1 (8),                    stage A0(t+2);  quadrant (1,1)
//         phase 4: (B0 still in registers)         stage B1(t+2);  vmcnt(6): all of tile t+1 has landed;  quadrant (1,0)
//     Hazards: a half-tile is re-staged >= 2 phases after its last ds_read, or 1 phase after (B0) with the reads retired
//     by an explicit lgkmcnt before the reading phase's first barrier; a staged buffer is read one phase after the wait
//     that retires it.  The two wave rows run half a phase apart (one extra barrier for wave row 1 up front): while one
//     wave of a SIMD issues MFMAs its partner reads LDS / issues DMA; s_setprio favours the MFMA wave.
//   * epilogue through the 32 KB of LDS the stages leave free, in four 64-row slabs (16-byte coalesced stores, optional
//     bias / residual / BatchNorm column statistics), while the DMAs of the next output tile are already in flight
//   * persistent workgroups, XCD-aware work order: all column tiles of a row block run on ONE XCD at about the same
//     time (workgroup id mod 8 = XCD), so the activation rows are fetched from HBM once and shared through that L2.
#include "common_hip.h"
#include "../../include/mammoclip_hip.h"

namespace g256 {

constexpr int BM = 256, BN = 256, BK = 64, NTHR = 512;
constexpr int HALF_BYTES = 128 * 128;            // 128 LDS rows of 128 bytes (64 bf16)
constexpr int OFF_A0 = 0, OFF_A1 = HALF_BYTES, OFF_B0 = 2 * HALF_BYTES, OFF_B1 = 3 * HALF_BYTES;
constexpr int STAGE_BYTES = 4 * HALF_BYTES;      // 64 KB
constexpr int EPI_OFF = 2 * STAGE_BYTES;         // 128 KB
constexpr int EPI_BYTES = 64 * BN * 2;           // one slab: 64 rows x 256 columns bf16 = 32 KB
constexpr int LDS_BYTES = EPI_OFF + EPI_BYTES;   // 160 KB

// direct-to-LDS DMA of 16 bytes per lane: wave-uniform 64-bit base in SGPRs + 32-bit per-lane byte offset; lane l's bytes
// land at lds_dst_wave_base + 16 l.  Inline assembly: the compiler keeps no book on it (the kernel counts vmcnt itself).
__device__ __forceinline__ void glds16_s(unsigned voff_bytes, const void* sbase, unsigned lds_dst_wave_base) {
    const unsigned lds = __builtin_amdgcn_readfirstlane(lds_dst_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff_bytes), "s"(lds), "s"(sbase) : "memory");
}

// Two main-loop schedules share everything else in this file:
//   G256_PHASED   the 4-phase / counted-vmcnt schedule described above
//   (default)     ONE barrier per K tile: wait for tile t, barrier, issue the whole of tile t+1 (8 DMA instructions per
//                 wave, 64 KB in flight per CU), then the four quadrants of tile t back to back; the two waves of a SIMD
//                 drift apart on their own and cover each other's LDS reads and DMA issue.
// Measured on MI355X (scripts/g256bench.hip, uniform random operands): the one-barrier schedule is 10-25 % faster on every
// shape of the model (8192^3: 853 vs 772 TFLOP/s; 44544x3072x512: 728 vs 657; 173280x176x1056: 471 vs 405): with 8 barriers
// per K tile the phased schedule's load phases (2 DMA issues of ~150 cycles each + up to 12 ds_read_b128) are longer than
// the 16-MFMA phases they are meant to hide under (phase profile, -DG256_PROF), so the barriers expose them.
#ifndef G256_PHASED
#define G256_SIMPLE
#endif
#ifdef G256_SIMPLE
#define G256_NOSTAGGER
#endif
#ifdef G256_PROF
__device__ unsigned long long g_g256_prof[2][12];   // developer phase profile (scripts/g256bench.hip): [wave row][segment]
#define GP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[i] += t_ - tprof; tprof = t_; } while (0)
#else
#define GP(i)
#endif

#define G256_BAR()                                   \
    do {                                             \
        asm volatile("" ::: "memory");               \
        __builtin_amdgcn_s_barrier();                \
        asm volatile("" ::: "memory");               \
        __builtin_amdgcn_sched_barrier(0);           \
    } while (0)

// position of one K tile in the flat stream of a workgroup (all wave-uniform)
struct TilePos {
    const unsigned char* a; // A + (z*sA + m0*lda + k0) elements
    const unsigned char* b; // B + (z*sB + n0*ldb + k0) elements
    int mrem, nrem, krem;   // rows / columns / k left from (m0, n0, k0)
    bool valid;
};

struct Work {               // per-workgroup work list (wave-uniform)
    int xcd, slot, S, NTl, MT, ktn;
    long long n_items;      // items of this workgroup
};

__device__ __forceinline__ void item_coords(const Work& w, long long it, int& z, int& mt, int& nt) {
    const long long u = w.slot + it * w.S;
    const long long ru = u / w.NTl;
    nt = (int)(u - ru * w.NTl);
    const long long unit = w.xcd + 8 * ru;
    z = (int)(unit / w.MT);
    mt = (int)(unit - (long long)z * w.MT);
}

// FP8: the operands are OCP e4m3 bytes (per-tensor scaled; the product of the two dequantisation scales arrives through
// alpha / alpha_dev).  A K tile is still 128 BYTES per row -- 128 fp8 values -- so staging, LDS image and fragment reads are
// byte for byte those of the bf16 kernel; a 16-byte fragment read holds 16 consecutive k values of the lane's row and
// feeds TWO v_mfma_f32_16x16x32_fp8_fp8 (bytes 0-7 and 8-15: the k order inside the 128-deep tile is a permutation that
// A and B share, which a dot product does not see).  Twice the flops per staged byte of the bf16 kernel.
template <bool STATS, bool FP8>
__global__ __launch_bounds__(NTHR, 2) void gemm256_kernel(const mc_gemm_args p, const int MT, const int NTl, const int ktn) {
    constexpr int ES = FP8 ? 1 : 2;             // bytes per operand element
    constexpr int KT = 128 / ES;                // K tile in elements (128 bytes per LDS row)
    constexpr int CH = 16 / ES;                 // elements per 16-byte chunk
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    Work wk;
    wk.xcd = blockIdx.x & 7; wk.slot = blockIdx.x >> 3; wk.S = gridDim.x >> 3; wk.NTl = NTl; wk.MT = MT; wk.ktn = ktn;
    {
        const long long RB = (long long)p.batch * MT;                         // (batch, row block) units
        const long long ux = RB > wk.xcd ? (RB - wk.xcd + 7) >> 3 : 0;        // units of this XCD
        const long long nloc = ux * NTl;
        wk.n_items = nloc > wk.slot ? (nloc - wk.slot + wk.S - 1) / wk.S : 0;
    }
    if (wk.n_items == 0) return;
    const long long total = wk.n_items * ktn;                                 // K tiles in this workgroup's stream

    auto make_pos = [&](long long it, int kt) __attribute__((always_inline)) {
        TilePos q;
        q.valid = it < wk.n_items;
        int z = 0, mt = 0, nt = 0;
        if (q.valid) item_coords(wk, it, z, mt, nt);
        const long long m0 = (long long)mt * BM, n0 = (long long)nt * BN, k0 = (long long)kt * KT;
        q.a = reinterpret_cast<const unsigned char*>(p.A) + ((long long)z * p.sA1 + m0 * p.lda + k0) * ES;
        q.b = reinterpret_cast<const unsigned char*>(p.B) + ((long long)z * p.sB1 + n0 * p.ldb + k0) * ES;
        const long long mr = p.M - m0, nr = p.N - n0, kr = p.K - k0;
        q.mrem = mr > BM ? BM : (int)mr; q.nrem = nr > BN ? BN : (int)nr; q.krem = kr > KT ? KT : (int)kr;
        return q;
    };

    // ---- DMA source geometry of this thread: 2 wave-instructions per half-tile, each fills 8 LDS rows (1 KiB)
    //   instruction q of wave w fills LDS rows lr = (q*8 + w)*8 + (lane >> 3); lane & 7 = slot s, chunk c = s ^ ((lr >> 1) & 7)
    //   A half h: LDS row lr <-> tile row (lr >> 6)*128 + h*64 + (lr & 63);  B half h: tile col (lr >> 5)*64 + h*32 + (lr & 31)
    //   => instruction q / half h move the tile row by q*128 + h*64 (A) resp. the tile column by q*128 + h*32 (B): a
    //   wave-uniform term that goes into the scalar base; the lane keeps ONE byte offset per operand
    auto geo = [&](int q, int h, int& ra, int& rb, int& ck) __attribute__((always_inline)) {
        const int g = q * 8 + wave;
        const int lr = g * 8 + (lane >> 3);
        ck = (lane & 7) ^ ((lr >> 1) & 7);
        ra = (lr >> 6) * 128 + h * 64 + (lr & 63);
        rb = (lr >> 5) * 64 + h * 32 + (lr & 31);
    };
    unsigned voffA, voffB;
    {
        int ra, rb, ck;
        geo(0, 0, ra, rb, ck);
        voffA = (unsigned)((ra * p.lda + ck * CH) * ES);
        voffB = (unsigned)((rb * p.ldb + ck * CH) * ES);
    }
    typedef __attribute__((address_space(3))) unsigned int lds_u32_t;
    const unsigned smem_lds = (unsigned)(uintptr_t)(lds_u32_t*)smem;

    // stage half-tile `which` (0 = A0, 1 = A1, 2 = B0, 3 = B1) of the tile at `q` into stage buffer `buf`
    auto stage = [&](const TilePos& q, int buf, auto which_c) __attribute__((always_inline)) {
        constexpr int which = decltype(which_c)::value;
        constexpr bool isA = which < 2;
        constexpr int h = which & 1;
        if (!q.valid) return;
#ifdef G256_NODMA
        if (q.valid) return;                    // diagnostic build: fragments + MFMA speed with no fill traffic (wrong results)
#endif
        const unsigned dst = smem_lds + buf * STAGE_BYTES + which * HALF_BYTES;
        const bool full = q.mrem == BM && q.nrem == BN && q.krem == KT;
        if (full) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned char* sb = isA ? q.a + (long long)(i * 128 + h * 64) * p.lda * ES : q.b + (long long)(i * 128 + h * 32) * p.ldb * ES;
                glds16_s(isA ? voffA : voffB, sb, dst + (unsigned)((i * 8 + wave) * 1024));
            }
        } else {
            // partial tile.  Rows / columns beyond the matrix are CLAMPED to the last valid one (row m of C depends only on
            // row m of A, column n only on row n of B, and those outputs are never stored); chunks beyond K must be ZERO
            // in LDS: their lanes are masked off the DMA (EXEC) and store zeros to their slot instead.  Every 8-lane group
            // covers all 8 chunks of a row and K % 8 == 0, so every wave-instruction keeps active lanes: the number of DMA
            // instructions per wave -- what the counted vmcnt relies on -- is the same as on the full-tile path.
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int ra, rb, ck;
                geo(i, h, ra, rb, ck);
                const int rr = isA ? (ra < q.mrem ? ra : q.mrem - 1) : (rb < q.nrem ? rb : q.nrem - 1);
                const unsigned voff = (unsigned)((rr * (isA ? p.lda : p.ldb) + ck * CH) * ES);
                const unsigned d = dst + (unsigned)((i * 8 + wave) * 1024);
                if (ck * CH < q.krem) glds16_s(voff, isA ? q.a : q.b, d);
                else *reinterpret_cast<uint4*>(smem + (d - smem_lds) + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };
    using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;

    // ---- fragment read addresses: LDS row r = base + (lane & 15), chunk (kk*4 + (lane >> 4)) ^ ((r >> 1) & 7); the row
    // bases are multiples of 16, so the swizzle term depends on the lane only; kk = 1 flips chunk bit 2 (byte 64)
    const int frow = lane & 15;
    const unsigned fsw0 = (unsigned)((((lane >> 4) ^ ((frow >> 1) & 7)) << 4));
    const unsigned fA0 = (unsigned)((wm * 64 + frow) * 128) + fsw0;     // + i*2048 (16 rows); kk = 1: byte bit 6 flipped
    const unsigned fB0 = (unsigned)((wn * 32 + frow) * 128) + fsw0;     // + j*2048
    const unsigned fA1 = fA0 ^ 64u, fB1 = fB0 ^ 64u;                    // (i*2048 / j*2048 never touch bit 6: plain immediates)

    f32x4_t acc[8][4];
    bf16x8_t a0[4][2], a1[4][2], b0[2][2], b1[2][2];

    long long t_cur = 0;
    auto read_a = [&](bf16x8_t (&af)[4][2], int buf, int h) __attribute__((always_inline)) {
#ifdef G256_NOREAD
        if (t_cur > 0) { asm volatile("" : "+v"(af[0][0]), "+v"(af[1][0]), "+v"(af[2][0]), "+v"(af[3][0])); return; }
#endif
        const unsigned char* base = smem + buf * STAGE_BYTES + (h ? OFF_A1 : OFF_A0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i][0] = *reinterpret_cast<const bf16x8_t*>(base + fA0 + i * 2048);
            af[i][1] = *reinterpret_cast<const bf16x8_t*>(base + fA1 + i * 2048);
        }
    };
    auto read_b = [&](bf16x8_t (&bf)[2][2], int buf, int h) __attribute__((always_inline)) {
#ifdef G256_NOREAD
        if (t_cur > 0) { asm volatile("" : "+v"(bf[0][0]), "+v"(bf[1][0])); return; }
#endif
        const unsigned char* base = smem + buf * STAGE_BYTES + (h ? OFF_B1 : OFF_B0);
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
#ifndef G256_NOPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
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
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j][kk], af[i][kk], acc[ih * 4 + i][jh * 2 + j], 0, 0, 0);
                    }
                }
#ifndef G256_NOPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };

    // alpha_dev: a device-resident factor (the product of the fp8 dequantisation scales); read through the scalar cache
    // and pinned before the loop (a vector load pending at the loop head would drain the DMA stream, see the epilogue)
    float alpha = p.alpha;
    if (p.alpha_dev) {
        alpha *= __builtin_nontemporal_load(p.alpha_dev);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(alpha));
    }

    // current (consumed) item
    long long cit = 0;
    int ckt = 0;
    int cz, cmt, cnt;
    item_coords(wk, 0, cz, cmt, cnt);

    // producer positions: p1 = tile t+1, p2 = tile t+2
    long long pit = 0; int pkt = 0;
    auto next_pos = [&]() __attribute__((always_inline)) {
        ++pkt;
        if (pkt == ktn) { pkt = 0; ++pit; }
        return make_pos(pit, pkt);
    };
    TilePos p0 = make_pos(0, 0);
    TilePos p1 = next_pos();
    TilePos p2 = next_pos();

    // ---- pipeline prologue: tile 0 completely, tile 1 without its last half (stream order per tile: B0, A0, B1, A1)
    stage(p0, 0, C2{}); stage(p0, 0, C0{}); stage(p0, 0, C3{}); stage(p0, 0, C1{});
#ifndef G256_SIMPLE
    stage(p1, 1, C2{}); stage(p1, 1, C0{}); stage(p1, 1, C3{});
    if (p1.valid) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G256_BAR();
#endif
#ifndef G256_NOSTAGGER
    if (wm == 1) G256_BAR();                    // wave row 1 runs half a phase behind wave row 0 from here on
#endif

#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    bool drain = false;                         // stores of an epilogue are in the queue: the next tile wait drains it
    int buf = 0;
#ifdef G256_PROF
    unsigned long long pacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprof = __builtin_amdgcn_s_memtime();
#endif
    for (long long t = 0; t < total; ++t) {
        t_cur = t;
#ifdef G256_SIMPLE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G256_BAR();
        stage(p1, buf ^ 1, C2{}); stage(p1, buf ^ 1, C0{}); stage(p1, buf ^ 1, C3{}); stage(p1, buf ^ 1, C1{});
        read_b(b0, buf, 0);
        read_a(a0, buf, 0);
        mma_quad(a0, b0, C0{}, C0{});
        read_b(b1, buf, 1);
        mma_quad(a0, b1, C0{}, C1{});
        read_a(a1, buf, 1);
        mma_quad(a1, b1, C1{}, C1{});
        mma_quad(a1, b0, C1{}, C0{});
        drain = false;
#else
        // ------------------------------------------------ phase 1: quadrant (0,0)
        read_b(b0, buf, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(a0, buf, 0);
        GP(8);
        stage(p1, buf ^ 1, C1{});                                // A1 of tile t+1
        GP(9);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");       // the 4 B0 reads have retired (B0 is re-staged next phase)
        GP(0);
        G256_BAR();
        GP(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        GP(2);
        mma_quad(a0, b0, C0{}, C0{});
        GP(3);
        G256_BAR();
        GP(4);
        // ------------------------------------------------ phase 2: quadrant (0,1)
        read_b(b1, buf, 1);
        GP(8);
        stage(p2, buf, C2{});                                    // B0 of tile t+2
        GP(9);
        G256_BAR();
        GP(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        GP(2);
        mma_quad(a0, b1, C0{}, C1{});
        GP(3);
        G256_BAR();
        GP(4);
        // ------------------------------------------------ phase 3: quadrant (1,1)
        read_a(a1, buf, 1);
        GP(8);
        stage(p2, buf, C0{});                                    // A0 of tile t+2
        GP(9);
        G256_BAR();
        GP(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        GP(2);
        mma_quad(a1, b1, C1{}, C1{});
        GP(3);
        G256_BAR();
        GP(4);
        // ------------------------------------------------ phase 4: quadrant (1,0)
        if (drain) {
            // first tile after an epilogue: its global stores share the counter with the DMAs and may complete out of
            // order with them, so nothing can be counted -- wait for everything issued so far (up to A0(t+2), issued a
            // phase ago), THEN issue B1(t+2): the next tile's vmcnt(6) is exact again
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stage(p2, buf, C3{});
            drain = false;
        } else {
            stage(p2, buf, C3{});                                // B1 of tile t+2
            if (p2.valid) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");           // everything up to A1(t+1) has landed
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        GP(5);
        G256_BAR();
        GP(1);
        mma_quad(a1, b0, C1{}, C0{});
        GP(3);
        G256_BAR();
        GP(4);
#ifdef G256_PROF
        pacc[7] += 1;
#endif
#endif  // G256_SIMPLE

        p1 = p2;
        p2 = next_pos();
        buf ^= 1;
        ++ckt;
        if (ckt == ktn) {
            // ============================================ epilogue of item (cz, cmt, cnt)
#ifndef G256_NOSTAGGER
            if (wm == 0) G256_BAR();            // un-stagger: wave row 0 waits for wave row 1 to finish its last quadrant
#endif
            unsigned char* const etile = smem + EPI_OFF;
            const long long m0 = (long long)cmt * BM;
            const int n0 = cnt * BN;
            bf16_t* const Cb = reinterpret_cast<bf16_t*>(p.C) + (long long)cz * p.sC1;
            const bf16_t* const Rb = p.R ? p.R + (long long)cz * p.sC1 : nullptr;
            const float* bias = p.bias ? p.bias + (long long)cz * p.bias_stride1 : nullptr;
            const int cc = tid & 31, r0 = tid >> 5;              // copy-out role: 16-byte chunk cc of rows r0 + 16*pass
            const int ncol = n0 + cc * 8;
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
                f32x4_t t[4];
                const float* bp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
                    bp[j] = bias + (n < p.N ? n : 0);
                }
                asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"
                             "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
                             : "v"(bp[0]), "v"(bp[1]), "v"(bp[2]), "v"(bp[3]) : "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[j][r] = n < p.N ? t[j][r] : 0.f;
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
                G256_BAR();
                if (ncol < p.N) {
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const int row = r0 + ps * 16;
                        const long long m = m0 + s * 64 + row;
                        if (m < p.M) {
                            uint4 v = *reinterpret_cast<const uint4*>(etile + row * (BN * 2) + ((cc ^ (row & 31)) << 4));
                            if (Rb) {
                                // (inline assembly with its own wait: a load the compiler tracks inside this loop would
                                // make it drain vmcnt at the loop head -- and with it the DMA stream -- on every K tile)
                                float f[8], g[8];
                                unpack8(v, f);
                                uint4 rv;
                                asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)"
                                             : "=&v"(rv) : "v"(Rb + m * p.ldr + ncol) : "memory");
                                unpack8(rv, g);
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
                            *reinterpret_cast<uint4*>(Cb + m * p.ldc + ncol) = v;
                        }
                    }
                }
                G256_BAR();
            }
            if (STATS) {
                // column statistics of this output tile: 16 row groups -> one value per column, fixed order
                float* red = reinterpret_cast<float*>(etile);            // [16][256][2] floats = 32 KB
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    red[(r0 * BN + cc * 8 + q) * 2 + 0] = csum[q];
                    red[(r0 * BN + cc * 8 + q) * 2 + 1] = csq[q];
                }
                G256_BAR();
                if (tid < BN) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { s1 += red[(r * BN + tid) * 2]; s2 += red[(r * BN + tid) * 2 + 1]; }
                    const int n = n0 + tid;
                    if (n < p.N) {
                        float* dst = p.stat_partials + ((long long)cz * MT + cmt) * 2 * p.N;
                        dst[n] = s1;
                        dst[p.N + n] = s2;
                    }
                }
                G256_BAR();
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            drain = true;
            ckt = 0;
            ++cit;
            if (cit < wk.n_items) item_coords(wk, cit, cz, cmt, cnt);
#ifndef G256_NOSTAGGER
            if (wm == 1) G256_BAR();            // restore the half-phase stagger
#endif
            GP(6);
        }
    }
#ifdef G256_PROF
    if (lane == 0 && wn == 0)
        for (int q = 0; q < 12; ++q) atomicAdd(&g_g256_prof[wm][q], pacc[q]);
#endif
#ifndef G256_NOSTAGGER
    if (wm == 0) G256_BAR();                    // balance the extra barrier of wave row 1
#endif
}

}  // namespace g256

// ---- host side ------------------------------------------------------------------------------------------------------
// eligibility + launch, called from mc_gemm_bf16 (gemm.hip) for plain NT bf16-output problems
#include <cstdlib>
// MC_GEMM_256: 0 = never, 1 = by the size rule below (default), 2 = whenever the layout allows (tests / A-B runs)
static int g256_mode() { const char* e = getenv("MC_GEMM_256"); return e ? atoi(e) : 1; }
extern "C" int mc_gemm256_eligible(const mc_gemm_args* a) {
    const mc_gemm_args& p = *a;
    const int mode = g256_mode();
    if (mode == 0 && !p.ab_fp8) return 0;
    if (p.a_kmajor || p.b_kmajor || p.c_f32 || p.pro_operand != 0 || p.splits > 1 || p.nb2 > 1) return 0;
    if (p.lda * 128 + 64 >= (1LL << 30) || p.ldb * 128 + 64 >= (1LL << 30)) return 0;      // 32-bit lane byte offsets
    if (p.ab_fp8) return 1;                     // the fp8 operand path exists only here
    if (mode == 2) return 1;
    if (p.N < 96 || p.K < 64 || p.M < 256) return 0;
    const long long MT = (p.M + 255) / 256, NTl = (p.N + 255) / 256;
    const long long items = (long long)(p.batch > 0 ? p.batch : 1) * MT * NTl;
    if (items < 200) return 0;
    // One 256 x 256 tile per CU and round.  Against the 128 x 128 kernel of gemm.hip (two workgroups per CU, four times
    // finer work granularity) the big tile wins where its rounds are well filled and little of the tile is padding
    // (A/B on MI355X, scripts/gemm256_bench.py): e.g. 44544 x 3072 x 512  742 vs 638 TFLOP/s, 16384 x 3072 x 768  869 vs
    // 740, 8192^3  1059 vs 873; it loses where the last round is mostly idle or N is a poor fit (44544 x 304 x 1824:
    // 348 tiles on 256 CUs, 41 % padding: 347 vs 489).  One column tile wide (N <= 256) with a long reduction is the
    // HBM-bound case -- the activation operand streams exactly once: 173280 x 176 x 1056  488 vs 442.
    const long long rounds = (items + 255) / 256;
    const double occ = (double)items / (double)(rounds * 256), useful = (double)p.N / (double)(NTl * 256);
    if (occ * useful >= 0.88) return 1;
    if (NTl == 1 && p.N >= 160 && p.K >= 768 && occ >= 0.85) return 1;
    return 0;
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
    const dim3 grid(256), block(g256::NTHR);
    if (p.ab_fp8) {
        MC_CHECK(p.K % 16 == 0 && p.lda % 16 == 0 && p.ldb % 16 == 0, "gemm (fp8 operands): K, lda, ldb must be multiples of 16");
        if (p.stat_partials) hipLaunchKernelGGL((g256::gemm256_kernel<true, true>), grid, block, 0, st, p, MT, NTl, ktn);
        else hipLaunchKernelGGL((g256::gemm256_kernel<false, true>), grid, block, 0, st, p, MT, NTl, ktn);
    } else if (p.stat_partials) hipLaunchKernelGGL((g256::gemm256_kernel<true, false>), grid, block, 0, st, p, MT, NTl, ktn);
    else hipLaunchKernelGGL((g256::gemm256_kernel<false, false>), grid, block, 0, st, p, MT, NTl, ktn);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
