// 256 x 256 x 64 bf16 MFMA GEMM for gfx950, TN form (both operands K-MAJOR), fp32 output, split-K through a workspace:
//     C[M,N] = sum_k A[k][M]^T . B[k][N]          A[k*lda + m], B[k*ldb + n], k = the long reduction index
// -- the weight gradients of the late-stage 1x1 convolutions and of the BERT linears, dW = dY^T . X with k = pixel / token
// [ref: autograd backward of model/modules/efficientnet_custom.py:104,122,283 and text_encoder.py:47-49 -> BertModel].
//
// Same skeleton as gemm256.hip (8 waves = 2 (M) x 4 (N), wave tile 128 x 64, four 16 KB half-tiles [A0 | A1 | B0 | B1] x 2 stage
// buffers, 4 phases per K tile with one half-tile DMA issue + 16 MFMAs per wave and phase, DMA stream 3 half-tiles ahead
// behind a counted vmcnt(6), wave rows half a phase apart, persistent workgroups with a flat K-tile stream).  Differences:
//   * a K tile is 64 ROWS of the operands; half h of A holds the 2 x 64 output rows (columns of the operand) of C-quadrant
//     row h as [64 k][128 m] bf16 (256 bytes per k row), half h of B the 4 x 32 columns of quadrant column h
//   * MFMA operand fragments come from the row-major image through gfx950's LDS transpose-read (ds_read_b64_tr_b16: a
//     16-lane group reads a 4 (k) x 16 (m) block, lane c receives column c): two reads per fragment.  Conflict-free with
//     the source-side swizzle "16-byte chunk c of k row r sits in slot c ^ 2((r & 3) | ((r >> 1) & 4))" -- the 8 k rows
//     {8g .. 8g+3} U {8g+8 .. 8g+11} a 32-lane bank group touches land in 8 different 32-byte columns; XOR by an EVEN
//     number keeps the 32-byte pairs a transpose-read needs together
//   * rows beyond the K range of a split deliver zeros through the descriptor's num_records; columns beyond M / N are one
//     per-lane flag per output tile (bit 31 of the offset); nothing else distinguishes a partial tile
//   * items = (output tile, K split); all tiles of one split run on ONE XCD at about the same time (they read the same
//     slab of both operands); each item writes its fp32 tile straight from the accumulators (a lane owns 4 consecutive
//     columns: 16-byte stores) into the split-K workspace, combined by gemm.hip's splitk_reduce_kernel (fixed order, optional
//     per-(group, column) factor = the SE gate of the grouped form), or into C when there is one split.
#include "common_hip.h"
#include "../../include/mammoclip_hip.h"

namespace g8t {

constexpr int BM = 256, BN = 256, BK = 64, NTHR = 512;
constexpr int HALF_BYTES = 64 * 256;             // 64 k rows of 256 bytes
constexpr int STAGE_BYTES = 4 * HALF_BYTES;      // 64 KB: [A0 | A1 | B0 | B1]
constexpr int LDS_BYTES = 2 * STAGE_BYTES;       // 128 KB

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(4))) short s4_t;
typedef __attribute__((ext_vector_type(8))) short s8_t;
typedef __attribute__((address_space(3))) s4_t lds_s4_t;

__device__ __forceinline__ void dma16(unsigned voff, u32x4 srd, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                 :: "v"(voff), "s"(srd), "s"(lds_dst) : "memory");
}

#define G8T_BAR()                                    \
    do {                                             \
        asm volatile("" ::: "memory");               \
        __builtin_amdgcn_s_barrier();                \
        asm volatile("" ::: "memory");               \
        __builtin_amdgcn_sched_barrier(0);           \
    } while (0)

struct Item { int mt, nt, split; long long kbeg, kend; int ktn; };

__global__ __launch_bounds__(NTHR, 2) void gemm256_tn_kernel(const mc_gemm_args p, const int MT, const int NT) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // ---- this workgroup's items.  >= 8 splits: XCD x owns the splits s = x (mod 8), its items are ordered split-major,
    // tiles inside (all tiles of a split share an L2); fewer splits: items (split-major) are dealt round-robin to the XCDs
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, S = gridDim.x >> 3;
    const int T = MT * NT;
    const bool by_split = p.splits >= 8;
    const int nloc = by_split ? (p.splits > xcd ? (p.splits - xcd + 7) >> 3 : 0) * T
                              : (T * p.splits > xcd ? (T * p.splits - xcd + 7) >> 3 : 0);
    const int n_items = nloc > slot ? (nloc - slot + S - 1) / S : 0;
    if (n_items == 0) return;

    auto item_at = [&](unsigned it) __attribute__((always_inline)) {
        Item q;
        const unsigned i = (unsigned)slot + it * (unsigned)S;
        const unsigned g = by_split ? i : (unsigned)xcd + 8u * i;
        const unsigned sl = g / (unsigned)T, tile = g - sl * (unsigned)T;
        q.split = by_split ? xcd + 8 * (int)sl : (int)sl;
        q.mt = (int)(tile / (unsigned)NT);
        q.nt = (int)(tile - (unsigned)q.mt * (unsigned)NT);
        // K range of the split (same rule as gemm.hip's gemm_kernel: the workspace reduction is shared)
        const long long ktiles = (p.K + BK - 1) / BK;
        const long long tps = (ktiles + p.splits - 1) / p.splits;
        q.kbeg = (long long)q.split * tps * BK;
        q.kend = q.kbeg + tps * BK;
        if (p.split_group_rows > 0) {
            const long long grp = q.split / p.split_sub, j = q.split % p.split_sub;
            const long long chunk = (p.split_group_rows + p.split_sub - 1) / p.split_sub;
            q.kbeg = grp * p.split_group_rows + j * chunk;
            q.kend = q.kbeg + chunk;
            if (q.kend > (grp + 1) * p.split_group_rows) q.kend = (grp + 1) * p.split_group_rows;
        }
        if (q.kend > p.K) q.kend = p.K;
        const long long len = q.kend - q.kbeg;
        q.ktn = len > 0 ? (int)((len + BK - 1) / BK) : 1;      // an empty split still owes its (zero) partial tile
        return q;
    };

    // ---- DMA source geometry: 2 wave-instructions per half-tile, each fills 4 k rows (1 KiB)
    //   instruction j = i*8 + wave covers k rows 4j + r, r = lane >> 4; LDS slot s = lane & 15 of the row's 256 bytes holds
    //   source chunk c = s ^ fk, fk = 2 (r | ((wave & 2) << 1))    (= 2((k & 3) | ((k >> 1) & 4)) for k = 4j + r)
    //   A half h: chunk c -> operand column (c >> 3)*128 + h*64 + (c & 7)*8;   B half h: (c >> 2)*64 + h*32 + (c & 3)*8
    const int r4 = lane >> 4;
    const int cs = (lane & 15) ^ (2 * (r4 | ((wave & 2) << 1)));
    const int colA = (cs >> 3) * 128 + (cs & 7) * 8, colB = (cs >> 2) * 64 + (cs & 3) * 8;      // + h*64 / h*32
    const unsigned voffA = (unsigned)((wave * 4 + r4) * (int)p.lda * 2 + colA * 2);
    const unsigned voffB = (unsigned)((wave * 4 + r4) * (int)p.ldb * 2 + colB * 2);
    const unsigned i1A = (unsigned)(32 * (int)p.lda * 2), i1B = (unsigned)(32 * (int)p.ldb * 2);  // second instruction: 32 k rows on
    typedef __attribute__((address_space(3))) unsigned int lds_u32_t;
    const unsigned smem_lds = (unsigned)(uintptr_t)(lds_u32_t*)smem;
    const unsigned dma_dst = smem_lds + (unsigned)(wave * 1024);               // + buf*STAGE + which*HALF + i*8192

    struct Prod {
        unsigned long long a, b;        // byte address of the K tile's origin (row k0, column m0 / n0)
        long long nra, nrb;             // bytes from the origin to the end of the last valid k row's valid columns (may be <= 0)
        int mcols, ncols;               // valid columns of this output tile (<= 256)
        int kt, ktn;
        unsigned it;
        bool valid;
    } pd;
    auto prod_tile = [&](unsigned it) __attribute__((always_inline)) {
        pd.it = it; pd.kt = 0;
        pd.valid = it < (unsigned)n_items;
        if (!pd.valid) return;
        const Item q = item_at(it);
        pd.ktn = q.ktn;
        const long long m0 = (long long)q.mt * BM, n0 = (long long)q.nt * BN;
        pd.mcols = p.M - m0 > BM ? BM : (int)(p.M - m0);
        pd.ncols = p.N - n0 > BN ? BN : (int)(p.N - n0);
        pd.a = (unsigned long long)(uintptr_t)p.A + (unsigned long long)((q.kbeg * p.lda + m0) * 2);
        pd.b = (unsigned long long)(uintptr_t)p.B + (unsigned long long)((q.kbeg * p.ldb + n0) * 2);
        const long long rows = q.kend - q.kbeg;
        pd.nra = rows > 0 ? ((rows - 1) * p.lda + pd.mcols) * 2 : 0;
        pd.nrb = rows > 0 ? ((rows - 1) * p.ldb + pd.ncols) * 2 : 0;
    };
    auto prod_next = [&]() __attribute__((always_inline)) {
        if (pd.kt + 1 < pd.ktn) {
            ++pd.kt;
            pd.a += (unsigned long long)(BK * p.lda * 2); pd.b += (unsigned long long)(BK * p.ldb * 2);
            pd.nra -= BK * p.lda * 2; pd.nrb -= BK * p.ldb * 2;
        } else {
            prod_tile(pd.it + 1);
        }
    };
    auto stage = [&](int buf, auto which_c) __attribute__((always_inline)) {
        constexpr int which = decltype(which_c)::value;
        constexpr bool isA = which < 2;
        constexpr int h = which & 1;
        if (!pd.valid) return;
        const unsigned long long base = isA ? pd.a : pd.b;
        const long long nr64 = isA ? pd.nra : pd.nrb;
        const unsigned nr = nr64 <= 0 ? 0u : (nr64 > 0x7fffffffLL ? 0x7fffffffu : (unsigned)nr64);
        const u32x4 srd = {(unsigned)base, (unsigned)(base >> 32) & 0xffffu, nr, 0x00020000u};
        const int col = (isA ? colA + h * 64 : colB + h * 32);
        const unsigned oob = col >= (isA ? pd.mcols : pd.ncols) ? 0x80000000u : 0u;
        const unsigned v0 = (isA ? voffA + h * 128u : voffB + h * 64u) + oob;
        const unsigned dst = dma_dst + (unsigned)(buf * STAGE_BYTES + which * HALF_BYTES);
        dma16(v0, srd, dst);
        dma16(v0 + (isA ? i1A : i1B), srd, dst + 8192u);
    };
    using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;

    // ---- fragment read addresses (transpose-reads).  Lane (i = lane & 15, g = lane >> 4) supplies, for MFMA k step kk and
    // read half q2 (k rows 0-3 / 4-7 of its group of 8), the address of k row 32 kk + 8 g + 4 q2 + (i >> 2), bytes
    // ((blk * 32) ^ swz) + (i & 3) * 8 of that row, where blk = the fragment's 16-column block of the half-tile (A: wm*4 + ii,
    // B: wn*2 + jj) and swz = 32 ((i >> 2) | ((g & 1) << 2)) -- the swizzle term does not depend on kk or q2.
    const int li = lane & 15, lg = lane >> 4;
    const unsigned swz = (unsigned)(32 * ((li >> 2) | ((lg & 1) << 2)));
    const unsigned rowb = (unsigned)((8 * lg + (li >> 2)) * 256 + (li & 3) * 8);
    const unsigned fA = rowb + (((unsigned)(wm * 128)) ^ swz);      // ^ (ii << 5);  + kk*8192 + q2*1024
    const unsigned fB = rowb + (((unsigned)(wn * 64)) ^ swz);       // ^ (jj << 5)

    f32x4_t acc[8][4];
    bf16x8_t a0[4][2], a1[4][2], b0[2][2], b1[2][2];

    auto tr8 = [&](const unsigned char* ptr) __attribute__((always_inline)) {
        const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(ptr));
        const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(ptr + 1024));
        const s8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    };
    auto read_a2 = [&](bf16x8_t (&af)[4][2], int buf, int h, int i_lo) __attribute__((always_inline)) {
        const unsigned char* base = smem + buf * STAGE_BYTES + h * HALF_BYTES;
#pragma unroll
        for (int ii = i_lo; ii < i_lo + 2; ++ii) {
            af[ii][0] = tr8(base + (fA ^ (unsigned)(ii << 5)));
            af[ii][1] = tr8(base + (fA ^ (unsigned)(ii << 5)) + 8192);
        }
    };
    auto read_b = [&](bf16x8_t (&bf)[2][2], int buf, int h) __attribute__((always_inline)) {
        const unsigned char* base = smem + buf * STAGE_BYTES + (2 + h) * HALF_BYTES;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            bf[jj][0] = tr8(base + (fB ^ (unsigned)(jj << 5)));
            bf[jj][1] = tr8(base + (fB ^ (unsigned)(jj << 5)) + 8192);
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
                for (int j = 0; j < 2; ++j)
                    acc[ih * 4 + i][jh * 2 + j] =
                        MC_MFMA_16x16x32(bf[j][kk], af[i][kk], acc[ih * 4 + i][jh * 2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    unsigned cit = 0;
    int ckt = 0;
    Item ci = item_at(0);

    // ---- pipeline prologue: tile 0 completely, tile 1 without its last half (stream order per tile: B0, A0, B1, A1)
    prod_tile(0);
    stage(0, C2{}); stage(0, C0{}); stage(0, C3{}); stage(0, C1{});
    prod_next();
    stage(1, C2{}); stage(1, C0{}); stage(1, C3{});
    if (pd.valid) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G8T_BAR();
    if (wm == 1) G8T_BAR();                     // wave row 1 runs half a phase behind wave row 0 from here on

#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    bool drain = false;
    int buf = 0;
    for (;;) {
        // ------------------------------------------------ phase 1: quadrant (0,0)
        read_b(b0, buf, 0);                                      // 8 transpose-reads
        __builtin_amdgcn_sched_barrier(0);
        read_a2(a0, buf, 0, 0);                                  // 8
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");       // the B0 reads have retired (B0 is re-staged next phase)
        __builtin_amdgcn_sched_barrier(0);
        read_a2(a0, buf, 0, 2);                                  // 8
        stage(buf ^ 1, C1{});                                    // A1 of tile t+1
        prod_next();                                             // pd = tile t+2
        G8T_BAR();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma_quad(a0, b0, C0{}, C0{});
        G8T_BAR();
        // ------------------------------------------------ phase 2: quadrant (0,1)
        read_b(b1, buf, 1);
        stage(buf, C2{});                                        // B0 of tile t+2
        G8T_BAR();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma_quad(a0, b1, C0{}, C1{});
        G8T_BAR();
        // ------------------------------------------------ phase 3: quadrant (1,1)
        read_a2(a1, buf, 1, 0);
        read_a2(a1, buf, 1, 2);
        stage(buf, C0{});                                        // A0 of tile t+2
        G8T_BAR();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma_quad(a1, b1, C1{}, C1{});
        G8T_BAR();
        // ------------------------------------------------ phase 4: quadrant (1,0)
        if (drain) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // epilogue stores share the counter: nothing can be counted
            stage(buf, C3{});
            drain = false;
        } else {
            stage(buf, C3{});                                    // B1 of tile t+2
            if (pd.valid) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        G8T_BAR();
        mma_quad(a1, b0, C1{}, C0{});
        G8T_BAR();

        buf ^= 1;
        ++ckt;
        if (ckt == ci.ktn) {
            // ============================================ epilogue of item ci: fp32 tile from the accumulators
            const long long m0 = (long long)ci.mt * BM;
            const int n0 = ci.nt * BN;
            const bool to_ws = p.splits > 1;
            float* const Cb = to_ws ? p.splitk_ws + (long long)ci.split * p.M * p.N : reinterpret_cast<float*>(p.C);
            const long long ldc = to_ws ? (long long)p.N : p.ldc;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const long long m = m0 + wm * 128 + i * 16 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
                    if (m < p.M && n < p.N) *reinterpret_cast<f32x4_t*>(Cb + m * ldc + n) = acc[i][j];
                    acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                }
            }
            drain = true;
            ckt = 0;
            ++cit;
            if (cit >= (unsigned)n_items) break;
            ci = item_at(cit);
        }
    }
    if (wm == 0) G8T_BAR();                     // balance the extra barrier of wave row 1
}

}  // namespace g8t

// ---- host side ------------------------------------------------------------------------------------------------------
#include <cstdlib>
static int g8t_mode() { const char* e = getenv("MC_GEMM_256TN"); return e ? atoi(e) : 1; }   // 0 never, 1 by rule, 2 whenever possible

// cost model for the number of K splits: XCD x runs the splits s = x (mod 8), 32 workgroups per XCD, one item per
// workgroup and round; an item costs its K tiles plus ~6 K-tile times of prologue / fp32 tile store
// group_rows > 0 (grouped form: the reduction is cut at multiples of group_rows, e.g. rows per image): returns the number
// of sub-splits PER GROUP (splits = groups * sub); otherwise the number of splits.
extern "C" int mc_gemm256_tn_splits(long long M, long long N, long long K, long long group_rows) {
    const long long T = ((M + 255) / 256) * ((N + 255) / 256);
    const long long groups = group_rows > 0 ? (K + group_rows - 1) / group_rows : 1;
    const long long len = group_rows > 0 ? group_rows : K;     // rows one group / the whole problem reduces over
    const long long ktiles = (len + 63) / 64;
    int best = 1;
    double best_cost = 1e300;
    for (int s = 1; s <= 64; ++s) {
        if (s > 1 && ktiles / s < 8) break;
        const long long splits = groups * s;
        if (splits > 4096) break;
        const long long ix = splits >= 8 ? ((splits + 7) / 8) * T : (splits * T + 7) / 8;   // items on the busiest XCD
        const long long rounds = (ix + 31) / 32;
        // + the split-K workspace: every split writes an fp32 [M, N] partial and the reduce reads it back -- at ~4 TB/s, in
        // units of one K-tile time (~1.55 us per 256 x 256 x 64 step at the kernel's in-loop rate).  Round 5 sweep
        // (scripts/tn_split_sweep.py): 512 x 3072 over 44544 rows ran 204 us with the 32 splits the round count alone asks for
        // (403 MB of partials against 319 MB of operands) and 163 us with 8; every other model shape keeps its choice.
        // (weight 0.5: with the full term 768 x 768 over 16384 rows moved from 24 to 16 splits, 43 -> 47 us in the same sweep)
        const double ws_units = 0.5 * (double)splits * (double)M * (double)N * 8.0 / 4.0e6 / 1.55;
        const double cost = (double)rounds * ((double)((ktiles + s - 1) / s) + 6.0) + ws_units;
        if (cost < best_cost * 0.999) { best_cost = cost; best = s; }
    }
    return best;
}

extern "C" int mc_gemm256_tn_eligible(const mc_gemm_args* a) {
    const mc_gemm_args& p = *a;
    const int mode = g8t_mode();
    if (mode == 0) return 0;
    if (!p.a_kmajor || !p.b_kmajor || !p.c_f32 || p.pro_operand != 0 || p.nb2 > 1 || p.batch > 1 || p.ab_fp8 || p.bias || p.R) return 0;
    if (p.alpha != 0.f && p.alpha != 1.f) return 0;
    if (p.splits > 1 && !p.splitk_ws) return 0;
    if (p.splits <= 1 && p.c_atomic) return 0;
    if (p.M % 8 || p.N % 8 || p.lda % 8 || p.ldb % 8 || p.lda < p.M || p.ldb < p.N) return 0;
    if (p.lda * 2 * 64 >= (1LL << 30) || p.ldb * 2 * 64 >= (1LL << 30)) return 0;             // 31-bit offsets inside a K tile
    if (p.splits <= 1 && p.ldc % 4) return 0;
    const long long T = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    if (T * (p.splits > 0 ? p.splits : 1) >= (1LL << 24)) return 0;
    if (mode == 2) return 1;
    // worth it when the tiles are reasonably filled and the reduction is long (128 x 384 over 173280 rows -- 37 % of its two
    // tiles -- still runs ~2x the 128 x 128 split-K instance's 116 TFLOP/s)
    const double useful = (double)(p.M * p.N) / (double)(T * 65536);
    return p.K >= 2048 && useful >= 0.35 && p.M >= 128 && p.N >= 128;
}

extern "C" int mc_gemm256_tn_launch(const mc_gemm_args* a, void* stream) {
    mc_gemm_args p = *a;
    if (p.splits <= 0) p.splits = 1;
    const int MT = (int)((p.M + 255) / 256), NT = (int)((p.N + 255) / 256);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(g8t::gemm256_tn_kernel, dim3(256), dim3(g8t::NTHR), 0, st, p, MT, NT);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
