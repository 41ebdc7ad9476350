// Row-streaming bf16 MFMA GEMM for the HBM-bound pointwise convolutions of the early EfficientNet stages:
//     C[M,N] = pro(X)[M,K] . W[N,K]^T  (+ R),   N <= 256, K <= 384,  M = pixels (millions)
// [ref: efficientnet_custom.py:104 (_expand_conv), :122 (_project_conv) and their data gradients]
//
// These layers move ~10x more bytes than a roofline-balanced GEMM (arithmetic intensity 20-110 FLOP/B against a
// ridge of ~310), so the kernel is built like a streaming kernel, not like a tiled GEMM:
//   * the whole weight matrix is staged ONCE per workgroup into LDS, already in MFMA operand-fragment order
//   * every wave owns whole output rows: it streams 32-row groups, loading the activation fragments straight
//     from global memory into registers (16 B per lane, next group prefetched while the current one computes)
//   * D = W_frag . X_frag^T (operands swapped) leaves each lane with 4 consecutive output columns of one row;
//     the 16 x N result tile is transposed through a small per-wave LDS buffer and leaves as fully contiguous
//     16-byte stores -- one workgroup writes complete cache lines, never a partial line shared with another XCD
//   * optional fused BN+SiLU(+SE gate) prologue on the activations, optional per-column sum / sum-of-squares
//     (training-mode BatchNorm statistics of the output) accumulated by persistent workgroups
// No __syncthreads() in the main loop: waves run independently.
#include "common_hip.h"
#include "../../include/mammoclip_hip.h"

namespace {

// FN = output fragments (exact: ceil(N/16)), KC = 32-wide K chunks (padded), RG = 16-row groups per wave iteration,
// PF = iterations in flight per wave
// EPI (round 3) -- fused elementwise epilogues for the data gradient of an MBConv projection conv, G = dP . Wp (the gradient
// of the SE-gated activation): G is rounded to bf16 exactly like the stored tensor was, but never goes to memory.
//   EPI = 1: the five per-image sums of mc_bnact_se_sums over (epi_x = depthwise output d, G); nothing is stored
//   EPI = 2: mc_bnact_bwd_apply on (d, G): C = k0*dz + k1*d + k2, dz = (G*mul[img] + add[img]) * silu'(d*scale + shift)
// Against dgrad -> se_sums -> apply (1 + 2 + 3 passes over the depthwise-site tensor) the two launches make 1 + 2.
// Waves take contiguous row ranges (an image change happens at most once per wave: rows_per_img % 16 == 0, so a 16-row group
// never straddles images); the d rows of the NEXT group are prefetched into registers while this one is computed.
template <int FN, int KC, int PF, int RG, int EPI = 0>
__global__ __launch_bounds__(256, 2) void gemm_rows_kernel(const mc_gemm_rows_args p, const int nblocks, const int ntiles) {
    constexpr int NP = FN * 16;                    // padded output width
    constexpr int CROW = (NP + 8) * 2;             // staging row bytes
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sW = smem;                                       // [KC][FN][64] x 16 B
    float* sScale = reinterpret_cast<float*>(sW + KC * FN * 1024);  // [KC*32]
    float* sShift = sScale + KC * 32;
    unsigned char* sC = reinterpret_cast<unsigned char*>(sShift + KC * 32);   // [4 waves][16][CROW]
    constexpr int SC_BYTES = (4 * 16 * CROW > 4 * 64 * 16 * 4) ? 4 * 16 * CROW : 4 * 64 * 16 * 4;
    float* sGate = reinterpret_cast<float*>(sC + SC_BYTES);                   // [4 waves][2 images][KC*32]: SE gate rows
    float* sFlush = sGate;                                                    // EPI = 1 (no gate prologue there): [4 waves][64 lanes][8]
    float* sPar = sFlush + 4 * 64 * 8;                                        // EPI = 1: scale | shift | mean | invstd, 256 floats each
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Wide outputs (N > 256, K small) are split into column tiles of FN*16: every tile's workgroup keeps ITS slice of
    // the weights in LDS and streams the same rows.  The column tiles of one row-block index get consecutive slots on
    // the same XCD (workgroup id % 8), so the activation rows come from HBM once and from that XCD's L2 afterwards.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nt = slot % ntiles, bxr = (slot / ntiles) * 8 + xcd;
    if (bxr >= nblocks) return;
    const int n0 = nt * NP;
    const int nw = p.N - n0 < NP ? p.N - n0 : NP;          // width of this column tile
    const int kc_used = (p.K + 31) >> 5;
    const bool has_pro = p.pro_scale != nullptr;

    // ---- stage weights as MFMA A-operand fragments: frag (kc, f), lane (i = l&15, kg = l>>4) = W[f*16+i][kc*32+kg*8 ..+8]
    for (int idx = tid; idx < KC * FN * 64; idx += 256) {
        int l = idx & 63, f = (idx >> 6) % FN, kc = (idx >> 6) / FN;
        int n = n0 + f * 16 + (l & 15), k = kc * 32 + (l >> 4) * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (n < p.N && k < p.K) v = *reinterpret_cast<const uint4*>(p.W + (long long)n * p.ldw + k);
        *reinterpret_cast<uint4*>(sW + (size_t)idx * 16) = v;
    }
    if (has_pro)
        for (int k = tid; k < KC * 32; k += 256) {
            sScale[k] = k < p.K ? p.pro_scale[k] : 0.f;
            sShift[k] = k < p.K ? p.pro_shift[k] : 0.f;
        }
    __syncthreads();

    unsigned char* myC = sC + wave * 16 * CROW;
    const int mrow = lane & 15, kg = lane >> 4;
    const long long ngroups = (p.M + RG * 16 - 1) / (RG * 16);
    // Work assignment.  Plain / BN+SiLU forms: cyclic by single 32-row groups -- the chip sweeps the tensor as one moving
    // window (best HBM rate: 8192 independent contiguous streams cost the plain form 8-16 %).  Gated form (per-image SE
    // gate in the prologue, round 3): every wave takes ONE contiguous range, so the image it works in changes once per
    // rows_per_img rows and the gate row lives in a small per-wave LDS cache (this image + the next) instead of being
    // fetched from global memory for every 16-byte chunk of every row -- those two extra vector-memory instructions per
    // activation load were the whole cost of the gated form (2.77 M x 40 x 240: 0.50 -> 0.40 ms, block-cyclic by 4: 0.44).
    const long long nwaves = (long long)nblocks * 4;
    const bool cyc = EPI == 0 && p.pro_gate == nullptr;
    const long long wid = (long long)bxr * 4 + wave;
    const long long gpw = (ngroups + nwaves - 1) / nwaves;
    const long long my_end = (wid + 1) * gpw < ngroups ? (wid + 1) * gpw : ngroups;       // contiguous form: this wave's range end
    auto gnext = [&](long long g) { return cyc ? g + nwaves : (g + 1 < my_end ? g + 1 : ngroups); };
    float* myG = sGate + wave * 2 * KC * 32;
    long long g_img = -1, g_end = 0;                        // image whose gate row sits in slot 0, first row of the next image
    auto gate_cache = [&](long long mbase) {
        if (mbase < g_end && g_img >= 0) return;
        g_img = mbase / p.pro_rows_per_img;
        g_end = (g_img + 1) * p.pro_rows_per_img;
        const long long last = (p.M - 1) / p.pro_rows_per_img;
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < 2 * KC * 32; k += 64) {
            const int sl = k >= KC * 32, kk = k - sl * KC * 32;
            const long long im = g_img + sl <= last ? g_img + sl : last;
            myG[k] = kk < p.K ? p.pro_gate[im * p.K + kk] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
    };
    // epilogue mapping: lane = (row slot, 16-byte column chunk); a lane always handles the same 8 output columns, so
    // the BatchNorm statistics of the stored tensor accumulate in 16 registers inside the store loop
    const int cpr = nw >> 3;                        // 16-byte chunks per output row of this tile (<= 32)
    const int slots = 64 / cpr;                     // rows written concurrently by one wave
    const int c8 = lane % cpr, rs = lane / cpr;
    const bool ep_active = rs < slots;
    float ssum[8], ssq[8], bias8[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { ssum[q] = 0.f; ssq[q] = 0.f; bias8[q] = 0.f; }
    if (EPI == 0 && p.bias && ep_active) load8f(p.bias + n0 + c8 * 8, bias8);

    // ---- EPI state (dead code for EPI = 0)
    constexpr int SLOTS_MIN = 64 / (2 * FN);                // slots = 64 / cpr, cpr <= 2 * FN
    constexpr int XCH = (16 + SLOTS_MIN - 1) / SLOTS_MIN;   // 16-byte chunks of a 16-row group a lane handles: ceil(16 / slots)
    float es[8], et[8], e0[8], e1[8], e2[8], emul[8], eadd[8], acc5[5][8];
    uint4 xe[PF][RG][XCH];                                  // d rows of the PF iterations in flight (slot u is reloaded as soon as its epilogue is done)
    long long img_cur = -1, img_end = 0;                    // image of the rows being processed, first row of the next image
    int eslot = 0;                                          // EPI = 1: workspace slot (0 / 1) of the image being accumulated
    if constexpr (EPI != 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { es[q] = et[q] = e0[q] = e1[q] = e2[q] = emul[q] = eadd[q] = 0.f; }
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int q = 0; q < 8; ++q) acc5[k][q] = 0.f;
        if constexpr (EPI == 1) {
            // the sums form carries 40 accumulator registers per lane: its four BatchNorm vectors live in LDS and are
            // re-read per 16-row group (re-reading them per 16-byte chunk made the kernel LDS-bound: 87 LDS instructions per group)
            for (int i = tid; i < 256; i += 256) {
                const bool ok = i < p.N;
                sPar[i] = ok ? p.epi_scale[i] : 0.f; sPar[256 + i] = ok ? p.epi_shift[i] : 0.f;
                sPar[512 + i] = ok ? p.epi_mean[i] : 0.f; sPar[768 + i] = ok ? p.epi_invstd[i] : 0.f;
            }
            __syncthreads();
        } else if (ep_active) {
            load8f(p.epi_scale + c8 * 8, es);
            load8f(p.epi_shift + c8 * 8, et);
            load8f(p.epi_coef + c8 * 8, e0); load8f(p.epi_coef + p.N + c8 * 8, e1); load8f(p.epi_coef + 2 * p.N + c8 * 8, e2);
        }
    }
    // d rows of iteration g for this lane's (row slot, chunk) role; unconditional loads (clamped addresses)
    auto load_x = [&](uint4 (&x)[RG][XCH], long long g) __attribute__((always_inline)) {
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int u = 0; u < XCH; ++u) {
                const int row = rs + u * slots;
                long long m = (g * RG + rg) * 16 + (row < 16 ? row : 0);
                if (m >= p.M) m = p.M - 1;
                x[rg][u] = *reinterpret_cast<const uint4*>(p.epi_x + m * p.epi_ldx + (ep_active ? c8 : 0) * 8);
            }
    };
    // EPI = 1: the wave's sums of the image just finished -> workspace slot (row slots combined through LDS, fixed order)
    auto flush_sums = [&]() __attribute__((always_inline)) {
        float* red = sFlush + wave * 64 * 8;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
#pragma unroll
            for (int q = 0; q < 8; ++q) red[lane * 8 + q] = ep_active ? acc5[k][q] : 0.f;
            __builtin_amdgcn_wave_barrier();
            if (ep_active && rs == 0) {
                float* dst = p.epi_ws + ((wid * 2 + eslot) * 5 + k) * p.N + c8 * 8;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float sm = 0.f;
                    for (int r = 0; r < slots; ++r) sm += red[(r * cpr + c8) * 8 + q];
                    dst[q] = sm;
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 8; ++q) acc5[k][q] = 0.f;
        }
        if (lane == 0) reinterpret_cast<int*>(p.epi_ws + (long long)nwaves * 2 * 5 * p.N)[wid * 2 + eslot] = (int)img_cur;
        ++eslot;
    };

    // PF 32-row groups per wave are in flight (narrow K = few bytes per group: latency needs several groups ahead)
    uint4 xn[PF][RG][KC];
    auto load_group = [&](uint4 (&x)[RG][KC], long long g) {
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
            long long m = (g * RG + rg) * 16 + mrow;
            if (m >= p.M) m = p.M - 1;              // clamped rows are computed but never stored
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                int k = kc * 32 + kg * 8;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (kc < kc_used && k < p.K) v = *reinterpret_cast<const uint4*>(p.X + m * p.ldx + k);
                x[rg][kc] = v;
            }
        }
    };
    // gx (EPI): the iteration whose d rows replace this one's in xd, chunk by chunk, as soon as each chunk is consumed
    auto process = [&](long long g, uint4 (&xf)[RG][KC], uint4 (&xd)[RG][XCH], long long gx) __attribute__((always_inline)) {
        if (EPI == 0 && has_pro) {
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                long long m = (g * RG + rg) * 16 + mrow;
                if (m >= p.M) m = p.M - 1;
                const float* gate = nullptr;
                if (p.pro_gate) {
                    gate_cache((g * RG + rg) * 16 < p.M ? (g * RG + rg) * 16 : p.M - 1);
                    gate = myG + (m >= g_end ? KC * 32 : 0);
                }
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    int k = kc * 32 + kg * 8;
                    if (kc < kc_used && k < p.K) {
                        float f[8], s[8], t[8];
                        unpack8(xf[rg][kc], f);
                        load8f(sScale + k, s);
                        load8f(sShift + k, t);
                        bn_silu8(f, s, t);
                        if (gate) {
                            float gv[8];
                            load8f(gate + k, gv);
#pragma unroll
                            for (int q = 0; q < 8; ++q) f[q] *= gv[q];
                        }
                        xf[rg][kc] = pack8(f);
                    }
                }
            }
        }

        f32x4_t acc[RG][FN];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
            for (int f = 0; f < FN; ++f) acc[rg][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            if (kc < kc_used) {
#pragma unroll
                for (int f = 0; f < FN; ++f) {
                    bf16x8_t wf = *reinterpret_cast<const bf16x8_t*>(sW + ((size_t)(kc * FN + f) * 64 + lane) * 16);
#pragma unroll
                    for (int rg = 0; rg < RG; ++rg)
                        acc[rg][f] = MC_MFMA_16x16x32(
                            wf, *reinterpret_cast<const bf16x8_t*>(&xf[rg][kc]), acc[rg][f], 0, 0, 0);
                }
            }
        }

        // ---- epilogue per 16-row group: registers -> per-wave LDS tile -> contiguous global stores (+ statistics)
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
            const long long mbase = (g * RG + rg) * 16;
            if (mbase >= p.M) break;
#pragma unroll
            for (int f = 0; f < FN; ++f) {
                uint2 pk = make_uint2(pack_bf2(acc[rg][f][0], acc[rg][f][1]), pack_bf2(acc[rg][f][2], acc[rg][f][3]));
                *reinterpret_cast<uint2*>(myC + mrow * CROW + (f * 16 + kg * 4) * 2) = pk;
            }
            __builtin_amdgcn_wave_barrier();
            if constexpr (EPI != 0) {
                if (mbase >= img_end) {                     // (wave-uniform) the group starts a new image
                    if (EPI == 1 && img_cur >= 0) flush_sums();
                    img_cur = mbase / p.epi_rows_per_img;
                    img_end = (img_cur + 1) * p.epi_rows_per_img;
                    if (EPI == 2 && ep_active) {
                        load8f(p.epi_mul + img_cur * p.N + c8 * 8, emul);
                        load8f(p.epi_add + img_cur * p.N + c8 * 8, eadd);
#pragma unroll
                        for (int q = 0; q < 8; ++q) eadd[q] *= p.epi_add_scale;
                    }
                }
                if (ep_active) {
                    if constexpr (EPI == 1) {               // once per 16-row group (the MFMA accumulators are dead here: no extra registers)
                        load8f(sPar + c8 * 8, es); load8f(sPar + 256 + c8 * 8, et);
                        load8f(sPar + 512 + c8 * 8, e0); load8f(sPar + 768 + c8 * 8, e1);
                    }
#pragma unroll
                    for (int u = 0; u < XCH; ++u) {
                        const int row = rs + u * slots;
                        const long long m = mbase + row;
                        float x[8];
                        unpack8(xd[rg][u], x);
                        {   // this chunk's register now takes the same (row slot, chunk) of iteration gx: a whole iteration of lead
                            long long mx = (gx * RG + rg) * 16 + (row < 16 ? row : 0);
                            if (mx >= p.M) mx = p.M - 1;
                            xd[rg][u] = *reinterpret_cast<const uint4*>(p.epi_x + mx * p.epi_ldx + c8 * 8);
                        }
                        if (row < 16 && m < p.M) {
                            float gq8[8];
                            unpack8(*reinterpret_cast<const uint4*>(myC + row * CROW + c8 * 16), gq8);     // G, bf16-rounded
                            if constexpr (EPI == 1) {
#pragma unroll
                                for (int q = 0; q < 8; q += 2) {                 // (same arithmetic as bnact_se_sums_k)
                                    const f32x2_t one = {1.f, 1.f};
                                    const f32x2_t xv = {x[q], x[q + 1]}, gv = {gq8[q], gq8[q + 1]};
                                    const f32x2_t z = __builtin_elementwise_fma(xv, f32x2_t{es[q], es[q + 1]}, f32x2_t{et[q], et[q + 1]});
                                    const f32x2_t sg = sigmoid2_f(z);
                                    const f32x2_t y = z * sg, yd = sg * __builtin_elementwise_fma(z, one - sg, one);
                                    const f32x2_t xh = (xv - f32x2_t{e0[q], e0[q + 1]}) * f32x2_t{e1[q], e1[q + 1]};
                                    const f32x2_t gyd = gv * yd;
                                    f32x2_t a2;
                                    a2 = __builtin_elementwise_fma(gv, y, f32x2_t{acc5[0][q], acc5[0][q + 1]}); acc5[0][q] = a2.x; acc5[0][q + 1] = a2.y;
                                    a2 = f32x2_t{acc5[1][q], acc5[1][q + 1]} + gyd; acc5[1][q] = a2.x; acc5[1][q + 1] = a2.y;
                                    a2 = __builtin_elementwise_fma(gyd, xh, f32x2_t{acc5[2][q], acc5[2][q + 1]}); acc5[2][q] = a2.x; acc5[2][q + 1] = a2.y;
                                    a2 = f32x2_t{acc5[3][q], acc5[3][q + 1]} + yd; acc5[3][q] = a2.x; acc5[3][q + 1] = a2.y;
                                    a2 = __builtin_elementwise_fma(yd, xh, f32x2_t{acc5[4][q], acc5[4][q + 1]}); acc5[4][q] = a2.x; acc5[4][q + 1] = a2.y;
                                }
                            } else {
                                float o[8];
#pragma unroll
                                for (int q = 0; q < 8; q += 2) {                 // (same arithmetic as bnact_bwd_k<true, true, true>)
                                    f32x2_t d2 = __builtin_elementwise_fma(f32x2_t{gq8[q], gq8[q + 1]}, f32x2_t{emul[q], emul[q + 1]}, f32x2_t{eadd[q], eadd[q + 1]});
                                    d2 = d2 * silu_grad2_f(__builtin_elementwise_fma(f32x2_t{x[q], x[q + 1]}, f32x2_t{es[q], es[q + 1]}, f32x2_t{et[q], et[q + 1]}));
                                    o[q] = e0[q] * d2.x + e1[q] * x[q] + e2[q];
                                    o[q + 1] = e0[q + 1] * d2.y + e1[q + 1] * x[q + 1] + e2[q + 1];
                                }
                                *reinterpret_cast<uint4*>(p.C + m * p.ldc + n0 + c8 * 8) = pack8(o);
                            }
                        }
                    }
                }
            } else
            if (ep_active) {
                for (int row = rs; row < 16; row += slots) {
                    const long long m = mbase + row;
                    if (m < p.M) {
                        uint4 v = *reinterpret_cast<const uint4*>(myC + row * CROW + c8 * 16);
                        float a[8];
                        unpack8(v, a);
                        if (p.stat_partials) {          // statistics of the conv output itself (before any residual)
#pragma unroll
                            for (int q = 0; q < 8; ++q) { ssum[q] += a[q]; ssq[q] += a[q] * a[q]; }
                        }
                        if (p.R || p.bias) {
                            float b[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) b[q] = 0.f;
                            if (p.R) unpack8(*reinterpret_cast<const uint4*>(p.R + m * p.ldr + n0 + c8 * 8), b);
#pragma unroll
                            for (int q = 0; q < 8; ++q) a[q] += b[q] + bias8[q];
                            v = pack8(a);
                        }
                        *reinterpret_cast<uint4*>(p.C + m * p.ldc + n0 + c8 * 8) = v;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    };

    long long gq[PF];                                       // the groups in flight, oldest first (static indices)
    gq[0] = cyc ? wid : (wid * gpw < ngroups ? wid * gpw : ngroups);
#pragma unroll
    for (int u = 1; u < PF; ++u) gq[u] = gnext(gq[u - 1]);
#pragma unroll
    for (int u = 0; u < PF; ++u)
        if (gq[u] < ngroups) load_group(xn[u], gq[u]);
    if constexpr (EPI != 0) {
#pragma unroll
        for (int u = 0; u < PF; ++u) load_x(xe[u], gq[u] < ngroups ? gq[u] : 0);
    }
    while (gq[0] < ngroups) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const long long gg = gq[u];
            if (gg < ngroups) {
                uint4 xf[RG][KC];
#pragma unroll
                for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                    for (int kc = 0; kc < KC; ++kc) xf[rg][kc] = xn[u][rg][kc];
                long long gn = gg;
#pragma unroll
                for (int t = 0; t < PF; ++t) gn = gnext(gn);
                gq[u] = gn;
                if (gn < ngroups) load_group(xn[u], gn);   // in flight during PF groups of work
                process(gg, xf, xe[u], gn < ngroups ? gn : gg);
            }
        }
    }

    if constexpr (EPI == 1) {
        if (img_cur >= 0) flush_sums();
        int* tags = reinterpret_cast<int*>(p.epi_ws + (long long)nwaves * 2 * 5 * p.N);
        if (lane == 0)
            for (int sl = eslot; sl < 2; ++sl) tags[wid * 2 + sl] = -1;
    }
    if (EPI == 0 && p.stat_partials) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(sC);        // [4 waves][64 lanes][16]
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            red[(wave * 64 + lane) * 16 + q] = ep_active ? ssum[q] : 0.f;
            red[(wave * 64 + lane) * 16 + 8 + q] = ep_active ? ssq[q] : 0.f;
        }
        __syncthreads();
        for (int c = tid; c < nw; c += 256) {
            float s = 0.f, s2 = 0.f;
            for (int w = 0; w < 4; ++w)
                for (int r = 0; r < slots; ++r) {
                    const float* e = red + (w * 64 + r * cpr + (c >> 3)) * 16 + (c & 7);
                    s += e[0]; s2 += e[8];
                }
            float* dst = p.stat_partials + (long long)bxr * 2 * p.N + n0;
            dst[c] = s;
            dst[p.N + c] = s2;
        }
    }
}

template <int FN, int KC> size_t lds_bytes() {
    size_t staging = 4 * 16 * ((FN * 16 + 8) * 2);
    size_t red = 4 * 64 * 16 * 4;
    return (size_t)KC * FN * 1024 + 2 * KC * 32 * 4 + (staging > red ? staging : red);
}

// Persistent workgroups per CU = what the instance's registers and LDS allow (round 3: the launch used to be capped at two
// per CU; the write-heavy expand convs (24 -> 144: 142 VGPRs) gain 16 % from a third, instances above 168 VGPRs stay at two)
template <int FN, int KC, int PF, int RG, int EPI> int rows_occupancy(bool gated) {
    static int occ[2] = {0, 0};
    if (!occ[gated]) {
        static unsigned long long attr_done = 0;
        const void* kfn = reinterpret_cast<const void*>(&gemm_rows_kernel<FN, KC, PF, RG, EPI>);
        const size_t lds_max = lds_bytes<FN, KC>() + (size_t)4 * 2 * KC * 32 * 4 + 12288;
        const size_t lds = lds_bytes<FN, KC>() + (gated ? (size_t)4 * 2 * KC * 32 * 4 : 0) + (EPI == 1 ? 12288 : 0);
        if (lds > 64 * 1024) MC_SET_MAX_LDS(attr_done, kfn, lds_max);
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kfn, 256, lds) != hipSuccess || nb < 1) nb = 2;
        occ[gated] = nb > 3 ? 3 : nb;
    }
    return occ[gated];
}

// per-image sums of the EPI = 1 launch: sums[k][img][c] = sum over the (wave, slot) workspace entries tagged with img, in wave order
__global__ __launch_bounds__(256) void rows_se_reduce_k(const float* __restrict__ ws, int nwaves, long long rows_per_wave,
                                                        long long rpi, int n_img, int n, float* __restrict__ sums) {
    const int img = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 5 * n) return;
    const int k = e / n, c = e % n;
    const int* tags = reinterpret_cast<const int*>(ws + (long long)nwaves * 2 * 5 * n);
    long long w_lo = ((long long)img * rpi) / rows_per_wave, w_hi = (((long long)img + 1) * rpi - 1) / rows_per_wave;
    if (w_hi > nwaves - 1) w_hi = nwaves - 1;
    float s = 0.f;
    for (long long w = w_lo; w <= w_hi; ++w)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
            if (tags[w * 2 + sl] == img) s += ws[((w * 2 + sl) * 5 + k) * n + c];
    sums[((long long)k * n_img + img) * n + c] = s;
}

// query != nullptr: report the workgroup count (= rows of stat_partials) instead of launching
template <int FN, int KC, int EPI> int launch_rows_e(const mc_gemm_rows_args& p, hipStream_t st, int* query) {
    // wide outputs / deep K: 16 rows per iteration (register budget); 16-32 column outputs over K <= 64: 64 rows per
    // iteration (a 16-row group is only 0.8-3 KB of traffic there: the per-iteration overhead was the limit)
    constexpr int RG = (FN >= 6 || KC >= 12) ? 1 : ((FN <= 2 && KC <= 2 && EPI == 0) ? 4 : 2);
    constexpr int PF0 = KC >= 12 ? 1 : (KC >= 6 ? (RG == 1 ? 2 : 1) : (RG == 4 ? 4 / KC : 8 / KC));   // ~8-16 KB of activations in flight per wave
    // epilogue forms: every iteration in flight also holds its d rows (RG * XCH 16-byte registers per lane): <= 48 registers
    constexpr int XCH_ = (16 + 64 / (2 * FN) - 1) / (64 / (2 * FN));
    constexpr int PFE_ = 12 / (RG * XCH_) < 1 ? 1 : 12 / (RG * XCH_);
    constexpr int PF = EPI == 0 ? PF0 : ((PF0 > 1 ? PF0 / 2 : 1) < PFE_ ? (PF0 > 1 ? PF0 / 2 : 1) : PFE_);
    const long long groups = (p.M + 15) / 16;
    long long b = (groups + 3) / 4;
    const int ntiles = (p.N + FN * 16 - 1) / (FN * 16);
    const long long cap = ntiles > 1 ? 512 : 256LL * rows_occupancy<FN, KC, PF, RG, EPI>(p.pro_gate != nullptr);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    const int blocks = (int)b;
    if (query) {
        if constexpr (EPI != 0) {                         // (query of an epilogue form: 0 workgroups = the launch would be refused)
            const long long nwaves_q = (long long)blocks * 4, iters_q = (p.M + RG * 16 - 1) / (RG * 16);
            if (ntiles != 1 || (iters_q + nwaves_q - 1) / nwaves_q * RG * 16 > p.epi_rows_per_img) { *query = 0; return MC_OK; }
        }
        *query = blocks;
        return MC_OK;
    }
    size_t lds = lds_bytes<FN, KC>() + (p.pro_gate ? (size_t)4 * 2 * KC * 32 * 4 : 0) + (EPI == 1 ? 12288 : 0);     // + per-wave SE gate cache / flush tile
    static unsigned long long attr_done = 0;
    const void* kfn = reinterpret_cast<const void*>(&gemm_rows_kernel<FN, KC, PF, RG, EPI>);
    const size_t lds_max = lds_bytes<FN, KC>() + (size_t)4 * 2 * KC * 32 * 4 + 12288;
    if (lds > 64 * 1024) MC_SET_MAX_LDS(attr_done, kfn, lds_max);
    if constexpr (EPI != 0) {
        // contiguous per-wave ranges of at most one image length (two workspace slots per wave)
        const long long nwaves = (long long)blocks * 4, iters = (p.M + RG * 16 - 1) / (RG * 16);
        const long long rows_per_wave = (iters + nwaves - 1) / nwaves * RG * 16;
        MC_CHECK(ntiles == 1 && rows_per_wave <= p.epi_rows_per_img, "gemm_rows (epilogue forms): N <= 256 and at least as many waves as images");
        hipLaunchKernelGGL((gemm_rows_kernel<FN, KC, PF, RG, EPI>), dim3((blocks + 7) / 8 * 8), dim3(256), lds, st, p, blocks, 1);
        MC_LAUNCH_CHECK();
        if constexpr (EPI == 1) {
            const int n_img = (int)((p.M + p.epi_rows_per_img - 1) / p.epi_rows_per_img);
            hipLaunchKernelGGL(rows_se_reduce_k, dim3(mc_div_up(5 * p.N, 256), n_img), dim3(256), 0, st, p.epi_ws, (int)nwaves, rows_per_wave,
                               p.epi_rows_per_img, n_img, p.N, p.epi_sums);
            MC_LAUNCH_CHECK();
        }
        return MC_OK;
    }
    hipLaunchKernelGGL((gemm_rows_kernel<FN, KC, PF, RG, EPI>), dim3((blocks + 7) / 8 * 8 * ntiles), dim3(256), lds, st, p, blocks, ntiles);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
template <int FN, int KC> int launch_rows(const mc_gemm_rows_args& p, hipStream_t st, int* query) {
    if constexpr (KC <= 4) {                               // epilogue forms: projection data gradients, K = c_out <= 128
        if (p.epi_mode == 1) return launch_rows_e<FN, KC, 1>(p, st, query);
        if (p.epi_mode == 2) return launch_rows_e<FN, KC, 2>(p, st, query);
    }
    return launch_rows_e<FN, KC, 0>(p, st, query);
}

template <int FN> int dispatch_kc(const mc_gemm_rows_args& p, hipStream_t st, int* query) {
    if (p.K <= 32) return launch_rows<FN, 1>(p, st, query);
    if (p.K <= 64) return launch_rows<FN, 2>(p, st, query);
    if (p.K <= 128) return launch_rows<FN, 4>(p, st, query);
    if constexpr (FN == 8) {
        if (p.K <= 192) return launch_rows<FN, 6>(p, st, query);     // 48 KiB weight slice: two workgroups per CU
    }
    if constexpr (FN <= 8) {
        if (p.K <= 256) return launch_rows<FN, 8>(p, st, query);
        if constexpr (FN <= 5) return launch_rows<FN, 12>(p, st, query);
    }
    mc_set_error("gemm_rows: internal: no instantiation");
    return MC_ERR_ARG;
}

int dispatch_fn(const mc_gemm_rows_args& p, hipStream_t st, int* query) {
    if (p.N > 256) return dispatch_kc<8>(p, st, query);   // wide output: 128-column tiles
    switch ((p.N + 15) / 16) {                     // exact fragment count: no dead accumulators, no predicated MFMAs
        case 1: return dispatch_kc<1>(p, st, query);
        case 2: return dispatch_kc<2>(p, st, query);
        case 3: return dispatch_kc<3>(p, st, query);
        case 4: return dispatch_kc<4>(p, st, query);
        case 5: return dispatch_kc<5>(p, st, query);
        case 6: return dispatch_kc<6>(p, st, query);
        case 7: return dispatch_kc<7>(p, st, query);
        case 8: return dispatch_kc<8>(p, st, query);
        case 9: return dispatch_kc<9>(p, st, query);
        case 10: return dispatch_kc<10>(p, st, query);
        case 11: return dispatch_kc<11>(p, st, query);
        case 12: return dispatch_kc<12>(p, st, query);
        case 13: return dispatch_kc<13>(p, st, query);
        case 14: return dispatch_kc<14>(p, st, query);
        case 15: return dispatch_kc<15>(p, st, query);
        default: return dispatch_kc<16>(p, st, query);
    }
}

}  // namespace

extern "C" int mc_gemm_rows_supported(int n, int k) {
    if (n <= 0 || k <= 0 || k > 384 || n % 8 || k % 8) return 0;
    if (n > 256) return k <= 256;                  // column tiles of 128 (FN = 8), weights slice <= 64 KiB of LDS
    int fn = (n + 15) / 16;
    int kcp = k <= 32 ? 1 : (k <= 64 ? 2 : (k <= 128 ? 4 : (k <= 256 ? 8 : 12)));
    if (kcp == 8 && fn > 8) return 0;
    if (kcp == 12 && fn > 5) return 0;
    return fn * kcp <= 64;                        // weight image <= 64 KiB of LDS
}

// rows of stat_partials = persistent workgroups of the launch mc_gemm_rows_bf16 makes for these arguments
extern "C" int mc_gemm_rows_blocks(const mc_gemm_rows_args* a) {
    int q = 0;
    if (!mc_gemm_rows_supported(a->N, a->K) || a->M <= 0) return 0;
    if (dispatch_fn(*a, nullptr, &q) != MC_OK) return 0;
    return q;
}

// can mc_gemm_rows_bf16 run the epilogue form (epi_mode 1 / 2) of this problem?  (the host routes on this
// instead of meeting the launch-time refusal: many tiny maps -- more images than waves, or maps under 16 pixels)
extern "C" int mc_gemm_rows_epi_supported(long long M, int N, int K, long long rows_per_img, int epi_mode) {
    if (epi_mode < 1 || epi_mode > 2 || M <= 0 || N > 256 || K > 128 || !mc_gemm_rows_supported(N, K)) return 0;
    if (rows_per_img < 16 || rows_per_img % 16 != 0) return 0;
    mc_gemm_rows_args a = {};
    a.M = M; a.N = N; a.K = K; a.epi_mode = epi_mode; a.epi_rows_per_img = rows_per_img;
    int q = 0;
    if (dispatch_fn(a, nullptr, &q) != MC_OK) return 0;
    return q > 0;
}

// floats of epi_ws for an epi_mode = 1 launch: [waves][2 slots][5][N] sums + [waves][2] image tags
extern "C" long long mc_gemm_rows_epi_ws_floats(const mc_gemm_rows_args* a) {
    const int blocks = mc_gemm_rows_blocks(a);
    return (long long)blocks * 4 * 2 * (5LL * a->N + 1);
}

extern "C" int mc_gemm_rows_bf16(const mc_gemm_rows_args* a, void* stream) {
    const mc_gemm_rows_args& p = *a;
    MC_CHECK(p.X && p.W && (p.C || p.epi_mode == 1) && p.M > 0 && p.N > 0 && p.K > 0, "gemm_rows: bad args");
    if (p.epi_mode != 0) {
        MC_CHECK(p.epi_mode == 1 || p.epi_mode == 2, "gemm_rows: epi_mode must be 0, 1 or 2");
        MC_CHECK(p.N <= 256 && p.K <= 128 && !p.R && !p.bias && !p.stat_partials && !p.pro_scale, "gemm_rows (epilogue forms): N <= 256, K <= 128, no residual / bias / statistics / prologue");
        MC_CHECK(p.epi_x && p.epi_scale && p.epi_shift && p.epi_ldx % 8 == 0 && mc_aligned16(p.epi_x), "gemm_rows (epilogue forms): epi_x / scale / shift");
        MC_CHECK(p.epi_rows_per_img >= 16 && p.epi_rows_per_img % 16 == 0, "gemm_rows (epilogue forms): rows per image must be a multiple of 16");
        if (p.epi_mode == 1) MC_CHECK(p.epi_mean && p.epi_invstd && p.epi_sums && p.epi_ws, "gemm_rows (epi_mode 1): mean / invstd / sums / workspace");
        else MC_CHECK(p.epi_coef && p.epi_mul && p.epi_add && p.C, "gemm_rows (epi_mode 2): coef / mul / add / C");
    }
    MC_CHECK(mc_gemm_rows_supported(p.N, p.K), "gemm_rows: unsupported shape (see mc_gemm_rows_supported)");
    MC_CHECK(p.ldx % 8 == 0 && p.ldw % 8 == 0 && p.ldc % 8 == 0, "gemm_rows: leading dims must be multiples of 8");
    MC_CHECK(mc_aligned16(p.X) && mc_aligned16(p.W) && (!p.C || mc_aligned16(p.C)), "gemm_rows: operands must be 16-byte aligned");
    MC_CHECK(!p.R || (p.ldr % 8 == 0 && mc_aligned16(p.R)), "gemm_rows: bad residual");
    MC_CHECK((p.pro_scale == nullptr) == (p.pro_shift == nullptr), "gemm_rows: prologue needs scale and shift");
    MC_CHECK(!p.pro_gate || (p.pro_scale && p.pro_rows_per_img >= 16), "gemm_rows: gate needs the BN prologue and >= 16 rows per image");
    return dispatch_fn(p, (hipStream_t)stream, nullptr);
}
