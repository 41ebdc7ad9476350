// Fused BERT self-attention core for gfx950: per (sequence, head)
//     ctx = dropout(softmax(alpha * Q K^T + mask_bias)) V
// forward and backward in one kernel each -- the [b, heads, T, T] score / probability tensors never reach HBM.
// [ref: model/modules/text_encoder.py:47-49 -> transformers BertSelfAttention.forward (scores, mask, softmax, dropout,
//       context); head size 64 (BERT-base), T <= 256 (the reference tokenises to max_length 256), T % 32 == 0]
//
// One 8-wave workgroup per (sequence, head); K and V rows (forward) live in LDS, row-major with a 144-byte stride.
// A wave owns 16 queries at a time and computes S^T = K Q^T on v_mfma_f32_16x16x32_bf16: with the A-operand ROWS of a
// tile pair mapped to keys 32J + (r>>2)*8 + (r&3) (+4), a lane ends up with 8 CONSECUTIVE keys of one query per 32-key
// block J -- exactly the k-slots an MFMA operand fragment holds -- so the probabilities feed P V (and dS feeds dS K)
// straight from registers, the row reductions are in-lane + two cross-lane steps, and one Philox draw covers a lane's
// 8 keys (same (seed, stream, element) function as the unfused softmax kernel: identical dropout masks).
// V^T / K^T / Q^T / dO^T fragments come from the row-major tiles through ds_read_b64_tr_b16.
// Backward: phase A = per-query work in the same layout (recompute P from the saved row max / 1/sum, dP = dO V^T,
// row dots, dQ = dS K; the keep-mask bits and the row dots are parked in LDS); phase B = per-key work in the
// transposed layout (a lane holds 8 consecutive queries of one key): dK = dS^T Q, dV = Pd^T dO with fp32 accumulators
// held by the wave that owns the 32 keys -- no atomics, bit-reproducible.
#include "common_hip.h"
#include "../../include/mammoclip_hip.h"

namespace {

constexpr int RS = 144;          // LDS row stride (bytes) of a 64-wide bf16 row: +16 B keeps ds_read_b64_tr_b16 conflict-free
constexpr int TMAX = 256;
constexpr int HD = 64;

typedef __attribute__((ext_vector_type(4))) short s4_t;
typedef __attribute__((ext_vector_type(8))) short s8_t;
typedef __attribute__((address_space(3))) s4_t lds_s4_t;

// fragment (index n = col0 + (lane & 15), k = row0 + (lane >> 4) * 8 .. +8) of a row-major [k][n] LDS tile
__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* tile, int row0, int col0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const unsigned char* a = tile + (row0 + g * 8 + (i >> 2)) * RS + (col0 + (i & 3) * 4) * 2;
    s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(a));
    s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t*)(a + 4 * RS));
    s8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}
// fragment of the row-major tile whose MFMA row r is tile row base + (r>>2)*8 + (r&3): lanes of a 16-row group end up
// owning rows base + g*8 + (0..3); the caller adds 4 rows for the second tile of the pair
__device__ __forceinline__ bf16x8_t perm_frag(const unsigned char* tile, int base, int ks, int lane) {
    const int g = lane >> 4, i = lane & 15;
    return *reinterpret_cast<const bf16x8_t*>(tile + (base + (i >> 2) * 8 + (i & 3)) * RS + (ks * 32 + g * 8) * 2);
}
__device__ __forceinline__ bf16x8_t as_frag(const float* f) {
    uint4 v = pack8(f);
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ float xor_sum16_32(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ void stage_rows(unsigned char* dst, const bf16_t* src, long long ld, int t, int tid) {
    for (int idx = tid; idx < t * 8; idx += 512) {
        const int row = idx >> 3, ch = idx & 7;
        *reinterpret_cast<uint4*>(dst + row * RS + ch * 16) = *reinterpret_cast<const uint4*>(src + (long long)row * ld + ch * 8);
    }
}

struct attn_args {
    const bf16_t* qkv;      // [b*t, 3H]: Q | K | V
    const float* maskb;     // [b, t] additive key bias
    const bf16_t* dctx;     // [b*t, H] (backward)
    bf16_t* ctx;            // [b*t, H] (forward)
    bf16_t* dqkv;           // [b*t, 3H] (backward)
    float* lse;             // [b*nh*t][2]: row max, 1 / row sum
    int t, nh;
    float alpha, p;
    unsigned long long seed;
    unsigned int sid;
};

template <bool DROP>
__global__ __launch_bounds__(512) void attn_fwd_k(attn_args a) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* const Ks = smem;
    unsigned char* const Vs = smem + TMAX * RS;
    float* const mb = reinterpret_cast<float*>(smem + 2 * TMAX * RS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int t = a.t, nh = a.nh, H = nh * HD, ld = 3 * H;
    const int bh = blockIdx.x, bi = bh / nh, h = bh % nh;
    const bf16_t* const base = a.qkv + (long long)bi * t * ld + h * HD;
    stage_rows(Ks, base + H, ld, t, tid);
    stage_rows(Vs, base + 2 * H, ld, t, tid);
    for (int k = tid; k < t; k += 512) mb[k] = a.maskb[(long long)bi * t + k];
    __syncthreads();
    const int nJ = t >> 5;
    for (int qb = wave; qb * 16 < t; qb += 8) {
        const int q = qb * 16 + li;
        bf16x8_t qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[ks] = *reinterpret_cast<const bf16x8_t*>(base + (long long)q * ld + ks * 32 + g * 8);
        float s[8][8];
        float mx = -3.4028234663852886e38f;
#pragma unroll
        for (int J = 0; J < 8; ++J) {
            if (J < nJ) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        acc = MC_MFMA_16x16x32(perm_frag(Ks, 32 * J + 4 * tt, ks, lane), qf[ks], acc, 0, 0, 0);
                    const float4 bv = *reinterpret_cast<const float4*>(mb + 32 * J + g * 8 + 4 * tt);
                    s[J][tt * 4 + 0] = acc[0] * a.alpha + bv.x;
                    s[J][tt * 4 + 1] = acc[1] * a.alpha + bv.y;
                    s[J][tt * 4 + 2] = acc[2] * a.alpha + bv.z;
                    s[J][tt * 4 + 3] = acc[3] * a.alpha + bv.w;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) mx = fmaxf(mx, s[J][i]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int J = 0; J < 8; ++J)
            if (J < nJ)
#pragma unroll
                for (int i = 0; i < 8; ++i) { s[J][i] = __expf(s[J][i] - mx); sum += s[J][i]; }
        const float inv = 1.f / xor_sum16_32(sum);
        const long long row = (long long)bh * t + q;
        if (g == 0) *reinterpret_cast<float2*>(a.lse + row * 2) = make_float2(mx, inv);
        f32x4_t o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int J = 0; J < 8; ++J) {
            if (J < nJ) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = s[J][i] * inv;
                if (DROP) {
                    float ds[8];
                    dropout_scale8(a.seed, a.sid, (unsigned long long)row * (t >> 3) + 4 * J + g, a.p, ds);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= ds[i];
                }
                const bf16x8_t pf = as_frag(v);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    o[dt] = MC_MFMA_16x16x32(tr_frag(Vs, 32 * J, dt * 16, lane), pf, o[dt], 0, 0, 0);
            }
        }
        bf16_t* const dst = a.ctx + ((long long)bi * t + q) * H + h * HD + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<uint2*>(dst + dt * 16) = make_uint2(pack_bf2(o[dt][0], o[dt][1]), pack_bf2(o[dt][2], o[dt][3]));
    }
}

template <bool DROP>
__global__ __launch_bounds__(512) void attn_bwd_k(attn_args a) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* const Ks = smem;                    // phase A: K rows, phase B: Q rows
    unsigned char* const Vs = smem + TMAX * RS;        // phase A: V rows, phase B: dO rows
    float* const mb = reinterpret_cast<float*>(smem + 2 * TMAX * RS);
    float* const lse_s = mb + TMAX;                    // [t][2]
    float* const dot_s = lse_s + 2 * TMAX;             // [t]
    unsigned char* const dmask = reinterpret_cast<unsigned char*>(dot_s + TMAX);   // [t][32] keep bits, 8 keys per byte
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int t = a.t, nh = a.nh, H = nh * HD, ld = 3 * H;
    const int bh = blockIdx.x, bi = bh / nh, h = bh % nh;
    const bf16_t* const base = a.qkv + (long long)bi * t * ld + h * HD;
    const bf16_t* const dbase = a.dctx + (long long)bi * t * H + h * HD;
    bf16_t* const gbase = a.dqkv + (long long)bi * t * ld + h * HD;
    stage_rows(Ks, base + H, ld, t, tid);
    stage_rows(Vs, base + 2 * H, ld, t, tid);
    for (int k = tid; k < t; k += 512) {
        mb[k] = a.maskb[(long long)bi * t + k];
        *reinterpret_cast<float2*>(lse_s + 2 * k) = *reinterpret_cast<const float2*>(a.lse + ((long long)bh * t + k) * 2);
    }
    __syncthreads();
    const int nJ = t >> 5;
    const float invkeep = 1.f / (1.f - a.p);
    // ---------------- phase A: a lane = one query x 8 consecutive keys per 32-key block
    for (int qb = wave; qb * 16 < t; qb += 8) {
        const int q = qb * 16 + li;
        bf16x8_t qf[2], dof[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8_t*>(base + (long long)q * ld + ks * 32 + g * 8);
            dof[ks] = *reinterpret_cast<const bf16x8_t*>(dbase + (long long)q * H + ks * 32 + g * 8);
        }
        const float mx = lse_s[2 * q], inv = lse_s[2 * q + 1];
        const long long row = (long long)bh * t + q;
        float pr[8][8], d[8][8];
        float dot = 0.f;
#pragma unroll
        for (int J = 0; J < 8; ++J) {
            if (J < nJ) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f}, dacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        acc = MC_MFMA_16x16x32(perm_frag(Ks, 32 * J + 4 * tt, ks, lane), qf[ks], acc, 0, 0, 0);
                        dacc = MC_MFMA_16x16x32(perm_frag(Vs, 32 * J + 4 * tt, ks, lane), dof[ks], dacc, 0, 0, 0);
                    }
                    const float4 bv = *reinterpret_cast<const float4*>(mb + 32 * J + g * 8 + 4 * tt);
                    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = __expf(acc[r] * a.alpha + bb[r] - mx) * inv;
                        pr[J][tt * 4 + r] = bf2f(f2bf(v));
                        d[J][tt * 4 + r] = dacc[r];
                    }
                }
                if (DROP) {
                    float ds[8];
                    dropout_scale8(a.seed, a.sid, (unsigned long long)row * (t >> 3) + 4 * J + g, a.p, ds);
                    unsigned int bits = 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        d[J][i] *= ds[i];
                        bits |= (ds[i] != 0.f ? 1u : 0u) << i;
                    }
                    dmask[q * 32 + 4 * J + g] = (unsigned char)bits;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) dot += pr[J][i] * d[J][i];
            }
        }
        dot = xor_sum16_32(dot);
        if (g == 0) dot_s[q] = dot;
        f32x4_t dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int J = 0; J < 8; ++J) {
            if (J < nJ) {
                float dsv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) dsv[i] = pr[J][i] * (d[J][i] - dot) * a.alpha;
                const bf16x8_t dsf = as_frag(dsv);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    dq[dt] = MC_MFMA_16x16x32(tr_frag(Ks, 32 * J, dt * 16, lane), dsf, dq[dt], 0, 0, 0);
            }
        }
        bf16_t* const dst = gbase + (long long)q * ld + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<uint2*>(dst + dt * 16) = make_uint2(pack_bf2(dq[dt][0], dq[dt][1]), pack_bf2(dq[dt][2], dq[dt][3]));
    }
    __syncthreads();
    // ---------------- phase B: a lane = one key x 8 consecutive queries per 32-query block; a wave owns 32 keys
    stage_rows(Ks, base, ld, t, tid);          // Q rows
    stage_rows(Vs, dbase, H, t, tid);          // dO rows
    __syncthreads();
    if (wave * 32 >= t) return;
    bf16x8_t kfr[2][2], vfr[2][2];
    float mbk[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = wave * 32 + kt * 16 + li;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kfr[kt][ks] = *reinterpret_cast<const bf16x8_t*>(base + H + (long long)key * ld + ks * 32 + g * 8);
            vfr[kt][ks] = *reinterpret_cast<const bf16x8_t*>(base + 2 * H + (long long)key * ld + ks * 32 + g * 8);
        }
        mbk[kt] = mb[key];
    }
    f32x4_t dk[2][4], dv[2][4];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[kt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[kt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    for (int I = 0; I < nJ; ++I) {
        float sv[2][8], dp[2][8];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            bf16x8_t qr[2], dor[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qr[ks] = perm_frag(Ks, 32 * I + 4 * tt, ks, lane);
                dor[ks] = perm_frag(Vs, 32 * I + 4 * tt, ks, lane);
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f}, dacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    acc = MC_MFMA_16x16x32(qr[ks], kfr[kt][ks], acc, 0, 0, 0);
                    dacc = MC_MFMA_16x16x32(dor[ks], vfr[kt][ks], dacc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { sv[kt][tt * 4 + r] = acc[r]; dp[kt][tt * 4 + r] = dacc[r]; }
            }
        }
        const int q0 = 32 * I + g * 8;
        float ls[16], dots[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 2 * q0 + 4 * i);
            ls[4 * i] = l4.x; ls[4 * i + 1] = l4.y; ls[4 * i + 2] = l4.z; ls[4 * i + 3] = l4.w;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 d4 = *reinterpret_cast<const float4*>(dot_s + q0 + 4 * i);
            dots[4 * i] = d4.x; dots[4 * i + 1] = d4.y; dots[4 * i + 2] = d4.z; dots[4 * i + 3] = d4.w;
        }
        bf16x8_t dsf[2], pdf[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int key = wave * 32 + kt * 16 + li;
            float pdv[8], dsv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float v = __expf(sv[kt][i] * a.alpha + mbk[kt] - ls[2 * i]) * ls[2 * i + 1];
                const float prr = bf2f(f2bf(v));
                float dsc = 1.f;
                if (DROP) dsc = ((dmask[(q0 + i) * 32 + (key >> 3)] >> (key & 7)) & 1) ? invkeep : 0.f;
                pdv[i] = v * dsc;
                dsv[i] = prr * (dp[kt][i] * dsc - dots[i]) * a.alpha;
            }
            pdf[kt] = as_frag(pdv);
            dsf[kt] = as_frag(dsv);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8_t qt = tr_frag(Ks, 32 * I, dt * 16, lane);
            const bf16x8_t dot_f = tr_frag(Vs, 32 * I, dt * 16, lane);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                dk[kt][dt] = MC_MFMA_16x16x32(qt, dsf[kt], dk[kt][dt], 0, 0, 0);
                dv[kt][dt] = MC_MFMA_16x16x32(dot_f, pdf[kt], dv[kt][dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int key = wave * 32 + kt * 16 + li;
        bf16_t* const dst = gbase + (long long)key * ld + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            *reinterpret_cast<uint2*>(dst + H + dt * 16) =
                make_uint2(pack_bf2(dk[kt][dt][0], dk[kt][dt][1]), pack_bf2(dk[kt][dt][2], dk[kt][dt][3]));
            *reinterpret_cast<uint2*>(dst + 2 * H + dt * 16) =
                make_uint2(pack_bf2(dv[kt][dt][0], dv[kt][dt][1]), pack_bf2(dv[kt][dt][2], dv[kt][dt][3]));
        }
    }
}

constexpr int FWD_LDS = 2 * TMAX * RS + TMAX * 4;
constexpr int BWD_LDS = 2 * TMAX * RS + TMAX * 4 * 4 + TMAX * 32;

int check_shape(int b, int t, int nh, float p) {
    MC_CHECK(b > 0 && nh > 0 && t >= 32 && t <= TMAX && t % 32 == 0, "attn: needs 32 <= t <= 256, t % 32 == 0 (head size 64)");
    MC_CHECK(p >= 0.f && p < 1.f, "attn: dropout p out of range");
    return MC_OK;
}

}  // namespace

extern "C" int mc_attn_supported(int t, int head_dim) { return head_dim == HD && t >= 32 && t <= TMAX && t % 32 == 0; }

extern "C" int mc_attn_fwd(const mc_bf16* qkv, const float* mask_bias, int b, int t, int nh, float alpha, float p,
                           unsigned long long seed, unsigned int stream_id, mc_bf16* ctx, float* lse, void* stream) {
    MC_CHECK(qkv && mask_bias && ctx && lse, "attn_fwd: null pointer");
    if (int e = check_shape(b, t, nh, p)) return e;
    attn_args a{};
    a.qkv = (const bf16_t*)qkv; a.maskb = mask_bias; a.ctx = (bf16_t*)ctx; a.lse = lse;
    a.t = t; a.nh = nh; a.alpha = alpha; a.p = p; a.seed = seed; a.sid = stream_id;
    static unsigned long long done_t = 0, done_f = 0;
    MC_SET_MAX_LDS(done_t, attn_fwd_k<true>, FWD_LDS);
    MC_SET_MAX_LDS(done_f, attn_fwd_k<false>, FWD_LDS);
    if (p > 0.f) hipLaunchKernelGGL(attn_fwd_k<true>, dim3(b * nh), dim3(512), FWD_LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(attn_fwd_k<false>, dim3(b * nh), dim3(512), FWD_LDS, (hipStream_t)stream, a);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

extern "C" int mc_attn_bwd(const mc_bf16* qkv, const float* mask_bias, const mc_bf16* dctx, const float* lse, int b, int t,
                           int nh, float alpha, float p, unsigned long long seed, unsigned int stream_id, mc_bf16* dqkv,
                           void* stream) {
    MC_CHECK(qkv && mask_bias && dctx && lse && dqkv, "attn_bwd: null pointer");
    if (int e = check_shape(b, t, nh, p)) return e;
    attn_args a{};
    a.qkv = (const bf16_t*)qkv; a.maskb = mask_bias; a.dctx = (const bf16_t*)dctx; a.dqkv = (bf16_t*)dqkv;
    a.lse = const_cast<float*>(lse);
    a.t = t; a.nh = nh; a.alpha = alpha; a.p = p; a.seed = seed; a.sid = stream_id;
    static unsigned long long done_t = 0, done_f = 0;
    MC_SET_MAX_LDS(done_t, attn_bwd_k<true>, BWD_LDS);
    MC_SET_MAX_LDS(done_f, attn_bwd_k<false>, BWD_LDS);
    if (p > 0.f) hipLaunchKernelGGL(attn_bwd_k<true>, dim3(b * nh), dim3(512), BWD_LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(attn_bwd_k<false>, dim3(b * nh), dim3(512), BWD_LDS, (hipStream_t)stream, a);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
