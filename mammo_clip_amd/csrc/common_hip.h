// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the Mammo-CLIP hot path.
// wave = 64 lanes; activations are bf16 (raw uint16 bits) in NHWC / row-major [rows, channels]
// layout with channels % 8 == 0, so every access is a 16-byte vector of 8 channels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The 16-bit storage / MFMA operand type is a property of the BUILD: bf16 (default: libmammoclip_hip.so) or IEEE f16
// (-DMC_F16: libmammoclip_hip_f16.so, the reference's AMP dtype [ref: trainer.py:271-278] -- 10 mantissa bits, same MFMA
// rate; the configuration in which |dloss| <= 1e-3 holds in train mode).  Kernels never look inside a
// 16-bit value except through the helpers below (bf2f / bf_lo / bf_hi / pack_bf2) and MC_MFMA_16x16x32, so both builds
// come from the same sources; names keep the "bf" of the default build.
typedef unsigned short bf16_t;
#ifdef MC_F16
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8_t;
#define MC_MFMA_16x16x32(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z)
#else
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
#define MC_MFMA_16x16x32(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z)
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define MC_OK 0
#define MC_ERR_ARG 1
#define MC_ERR_LAUNCH 2

extern "C" void mc_set_error(const char* msg);   // api_util.hip

#define MC_CHECK(cond, msg)                     \
    do {                                        \
        if (!(cond)) {                          \
            mc_set_error(msg);                  \
            return MC_ERR_ARG;                  \
        }                                       \
    } while (0)

#define MC_LAUNCH_CHECK()                                     \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) {                              \
            mc_set_error(hipGetErrorString(e__));             \
            return MC_ERR_LAUNCH;                             \
        }                                                     \
    } while (0)

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
#ifdef MC_F16
typedef __attribute__((ext_vector_type(2))) _Float16 bf16x2_t;
__device__ __forceinline__ float bf2f(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ float bf_lo(uint32_t w) { return (float)__builtin_bit_cast(bf16x2_t, w).x; }
__device__ __forceinline__ float bf_hi(uint32_t w) { return (float)__builtin_bit_cast(bf16x2_t, w).y; }
#else
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
#endif

// fp32 -> 16 bit, round-to-nearest-even (same rounding as torch's float -> bfloat16 / float16): gfx950 has a packed
// hardware convert (v_cvt_pk_bf16_f32) -- one instruction per two elements instead of ~6 integer ops per element.
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
    f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]);
    v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
    return v;
}
__device__ __forceinline__ void load8f(const float* p, float* f) {   // p 16-byte aligned
    float4 a = *reinterpret_cast<const float4*>(p);
    float4 b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// streaming (read-once / write-once) 16-byte accesses: non-temporal cache policy
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store16(void* p, const uint4& v) {
    u32x4_t w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t*>(p));
}
__device__ __forceinline__ uint4 nt_load16(const void* p) {
    u32x4_t w = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(w.x, w.y, w.z, w.w);
}

// v_exp_f32 + v_rcp_f32 (1 ulp): no IEEE division sequence on the streaming paths.
// The streaming kernels run at 2-3 waves per SIMD, where every VALU instruction costs the same ~5 issue cycles whether it
// is packed or not (scripts/valu_dep_ubench.hip): the arithmetic AROUND the two transcendentals is therefore written on
// fp32 PAIRS (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) -- 7 instructions per two sigmoids instead of 10.  The scalar
// forms evaluate the same expression, bit for bit.
#define MC_NLOG2E -1.4426950408889634f
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * MC_NLOG2E)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
// d silu(z)/dz = s(1 + z(1-s))  (efficient_net_custom_utils.py:71-75)
__device__ __forceinline__ float silu_grad_f(float z) {
    float s = sigmoid_f(z);
    return s * (1.0f + z * (1.0f - s));
}
__device__ __forceinline__ f32x2_t sigmoid2_f(f32x2_t z) {
    const f32x2_t t = z * f32x2_t{MC_NLOG2E, MC_NLOG2E};
    const f32x2_t d = f32x2_t{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + f32x2_t{1.f, 1.f};
    return f32x2_t{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
__device__ __forceinline__ f32x2_t silu2_f(f32x2_t z) { return z * sigmoid2_f(z); }
__device__ __forceinline__ f32x2_t silu_grad2_f(f32x2_t z) {
    const f32x2_t s = sigmoid2_f(z);
    return s * __builtin_elementwise_fma(z, f32x2_t{1.f, 1.f} - s, f32x2_t{1.f, 1.f});
}
// f[q] = silu(f[q] * s[q] + t[q]), q < 8
__device__ __forceinline__ void bn_silu8(float* f, const float* s, const float* t) {
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        const f32x2_t z = silu2_f(__builtin_elementwise_fma(f32x2_t{f[q], f[q + 1]}, f32x2_t{s[q], s[q + 1]}, f32x2_t{t[q], t[q + 1]}));
        f[q] = z.x; f[q + 1] = z.y;
    }
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- Philox4x32-10 counter-based RNG: dropout masks are a pure function of (seed, stream, index),
// so the backward pass (and a re-forward) regenerates them instead of storing them.
__device__ __forceinline__ uint4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
// 8 keep/scale factors for the 8 consecutive elements starting at element index idx8*8
__device__ __forceinline__ void dropout_scale8(unsigned long long seed, uint32_t stream,
                                               unsigned long long idx8, float p, float* s) {
    float inv = 1.0f / (1.0f - p);
    uint32_t thr = (uint32_t)(p * 65536.0f);
    uint4 r = philox4x32((uint32_t)idx8, (uint32_t)(idx8 >> 32), stream, 0x5bd1e995u,
                         (uint32_t)seed, (uint32_t)(seed >> 32));
    uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s[2 * i] = ((w[i] & 0xffffu) >= thr) ? inv : 0.0f;
        s[2 * i + 1] = ((w[i] >> 16) >= thr) ? inv : 0.0f;
    }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: a process that touches more than one device (the
// shared-device multi-rank modes, a test on cuda:1) must set it on each.  `done` = a 64-bit per-call-site device mask.
#define MC_SET_MAX_LDS(done, func, bytes)                                                                          \
    do {                                                                                                           \
        int dev__ = 0;                                                                                             \
        (void)hipGetDevice(&dev__);                                                                                \
        const unsigned long long bit__ = 1ULL << (dev__ & 63);                                                     \
        if (!((done) & bit__)) {                                                                                   \
            if (hipFuncSetAttribute((const void*)(func), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)) != hipSuccess) { \
                mc_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");                            \
                return MC_ERR_LAUNCH;                                                                              \
            }                                                                                                      \
            (done) |= bit__;                                                                                       \
        }                                                                                                          \
    } while (0)

static inline int mc_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline bool mc_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
