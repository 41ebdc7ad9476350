// Per-tensor-scaled OCP e4m3 quantisation of bf16 tensors for the fp8 pointwise-convolution path (BASELINE config #5:
// "EfficientNet-B5 fp8 weights/activations", the 1x1 convolutions of efficientnet_custom.py:104,122,283).
//   scale  = 448 / amax           (448 = largest finite e4m3 value; gfx950 implements the OCP format, not fnuz)
//   q      = e4m3(clamp(x * scale, -448, 448))        v_cvt_pk_fp8_f32, round to nearest even
//   x      ~ q * (amax / 448)
// amax comes from the caller: either measured on this tensor first (mc_amax_bf16, "current scaling") or carried over from
// the previous step (delayed scaling: the quantisation pass also measures the amax of the tensor it converts).
// HBM-bound streaming kernels: 16-byte loads, 8-byte stores, one integer atomic per workgroup for the amax
// (non-negative floats order like their bit patterns: deterministic).
#include "common_hip.h"
#include "../../include/mammoclip_hip.h"

namespace {

// NaN-propagating max of non-negative values (fmaxf drops a NaN operand: a NaN activation would be quantised to a
// finite value and the divergence hidden).  A NaN amax has the largest bit pattern, so the integer atomicMax keeps it;
// with a NaN amax the scale is NaN and every quantised value of the tensor comes out NaN (e4m3 0x7f).
__device__ __forceinline__ float nmax(float a, float b) { return (a != a || a > b) ? a : b; }
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = nmax(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(256) void amax_bf16_k(const bf16_t* __restrict__ x, long long nv, unsigned* __restrict__ amax) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
        float f[8];
        unpack8(nt_load16(x + i * 8), f);
#pragma unroll
        for (int q = 0; q < 8; ++q) m = nmax(fabsf(f[q]), m);
    }
    m = wave_max_f(m);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(amax, __float_as_uint(nmax(nmax(red[0], red[1]), nmax(red[2], red[3]))));
}

__global__ __launch_bounds__(256) void quant_fp8_bf16_k(const bf16_t* __restrict__ x, long long nv, const float* __restrict__ amax_in,
                                                        unsigned char* __restrict__ y, float* __restrict__ scale_out,
                                                        unsigned* __restrict__ amax_next) {
    const float amax = amax_in[0];
    const float scale = amax != amax ? amax : (amax > 0.f ? 448.0f / amax : 1.0f);
    if (blockIdx.x == 0 && threadIdx.x == 0 && scale_out) scale_out[0] = amax > 0.f ? amax / 448.0f : 1.0f;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
        float f[8];
        unpack8(nt_load16(x + i * 8), f);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            m = nmax(fabsf(f[q]), m);
            const float v = f[q] * scale;
            f[q] = v != v ? v : fminf(fmaxf(v, -448.0f), 448.0f);          // clamp finite values, let NaN through
        }
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
        *reinterpret_cast<uint2*>(y + i * 8) = make_uint2((unsigned)lo, (unsigned)hi);
    }
    if (amax_next) {
        m = wave_max_f(m);
        __shared__ float red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(amax_next, __float_as_uint(nmax(nmax(red[0], red[1]), nmax(red[2], red[3]))));
    }
}

}  // namespace

static int fp8_grid(long long nv) {
    long long g = (nv + 255) / 256;
    return (int)(g < 2048 ? (g < 1 ? 1 : g) : 2048);
}

extern "C" int mc_amax_bf16(const mc_bf16* x, long long n, float* amax, void* stream) {
    MC_CHECK(x && amax && n > 0 && n % 8 == 0 && mc_aligned16(x), "amax_bf16: x (16-byte aligned, n % 8 == 0) and amax are required");
    hipLaunchKernelGGL(amax_bf16_k, dim3(fp8_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, x, n / 8, reinterpret_cast<unsigned*>(amax));
    MC_LAUNCH_CHECK();
    return MC_OK;
}

extern "C" int mc_quant_fp8_bf16(const mc_bf16* x, long long n, const float* amax_in, unsigned char* y, float* scale_out,
                                 float* amax_next, void* stream) {
    MC_CHECK(x && y && amax_in && n > 0 && n % 8 == 0 && mc_aligned16(x) && (((uintptr_t)y) & 7u) == 0,
             "quant_fp8_bf16: x (16-byte aligned), y (8-byte aligned), amax_in are required, n % 8 == 0");
    hipLaunchKernelGGL(quant_fp8_bf16_k, dim3(fp8_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, x, n / 8, amax_in, y, scale_out,
                       reinterpret_cast<unsigned*>(amax_next));
    MC_LAUNCH_CHECK();
    return MC_OK;
}
