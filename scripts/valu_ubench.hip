#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(2))) float f2;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a[16]; f2 a2[16];
    unsigned w = __float_as_uint(seed) | 0x3f803f80u;
    for (int i = 0; i < 16; ++i) { a[i] = seed + i; a2[i] = f2{seed + i, seed - i}; }
    f2 b2 = {seed, seed * 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], seed, 1.0f);
            if (MODE == 1) a2[i] = __builtin_elementwise_fma(a2[i], b2, b2);
            if (MODE == 2) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(w + i));
            if (MODE == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (MODE == 6) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(a[i]));
            if (MODE == 7) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(a[i]));
            if (MODE == 8) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (MODE == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (MODE == 10) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2[i]) : "v"(b2));
            if (MODE == 11) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a2[i]) : "v"(b2));
            if (MODE == 12) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (MODE == 13) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2[i]) : "v"(a2[(i + 5) & 15]), "v"(b2));
            if (MODE == 14) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(seed));
            if (MODE == 15) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(a[(i + 5) & 15]));
#if defined(TRY_DOT2)
            if (MODE == 3) a[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, w), __builtin_bit_cast(bf2, w + i), a[i], false);
#endif
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i] + a2[i].x + a2[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* d, int perinst) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 256 * 8;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst = (double)grid * 256 / 64 * iters * 16;      // wave-instructions
    double cyc = ms * 1e-3 * 2.4e9 * 1024;                    // SIMD-cycles available at 2.4 GHz
    printf("%-22s %8.3f ms  %.2f SIMD-cycles per wave-instruction (at 2.4 GHz), %.1f T MAC/s\n", name, ms, cyc / inst, inst * 64 * perinst / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", d, 1); run<1>("v_pk_fma_f32", d, 2); run<2>("v_dot2c_f32_bf16", d, 2);
    run<4>("v_exp_f32", d, 1); run<5>("v_rcp_f32", d, 1); run<6>("v_lshlrev_b32", d, 1); run<7>("v_and_b32 lit", d, 1);
    run<8>("v_cvt_pk_bf16_f32", d, 1); run<9>("v_mul_f32", d, 1); run<10>("v_pk_mul_f32", d, 2); run<11>("v_pk_add_f32", d, 2);
    run<12>("v_add_f32", d, 1); run<13>("v_pk_fma 3 distinct", d, 2); run<14>("v_fma 3 distinct", d, 1); run<15>("v_mov_b32", d, 1);
#if defined(TRY_DOT2)
    run<3>("v_dot2_f32_bf16", d, 2);
#endif
    return 0;
}
