// Dependent-chain issue cost of VALU instructions at 1 / 2 / 3 waves per SIMD (occupancy limited through LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) float f2;
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    extern __shared__ float sm[];
    f2 a2[8]; float a[8];
    for (int i = 0; i < 8; ++i) { a2[i] = f2{seed + i, seed - i}; a[i] = seed + i; }
    f2 b2 = {seed, seed * 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2[i % NACC]) : "v"(b2), "v"(b2));
            if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i % NACC]) : "v"(seed), "v"(seed));
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + a2[i].x + a2[i].y;
    if (s == 12345.f) sm[threadIdx.x] = s;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int NACC> void run(const char* name, float* d, int wps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 256 * wps;
    const int lds = wps == 1 ? 100 * 1024 : wps == 2 ? 70 * 1024 : 50 * 1024;
    hipFuncSetAttribute((const void*)k<MODE, NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(grid), dim3(256), lds, 0, d, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(grid), dim3(256), lds, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst = (double)iters * 16 * wps;                  // wave-instructions per SIMD
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-14s acc=%d waves/SIMD=%d  %8.3f ms  %.2f cycles per wave-instruction per SIMD\n", name, NACC, wps, ms, cyc / inst);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int wps = 1; wps <= 3; ++wps) {
        run<0, 1>("v_pk_fma_f32", d, wps); run<0, 2>("v_pk_fma_f32", d, wps); run<0, 4>("v_pk_fma_f32", d, wps); run<0, 8>("v_pk_fma_f32", d, wps);
        run<1, 1>("v_fma_f32", d, wps); run<1, 2>("v_fma_f32", d, wps); run<1, 4>("v_fma_f32", d, wps); run<1, 8>("v_fma_f32", d, wps);
    }
    return 0;
}
