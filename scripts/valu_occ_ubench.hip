// VALU issue rate by instruction form and by resident waves per SIMD (developer micro-benchmark, gfx950).
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_occ_ubench.hip -o /tmp/valu_occ && /tmp/valu_occ
// Occupancy is set through the dynamic LDS size of 256-thread workgroups (k workgroups per CU = k waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) float f2;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed, const float* wsrc) {
    extern __shared__ float dyn[];
    float a[16]; f2 a2[16];
    unsigned w = __float_as_uint(seed) | 0x3f803f80u;
    for (int i = 0; i < 16; ++i) { a[i] = seed + i; a2[i] = f2{seed + i, seed - i}; }
    f2 b2 = {seed, seed * 0.5f};
    // wave-uniform operands (scalar loads -> SGPRs)
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float s0 = wsrc[wv], s1 = wsrc[wv + 4];
    unsigned su = __float_as_uint(s0) | 0x3f803f80u;
    f2 sp = {s0, s1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(seed));
            if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 5) & 15]), "s"(s0));
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2[i]) : "v"(a2[(i + 5) & 15]), "v"(b2));
            if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2[i]) : "v"(a2[(i + 5) & 15]), "s"(sp));
            if (MODE == 4) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(w + i));
            if (MODE == 5) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "s"(su), "v"(w + i));
            if (MODE == 6) asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(seed));
            if (MODE == 7) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(seed));
            if (MODE == 8) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w), "v"(w + i));
            if (MODE == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2[i]) : "v"(b2));
            if (MODE == 10) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "v"(w));
            if (MODE == 11) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (MODE == 12) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 5) & 15]));
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i] + a2[i].x + a2[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s + dyn[threadIdx.x & 7];
}
template <int MODE> void run(const char* name, float* d, const float* ws, int macs) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-28s", name);
    const int occs[] = {1, 2, 3, 4, 6, 8};
    for (int oc : occs) {
        const int lds = 160 * 1024 / oc - 1024;
        hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        const int iters = 8000, grid = 256 * oc;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), lds, 0, d, 10, 1.0f, ws);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), lds, 0, d, iters, 1.0f, ws);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double inst = (double)grid * 4 * iters * 16;              // wave-instructions
        double cyc = ms * 1e-3 * 2.4e9 * 1024;                    // SIMD-cycles available at 2.4 GHz
        printf("  w%d %5.2f", oc, cyc / inst);
    }
    printf("   (SIMD cycles per wave-instruction at 2.4 GHz; %d MAC/lane)\n", macs);
}
int main() {
    float *d, *ws; hipMalloc(&d, 256 * 8 * 256 * 4); hipMalloc(&ws, 64); hipMemset(ws, 0, 64);
    run<0>("v_fma_f32 vgpr", d, ws, 1); run<1>("v_fma_f32 sgpr tap", d, ws, 1);
    run<2>("v_pk_fma_f32 vgpr", d, ws, 2); run<3>("v_pk_fma_f32 sgpr pair", d, ws, 2);
    run<4>("v_dot2c_f32_bf16 vgpr", d, ws, 2); run<5>("v_dot2c_f32_bf16 sgpr", d, ws, 2);
    run<8>("v_dot2_f32_bf16 (vop3p)", d, ws, 2);
    run<6>("v_fmac_f32_dpp wave_shr:1", d, ws, 1); run<7>("v_fmac_f32_dpp row_shr:1", d, ws, 1);
    run<12>("v_mov_b32_dpp wave_shr:1", d, ws, 0);
    run<9>("v_pk_mul_f32", d, ws, 0); run<10>("v_perm_b32", d, ws, 0); run<11>("v_cvt_pk_bf16_f32", d, ws, 0);
    return 0;
}
