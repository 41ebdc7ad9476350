// Does the 256 MB memory-side cache (MALL / Infinity Cache) serve a tensor that is re-read right after it was read or
// written?  read-after-read and read-after-write rates by tensor size (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_read(const uint4* a, size_t n, unsigned* out) {
    unsigned s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { uint4 v = a[i]; s += v.x ^ v.y ^ v.z ^ v.w; }
    if (s == 0x12345u) out[0] = s;
}
__global__ __launch_bounds__(256) void k_fill(uint4* o, size_t n, unsigned v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = make_uint4(v, 2, 3, 4);
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_fill_nt(uint4* o, size_t n, unsigned v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        u32x4 w = {v, 2, 3, 4};
        __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(o + i));
    }
}
__global__ __launch_bounds__(256) void k_read_nt(const uint4* a, size_t n, unsigned* out) {
    unsigned s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a + i)); s += v.x ^ v.y ^ v.z ^ v.w; }
    if (s == 0x12345u) out[0] = s;
}
int main() {
    const size_t maxb = 2ull << 30;
    uint4 *a, *big; unsigned* out;
    hipMalloc(&a, maxb); hipMalloc(&big, maxb); hipMalloc(&out, 4);
    hipMemset(a, 1, maxb); hipMemset(big, 1, maxb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        float ms;
        // (1) flush with a 2 GB read of another buffer, then read `a` twice: second read = read-after-read
        hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, big, maxb / 16, out);
        hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, a, n, out);
        hipEventRecord(e0); hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, a, n, out); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); const double rar = bytes / ms / 1e9;
        // (2) flush, write `a`, read it: read-after-write
        hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, big, maxb / 16, out);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, a, n, 7u);
        hipEventRecord(e0); hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, a, n, out); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); const double raw = bytes / ms / 1e9;
        // (3) cold read
        hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, big, maxb / 16, out);
        hipEventRecord(e0); hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, a, n, out); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); const double cold = bytes / ms / 1e9;
        // (4) flush, NON-TEMPORAL write of `a`, read it
        hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, big, maxb / 16, out);
        hipLaunchKernelGGL(k_fill_nt, dim3(4096), dim3(256), 0, 0, a, n, 7u);
        hipEventRecord(e0); hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, a, n, out); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); const double rawnt = bytes / ms / 1e9;
        // (5) plain write, non-temporal read
        hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, big, maxb / 16, out);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, a, n, 7u);
        hipEventRecord(e0); hipLaunchKernelGGL(k_read_nt, dim3(4096), dim3(256), 0, 0, a, n, out); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); const double rntaw = bytes / ms / 1e9;
        // (6) write + read pair timed together: plain vs nt write
        hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, big, maxb / 16, out);
        hipEventRecord(e0); hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, a, n, 7u); hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, a, n, out); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); const double pair = ms;
        hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, big, maxb / 16, out);
        hipEventRecord(e0); hipLaunchKernelGGL(k_fill_nt, dim3(4096), dim3(256), 0, 0, a, n, 7u); hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, a, n, out); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); const double pairnt = ms;
        printf("%5zu MB: cold %6.2f TB/s   read-after-read %6.2f   read-after-write %6.2f   read-after-NT-write %6.2f   NT-read-after-write %6.2f | write+read %.3f ms, NT-write+read %.3f ms\n", mb, cold, rar, raw, rawnt, rntaw, pair, pairnt);
    }
    return 0;
}
