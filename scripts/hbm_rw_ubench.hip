// HBM ceilings by read : write mix (developer tool): pure read (sum), pure write (fill), copy (1:1), 2 reads : 1 write.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_fill(uint4* o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = make_uint4(1, 2, 3, 4);
}
__global__ __launch_bounds__(256) void k_read(const uint4* a, size_t n, unsigned* out) {
    unsigned s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { uint4 v = a[i]; s += v.x ^ v.y ^ v.z ^ v.w; }
    if (s == 0x12345u) out[0] = s;
}
__global__ __launch_bounds__(256) void k_copy(const uint4* a, uint4* o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = a[i];
}
__global__ __launch_bounds__(256) void k_2r1w(const uint4* a, const uint4* b, uint4* o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 x = a[i], y = b[i];
        o[i] = make_uint4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}
__global__ __launch_bounds__(256) void k_1r6w(const uint4* a, uint4* o, size_t n) {      // expand-like: 1 read, 6 writes
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 x = a[i];
        for (int j = 0; j < 6; ++j) o[i * 6 + j] = make_uint4(x.x + j, x.y, x.z, x.w);
    }
}
int main() {
    const size_t bytes = 3ull << 30, n = bytes / 16;
    uint4 *a, *b, *o; unsigned* out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, bytes); hipMalloc(&out, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192, 32768}) {
        auto t = [&](const char* nm, double gb, auto fn) {
            fn(); hipEventRecord(e0); for (int i = 0; i < 5; ++i) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            printf("grid %6d %-10s %7.3f ms  %6.2f TB/s\n", grid, nm, ms, gb / ms);
        };
        t("fill", 3.221, [&] { hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, o, n); });
        t("read", 3.221, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, out); });
        t("copy", 6.442, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, o, n); });
        t("2r1w", 9.663, [&] { hipLaunchKernelGGL(k_2r1w, dim3(grid), dim3(256), 0, 0, a, b, o, n); });
        t("1r6w", 3.758, [&] { hipLaunchKernelGGL(k_1r6w, dim3(grid), dim3(256), 0, 0, a, o, n / 6); });
    }
    hipEventRecord(e0); for (int i = 0; i < 5; ++i) hipMemsetAsync(o, 0, bytes, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("hipMemsetAsync %7.3f ms %6.2f TB/s\n", ms / 5, 3.221 / (ms / 5));
    return 0;
}
