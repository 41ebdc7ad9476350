// Standalone micro-benchmark / phase profiler for the 256 x 256 tile GEMM (developer tool, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DG256_PROF] scripts/g256bench.hip -o scripts/g256bench.bin
#include "../mammo_clip_amd/csrc/gemm256.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
static char g_err_msg[256];
extern "C" void mc_set_error(const char* m) { snprintf(g_err_msg, sizeof g_err_msg, "%s", m); }
#define HC(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void fill_rand(bf16_t* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = f2bf(((int)(x & 0xffff) - 32768) / 32768.0f);       // uniform [-1, 1)
    }
}

int main() {
    struct Shape { long long M; int N; long long K; int batch; };
    std::vector<Shape> shapes = {{8192, 8192, 8192, 1}, {4096, 4096, 4096, 1}, {44544, 1824, 304, 1}, {44544, 304, 1824, 1}, {44544, 3072, 512, 1},
                                 {44544, 512, 3072, 1}, {173280, 176, 1056, 1}, {16384, 2304, 768, 1}, {16384, 768, 3072, 1}, {1392, 304, 1824, 32}};
    if (getenv("G256_FEW")) shapes.resize(3);
    size_t maxel = (size_t)173280 * 3072;
    bf16_t *A, *B, *C;
    HC(hipMalloc(&A, maxel * 2)); HC(hipMalloc(&B, maxel * 2)); HC(hipMalloc(&C, maxel * 2));
    hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, A, maxel, 1u);
    hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, B, maxel, 7u);
    HC(hipDeviceSynchronize());
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    for (auto s : shapes) {
        mc_gemm_args a = {};
        a.A = A; a.B = B; a.C = C; a.M = s.M; a.N = s.N; a.K = s.K; a.batch = s.batch; a.nb2 = 1; a.splits = 1;
        a.lda = s.K; a.ldb = s.K; a.ldc = s.N; a.sA1 = s.M * s.K; a.sB1 = (long long)s.N * s.K; a.sC1 = s.M * s.N;
        if (mc_gemm256_launch(&a, nullptr)) { printf("err %s\n", g_err_msg); return 1; }
        HC(hipDeviceSynchronize());
#ifdef G256_PROF
        unsigned long long z[2][12] = {}; HC(hipMemcpyToSymbol(HIP_SYMBOL(g256::g_g256_prof), z, sizeof(z)));
#endif
        HC(hipEventRecord(e0));
        const int it = 5;
        for (int i = 0; i < it; ++i) mc_gemm256_launch(&a, nullptr);
        HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
        float ms; HC(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
        double fl = 2.0 * s.batch * s.M * s.N * s.K, by = 2.0 * s.batch * (s.M * s.K + s.N * s.K + s.M * s.N);
        printf("b=%2d M=%7lld N=%5d K=%5lld %8.1f us %7.1f TF %7.0f GB/s", s.batch, s.M, s.N, s.K, ms * 1e3, fl / ms / 1e9, by / ms / 1e6);
#ifdef G256_PROF
        HC(hipMemcpyFromSymbol(z, HIP_SYMBOL(g256::g_g256_prof), sizeof(z)));
        for (int g = 0; g < 2; ++g) {
            double kt = (double)z[g][7];
            printf("\n    wave row %d per K tile (cycles): ds_read issue %.0f  dma issue %.0f  lgkm8 %.0f  bar1 %.0f  lgkm %.0f  mfma %.0f  bar2 %.0f  ph4-issue+vmcnt %.0f  epilogue %.0f   sum %.0f",
                   g, z[g][8] / kt, z[g][9] / kt, z[g][0] / kt, z[g][1] / kt, z[g][2] / kt, z[g][3] / kt, z[g][4] / kt, z[g][5] / kt, z[g][6] / kt,
                   (z[g][0] + z[g][1] + z[g][2] + z[g][3] + z[g][4] + z[g][5] + z[g][6] + z[g][8] + z[g][9]) / kt);
        }
#endif
        printf("\n");
    }
    return 0;
}
