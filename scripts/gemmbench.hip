// Standalone micro-benchmark / phase profiler for the tiled GEMM (developer tool, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DGEMM_PROF] scripts/gemmbench.hip -o scripts/gemmbench.bin
#include "../mammo_clip_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
static char g_err_msg[256];
extern "C" void mc_set_error(const char* m) { snprintf(g_err_msg, sizeof g_err_msg, "%s", m); }
extern "C" const char* mc_last_error(void) { return g_err_msg; }
#define HC(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    struct Shape { const char* kind; long long M; int N; long long K; };   // GEMM dims as passed to mc_gemm_bf16
    std::vector<Shape> shapes = {
        {"fwd", 173280, 176, 1056}, {"fwd", 173280, 1056, 176}, {"fwd", 44544, 304, 1824}, {"fwd", 44544, 3072, 512}, {"fwd", 44544, 512, 3072},
        {"fwd", 8192, 768, 768}, {"fwd", 8192, 3072, 768}, {"fwd", 8192, 768, 3072}, {"fwd", 44544, 1824, 304}, {"fwd", 173280, 768, 128}, {"fwd", 173280, 128, 768},
        {"wgrad", 1056, 176, 173280}, {"wgrad", 176, 1056, 173280}, {"wgrad", 1824, 304, 44544}, {"wgrad", 512, 3072, 44544}};
    size_t maxel = (size_t)173280 * 3072;
    bf16_t *A, *B, *C; float* ws;
    HC(hipMalloc(&A, maxel * 2)); HC(hipMalloc(&B, maxel * 2)); HC(hipMalloc(&C, maxel * 2));
    HC(hipMemset(A, 0x3c, maxel * 2)); HC(hipMemset(B, 0x3c, maxel * 2));
    HC(hipMalloc(&ws, (size_t)64 * 3072 * 512 * 4));
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    for (auto s : shapes) {
        mc_gemm_args a = {};
        a.A = A; a.B = B; a.C = C; a.M = s.M; a.N = s.N; a.K = s.K; a.batch = 1; a.nb2 = 1; a.splits = 1;
        bool wg = s.kind[0] == 'w';
        if (wg) {   // dW[M x N] = A[K x M]^T B[K x N], fp32 out, split-K through the workspace
            a.a_kmajor = 1; a.b_kmajor = 1; a.c_f32 = 1; a.lda = s.M; a.ldb = s.N; a.ldc = s.N;
            long long tiles = ((s.M + 127) / 128) * ((s.N + 127) / 128), kt = (s.K + 63) / 64;
            const char* tg = getenv("WG_TARGET"); long long target = tg ? atoll(tg) : 1024; long long sp = (target + tiles - 1) / tiles; if (sp > kt / 8) sp = kt / 8; if (sp < 1) sp = 1; if (sp >= 16) sp -= sp % 8;
            a.splits = (int)sp; a.splitk_ws = ws;
        } else { a.lda = s.K; a.ldb = s.K; a.ldc = s.N; }
        if (mc_gemm_bf16(&a, nullptr)) { printf("err %s\n", mc_last_error()); return 1; }
        HC(hipDeviceSynchronize());
#ifdef GEMM_PROF
        unsigned long long z[8] = {0}; HC(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_prof), z, sizeof(z)));
#endif
        HC(hipEventRecord(e0));
        const int it = 5;
        for (int i = 0; i < it; ++i) mc_gemm_bf16(&a, nullptr);
        HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
        float ms; HC(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
        printf("%-5s M=%7lld N=%5d K=%7lld splits=%2d %7.3f ms %6.1f TF", s.kind, s.M, s.N, s.K, a.splits, ms, 2.0 * s.M * s.N * s.K / ms / 1e9);
#ifdef GEMM_PROF
        HC(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_gemm_prof), sizeof(z)));
        double tot = (double)(z[0] + z[1] + z[2] + z[3] + z[4]);
        printf("  | store %4.1f%% barrier %4.1f%% issue %4.1f%% mfma %4.1f%% epilogue %4.1f%%  ksteps %llu cyc/step %.0f", 100 * z[0] / tot,
               100 * z[1] / tot, 100 * z[2] / tot, 100 * z[3] / tot, 100 * z[4] / tot, z[5], tot / (z[5] ? z[5] : 1));
#endif
        printf("\n");
    }
    return 0;
}
