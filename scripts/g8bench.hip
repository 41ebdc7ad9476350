// Standalone benchmark + correctness screen of the 256 x 256 tile GEMM (gemm256.hip).  Developer tool, not part of the
// library.  (Round 3 used it to A/B the rewrite against round 2's kernel: 8192^3 1374 vs 1024 TFLOP/s, see DESIGN.md.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/g8bench.hip -o scripts/g8bench.bin
#include "../mammo_clip_amd/csrc/gemm256.hip"
#include "../mammo_clip_amd/csrc/gemm256_tn.hip"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
static char g_err_msg[256];
extern "C" void mc_set_error(const char* m) { snprintf(g_err_msg, sizeof g_err_msg, "%s", m); }
#define HC(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void fill_rand(bf16_t* p, size_t n, unsigned seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = f2bf(scale * (((int)(x & 0xffff) - 32768) / 32768.0f));       // uniform [-scale, scale)
    }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((int)(x & 0xffff) - 32768) / 32768.0f;
    }
}
// reference for sampled outputs: sample s -> (z, m, n); err[s] = |C - ref| / (1 + |ref|)
__global__ void ref_check(const bf16_t* A, const bf16_t* B, const bf16_t* C, const float* bias, const bf16_t* R, long long M, int N, long long K,
                          long long lda, long long ldb, long long ldc, int batch, int nsamp, int full, float* err) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsamp) return;
    long long z, m, n;
    if (full) { z = s / (M * N); long long r = s - z * M * N; m = r / N; n = r % N; }
    else {
        unsigned x = (unsigned)s * 2654435761u + 12345u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        unsigned y = x * 3266489917u; y ^= y >> 16;
        z = x % batch; m = (long long)(y % (unsigned)M); n = (x >> 8) % N;
        if ((s & 7) == 0) m = M - 1 - (y % 256 < M ? y % 256 : 0);         // bias samples to the last row block / column tile
        if ((s & 7) == 1) n = N - 1 - ((x >> 8) % 256 < (unsigned)N ? (x >> 8) % 256 : 0);
    }
    const bf16_t* a = A + z * M * lda + m * lda;
    const bf16_t* b = B + z * (long long)N * ldb + n * ldb;
    float acc = 0.f;
    for (long long k = 0; k < K; ++k) acc += bf2f(a[k]) * bf2f(b[k]);
    if (bias) acc += bias[n];
    acc = bf2f(f2bf(acc));
    if (R) acc = bf2f(f2bf(acc + bf2f(R[z * M * ldc + m * ldc + n])));
    float c = bf2f(C[z * M * ldc + m * ldc + n]);
    err[s] = fabsf(c - acc) / (1.f + fabsf(acc));
}

// TN reference for sampled outputs of C[M,N] = sum_k A[k][m] B[k][n], C given as `splits` fp32 partial tiles
__global__ void ref_check_tn(const bf16_t* A, const bf16_t* B, const float* ws, int splits, long long M, long long N, long long K,
                             long long lda, long long ldb, int nsamp, int full, float* err) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsamp) return;
    long long m, n;
    if (full) { m = s / N; n = s % N; }
    else {
        unsigned x = (unsigned)s * 2654435761u + 12345u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        unsigned y = x * 3266489917u; y ^= y >> 16;
        m = y % (unsigned)M; n = (x >> 8) % (unsigned)N;
        if ((s & 7) == 0) m = M - 1 - (y % 64);
        if ((s & 7) == 1) n = N - 1 - ((x >> 8) % 64);
    }
    float acc = 0.f;
    for (long long k = 0; k < K; ++k) acc += bf2f(A[k * lda + m]) * bf2f(B[k * ldb + n]);
    float c = 0.f;
    for (int q = 0; q < splits; ++q) c += ws[(long long)q * M * N + m * N + n];
    err[s] = fabsf(c - acc) / (1.f + fabsf(acc));
}

struct Shape { long long M; int N; long long K; int batch; int extra; };   // extra: 1 = bias + residual + stats

static int launch(int, mc_gemm_args* a) { return mc_gemm256_launch(a, nullptr); }

int main(int argc, char** argv) {
    std::vector<Shape> check = {{256, 256, 64, 1, 0}, {300, 200, 72, 1, 0}, {257, 8, 8, 1, 0}, {512, 176, 304, 1, 1}, {1000, 304, 1824, 2, 1},
                                {5415, 176, 1056, 3, 1}, {2048, 768, 3072, 1, 1}, {4096, 4096, 512, 1, 0}, {44544, 3072, 512, 1, 1}};
    std::vector<Shape> shapes = {{8192, 8192, 8192, 1, 0}, {4096, 4096, 4096, 1, 0}, {44544, 1824, 304, 1, 0}, {44544, 304, 1824, 1, 0}, {44544, 3072, 512, 1, 0},
                                 {44544, 512, 3072, 1, 0}, {173280, 176, 1056, 1, 0}, {16384, 2304, 768, 1, 0}, {16384, 3072, 768, 1, 0}, {16384, 768, 3072, 1, 0},
                                 {1392, 304, 1824, 32, 0}, {5415, 176, 1056, 32, 0},
                                 {173280, 1056, 176, 1, 0}, {173280, 768, 128, 1, 0}, {173280, 176, 768, 1, 0}, {173280, 128, 768, 1, 0}, {693120, 384, 64, 1, 0}, {44544, 304, 1056, 1, 0}};
    size_t maxel = (size_t)693120 * 1024;
    bf16_t *A, *B, *C, *R; float *bias, *stats, *err;
    HC(hipMalloc(&A, maxel * 2)); HC(hipMalloc(&B, maxel * 2)); HC(hipMalloc(&C, maxel * 2)); HC(hipMalloc(&R, maxel * 2));
    HC(hipMalloc(&bias, 65536 * 4)); HC(hipMalloc(&stats, (size_t)64 << 20)); HC(hipMalloc(&err, (size_t)4 << 20));
    hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, A, maxel, 1u, 1.0f);
    hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, B, maxel, 7u, 1.0f);
    hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, R, maxel, 9u, 4.0f);
    hipLaunchKernelGGL(fill_f32, dim3(64), dim3(256), 0, 0, bias, (size_t)65536, 3u);
    HC(hipDeviceSynchronize());
    auto mk = [&](const Shape& s) {
        mc_gemm_args a = {};
        a.A = A; a.B = B; a.C = C; a.M = s.M; a.N = s.N; a.K = s.K; a.batch = s.batch; a.nb2 = 1; a.splits = 1;
        a.lda = s.K; a.ldb = s.K; a.ldc = s.N; a.sA1 = s.M * s.K; a.sB1 = (long long)s.N * s.K; a.sC1 = s.M * s.N;
        if (s.extra) { a.bias = bias; a.R = R; a.ldr = s.N; a.stat_partials = stats; }
        return a;
    };
    // ---------------- correctness (both kernels)
    int bad = 0;
    for (int which = 1; which < 2; ++which)
        for (auto s : check) {
            mc_gemm_args a = mk(s);
            HC(hipMemset(C, 0xff, (size_t)s.batch * s.M * s.N * 2));
            if (launch(which, &a)) { printf("launch error %s\n", g_err_msg); return 1; }
            HC(hipDeviceSynchronize());
            const long long tot = (long long)s.batch * s.M * s.N;
            const int full = tot <= (1 << 20);
            const int ns = full ? (int)tot : (1 << 18);
            hipLaunchKernelGGL(ref_check, dim3((ns + 255) / 256), dim3(256), 0, 0, A, B, C, a.bias, a.R, s.M, s.N, s.K, a.lda, a.ldb, a.ldc, s.batch, ns, full, err);
            std::vector<float> h(ns);
            HC(hipMemcpy(h.data(), err, (size_t)ns * 4, hipMemcpyDeviceToHost));
            float mx = 0; int nbad = 0;
            for (float e : h) { if (!(e <= 2e-2f)) ++nbad; if (e > mx || e != e) mx = e; }
            float sdev = 0.f;
            if (s.extra) {   // column statistics: sum over rows of C == sum of partial rows (checked loosely against the stored C)
                const int rows = (int)(s.batch * ((s.M + 255) / 256));
                std::vector<float> st((size_t)rows * 2 * s.N);
                HC(hipMemcpy(st.data(), stats, st.size() * 4, hipMemcpyDeviceToHost));
                std::vector<bf16_t> hc((size_t)tot);
                HC(hipMemcpy(hc.data(), C, (size_t)tot * 2, hipMemcpyDeviceToHost));
                for (int n = 0; n < s.N; n += std::max(1, s.N / 7)) {
                    double s1 = 0, s2 = 0, r1 = 0, r2 = 0;
                    for (int r = 0; r < rows; ++r) { s1 += st[((size_t)r * 2) * s.N + n]; s2 += st[((size_t)r * 2 + 1) * s.N + n]; }
                    for (long long m = 0; m < s.batch * s.M; ++m) { unsigned u = (unsigned)hc[m * s.N + n] << 16; float f; memcpy(&f, &u, 4); r1 += f; r2 += (double)f * f; }
                    sdev = std::max(sdev, (float)(fabs(s1 - r1) / (1 + fabs(r1)) + fabs(s2 - r2) / (1 + r2)));
                }
            }
            printf("check %s  b=%d M=%lld N=%d K=%lld extra=%d: max rel err %.3g, bad %d / %d, stats dev %.2g %s\n", which ? "g8p " : "g256", s.batch, s.M, s.N, s.K, s.extra,
                   mx, nbad, ns, sdev, (nbad || sdev > 1e-3f) ? "<<<<<< FAIL" : "ok");
            bad += nbad + (sdev > 1e-3f);
        }
    // ---------------- TN (weight gradient) kernel: correctness + timing
    {
        struct TShape { long long M, N, K; int splits; };
        std::vector<TShape> tcheck = {{256, 256, 64, 1}, {304, 1824, 1392, 1}, {176, 1056, 5415, 3}, {304, 1824, 44544, 16}, {768, 3072, 4096, 8}, {264, 40, 1000, 2}};
        float* ws = reinterpret_cast<float*>(R);            // ~1 GB of scratch
        for (auto s : tcheck) {
            mc_gemm_args a = {};
            a.A = A; a.B = B; a.C = ws; a.M = s.M; a.N = (int)s.N; a.K = s.K; a.batch = 1; a.nb2 = 1; a.splits = s.splits;
            a.lda = s.M; a.ldb = s.N; a.ldc = s.N; a.a_kmajor = 1; a.b_kmajor = 1; a.c_f32 = 1; a.splitk_ws = ws;
            if ((size_t)s.splits * s.M * s.N * 4 > maxel * 2) { printf("ws too small\n"); return 1; }
            HC(hipMemset(ws, 0xff, (size_t)s.splits * s.M * s.N * 4));
            if (mc_gemm256_tn_launch(&a, nullptr)) { printf("launch error %s\n", g_err_msg); return 1; }
            HC(hipDeviceSynchronize());
            const long long tot = s.M * s.N;
            const int full = tot <= (1 << 18);
            const int ns = full ? (int)tot : (1 << 16);
            hipLaunchKernelGGL(ref_check_tn, dim3((ns + 255) / 256), dim3(256), 0, 0, A, B, ws, s.splits, s.M, s.N, s.K, a.lda, a.ldb, ns, full, err);
            std::vector<float> h(ns);
            HC(hipMemcpy(h.data(), err, (size_t)ns * 4, hipMemcpyDeviceToHost));
            float mx = 0; int nbad = 0;
            for (float e : h) { if (!(e <= 2e-3f)) ++nbad; if (e > mx || e != e) mx = e; }
            printf("check g8tn M=%lld N=%lld K=%lld splits=%d: max rel err %.3g, bad %d / %d %s\n", s.M, s.N, s.K, s.splits, mx, nbad, ns, nbad ? "<<<<<< FAIL" : "ok");
            bad += nbad;
        }
        hipEvent_t t0, t1; HC(hipEventCreate(&t0)); HC(hipEventCreate(&t1));
        std::vector<TShape> tshapes = {{304, 1824, 44544, 0}, {1824, 304, 44544, 0}, {512, 3072, 44544, 0}, {3072, 512, 44544, 0}, {176, 1056, 173280, 0},
                                       {1056, 176, 173280, 0}, {2048, 512, 44544, 0}, {768, 768, 16384, 0}, {3072, 768, 16384, 0}, {768, 3072, 16384, 0}, {2304, 768, 16384, 0},
                                       {4096, 4096, 4096, 1}, {8192, 8192, 8192, 1}};
        for (auto s : tshapes) {
            const int sp = s.splits ? s.splits : mc_gemm256_tn_splits(s.M, s.N, s.K, 0);
            mc_gemm_args a = {};
            a.A = A; a.B = B; a.C = C; a.M = s.M; a.N = (int)s.N; a.K = s.K; a.batch = 1; a.nb2 = 1; a.splits = sp;
            a.lda = s.M; a.ldb = s.N; a.ldc = s.N; a.a_kmajor = 1; a.b_kmajor = 1; a.c_f32 = 1; a.splitk_ws = reinterpret_cast<float*>(R);
            if ((size_t)sp * s.M * s.N * 4 > maxel * 2) { printf("skip (ws)\n"); continue; }
            std::vector<float> ms_all;
            for (int r = 0; r < 5; ++r) {
                mc_gemm256_tn_launch(&a, nullptr);
                HC(hipEventRecord(t0));
                for (int i = 0; i < 5; ++i) mc_gemm256_tn_launch(&a, nullptr);
                HC(hipEventRecord(t1)); HC(hipEventSynchronize(t1));
                float ms; HC(hipEventElapsedTime(&ms, t0, t1)); ms_all.push_back(ms / 5);
            }
            std::sort(ms_all.begin(), ms_all.end());
            const double fl = 2.0 * s.M * s.N * s.K;
            printf("TN M=%5lld N=%5lld K=%7lld splits=%3d | %8.1f us %7.1f TF (kernel only, reduce not included)\n", s.M, s.N, s.K, sp, ms_all[2] * 1e3, fl / ms_all[2] / 1e9);
        }
    }
    // ---------------- timing, interleaved rounds
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    const int rounds = 5, it = 5;
    for (auto s : shapes) {
        mc_gemm_args a = mk(s);
        std::vector<float> best(2, 1e30f), med[2];
        for (int r = 0; r < rounds; ++r)
            for (int which = 0; which < 2; ++which) {
                launch(which, &a);
                HC(hipEventRecord(e0));
                for (int i = 0; i < it; ++i) launch(which, &a);
                HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
                float ms; HC(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
                med[which].push_back(ms); best[which] = std::min(best[which], ms);
            }
        double fl = 2.0 * s.batch * s.M * s.N * s.K, by = 2.0 * s.batch * (s.M * s.K + s.N * s.K + s.M * s.N);
        for (int w = 0; w < 2; ++w) std::sort(med[w].begin(), med[w].end());
        printf("b=%2d M=%7lld N=%5d K=%5lld | g256 %8.1f us %7.1f TF | g8p %8.1f us %7.1f TF (best %7.1f) %6.0f GB/s | x%.3f\n", s.batch, s.M, s.N, s.K,
               med[0][rounds / 2] * 1e3, fl / med[0][rounds / 2] / 1e9, med[1][rounds / 2] * 1e3, fl / med[1][rounds / 2] / 1e9, fl / best[1] / 1e9,
               by / med[1][rounds / 2] / 1e6, med[0][rounds / 2] / med[1][rounds / 2]);
    }
    return bad ? 2 : 0;
}
