// Standalone micro-benchmark / phase profiler for the depthwise kernels (developer tool, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMARCH_PROF] scripts/dwbench.hip -o /tmp/dwbench && /tmp/dwbench
#include "../mammo_clip_amd/csrc/conv.hip"
#include "../mammo_clip_amd/csrc/conv_lane.hip"
#include <cstdio>
#include <vector>
static char g_err_msg[256];
extern "C" void mc_set_error(const char* m) { snprintf(g_err_msg, sizeof g_err_msg, "%s", m); }
extern "C" const char* mc_last_error(void) { return g_err_msg; }

#define HC(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    struct Shape { int c, k, s, h, w; };
    std::vector<Shape> shapes = {{24, 3, 1, 760, 456}, {48, 3, 1, 760, 456}, {144, 3, 2, 760, 456}, {240, 3, 1, 380, 228},
                                 {240, 5, 2, 380, 228}, {384, 5, 1, 190, 114}, {384, 3, 2, 190, 114}, {768, 3, 1, 95, 57},
                                 {768, 5, 1, 95, 57}, {1056, 5, 1, 95, 57}, {1056, 5, 2, 95, 57}, {1824, 5, 1, 48, 29},
                                 {1824, 3, 1, 48, 29}, {3072, 3, 1, 48, 29}};
    const int n = 32;
    size_t maxel = (size_t)n * 760 * 456 * 144;
    bf16_t *x, *y; float *w, *sc, *sh, *part;
    HC(hipMalloc(&x, maxel * 2)); HC(hipMalloc(&y, maxel * 2));
    HC(hipMemset(x, 0x3c, maxel * 2)); HC(hipMemset(y, 0, maxel * 2));
    HC(hipMalloc(&w, 25 * 4096 * 4)); HC(hipMemset(w, 0, 25 * 4096 * 4));
    HC(hipMalloc(&sc, 4096 * 4)); HC(hipMalloc(&sh, 4096 * 4)); HC(hipMemset(sc, 0, 4096 * 4)); HC(hipMemset(sh, 0, 4096 * 4));
    HC(hipMalloc(&part, 2048 * 2 * 4096 * 4));
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    for (auto s : shapes) {
        mc_dwconv_args a = {};
        a.x = x; a.out = y; a.w_kkc = w; a.n = n; a.h = s.h; a.w = s.w; a.c = s.c; a.k = s.k; a.stride = s.s;
        a.oh = (s.h + s.s - 1) / s.s; a.ow = (s.w + s.s - 1) / s.s;
        a.pad_l = a.pad_t = (s.k - 1) / 2;
        double bytes = 2.0 * s.c * n * ((double)s.h * s.w + (double)a.oh * a.ow);
        for (int pro = 0; pro < 2; ++pro) {
            a.pro_scale = pro ? sc : nullptr; a.pro_shift = pro ? sh : nullptr; a.stat_partials = pro ? part : nullptr;
            if (mc_dwconv_fwd(&a, nullptr)) { printf("err %s\n", mc_last_error()); return 1; }
            HC(hipDeviceSynchronize());
#ifdef MARCH_PROF
            unsigned long long z[8] = {0}; HC(hipMemcpyToSymbol(HIP_SYMBOL(g_march_prof), z, sizeof(z)));
#endif
            HC(hipEventRecord(e0));
            const int it = 5;
            for (int i = 0; i < it; ++i) mc_dwconv_fwd(&a, nullptr);
            HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
            float ms; HC(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
            printf("fwd k%ds%d c=%4d %3dx%3d pro=%d  %7.3f ms %7.1f GB/s", s.k, s.s, s.c, s.h, s.w, pro, ms, bytes / ms / 1e6);
#ifdef MARCH_PROF
            HC(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_march_prof), sizeof(z)));
            double tot = (double)(z[0] + z[1] + z[2] + z[3] + z[4]);
            printf("  | bar1 %4.1f%% store %4.1f%% bar2 %4.1f%% issue %4.1f%% compute %4.1f%%  blocks %llu cyc/blk %.0f", 100 * z[0] / tot,
                   100 * z[1] / tot, 100 * z[2] / tot, 100 * z[3] / tot, 100 * z[4] / tot, z[5], tot / (z[5] ? z[5] : 1));
#endif
            printf("\n");
        }
        {   // weight gradient (x with prologue, dy = y buffer)
            a.pro_scale = sc; a.pro_shift = sh; a.stat_partials = nullptr; a.dy = y; a.out = w;
            if (mc_dwconv_bwd_weight(&a, nullptr)) { printf("err %s\n", mc_last_error()); return 1; }
            HC(hipDeviceSynchronize());
            HC(hipEventRecord(e0));
            const int it = 5;
            for (int i = 0; i < it; ++i) mc_dwconv_bwd_weight(&a, nullptr);
            HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
            float ms; HC(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
            printf("bww k%ds%d c=%4d %3dx%3d pro=1  %7.3f ms %7.1f GB/s\n", s.k, s.s, s.c, s.h, s.w, ms, bytes / ms / 1e6);
            a.out = y;
        }
    }
    return 0;
}
