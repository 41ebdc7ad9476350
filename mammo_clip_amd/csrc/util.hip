// error state, casts, transposes, stem im2col / weight prep, fp32 dropout, bf16 column sums.
#include "common_hip.h"
#include "../../include/mammoclip_hip.h"
#include <string.h>

static thread_local char g_err[512] = "";
extern "C" void mc_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* mc_last_error(void) { return g_err; }
extern "C" int mc_version(void) { return 100; }
#ifdef MC_F16
extern "C" int mc_storage_is_f16(void) { return 1; }
#else
extern "C" int mc_storage_is_f16(void) { return 0; }
#endif

namespace {

__global__ void cast_f32_bf16_k(const float* __restrict__ s, bf16_t* __restrict__ d, long long n) {
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const long long stride = (long long)gridDim.x * blockDim.x * 8;
    for (; i < n; i += stride) {
        if (i + 8 <= n && ((((uintptr_t)(s + i)) & 15u) == 0) && ((((uintptr_t)(d + i)) & 15u) == 0)) {
            float f[8];
            load8f(s + i, f);
            *reinterpret_cast<uint4*>(d + i) = pack8(f);
        } else {
            for (long long j = i; j < n && j < i + 8; ++j) d[j] = f2bf(s[j]);
        }
    }
}
// low part of the two-term bf16 split of an fp32 value: d = bf16(s - float(bf16(s)))  (s ~ hi + lo to 2^-17 relative)
__global__ void cast_f32_bf16_lo_k(const float* __restrict__ s, bf16_t* __restrict__ d, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) d[i] = f2bf(s[i] - bf2f(f2bf(s[i])));
}
__global__ void cast_bf16_f32_k(const bf16_t* __restrict__ s, float* __restrict__ d, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) d[i] = bf2f(s[i]);
}
__global__ void transpose_f32_k(const float* __restrict__ s, float* __restrict__ d, int rows, int cols) {
    __shared__ float tile[32][33];
    int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        int r = by + j, c = bx + threadIdx.x;
        if (r < rows && c < cols) tile[j][threadIdx.x] = s[(long long)r * cols + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        int c = bx + j, r = by + threadIdx.x;
        if (r < rows && c < cols) d[(long long)c * rows + r] = tile[threadIdx.x][j];
    }
}
__global__ void cast_transpose_k(const float* __restrict__ s, bf16_t* __restrict__ d, int rows, int cols) {
    __shared__ float tile[32][33];
    int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        int r = by + j, c = bx + threadIdx.x;
        if (r < rows && c < cols) tile[j][threadIdx.x] = s[(long long)r * cols + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        int c = bx + j, r = by + threadIdx.x;
        if (r < rows && c < cols) d[(long long)c * rows + r] = f2bf(tile[threadIdx.x][j]);
    }
}
__global__ void stem_weight_prep_k(const float* __restrict__ w, bf16_t* __restrict__ out, int c0) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c0 * 32) {
        int o = i / 32, k = i % 32;
        out[i] = (k < 27) ? f2bf(w[o * 27 + k]) : (bf16_t)0;
    }
}

// one thread per output pixel: gathers the 3x3x3 patch (k = cin*9 + kh*3 + kw) and writes 64 bytes.
// SRC = float: the batch is already normalised.  SRC = unsigned char: raw 8-bit pixels; the dataset's normalisation
// [ref: data/datasets/imagetext.py:131-135: x -= x.min(); x /= x.max(); (x - mean) / std, all in float32] is applied
// on the fly with the same operation order and roundings (mm = per-image {min}[n], {max}[n] from image_minmax_u8_k).
template <typename SRC>
__global__ void stem_im2col_k(const SRC* __restrict__ x, long long sn, long long sc, long long sh,
                              long long sw, int n, int h, int w, int pad_l, int pad_t, int oh, int ow,
                              bf16_t* __restrict__ out, const unsigned int* __restrict__ mm, float mean, float stdv) {
    long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)n * oh * ow;
    if (pix >= total) return;
    int ox = (int)(pix % ow);
    int oy = (int)((pix / ow) % oh);
    long long img = pix / ((long long)ow * oh);
    float lo = 0.f, range = 1.f;
    if constexpr (sizeof(SRC) == 1) {
        lo = (float)mm[img];
        range = (float)mm[n + img] - lo;               // = max of the shifted image
    }
    float v[32];
#pragma unroll
    for (int i = 27; i < 32; ++i) v[i] = 0.f;
    const SRC* base = x + img * sn;
    // channels-last memory (the trainer's permuted view: sc == 1, sw == 3): the 3 pixels x 3 channels of one kernel row
    // are 9 contiguous elements -- interior pixels fetch them as three 3-element loads per row instead of 27 scalars
    const int ix0 = ox * 2 - pad_l, iy0 = oy * 2 - pad_t;
    if (sc == 1 && sw == 3 && ix0 >= 0 && ix0 + 2 < w && iy0 >= 0 && iy0 + 2 < h) {
        struct __attribute__((packed, aligned(sizeof(SRC)))) T3 { SRC a, b, c; };
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const SRC* rp = base + (long long)(iy0 + kh) * sh + (long long)ix0 * 3;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const T3 t3 = *reinterpret_cast<const T3*>(rp + kw * 3);
                const SRC e[3] = {t3.a, t3.b, t3.c};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float val;
                    if constexpr (sizeof(SRC) == 1) {
                        float t = (float)e[c] - lo;
                        t = __fdiv_rn(t, range);
                        val = __fdiv_rn(__fsub_rn(t, mean), stdv);
                    } else {
                        val = e[c];
                    }
                    v[c * 9 + kh * 3 + kw] = val;
                }
            }
        }
    } else
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            int iy = oy * 2 + kh - pad_t;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                int ix = ox * 2 + kw - pad_l;
                float val = 0.f;
                if (iy >= 0 && iy < h && ix >= 0 && ix < w) {
                    if constexpr (sizeof(SRC) == 1) {
                        float t = (float)base[c * sc + iy * sh + ix * sw] - lo;
                        t = __fdiv_rn(t, range);
                        val = __fdiv_rn(__fsub_rn(t, mean), stdv);
                    } else {
                        val = base[c * sc + iy * sh + ix * sw];
                    }
                }
                v[c * 9 + kh * 3 + kw] = val;
            }
        }
    uint4* o = reinterpret_cast<uint4*>(out + pix * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = pack8(v + q * 8);
}

// per-image min / max of a dense 8-bit image block (integer atomics: exact and order-independent)
__global__ void image_minmax_u8_k(const unsigned char* __restrict__ x, long long sn, long long elems, int n,
                                  unsigned int* __restrict__ mm) {
    const int img = blockIdx.y;
    const unsigned char* base = x + img * sn;
    unsigned int lo = 255u, hi = 0u;
    const long long nv = elems / 16;
    const bool al = ((reinterpret_cast<uintptr_t>(base)) & 15) == 0;
    if (al) {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
            const uint4 q = reinterpret_cast<const uint4*>(base)[i];
            const unsigned int wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const unsigned int u = (wds[j] >> (8 * b)) & 0xffu;
                    lo = u < lo ? u : lo; hi = u > hi ? u : hi;
                }
        }
    }
    for (long long i = (al ? nv * 16 : 0) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < elems;
         i += (long long)gridDim.x * blockDim.x) {
        const unsigned int u = base[i];
        lo = u < lo ? u : lo; hi = u > hi ? u : hi;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&mm[img], lo); atomicMax(&mm[n + img], hi); }
}

// wg[img][n][k] = bf16( w[n][k] * gate[img][k] ): per-image gated copies of a (small) weight matrix, so that the
// project GEMM of the late stages runs as a plain batched GEMM (one image per batch entry) with direct-to-LDS staging.
// Same rounding as scaling the weight tile while it is staged (fp32 product, one round to bf16).
__global__ void gate_weights_k(const bf16_t* __restrict__ w, const float* __restrict__ gate, int n_img, int n, int k,
                               bf16_t* __restrict__ out) {
    const long long kv = k / 8, per = (long long)n * kv;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per * n_img) return;
    const long long img = i / per, r = i % per;
    const int kc = (int)(r % kv);
    float f[8], g[8];
    unpack8(*reinterpret_cast<const uint4*>(w + r * 8), f);
    load8f(gate + img * k + kc * 8, g);
#pragma unroll
    for (int q = 0; q < 8; ++q) f[q] *= g[q];
    *reinterpret_cast<uint4*>(out + i * 8) = pack8(f);
}

__global__ void dropout_f32_k(const float* __restrict__ x, float* __restrict__ y, long long n, float p,
                              unsigned long long seed, unsigned int sid) {
    long long i8 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long n8 = (n + 7) / 8;
    if (i8 >= n8) return;
    float s[8];
    dropout_scale8(seed, sid, (unsigned long long)i8, p, s);
    for (int j = 0; j < 8; ++j) {
        long long i = i8 * 8 + j;
        if (i < n) y[i] = x[i] * s[j];
    }
}

// column sums of bf16 [m, c] (row stride ld): thread = (row group, 8-channel vector)
__global__ __launch_bounds__(256) void colsum_partial_k(const bf16_t* __restrict__ x, long long m, int c,
                                                        long long ld, float* __restrict__ partials) {
    const int cv = c / 8;
    const int cvp = cv < 256 ? cv : 256;
    const int rpb = 256 / cvp;                 // row lanes per block iteration
    const int tid = threadIdx.x;
    const int rl = tid / cvp, cl = tid % cvp;
    __shared__ float red[256 * 8];
    for (int cbase = 0; cbase < cv; cbase += cvp) {
        int v = cbase + cl;
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        if (rl < rpb && v < cv) {
            const long long rstep = (long long)gridDim.x * rpb;
            long long r = (long long)blockIdx.x * rpb + rl;
            for (; r + 3 * rstep < m; r += 4 * rstep) {            // four independent 16-byte loads in flight per thread
                uint4 u[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = *reinterpret_cast<const uint4*>(x + (r + i * rstep) * ld + v * 8);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float f[8];
                    unpack8(u[i], f);
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[q] += f[q];
                }
            }
            for (; r < m; r += rstep) {
                float f[8];
                unpack8(*reinterpret_cast<const uint4*>(x + r * ld + v * 8), f);
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] += f[q];
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) red[tid * 8 + q] = acc[q];
        __syncthreads();
        if (rl == 0 && v < cv) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float s = 0.f;
                for (int r = 0; r < rpb; ++r) s += red[(r * cvp + cl) * 8 + q];
                partials[(long long)blockIdx.x * c + v * 8 + q] = s;
            }
        }
        __syncthreads();
    }
}
__global__ void colsum_final_k(const float* __restrict__ partials, int rows, int c, float* __restrict__ out,
                               int accumulate) {
    __shared__ double sh[16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + cl;
    double s = 0.0;
    if (i < c) {
        int r = rl;
        for (; r + 48 < rows; r += 64) {                      // four independent loads in flight, summed in row order
            const float a = partials[(long long)r * c + i], b = partials[(long long)(r + 16) * c + i];
            const float d = partials[(long long)(r + 32) * c + i], e = partials[(long long)(r + 48) * c + i];
            s += (double)a; s += (double)b; s += (double)d; s += (double)e;
        }
        for (; r < rows; r += 16) s += (double)partials[(long long)r * c + i];
    }
    sh[rl][cl] = s;
    __syncthreads();
    if (rl != 0 || i >= c) return;
    s = 0.0;
    for (int r = 0; r < 16; ++r) s += sh[r][cl];
    out[i] = accumulate ? out[i] + (float)s : (float)s;
}

}  // namespace

extern "C" int mc_cast_f32_bf16(const float* src, mc_bf16* dst, long long n, void* stream) {
    if (n <= 0) return MC_OK;
    MC_CHECK(src && dst, "cast: null pointer");
    int blocks = mc_div_up(n, 256 * 8);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cast_f32_bf16_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_cast_f32_bf16_lo(const float* src, mc_bf16* dst, long long n, void* stream) {
    if (n <= 0) return MC_OK;
    MC_CHECK(src && dst, "cast: null pointer");
    int blocks = mc_div_up(n, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cast_f32_bf16_lo_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_cast_bf16_f32(const mc_bf16* src, float* dst, long long n, void* stream) {
    if (n <= 0) return MC_OK;
    MC_CHECK(src && dst, "cast: null pointer");
    int blocks = mc_div_up(n, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cast_bf16_f32_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_transpose_f32(const float* src, float* dst, int rows, int cols, void* stream) {
    MC_CHECK(src && dst && rows > 0 && cols > 0, "transpose: bad args");
    dim3 grid(mc_div_up(cols, 32), mc_div_up(rows, 32));
    hipLaunchKernelGGL(transpose_f32_k, grid, dim3(32, 8), 0, (hipStream_t)stream, src, dst, rows, cols);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_cast_transpose_f32_bf16(const float* src, mc_bf16* dst, int rows, int cols, void* stream) {
    MC_CHECK(src && dst && rows > 0 && cols > 0, "cast_transpose: bad args");
    dim3 grid(mc_div_up(cols, 32), mc_div_up(rows, 32));
    hipLaunchKernelGGL(cast_transpose_k, grid, dim3(32, 8), 0, (hipStream_t)stream, src, dst, rows, cols);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_stem_weight_prep(const float* w, mc_bf16* out, int c0, void* stream) {
    MC_CHECK(w && out && c0 > 0, "stem_weight_prep: bad args");
    hipLaunchKernelGGL(stem_weight_prep_k, dim3(mc_div_up(c0 * 32, 256)), dim3(256), 0, (hipStream_t)stream, w, out, c0);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_stem_im2col(const float* x, long long sn, long long sc, long long sh, long long sw, int n,
                              int h, int w, int pad_l, int pad_t, int oh, int ow, mc_bf16* out, void* stream) {
    MC_CHECK(x && out && n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "stem_im2col: bad args");
    MC_CHECK(mc_aligned16(out), "stem_im2col: out must be 16-byte aligned");
    long long total = (long long)n * oh * ow;
    hipLaunchKernelGGL(stem_im2col_k<float>, dim3(mc_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, x, sn, sc,
                       sh, sw, n, h, w, pad_l, pad_t, oh, ow, out, (const unsigned int*)nullptr, 0.f, 1.f);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_image_minmax_u8(const unsigned char* x, long long sn, long long elems_per_image, int n,
                                  unsigned int* minmax, void* stream) {
    MC_CHECK(x && minmax && n > 0 && elems_per_image > 0 && sn >= elems_per_image, "image_minmax_u8: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(minmax, 0xff, (size_t)n * 4, st) != hipSuccess || hipMemsetAsync(minmax + n, 0, (size_t)n * 4, st) != hipSuccess) {
        mc_set_error("image_minmax_u8: memset failed");
        return MC_ERR_LAUNCH;
    }
    long long per = mc_div_up(elems_per_image, 256 * 16 * 8);
    int gx = (int)(per < 1 ? 1 : (per > 64 ? 64 : per));
    hipLaunchKernelGGL(image_minmax_u8_k, dim3(gx, n), dim3(256), 0, st, x, sn, elems_per_image, n, minmax);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_stem_im2col_u8(const unsigned char* x, long long sn, long long sc, long long sh, long long sw,
                                 const unsigned int* minmax, float mean, float std, int n, int h, int w, int pad_l,
                                 int pad_t, int oh, int ow, mc_bf16* out, void* stream) {
    MC_CHECK(x && minmax && out && n > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && std != 0.f, "stem_im2col_u8: bad args");
    MC_CHECK(mc_aligned16(out), "stem_im2col_u8: out must be 16-byte aligned");
    long long total = (long long)n * oh * ow;
    hipLaunchKernelGGL(stem_im2col_k<unsigned char>, dim3(mc_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, x, sn,
                       sc, sh, sw, n, h, w, pad_l, pad_t, oh, ow, out, minmax, mean, std);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_gate_weights_bf16(const mc_bf16* w, const float* gate, int n_img, int n, int k, mc_bf16* out,
                                    void* stream) {
    MC_CHECK(w && gate && out && n_img > 0 && n > 0 && k > 0 && k % 8 == 0, "gate_weights: bad args");
    MC_CHECK(mc_aligned16(w) && mc_aligned16(out) && mc_aligned16(gate), "gate_weights: operands must be 16-byte aligned");
    const long long total = (long long)n_img * n * (k / 8);
    hipLaunchKernelGGL(gate_weights_k, dim3(mc_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, w, gate, n_img, n, k, out);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_dropout_f32(const float* x, float* y, long long n, float p, unsigned long long seed,
                              unsigned int stream_id, void* stream) {
    if (n <= 0) return MC_OK;
    MC_CHECK(x && y && p >= 0.f && p < 1.f, "dropout: bad args");
    long long n8 = (n + 7) / 8;
    hipLaunchKernelGGL(dropout_f32_k, dim3(mc_div_up(n8, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, p,
                       seed, stream_id);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_colsum_rows(long long m, int c) {
    int cv = c / 8;
    int cvp = cv < 256 ? cv : 256;
    int rpb = 256 / (cvp > 0 ? cvp : 1);
    long long blocks = (m + (long long)rpb * 16 - 1) / ((long long)rpb * 16);
    if (blocks < 1) blocks = 1;
    if (blocks > 512) blocks = 512;            // two workgroups per CU; the finishing kernel reads `blocks` rows per column
    return (int)blocks;
}
extern "C" int mc_colsum_bf16(const mc_bf16* x, long long m, int c, long long ld, float* partials, float* out,
                              int accumulate, void* stream) {
    MC_CHECK(x && partials && out && m > 0 && c > 0, "colsum: bad args");
    MC_CHECK(c % 8 == 0 && ld % 8 == 0 && mc_aligned16(x), "colsum: c and ld must be multiples of 8, x aligned");
    int rows = mc_colsum_rows(m, c);
    hipLaunchKernelGGL(colsum_partial_k, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, m, c, ld, partials);
    MC_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_k, dim3(mc_div_up(c, 16)), dim3(256), 0, (hipStream_t)stream, partials, rows, c,
                       out, accumulate);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
