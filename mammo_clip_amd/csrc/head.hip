// Projection heads, L2 normalisation and the InfoNCE cross-entropy, all fp32 (the reference keeps
// embeddings / logits / loss in fp32 even under autocast).
// [ref: model/modules/projection.py:23-29, model/clip.py:86-91, loss/breast_clip.py:46-100]
// These are tiny (b x 512 x W*b); a simple LDS-tiled VALU kernel with arbitrary strides serves every layout.
#include "common_hip.h"
#include "../../include/mammoclip_hip.h"

namespace {

constexpr int TS = 32;

__global__ __launch_bounds__(256) void sgemm_k(const float* __restrict__ a, long long ars, long long acs,
                                               const float* __restrict__ b, long long brs, long long bcs,
                                               float* __restrict__ c, long long ldc, int m, int n, int k, float alpha,
                                               float beta, const float* __restrict__ bias,
                                               const float* __restrict__ alpha_dev, float* __restrict__ ws) {
    if (alpha_dev) alpha *= alpha_dev[0];
    __shared__ float sa[TS][TS + 1], sb[TS][TS + 1];
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;       // 16 x 16 threads, 2 x 2 outputs each
    const int m0 = blockIdx.y * TS, n0 = blockIdx.x * TS;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    // split-K (gridDim.z > 1): this workgroup reduces k in [kb, ke) and leaves its partial in ws[z][m][n]
    const int kper = ((k + (int)gridDim.z - 1) / (int)gridDim.z + TS - 1) / TS * TS;
    const int kb = blockIdx.z * kper, ke = kb + kper < k ? kb + kper : k;
    for (int k0 = kb; k0 < ke; k0 += TS) {
        for (int e = threadIdx.x; e < TS * TS; e += 256) {
            int r = e / TS, cc = e % TS;
            // choose the faster-varying index by stride to keep loads coalesced where possible
            int mi = m0 + r, ki = k0 + cc;
            sa[r][cc] = (mi < m && ki < ke) ? a[mi * ars + ki * acs] : 0.f;
            int kj = k0 + r, nj = n0 + cc;
            sb[r][cc] = (kj < ke && nj < n) ? b[kj * brs + nj * bcs] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < TS; ++kk) {
            float a0 = sa[ty][kk], a1 = sa[ty + 16][kk];
            float b0 = sb[kk][tx], b1 = sb[kk][tx + 16];
            acc[0][0] = fmaf(a0, b0, acc[0][0]); acc[0][1] = fmaf(a0, b1, acc[0][1]);
            acc[1][0] = fmaf(a1, b0, acc[1][0]); acc[1][1] = fmaf(a1, b1, acc[1][1]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int mi = m0 + ty + i * 16, nj = n0 + tx + j * 16;
            if (mi < m && nj < n) {
                if (gridDim.z > 1) { ws[((long long)blockIdx.z * m + mi) * n + nj] = acc[i][j]; continue; }
                float v = alpha * acc[i][j];
                if (bias) v += bias[nj];
                if (beta != 0.f) v += beta * c[mi * ldc + nj];
                c[mi * ldc + nj] = v;
            }
        }
}
// combine the split-K partials in split order (deterministic) and apply the epilogue
__global__ __launch_bounds__(256) void sgemm_splitk_finish_k(const float* __restrict__ ws, int splits, float* __restrict__ c,
                                                             long long ldc, int m, int n, float alpha, float beta,
                                                             const float* __restrict__ bias, const float* __restrict__ alpha_dev) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)m * n) return;
    if (alpha_dev) alpha *= alpha_dev[0];
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += ws[(long long)z * m * n + i];
    const int mi = (int)(i / n), nj = (int)(i % n);
    float v = alpha * s;
    if (bias) v += bias[nj];
    if (beta != 0.f) v += beta * c[mi * ldc + nj];
    c[mi * ldc + nj] = v;
}

__global__ void l2norm_fwd_k(const float* __restrict__ x, int rows, int d, float* __restrict__ y, float* __restrict__ norm) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int i = lane; i < d; i += 64) { float v = x[(long long)row * d + i]; s += v * v; }
    s = sqrtf(wave_sum(s));
    if (lane == 0) norm[row] = s;
    for (int i = lane; i < d; i += 64) y[(long long)row * d + i] = x[(long long)row * d + i] / s;
}
__global__ void l2norm_bwd_k(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ norm,
                             int rows, int d, float* __restrict__ dx) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int i = lane; i < d; i += 64) s += dy[(long long)row * d + i] * y[(long long)row * d + i];
    s = wave_sum(s);
    float inv = 1.f / norm[row];
    for (int i = lane; i < d; i += 64)
        dx[(long long)row * d + i] = (dy[(long long)row * d + i] - y[(long long)row * d + i] * s) * inv;
}

// one wave per row: lse, per-row loss contribution (into row_loss), dlogits in place.
// labels: optional int64 class index per row (the `labels` the loss was called with, loss/breast_clip.py:43-44);
// the target of row r is labels[r] + label_offset (null: r + label_offset).
__global__ void ce_fwd_bwd_k(float* __restrict__ logits, int rows, int n, const long long* __restrict__ labels,
                             int label_offset, float w, float eps, float* __restrict__ row_loss) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* lr = logits + (long long)row * n;
    float mx = -3.4e38f;
    for (int i = lane; i < n; i += 64) mx = fmaxf(mx, lr[i]);
    mx = wave_max(mx);
    float s = 0.f, sx = 0.f;
    for (int i = lane; i < n; i += 64) { s += __expf(lr[i] - mx); sx += lr[i]; }
    s = wave_sum(s);
    sx = wave_sum(sx);
    const float lse = mx + __logf(s);
    const int label = (labels ? (int)labels[row] : row) + label_offset;
    const float scale = w / (float)rows;
    // caller-supplied labels are validated here (the host can only range-check the arange form): an out-of-range label
    // makes the row's loss -- and with it the step's loss -- NaN instead of reading beside the row
    const bool ok = label >= 0 && label < n;
    // label smoothing eps: target = (1-eps) * onehot + eps / n   (torch F.cross_entropy semantics)
    if (lane == 0)
        row_loss[row] = ok ? ((1.f - eps) * (lse - lr[label]) + eps * (lse - sx / (float)n)) * scale : __uint_as_float(0x7fc00000u);
    const float inv = 1.f / s;
    const float un = eps / (float)n;
    for (int i = lane; i < n; i += 64) {
        float pr = __expf(lr[i] - mx) * inv;
        lr[i] = (pr - (i == label ? 1.f - eps : 0.f) - un) * scale;
    }
}
// loss_out[0] += sum_r row_loss[r], in a fixed order (bit-reproducible: no float atomics)
__global__ void ce_sum_rows_k(const float* __restrict__ row_loss, int rows, float* __restrict__ loss_out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < rows; i += 256) s += row_loss[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[0] += red[0];
}

}  // namespace

static int sgemm_splits(int m, int n, int k) {
    const long long tiles = (long long)mc_div_up(n, TS) * mc_div_up(m, TS);
    long long s = 512 / tiles;                      // aim at ~2 workgroups per CU
    const long long kmax = k / (4 * TS);            // at least 4 k-tiles per split
    if (s > kmax) s = kmax;
    if (s > 32) s = 32;
    return s < 2 ? 1 : (int)s;
}
extern "C" long long mc_sgemm_ws_floats(int m, int n, int k) {
    const int s = sgemm_splits(m, n, k);
    return s > 1 ? (long long)s * m * n : 0;
}
extern "C" int mc_sgemm(const float* a, long long ars, long long acs, const float* b, long long brs, long long bcs,
                        float* c, long long ldc, int m, int n, int k, float alpha, float beta, const float* bias,
                        const float* alpha_dev, float* ws, void* stream) {
    MC_CHECK(a && b && c && m > 0 && n > 0 && k > 0, "sgemm: bad args");
    const int splits = ws ? sgemm_splits(m, n, k) : 1;
    dim3 grid(mc_div_up(n, TS), mc_div_up(m, TS), splits);
    hipLaunchKernelGGL(sgemm_k, grid, dim3(256), 0, (hipStream_t)stream, a, ars, acs, b, brs, bcs, c, ldc, m, n, k, alpha,
                       beta, bias, alpha_dev, ws);
    MC_LAUNCH_CHECK();
    if (splits > 1) {
        hipLaunchKernelGGL(sgemm_splitk_finish_k, dim3(mc_div_up((long long)m * n, 256)), dim3(256), 0, (hipStream_t)stream, ws,
                           splits, c, ldc, m, n, alpha, beta, bias, alpha_dev);
        MC_LAUNCH_CHECK();
    }
    return MC_OK;
}
__global__ void scale_f32_k(const float* __restrict__ x, const float* __restrict__ sd, float alpha, float* __restrict__ y,
                            long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float s = alpha * (sd ? sd[0] : 1.f);
    if (i < n) y[i] = x[i] * s;
}
extern "C" int mc_scale_f32(const float* x, const float* scalar_dev, float alpha, float* y, long long n, void* stream) {
    MC_CHECK(x && y && n > 0, "scale: bad args");
    hipLaunchKernelGGL(scale_f32_k, dim3(mc_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, x, scalar_dev, alpha, y, n);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_l2norm_fwd(const float* x, int rows, int d, float* y, float* norm, void* stream) {
    MC_CHECK(x && y && norm && rows > 0 && d > 0, "l2norm_fwd: bad args");
    hipLaunchKernelGGL(l2norm_fwd_k, dim3(mc_div_up(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, d, y, norm);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_l2norm_bwd(const float* dy, const float* y, const float* norm, int rows, int d, float* dx, void* stream) {
    MC_CHECK(dy && y && norm && dx && rows > 0 && d > 0, "l2norm_bwd: bad args");
    hipLaunchKernelGGL(l2norm_bwd_k, dim3(mc_div_up(rows, 4)), dim3(256), 0, (hipStream_t)stream, dy, y, norm, rows, d, dx);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_ce_fwd_bwd(float* logits, int rows, int n, const long long* labels, int label_offset, float w,
                             float smoothing, float* loss_out, float* row_ws, void* stream) {
    MC_CHECK(logits && loss_out && row_ws && rows > 0 && n > 0, "ce: bad args");
    MC_CHECK(label_offset >= 0 && (labels || label_offset + rows <= n), "ce: labels out of range");
    hipLaunchKernelGGL(ce_fwd_bwd_k, dim3(mc_div_up(rows, 4)), dim3(256), 0, (hipStream_t)stream, logits, rows, n, labels,
                       label_offset, w, smoothing, row_ws);
    MC_LAUNCH_CHECK();
    hipLaunchKernelGGL(ce_sum_rows_k, dim3(1), dim3(256), 0, (hipStream_t)stream, row_ws, rows, loss_out);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
