// Training-mode BatchNorm backward folded into the 1x1 convolution in front of it (MBConv expand conv + _bn0):
//     e = x We^T,  z = bn0(e),  dz = dL/dz (already through the activation),
//     de = A (dz - m) + B (e - mu),   A = gamma*invstd,  B = -gamma*invstd^2 * mean(dz*xhat),  m = mean(dz)
// is LINEAR in dz and e, and e - mu = (x - xbar) We^T, so the two consumers of de need neither de nor e:
//     dx  = de We      = dz (A.We) + x G + cvec,         G = We^T diag(B) We,  cvec = -(A.m) We - xbar G
//     dWe = de^T x     = A.(dz^T x - m (x) colsum(x)) + (B.We) Sxx,      Sxx = x^T x - colsum(x) (x) colsum(x) / rows
// The three small-matrix kernels below build the folded operands; the big passes are plain GEMMs over dz and x
// (the expanded tensors e / de are not touched again: three passes over the expanded tensor per block are gone).
// [ref: efficientnet_custom.py:104-107 (_expand_conv, _bn0, swish), torch BatchNorm2d training-mode backward]
#include "common_hip.h"
#include "../../include/mammoclip_hip.h"

namespace {

// w1t[j][i] = bf16(We[i][j] * A[i]);  wb[i][j] = bf16(We[i][j] * B[i]);  sxx[i][j] = bf16(xtx[i][j] - cs[i] cs[j] / rows)
// f16 storage build (MC_F16): B ~ 1e-8 and Sxx ~ rows * var ~ 1e6 lie outside f16's exponent range, so the two operands
// carry the factor `rows` the other way round -- wb = f16(We * B * rows), sxx = f16(Sxx / rows): their product (B.We) Sxx
// is unchanged, and G' = We^T wb = rows * G is undone by alpha = 1 / rows of the x G' GEMM (ops.bn_fold_expand_bwd) and
// by fold_cvec_k below.
#ifdef MC_F16
#define MC_FOLD_ROWS(rows) (rows)
#else
#define MC_FOLD_ROWS(rows) 1.0
#endif
__global__ void fold_prepare_k(const float* __restrict__ we, const float* __restrict__ coef, const float* __restrict__ xtx,
                               const float* __restrict__ cs, double rows, int n, int k, bf16_t* __restrict__ w1t,
                               bf16_t* __restrict__ wb, bf16_t* __restrict__ sxx) {
    const long long nk = (long long)n * k, total = nk + (long long)k * k;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        if (idx < nk) {
            const int i = (int)(idx / k), j = (int)(idx % k);
            const float w = we[idx];
            w1t[(long long)j * n + i] = f2bf(w * coef[i]);
            wb[idx] = f2bf((float)((double)(w * coef[n + i]) * MC_FOLD_ROWS(rows)));
        } else {
            const long long r = idx - nk;
            const int i = (int)(r / k), j = (int)(r % k);
            sxx[r] = f2bf((float)(((double)xtx[r] - (double)cs[i] * (double)cs[j] / rows) / MC_FOLD_ROWS(rows)));
        }
    }
}

// one workgroup per output column j: gtb[j][:] = bf16(gt[j][:]),
// cvec[j] = -sum_i m[i] bf16(A[i] We[i][j]) - sum_i xbar[i] gtb[j][i]      (m = dbeta / rows, xbar = cs / rows)
__global__ __launch_bounds__(256) void fold_cvec_k(const float* __restrict__ gt, const float* __restrict__ we,
                                                   const float* __restrict__ coef, const float* __restrict__ dbeta,
                                                   const float* __restrict__ cs, double rows, int n, int k,
                                                   bf16_t* __restrict__ gtb, float* __restrict__ cvec) {
    const int j = blockIdx.x;
    double acc = 0.0;
    // with the ROUNDED operand the dz GEMM uses (w1t = bf16(We * A)): the column sums of dx then cancel exactly, like the
    // column sums of de do -- a constant offset of 2^-9 |m (A.We)| per row would otherwise survive into every bias /
    // BatchNorm-shift gradient upstream, which are sums over all pixels
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)dbeta[i] * (double)bf2f(f2bf(we[(long long)i * k + j] * coef[i]));
    for (int i = threadIdx.x; i < k; i += 256) {
        const bf16_t g = f2bf(gt[(long long)j * k + i]);
        gtb[(long long)j * k + i] = g;
        acc += (double)cs[i] * (double)bf2f(g) / MC_FOLD_ROWS(rows);      // (f16 build: gt holds rows * G)
    }
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) cvec[j] = (float)(-red[0] / rows);
}

// dwe[i][j] = A[i] * (t1[i][j] - m[i] cs[j]) + wx[i][j]
__global__ void fold_wgrad_k(const float* __restrict__ t1, const float* __restrict__ wx, const float* __restrict__ coef,
                             const float* __restrict__ dbeta, const float* __restrict__ cs, double rows, int n, int k,
                             float* __restrict__ dwe) {
    const long long total = (long long)n * k;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int i = (int)(idx / k), j = (int)(idx % k);
        const float m = (float)((double)dbeta[i] / rows);
        dwe[idx] = coef[i] * (t1[idx] - m * cs[j]) + wx[idx];
    }
}

// Round 6 (mc_mbconv_xdw_fwd runs the expand conv inside the depthwise kernel: e never exists).  BatchNorm0 batch statistics of
// e = x W^T from the Gram matrix of x: one workgroup per expanded channel c,
//     sum_e[c] = w_c . cs,      sumsq_e[c] = w_c^T S w_c + sum_e[c]^2 / rows,      S = x^T x - cs cs^T / rows   (centred, fp64)
// written as TWO partials rows (value rounded to fp32 + the rounding remainder): mc_bn_finalize adds its rows in fp64, so the
// variance does not lose the digits a single fp32 sum of squares would drop when mean^2 >> var.
__global__ __launch_bounds__(256) void gram_partials_k(const bf16_t* __restrict__ w, int ldw, const float* __restrict__ xtx,
                                                       const float* __restrict__ cs, double rows, int n, int k,
                                                       float* __restrict__ part) {
    const int c = blockIdx.x;
    const bf16_t* wc = w + (long long)c * ldw;
    double s = 0.0, q = 0.0;
    for (int j = threadIdx.x; j < k; j += 256) s += (double)bf2f(wc[j]) * (double)cs[j];
    for (int idx = threadIdx.x; idx < k * k; idx += 256) {
        const int i = idx / k, j = idx - i * k;
        const double sij = (double)xtx[idx] - (double)cs[i] * (double)cs[j] / rows;
        q += (double)bf2f(wc[i]) * sij * (double)bf2f(wc[j]);
    }
    __shared__ double red[2][256];
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) { red[0][threadIdx.x] += red[0][threadIdx.x + st]; red[1][threadIdx.x] += red[1][threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double sum = red[0][0];
        double sq = red[1][0];
        if (sq < 0.0) sq = 0.0;
        sq += sum * sum / rows;
        const float s_hi = (float)sum, q_hi = (float)sq;
        part[c] = s_hi; part[n + c] = q_hi;                                    // row 0: (sum, sumsq)
        part[2 * n + c] = (float)(sum - (double)s_hi); part[3 * n + c] = (float)(sq - (double)q_hi);   // row 1: the remainders
    }
}

}  // namespace

extern "C" int mc_bn_gram_partials(const mc_bf16* w, int ldw, const float* xtx, const float* colsum_x, double rows, int n, int k,
                                   float* partials, void* stream) {
    MC_CHECK(w && xtx && colsum_x && partials && n > 0 && k > 0 && ldw >= k && rows > 0, "bn_gram_partials: bad args");
    hipLaunchKernelGGL(gram_partials_k, dim3(n), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, ldw, xtx, colsum_x, rows, n, k,
                       partials);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

extern "C" int mc_bn_fold_prepare(const float* we, const float* coef, const float* xtx, const float* colsum_x, double rows,
                                  int n, int k, mc_bf16* w1t, mc_bf16* wb, mc_bf16* sxx, void* stream) {
    MC_CHECK(we && coef && xtx && colsum_x && w1t && wb && sxx && n > 0 && k > 0 && rows > 0, "bn_fold_prepare: bad args");
    const long long total = (long long)n * k + (long long)k * k;
    hipLaunchKernelGGL(fold_prepare_k, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0,
                       (hipStream_t)stream, we, coef, xtx, colsum_x, rows, n, k, (bf16_t*)w1t, (bf16_t*)wb, (bf16_t*)sxx);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

extern "C" int mc_bn_fold_cvec(const float* gt, const float* we, const float* coef, const float* dbeta,
                               const float* colsum_x, double rows, int n, int k, mc_bf16* gtb, float* cvec, void* stream) {
    MC_CHECK(gt && we && coef && dbeta && colsum_x && gtb && cvec && n > 0 && k > 0 && rows > 0, "bn_fold_cvec: bad args");
    hipLaunchKernelGGL(fold_cvec_k, dim3(k), dim3(256), 0, (hipStream_t)stream, gt, we, coef, dbeta, colsum_x, rows, n, k,
                       (bf16_t*)gtb, cvec);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

extern "C" int mc_bn_fold_wgrad(const float* t1, const float* wx, const float* coef, const float* dbeta,
                                const float* colsum_x, double rows, int n, int k, float* dwe, void* stream) {
    MC_CHECK(t1 && wx && coef && dbeta && colsum_x && dwe && n > 0 && k > 0 && rows > 0, "bn_fold_wgrad: bad args");
    const long long total = (long long)n * k;
    hipLaunchKernelGGL(fold_wgrad_k, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0,
                       (hipStream_t)stream, t1, wx, coef, dbeta, colsum_x, rows, n, k, dwe);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
