// Training-mode BatchNorm + SiLU family on NHWC bf16 tensors x[n_img*hw, c], and the squeeze-excite MLP.
// [ref: efficientnet_custom.py:64-88 (BatchNorm2d momentum 0.01 eps 1e-3), :104-131 (MBConvBlock.forward),
//       efficient_net_custom_utils.py:64-80 (SwishImplementation fwd/bwd), :129-154 (drop_connect)]
// All kernels are HBM-bound streaming passes: 16-byte (8-channel) vector accesses, per-channel parameters
// in registers, reductions leave as small partial buffers finished by a tiny finalize kernel (fp64 sums).
#include "common_hip.h"
#include "../../include/mammoclip_hip.h"

namespace {

// Thread layout of the row-streaming kernels: a 256-thread workgroup is rpb row-lanes x cvp channel-vector lanes (8
// channels = 16 bytes per lane) and makes one pass over its rows per channel chunk.  Chunk widths: whole rows when that
// keeps >= 70 % of the lanes busy; otherwise (squeeze pass) a power-of-two main chunk plus the remainder, each pass with
// as many row-lanes as fit (cv = 132, 1056 channels: 128 x 2 row-lanes + 4 x 64 instead of 132 lanes of 256); above 256
// vectors every kernel gives the remainder pass its own row-lanes (cv = 384: 256 x 1 + 128 x 2 instead of a half-empty
// second pass).
// The power-of-two split pays for the one-tensor squeeze pass only (1056 channels: 108 -> 89 us); the two-tensor
// passes measured 3-10 % slower with it and keep whole rows up to 256 vectors (split_small = false).
__host__ __device__ inline int rowmap_width(int cv, int cbase, bool split_small) {
    if (cbase > 0) return (cv - cbase) < 256 ? (cv - cbase) : 256;
    if (cv > 256) return 256;
    if (!split_small || cv * (256 / cv) * 10 >= 256 * 7) return cv;
    int a = 1;
    while (a * 2 <= cv) a *= 2;
    return a;
}

struct RowMap {
    int cv, cvp, rpb, rl, cl;
    bool split_small;
    __device__ RowMap(int c, bool split_small_ = false) {
        cv = c / 8;
        split_small = split_small_;
        set(0);
    }
    __device__ void set(int cbase) {          // layout of the pass that starts at channel vector cbase
        cvp = rowmap_width(cv, cbase, split_small);
        rpb = 256 / cvp;
        rl = threadIdx.x / cvp;
        cl = threadIdx.x % cvp;
    }
};

// Column sums of the [rpb][O = cvp*8] matrix the row-lanes of a workgroup left in red[] (red[thread*8 + q], thread =
// rl*cvp + cl).  Narrow tensors have many row-lanes (24 channels: 85) and the single-thread loop over them was 15-30 % of
// the kernel: here thread t < O*P sums the rows part, part+P, ... of column o = t % O (P = 256 / O) and the P partials are
// combined in part order -- deterministic.  Used when rpb >= 8 (then O <= 256).  Returns column t's sum for t < O.
__device__ inline float rowlane_colsum(const float* red, float* tmp, int cvp, int rpb) {
    const int O = cvp * 8, P = 256 / O, t = threadIdx.x;
    const int o = t % O, part = t / O;
    float s = 0.f;
    if (part < P)
        for (int r = part; r < rpb; r += P) s += red[r * O + o];
    tmp[t] = s;
    __syncthreads();
    float tot = 0.f;
    if (t < O)
        for (int k = 0; k < P; ++k) tot += tmp[k * O + t];
    return tot;
}

__global__ void bn_finalize_k(const float* __restrict__ partials, int rows, int c, double count,
                              const float* __restrict__ gamma, const float* __restrict__ beta,
                              float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
                              int update, float* __restrict__ mean, float* __restrict__ invstd,
                              float* __restrict__ scale, float* __restrict__ shift) {
    // 256 threads = 16 row-lanes x 16 channels: coalesced 64-byte reads, row-lanes reduced through LDS
    __shared__ double sh[2][16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + cl;
    double s = 0.0, s2 = 0.0;
    if (i < c) {
        int r = rl;
        for (; r + 7 * 16 < rows; r += 8 * 16) {           // 16 independent loads in flight per thread
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = partials[((long long)(r + u * 16) * 2) * c + i];
                b[u] = partials[((long long)(r + u * 16) * 2 + 1) * c + i];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += (double)a[u]; s2 += (double)b[u]; }
        }
        for (; r < rows; r += 16) {
            s += (double)partials[((long long)r * 2) * c + i];
            s2 += (double)partials[((long long)r * 2 + 1) * c + i];
        }
    }
    sh[0][rl][cl] = s; sh[1][rl][cl] = s2;
    __syncthreads();
    if (rl != 0 || i >= c) return;
    s = 0.0; s2 = 0.0;
    for (int r = 0; r < 16; ++r) { s += sh[0][r][cl]; s2 += sh[1][r][cl]; }
    double m = s / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[i] = (float)m;
    invstd[i] = is;
    float sc = gamma[i] * is;
    scale[i] = sc;
    shift[i] = beta[i] - (float)m * sc;
    if (update) {
        double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        rmean[i] = (1.f - momentum) * rmean[i] + momentum * (float)m;
        rvar[i] = (1.f - momentum) * rvar[i] + momentum * (float)unbiased;
    }
}

__global__ void bn_eval_coeffs_k(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                 int c, float* scale, float* shift) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    float sc = gamma[i] / sqrtf(rv[i] + eps);
    scale[i] = sc;
    shift[i] = beta[i] - rm[i] * sc;
}

__global__ __launch_bounds__(256) void bnact_apply_k(const mc_bnact_args p) {
    const int cvn = p.c / 8;
    const long long total = p.n_img * p.hw * cvn;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int cv = (int)(i % cvn);
        long long pix = i / cvn;
        float f[8], s[8], t[8];
        unpack8(nt_load16(p.x + pix * p.c + cv * 8), f);        // pre-BN tensor and skip input: next read in the backward pass
        load8f(p.scale + cv * 8, s);
        load8f(p.shift + cv * 8, t);
        float rs = p.rowscale ? p.rowscale[pix / p.hw] : 1.f;
        if (p.act == 1) bn_silu8(f, s, t);
        else {
#pragma unroll
            for (int q = 0; q < 8; ++q) f[q] = f[q] * s[q] + t[q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] *= rs;
        if (p.res) {
            float r[8];
            unpack8(nt_load16(p.res + pix * p.c + cv * 8), r);
#pragma unroll
            for (int q = 0; q < 8; ++q) f[q] += r[q];
        }
        *reinterpret_cast<uint4*>(p.out + pix * p.c + cv * 8) = pack8(f);
    }
}

// per-image reductions over hw: MODE 0 = pool (mean of act(z)), MODE 1 = SE dgate (sum g*act(z))
template <int MODE>
__global__ __launch_bounds__(256) void bnact_img_reduce_k(const mc_bnact_args p) {
    RowMap rm(p.c, MODE == 0);
    __shared__ float red[256 * 8];
    __shared__ float tmp[256];
    const long long img = blockIdx.x;
    const bf16_t* xb = p.x + img * p.hw * p.c;
    const bf16_t* gb = (MODE == 1) ? p.g + img * p.hw * p.c : nullptr;
    float* dst = (MODE == 0 ? p.pooled : p.dgate) + img * p.c;
    const float post = (MODE == 0) ? 1.0f / (float)p.hw : 1.0f;
    for (int cbase = 0; cbase < rm.cv; cbase += rm.cvp) {
        rm.set(cbase);
        int v = cbase + rm.cl;
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
        if (rm.rl < rm.rpb && v < rm.cv) {
            float s[8], t[8];
            load8f(p.scale + v * 8, s);
            load8f(p.shift + v * 8, t);
            bf16_t* ob = (MODE == 0 && p.out) ? p.out + img * p.hw * p.c + v * 8 : nullptr;   // optional: keep act(z)
            auto body = [&](long long r, const uint4& xv, const uint4& gv) {
                float f[8];
                unpack8(xv, f);
                if (MODE == 0) {
                    if (p.act == 1) bn_silu8(f, s, t);
                    else {
#pragma unroll
                        for (int q = 0; q < 8; ++q) f[q] = f[q] * s[q] + t[q];
                    }
                    if (ob) {                       // the pooled mean is taken over the stored (bf16-rounded) values
                        const uint4 o = pack8(f);
                        *reinterpret_cast<uint4*>(ob + r * p.c) = o;
                        unpack8(o, f);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[q] += f[q];
                } else {
                    float g[8];
                    unpack8(gv, g);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float z = f[q] * s[q] + t[q];
                        acc[q] += g[q] * (p.act == 1 ? silu_f(z) : z);
                    }
                }
            };
            const long long rstride = (long long)gridDim.y * rm.rpb;
            const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
            long long r = (long long)blockIdx.y * rm.rpb + rm.rl;
            for (; r + 3 * rstride < p.hw; r += 4 * rstride) {      // four independent rows in flight
                uint4 xv[4], gv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xv[u] = *reinterpret_cast<const uint4*>(xb + (r + u * rstride) * p.c + v * 8);
                    gv[u] = (MODE == 1) ? *reinterpret_cast<const uint4*>(gb + (r + u * rstride) * p.c + v * 8) : z4;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) body(r + u * rstride, xv[u], gv[u]);
            }
            for (; r < p.hw; r += rstride)
                body(r, *reinterpret_cast<const uint4*>(xb + r * p.c + v * 8),
                     (MODE == 1) ? *reinterpret_cast<const uint4*>(gb + r * p.c + v * 8) : z4);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) red[threadIdx.x * 8 + q] = acc[q];
        __syncthreads();
        if (rm.rpb >= 8) {
            const float s = rowlane_colsum(red, tmp, rm.cvp, rm.rpb);
            const int vo = cbase + (int)threadIdx.x / 8, q = threadIdx.x & 7;
            if ((int)threadIdx.x < rm.cvp * 8 && vo < rm.cv) {
                if (gridDim.y == 1) dst[vo * 8 + q] = s * post;
                else p.split_ws[((long long)blockIdx.y * p.n_img + img) * p.c + vo * 8 + q] = s * post;
            }
        } else if (rm.rl == 0 && v < rm.cv) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float s = 0.f;
                for (int r = 0; r < rm.rpb; ++r) s += red[(r * rm.cvp + rm.cl) * 8 + q];
                if (gridDim.y == 1) dst[v * 8 + q] = s * post;
                else p.split_ws[((long long)blockIdx.y * p.n_img + img) * p.c + v * 8 + q] = s * post;   // summed in fixed order
            }
        }
        __syncthreads();
    }
}

// Squeeze-excite backward sums in ONE pass over (x = depthwise output, g = d loss / d (gated activation)):
//   sums[0] = sum_hw g*y          (d loss / d gate)                       y  = silu(z), z = x*scale + shift
//   sums[1] = sum_hw g*y'         sums[2] = sum_hw g*y'*xhat              y' = silu'(z), xhat = (x - mean)*invstd
//   sums[3] = sum_hw y'           sums[4] = sum_hw y'*xhat
// With these, the BatchNorm-backward reductions of the block follow WITHOUT another pass over the big tensors:
//   dz = (g*gate + dpooled/HW) * y'  =>  sum dz = sum_img gate*sums[1] + dpooled/HW*sums[3]   (same with xhat: [2],[4])
// (dpooled only exists after the SE MLP backward, which itself needs sums[0] -- hence the decomposition).
__global__ __launch_bounds__(256) void bnact_se_sums_k(const mc_bnact_args p) {
    RowMap rm(p.c);
    __shared__ float red[256 * 8];
    __shared__ float tmp[256];
    const long long img = blockIdx.x;
    const bf16_t* xb = p.x + img * p.hw * p.c;
    const bf16_t* gb = p.g + img * p.hw * p.c;
    const long long plane = p.n_img * p.c;
    for (int cbase = 0; cbase < rm.cv; cbase += rm.cvp) {
        rm.set(cbase);
        int v = cbase + rm.cl;
        float acc[5][8];
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[k][q] = 0.f;
        if (rm.rl < rm.rpb && v < rm.cv) {
            float s[8], t[8], mu[8], is[8];
            load8f(p.scale + v * 8, s);
            load8f(p.shift + v * 8, t);
            load8f(p.mean + v * 8, mu);
            load8f(p.invstd + v * 8, is);
            auto body = [&](const uint4& xv, const uint4& gv) {
                float x[8], g[8];
                unpack8(xv, x);
                unpack8(gv, g);
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    const f32x2_t one = {1.f, 1.f};
                    const f32x2_t xv = {x[q], x[q + 1]}, gv = {g[q], g[q + 1]};
                    const f32x2_t z = __builtin_elementwise_fma(xv, f32x2_t{s[q], s[q + 1]}, f32x2_t{t[q], t[q + 1]});
                    const f32x2_t sg = sigmoid2_f(z);
                    const f32x2_t y = z * sg, yd = sg * __builtin_elementwise_fma(z, one - sg, one);
                    const f32x2_t xh = (xv - f32x2_t{mu[q], mu[q + 1]}) * f32x2_t{is[q], is[q + 1]};
                    const f32x2_t gyd = gv * yd;
                    f32x2_t a;
                    a = __builtin_elementwise_fma(gv, y, f32x2_t{acc[0][q], acc[0][q + 1]}); acc[0][q] = a.x; acc[0][q + 1] = a.y;
                    a = f32x2_t{acc[1][q], acc[1][q + 1]} + gyd; acc[1][q] = a.x; acc[1][q + 1] = a.y;
                    a = __builtin_elementwise_fma(gyd, xh, f32x2_t{acc[2][q], acc[2][q + 1]}); acc[2][q] = a.x; acc[2][q + 1] = a.y;
                    a = f32x2_t{acc[3][q], acc[3][q + 1]} + yd; acc[3][q] = a.x; acc[3][q + 1] = a.y;
                    a = __builtin_elementwise_fma(yd, xh, f32x2_t{acc[4][q], acc[4][q + 1]}); acc[4][q] = a.x; acc[4][q + 1] = a.y;
                }
            };
            const long long rstride = (long long)gridDim.y * rm.rpb;
            long long r = (long long)blockIdx.y * rm.rpb + rm.rl;
            if (r + rstride < p.hw) {
                // two rows per step, the NEXT step's four loads issued before this step's arithmetic (the body is ~900
                // VALU instructions per pair of rows: without the prefetch a wave has nothing in flight while it computes)
                uint4 x0 = *reinterpret_cast<const uint4*>(xb + r * p.c + v * 8);
                uint4 x1 = *reinterpret_cast<const uint4*>(xb + (r + rstride) * p.c + v * 8);
                uint4 g0 = *reinterpret_cast<const uint4*>(gb + r * p.c + v * 8);
                uint4 g1 = *reinterpret_cast<const uint4*>(gb + (r + rstride) * p.c + v * 8);
                r += 2 * rstride;
                for (; r + rstride < p.hw; r += 2 * rstride) {
                    const uint4 nx0 = *reinterpret_cast<const uint4*>(xb + r * p.c + v * 8);
                    const uint4 nx1 = *reinterpret_cast<const uint4*>(xb + (r + rstride) * p.c + v * 8);
                    const uint4 ng0 = *reinterpret_cast<const uint4*>(gb + r * p.c + v * 8);
                    const uint4 ng1 = *reinterpret_cast<const uint4*>(gb + (r + rstride) * p.c + v * 8);
                    __builtin_amdgcn_sched_barrier(0);          // keep the four loads in front of the arithmetic
                    body(x0, g0);
                    body(x1, g1);
                    x0 = nx0; x1 = nx1; g0 = ng0; g1 = ng1;
                }
                body(x0, g0);
                body(x1, g1);
            }
            if (r < p.hw)
                body(*reinterpret_cast<const uint4*>(xb + r * p.c + v * 8), *reinterpret_cast<const uint4*>(gb + r * p.c + v * 8));
        }
        for (int k = 0; k < 5; ++k) {
#pragma unroll
            for (int q = 0; q < 8; ++q) red[threadIdx.x * 8 + q] = acc[k][q];
            __syncthreads();
            if (rm.rpb >= 8) {
                const float sm = rowlane_colsum(red, tmp, rm.cvp, rm.rpb);
                const int vo = cbase + (int)threadIdx.x / 8, q = threadIdx.x & 7;
                if ((int)threadIdx.x < rm.cvp * 8 && vo < rm.cv) {
                    if (gridDim.y == 1) p.dgate[k * plane + img * p.c + vo * 8 + q] = sm;
                    else p.split_ws[((long long)blockIdx.y * 5 + k) * plane + img * p.c + vo * 8 + q] = sm;
                }
            } else if (rm.rl == 0 && v < rm.cv) {
                float* dst = p.dgate + k * plane + img * p.c;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float sm = 0.f;
                    for (int r = 0; r < rm.rpb; ++r) sm += red[(r * rm.cvp + rm.cl) * 8 + q];
                    if (gridDim.y == 1) dst[v * 8 + q] = sm;
                    else p.split_ws[((long long)blockIdx.y * 5 + k) * plane + img * p.c + v * 8 + q] = sm;
                }
            }
            __syncthreads();
        }
    }
}

// BN-backward partials [n_img][2][c] from the per-image sums above, the SE gate and d loss / d pooled
__global__ void bn_partials_from_se_sums_k(const float* __restrict__ sums, const float* __restrict__ gate,
                                           const float* __restrict__ dpooled, float add_scale, long long n_img, int c,
                                           float* __restrict__ partials) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long plane = n_img * c;
    if (i >= plane) return;
    long long img = i / c;
    int ch = (int)(i % c);
    float gt = gate[i], dp = dpooled[i] * add_scale;
    partials[(img * 2 + 0) * c + ch] = gt * sums[1 * plane + i] + dp * sums[3 * plane + i];
    partials[(img * 2 + 1) * c + ch] = gt * sums[2 * plane + i] + dp * sums[4 * plane + i];
}

// Backward through y = act(x*scale + shift) * rowscale with upstream gradient g*mul[img,c] + add[img,c]*add_scale:
//   REDUCE pass:  per-channel sums (sum dz, sum dz*xhat)          -> partials
//   APPLY  pass:  dx = A*dz + B*x + C (coefficients from mc_bn_bwd_finalize) -> bf16
// Grid = (row blocks, image): a workgroup stays inside ONE image, so the per-(image, channel) factors are loaded
// once; a thread owns one 8-channel vector and walks down the rows, so the per-channel parameters live in registers
// and the loop body is two 16-byte loads, the activation derivative and (APPLY) one 16-byte store, two rows in
// flight per iteration.
// ACT: SiLU in front of the BatchNorm output (p.act == 1); MA: per-(image, channel) factor / addend on the upstream gradient
// (p.mul / p.add).  Compile-time so that the plain form (BatchNorm2 of the projection: no activation, no factors) carries
// 24 instead of 56 parameter registers and keeps four rows in flight in the reduce pass too.
template <bool APPLY, bool ACT, bool MA>
__global__ __launch_bounds__(256) void bnact_bwd_k(const mc_bnact_args p) {
    RowMap rm(p.c);
    __shared__ float red[APPLY ? 1 : 256 * 16];
    const long long img = blockIdx.y;
    const bf16_t* xb = p.x + img * p.hw * p.c;
    const bf16_t* gb = p.g ? p.g + img * p.hw * p.c : nullptr;
    bf16_t* dxb = p.dx ? p.dx + img * p.hw * p.c : nullptr;      // reduce pass: optional store of dz (folded BatchNorm backward)
    const float rs = p.rowscale ? p.rowscale[img] : 1.f;
    const float asc = (p.add_scale == 0.f) ? 1.f : p.add_scale;
    for (int cbase = 0; cbase < rm.cv; cbase += rm.cvp) {
        rm.set(cbase);
        const long long rstride = (long long)gridDim.x * rm.rpb;
        const int v = cbase + rm.cl;
        const bool active = rm.rl < rm.rpb && v < rm.cv;
        float a0[8], a1[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { a0[q] = 0.f; a1[q] = 0.f; }
        if (active) {
            float s[8], t[8], k0[8], k1[8], k2[8], mulv[8], addv[8];
            if (ACT) { load8f(p.scale + v * 8, s); load8f(p.shift + v * 8, t); }
            if (APPLY) { load8f(p.coef + v * 8, k0); load8f(p.coef + p.c + v * 8, k1); load8f(p.coef + 2 * p.c + v * 8, k2); }
            else { load8f(p.mean + v * 8, k0); load8f(p.invstd + v * 8, k1); }
#pragma unroll
            for (int q = 0; q < 8; ++q) { mulv[q] = rs; addv[q] = 0.f; }
            if (MA && p.mul) {
                float m[8];
                load8f(p.mul + img * p.c + v * 8, m);
#pragma unroll
                for (int q = 0; q < 8; ++q) mulv[q] = m[q] * rs;
            }
            if (MA && p.add) {
                float m[8];
                load8f(p.add + img * p.c + v * 8, m);
#pragma unroll
                for (int q = 0; q < 8; ++q) addv[q] = m[q] * asc * rs;
            }
            auto body = [&](long long r, const uint4& xv, const uint4& gv) {
                float x[8], g[8], dz[8];
                unpack8(xv, x);
                unpack8(gv, g);
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    f32x2_t d = MA ? __builtin_elementwise_fma(f32x2_t{g[q], g[q + 1]}, f32x2_t{mulv[q], mulv[q + 1]}, f32x2_t{addv[q], addv[q + 1]})
                                   : f32x2_t{g[q], g[q + 1]} * f32x2_t{rs, rs};
                    if (ACT) d = d * silu_grad2_f(__builtin_elementwise_fma(f32x2_t{x[q], x[q + 1]}, f32x2_t{s[q], s[q + 1]}, f32x2_t{t[q], t[q + 1]}));
                    dz[q] = d.x; dz[q + 1] = d.y;
                }
                if (APPLY) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) dz[q] = k0[q] * dz[q] + k1[q] * x[q] + k2[q];
                    nt_store16(dxb + r * p.c + v * 8, pack8(dz));
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { a0[q] += dz[q]; a1[q] += dz[q] * (x[q] - k0[q]) * k1[q]; }
                    if (dxb) nt_store16(dxb + r * p.c + v * 8, pack8(dz));
                }
            };
            const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
            long long r = (long long)blockIdx.x * rm.rpb + rm.rl;
            if constexpr (APPLY || !(ACT && MA))                     // (the full reduce variant has no registers to spare for this)
            for (; r + 3 * rstride < p.hw; r += 4 * rstride) {      // four independent rows (8 loads) in flight
                uint4 xv[4], gv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xv[u] = nt_load16(xb + (r + u * rstride) * p.c + v * 8);      // x and g are dead after this pass
                    gv[u] = gb ? nt_load16(gb + (r + u * rstride) * p.c + v * 8) : z4;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) body(r + u * rstride, xv[u], gv[u]);
            }
            for (; r + rstride < p.hw; r += 2 * rstride) {
                const long long r2 = r + rstride;
                uint4 x0 = *reinterpret_cast<const uint4*>(xb + r * p.c + v * 8);
                uint4 x1 = *reinterpret_cast<const uint4*>(xb + r2 * p.c + v * 8);
                uint4 g0 = gb ? *reinterpret_cast<const uint4*>(gb + r * p.c + v * 8) : z4;
                uint4 g1 = gb ? *reinterpret_cast<const uint4*>(gb + r2 * p.c + v * 8) : z4;
                body(r, x0, g0);
                body(r2, x1, g1);
            }
            if (r < p.hw) {
                uint4 x0 = *reinterpret_cast<const uint4*>(xb + r * p.c + v * 8);
                uint4 g0 = gb ? *reinterpret_cast<const uint4*>(gb + r * p.c + v * 8) : z4;
                body(r, x0, g0);
            }
        }
        if (!APPLY && rm.rpb >= 8) {
            const long long prow = img * gridDim.x + blockIdx.x;
            float* const tmp = red + 256 * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int q = 0; q < 8; ++q) red[threadIdx.x * 8 + q] = h == 0 ? a0[q] : a1[q];
                __syncthreads();
                const float sm = rowlane_colsum(red, tmp, rm.cvp, rm.rpb);
                const int vo = cbase + (int)threadIdx.x / 8;
                if ((int)threadIdx.x < rm.cvp * 8 && vo < rm.cv) p.partials[(prow * 2 + h) * p.c + vo * 8 + (threadIdx.x & 7)] = sm;
                __syncthreads();
            }
        } else if (!APPLY) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { red[threadIdx.x * 16 + q] = a0[q]; red[threadIdx.x * 16 + 8 + q] = a1[q]; }
            __syncthreads();
            if (rm.rl == 0 && v < rm.cv) {
                const long long prow = img * gridDim.x + blockIdx.x;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float sm = 0.f;
                    for (int r = 0; r < rm.rpb; ++r) sm += red[(r * rm.cvp + rm.cl) * 16 + q];
                    p.partials[(prow * 2 + (q >> 3)) * p.c + v * 8 + (q & 7)] = sm;
                }
            }
            __syncthreads();
        }
    }
}

__global__ void bn_bwd_finalize_k(const float* __restrict__ partials, int rows, int c, double count,
                                  const float* __restrict__ gamma, const float* __restrict__ mean,
                                  const float* __restrict__ invstd, float* __restrict__ dgamma,
                                  float* __restrict__ dbeta, float* __restrict__ coef) {
    __shared__ double sh[2][16][17];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + cl;
    double s0 = 0.0, s1 = 0.0;
    if (i < c) {
        int r = rl;
        for (; r + 7 * 16 < rows; r += 8 * 16) {           // 16 independent loads in flight per thread
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = partials[((long long)(r + u * 16) * 2) * c + i];
                b[u] = partials[((long long)(r + u * 16) * 2 + 1) * c + i];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s0 += (double)a[u]; s1 += (double)b[u]; }
        }
        for (; r < rows; r += 16) {
            s0 += (double)partials[((long long)r * 2) * c + i];
            s1 += (double)partials[((long long)r * 2 + 1) * c + i];
        }
    }
    sh[0][rl][cl] = s0; sh[1][rl][cl] = s1;
    __syncthreads();
    if (rl != 0 || i >= c) return;
    s0 = 0.0; s1 = 0.0;
    for (int r = 0; r < 16; ++r) { s0 += sh[0][r][cl]; s1 += sh[1][r][cl]; }
    dbeta[i] = (float)s0;
    dgamma[i] = (float)s1;
    double gi = (double)gamma[i] * (double)invstd[i];
    coef[i] = (float)gi;
    coef[c + i] = (float)(-gi * (double)invstd[i] * s1 / count);
    coef[2 * c + i] = (float)(gi * ((double)invstd[i] * s1 * (double)mean[i] - s0) / count);
}

// ---------------------------------------------------------------- squeeze-excite MLP
// The MLP is tiny (c x c/24 weights) but sits on the critical path twice per block: spread it over many
// workgroups (one wave per hidden unit / one thread per channel) instead of one workgroup per image.
__global__ __launch_bounds__(256) void se_hidden_k(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                   const float* __restrict__ b1, int c, int cs,
                                                   float* __restrict__ u_out, float* __restrict__ r_out) {
    const int img = blockIdx.y, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= cs) return;
    const float* pv = pooled + (long long)img * c;
    float s = 0.f;
    for (int i = lane; i < c; i += 64) s += w1[(long long)j * c + i] * pv[i];
    s = wave_sum(s);
    if (lane == 0) {
        float u = s + b1[j];
        if (u_out) u_out[(long long)img * cs + j] = u;
        r_out[(long long)img * cs + j] = silu_f(u);
    }
}
__global__ __launch_bounds__(256) void se_gate_k(const float* __restrict__ r, const float* __restrict__ w2,
                                                 const float* __restrict__ b2, int c, int cs, float* __restrict__ gate) {
    extern __shared__ float sh[];
    const int img = blockIdx.y;
    for (int j = threadIdx.x; j < cs; j += 256) sh[j] = r[(long long)img * cs + j];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= c) return;
    float s = b2[i];
    for (int j = 0; j < cs; ++j) s += w2[(long long)i * cs + j] * sh[j];
    gate[(long long)img * c + i] = sigmoid_f(s);
}

// ws layout per image: ds[c] | r[cs] | du[cs] ; u is staged in the du slot between the two hidden kernels
// (round 4: the elementwise ds = dgate * g * (1 - g) pass is folded in here -- every wave needs the whole ds row of its image
// for its dot product anyway; the wave of hidden unit 0 stores it for the weight-gradient kernel)
__global__ __launch_bounds__(256) void se_bwd_hidden_k(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, const float* __restrict__ w2,
                                                       const float* __restrict__ gate, const float* __restrict__ dgate, int c,
                                                       int cs, float* __restrict__ ws) {
    const int img = blockIdx.y, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= cs) return;
    const float* pv = pooled + (long long)img * c;
    const float* gv = gate + (long long)img * c;
    const float* dgv = dgate + (long long)img * c;
    float* wsi = ws + (long long)img * (c + 2 * cs);
    float s = 0.f, d = 0.f;
#pragma unroll 4
    for (int i = lane; i < c; i += 64) {
        const float g = gv[i];
        const float ds = dgv[i] * g * (1.f - g);
        if (j == 0) wsi[i] = ds;
        s += w1[(long long)j * c + i] * pv[i];
        d += w2[(long long)i * cs + j] * ds;
    }
    s = wave_sum(s);
    d = wave_sum(d);
    if (lane == 0) {
        float u = s + b1[j];
        wsi[c + j] = silu_f(u);
        wsi[c + cs + j] = d * silu_grad_f(u);
    }
}
__device__ __forceinline__ void se_bwd_dpool_body(const float* __restrict__ w1, int c, int cs, const float* __restrict__ ws,
                                                  float* __restrict__ dpooled, int bx, int img, float* sh) {
    const float* wsi = ws + (long long)img * (c + 2 * cs);
    for (int j = threadIdx.x; j < cs; j += 256) sh[j] = wsi[c + cs + j];
    __syncthreads();
    const int i = bx * 256 + threadIdx.x;
    if (i >= c) return;
    float s = 0.f;
    for (int j = 0; j < cs; ++j) s += w1[(long long)j * c + i] * sh[j];
    dpooled[(long long)img * c + i] = s;
}

// weight / bias gradients, reduced over the n images (no atomics)
// (round 4: one launch with the d pooled kernel -- both only read what se_bwd_hidden_k left in ws: the first dpool_blocks
// workgroups are (channel block, image) pairs of the d pooled matvec, the others the weight-gradient elements)
__global__ __launch_bounds__(256) void se_bwd_tail_k(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                     const float* __restrict__ ws, int n, int c, int cs, int dpool_blocks,
                                                     float* __restrict__ dpooled, float* __restrict__ dw1,
                                                     float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2) {
    extern __shared__ float sh[];
    if ((int)blockIdx.x < dpool_blocks) {
        const int cb = (c + 255) / 256;
        se_bwd_dpool_body(w1, c, cs, ws, dpooled, blockIdx.x % cb, blockIdx.x / cb, sh);
        return;
    }
    const long long stride = c + 2 * cs;
    long long e = (long long)(blockIdx.x - dpool_blocks) * blockDim.x + threadIdx.x;
    const long long n1 = (long long)c * cs;
    if (e < n1) {                       // dw2[ci, j] = sum_n ds[n,ci] * r[n,j]
        int ci = (int)(e / cs), j = (int)(e % cs);
        float s = 0.f;
#pragma unroll 8
        for (int i = 0; i < n; ++i) s += ws[i * stride + ci] * ws[i * stride + c + j];
        dw2[e] = s;
    } else if (e < 2 * n1) {            // dw1[j, ci] = sum_n du[n,j] * pooled[n,ci]
        long long e2 = e - n1;
        int j = (int)(e2 / c), ci = (int)(e2 % c);
        float s = 0.f;
#pragma unroll 8
        for (int i = 0; i < n; ++i) s += ws[i * stride + c + cs + j] * pooled[(long long)i * c + ci];
        dw1[e2] = s;
    } else if (e < 2 * n1 + c) {
        int ci = (int)(e - 2 * n1);
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += ws[i * stride + ci];
        db2[ci] = s;
    } else if (e < 2 * n1 + c + cs) {
        int j = (int)(e - 2 * n1 - c);
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += ws[i * stride + c + cs + j];
        db1[j] = s;
    }
}

int check_bnact(const mc_bnact_args& p) {
    MC_CHECK(p.x && p.scale && p.shift, "bnact: null x/scale/shift");
    MC_CHECK(p.n_img > 0 && p.hw > 0 && p.c > 0 && p.c % 8 == 0, "bnact: bad shape (c % 8 == 0 required)");
    MC_CHECK(mc_aligned16(p.x), "bnact: x must be 16-byte aligned");
    return MC_OK;
}
int stream_blocks(long long total_vec) {
    long long b = (total_vec + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int mc_bn_finalize(const float* partials, int rows, int c, double count, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                              int update_running, float* mean, float* invstd, float* scale, float* shift, void* stream) {
    MC_CHECK(partials && gamma && beta && mean && invstd && scale && shift && rows > 0 && c > 0 && count > 0, "bn_finalize: bad args");
    MC_CHECK(!update_running || (running_mean && running_var), "bn_finalize: running buffers missing");
    hipLaunchKernelGGL(bn_finalize_k, dim3(mc_div_up(c, 16)), dim3(256), 0, (hipStream_t)stream, partials, rows, c, count,
                       gamma, beta, running_mean, running_var, momentum, eps, update_running, mean, invstd, scale, shift);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float eps, int c, float* scale, float* shift, void* stream) {
    MC_CHECK(gamma && beta && running_mean && running_var && scale && shift && c > 0, "bn_eval_coeffs: bad args");
    hipLaunchKernelGGL(bn_eval_coeffs_k, dim3(mc_div_up(c, 128)), dim3(128), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, c, scale, shift);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
static int bwd_grid_x(const mc_bnact_args& p) {
    int cv = p.c / 8;
    int cvp = rowmap_width(cv, 0, false);
    int rpb = 256 / (cvp > 0 ? cvp : 1);
    long long per = (p.hw + (long long)rpb * 32 - 1) / ((long long)rpb * 32);      // >= ~32 rows per thread
    long long cap = 4096 / (p.n_img > 0 ? p.n_img : 1);
    if (cap < 1) cap = 1;
    long long gx = per < cap ? per : cap;
    return (int)(gx < 1 ? 1 : gx);
}
extern "C" int mc_bnact_rows(const mc_bnact_args* a) { return (int)(bwd_grid_x(*a) * a->n_img); }
extern "C" int mc_bnact_apply(const mc_bnact_args* a, void* stream) {
    const mc_bnact_args& p = *a;
    if (int e = check_bnact(p)) return e;
    MC_CHECK(p.out, "bnact_apply: null out");
    hipLaunchKernelGGL(bnact_apply_k, dim3(stream_blocks(p.n_img * p.hw * (p.c / 8))), dim3(256), 0, (hipStream_t)stream, p);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
// out[i] = sum over splits of ws[split][i], in split order (deterministic, unlike float atomics)
// 64 elements x 4 split-lanes per workgroup (the old one-thread-per-element loop ran 64 dependent-latency loads per
// thread on 150 workgroups: 1 TB/s); the four partials of an element are combined in lane order -- deterministic
__global__ __launch_bounds__(256) void split_sum_k(const float* __restrict__ ws, float* __restrict__ out, long long n, int splits) {
    __shared__ float sh[4][64];
    const int el = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + el;
    float s = 0.f;
    if (i < n) {
        int k = pl;
        for (; k + 12 < splits; k += 16) {                    // four independent loads in flight
            const float a = ws[(long long)k * n + i], b = ws[(long long)(k + 4) * n + i];
            const float c = ws[(long long)(k + 8) * n + i], d = ws[(long long)(k + 12) * n + i];
            s += a; s += b; s += c; s += d;
        }
        for (; k < splits; k += 4) s += ws[(long long)k * n + i];
    }
    sh[pl][el] = s;
    __syncthreads();
    if (pl == 0 && i < n) out[i] = ((sh[0][el] + sh[1][el]) + sh[2][el]) + sh[3][el];
}
// row splits per image of the per-image reductions: enough workgroups to fill the chip (target / n_img), but not so many
// that a workgroup's stream gets short.  Measured on the B5 shapes: the one-tensor pool pass is best around 1024
// workgroups (c = 768: 72 -> 59 us), the two-tensor passes around 2048.  mc_bnact_img_splits() (workspace sizing)
// reports the larger count.
static int img_splits(const mc_bnact_args& p, long long target = 2048) {
    int cv = p.c / 8;
    int cvp = rowmap_width(cv, 0, false);
    int rpb = 256 / cvp;
    long long per = (p.hw + (long long)rpb * 64 - 1) / ((long long)rpb * 64);
    long long want = target / (p.n_img > 0 ? p.n_img : 1);
    if (want < 1) want = 1;
    long long s = per < want ? per : want;
    if (s < 1) s = 1;
    return (int)s;
}
extern "C" int mc_bnact_img_splits(const mc_bnact_args* a) { return img_splits(*a); }
extern "C" int mc_bnact_pool(const mc_bnact_args* a, void* stream) {
    const mc_bnact_args& p = *a;
    if (int e = check_bnact(p)) return e;
    MC_CHECK(p.pooled, "bnact_pool: null pooled");
    int sp = img_splits(p, 1024);
    MC_CHECK(sp == 1 || p.split_ws, "bnact_pool: split_ws needed (mc_bnact_img_splits() > 1)");
    hipLaunchKernelGGL((bnact_img_reduce_k<0>), dim3((unsigned)p.n_img, sp), dim3(256), 0, (hipStream_t)stream, p);
    MC_LAUNCH_CHECK();
    if (sp > 1) {
        long long n = p.n_img * p.c;
        hipLaunchKernelGGL(split_sum_k, dim3(mc_div_up(n, 64)), dim3(256), 0, (hipStream_t)stream, p.split_ws, p.pooled, n, sp);
        MC_LAUNCH_CHECK();
    }
    return MC_OK;
}
extern "C" int mc_bnact_se_dgate(const mc_bnact_args* a, void* stream) {
    const mc_bnact_args& p = *a;
    if (int e = check_bnact(p)) return e;
    MC_CHECK(p.g && p.dgate, "bnact_se_dgate: null g/dgate");
    int sp = img_splits(p);
    MC_CHECK(sp == 1 || p.split_ws, "bnact_se_dgate: split_ws needed (mc_bnact_img_splits() > 1)");
    hipLaunchKernelGGL((bnact_img_reduce_k<1>), dim3((unsigned)p.n_img, sp), dim3(256), 0, (hipStream_t)stream, p);
    MC_LAUNCH_CHECK();
    if (sp > 1) {
        long long n = p.n_img * p.c;
        hipLaunchKernelGGL(split_sum_k, dim3(mc_div_up(n, 64)), dim3(256), 0, (hipStream_t)stream, p.split_ws, p.dgate, n, sp);
        MC_LAUNCH_CHECK();
    }
    return MC_OK;
}
extern "C" int mc_bnact_se_sums(const mc_bnact_args* a, void* stream) {
    const mc_bnact_args& p = *a;
    if (int e = check_bnact(p)) return e;
    MC_CHECK(p.g && p.dgate && p.mean && p.invstd, "bnact_se_sums: null g / sums / mean / invstd");
    int sp = img_splits(p);
    MC_CHECK(sp == 1 || p.split_ws, "bnact_se_sums: split_ws needed (mc_bnact_img_splits() > 1)");
    hipLaunchKernelGGL(bnact_se_sums_k, dim3((unsigned)p.n_img, sp), dim3(256), 0, (hipStream_t)stream, p);
    MC_LAUNCH_CHECK();
    if (sp > 1) {
        long long n = 5 * p.n_img * p.c;
        hipLaunchKernelGGL(split_sum_k, dim3(mc_div_up(n, 64)), dim3(256), 0, (hipStream_t)stream, p.split_ws, p.dgate, n, sp);
        MC_LAUNCH_CHECK();
    }
    return MC_OK;
}
extern "C" int mc_bn_partials_from_se_sums(const float* sums, const float* gate, const float* dpooled, float add_scale,
                                           long long n_img, int c, float* partials, void* stream) {
    MC_CHECK(sums && gate && dpooled && partials && n_img > 0 && c > 0, "bn_partials_from_se_sums: bad args");
    hipLaunchKernelGGL(bn_partials_from_se_sums_k, dim3(mc_div_up(n_img * c, 256)), dim3(256), 0, (hipStream_t)stream, sums,
                       gate, dpooled, add_scale, n_img, c, partials);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
template <bool APPLY>
static void launch_bwd(const mc_bnact_args& p, void* stream) {
    const dim3 grid(bwd_grid_x(p), (unsigned)p.n_img), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bool act = p.act == 1, ma = p.mul || p.add;
    if (act && ma) hipLaunchKernelGGL((bnact_bwd_k<APPLY, true, true>), grid, block, 0, st, p);
    else if (act) hipLaunchKernelGGL((bnact_bwd_k<APPLY, true, false>), grid, block, 0, st, p);
    else if (ma) hipLaunchKernelGGL((bnact_bwd_k<APPLY, false, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((bnact_bwd_k<APPLY, false, false>), grid, block, 0, st, p);
}
extern "C" int mc_bnact_bwd_reduce(const mc_bnact_args* a, void* stream) {
    const mc_bnact_args& p = *a;
    if (int e = check_bnact(p)) return e;
    MC_CHECK(p.partials && p.mean && p.invstd, "bnact_bwd_reduce: null partials/mean/invstd");
    launch_bwd<false>(p, stream);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_bnact_bwd_apply(const mc_bnact_args* a, void* stream) {
    const mc_bnact_args& p = *a;
    if (int e = check_bnact(p)) return e;
    MC_CHECK(p.coef && p.dx, "bnact_bwd_apply: null coef/dx");
    launch_bwd<true>(p, stream);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_bn_bwd_finalize(const float* partials, int rows, int c, double count, const float* gamma,
                                  const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef,
                                  void* stream) {
    MC_CHECK(partials && gamma && mean && invstd && dgamma && dbeta && coef && rows > 0 && c > 0, "bn_bwd_finalize: bad args");
    hipLaunchKernelGGL(bn_bwd_finalize_k, dim3(mc_div_up(c, 16)), dim3(256), 0, (hipStream_t)stream, partials, rows, c,
                       count, gamma, mean, invstd, dgamma, dbeta, coef);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_se_fwd(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, int n,
                         int c, int cs, float* gate, float* ws, void* stream) {
    MC_CHECK(pooled && w1 && b1 && w2 && b2 && gate && ws && n > 0 && c > 0 && cs > 0, "se_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(se_hidden_k, dim3(mc_div_up(cs, 4), n), dim3(256), 0, st, pooled, w1, b1, c, cs, (float*)nullptr, ws);
    MC_LAUNCH_CHECK();
    hipLaunchKernelGGL(se_gate_k, dim3(mc_div_up(c, 256), n), dim3(256), cs * sizeof(float), st, ws, w2, b2, c, cs, gate);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_se_bwd(const float* pooled, const float* gate, const float* dgate, const float* w1, const float* b1,
                         const float* w2, const float* b2, int n, int c, int cs, float* dpooled, float* dw1, float* db1,
                         float* dw2, float* db2, float* ws, void* stream) {
    MC_CHECK(pooled && gate && dgate && w1 && b1 && w2 && dpooled && dw1 && db1 && dw2 && db2 && ws, "se_bwd: null arg");
    MC_CHECK(n > 0 && c > 0 && cs > 0, "se_bwd: bad shape");
    (void)b2;
    hipStream_t st = (hipStream_t)stream;
    // two launches (round 3: four): [ds + hidden-layer backward] -> [d pooled | weight gradients]
    hipLaunchKernelGGL(se_bwd_hidden_k, dim3(mc_div_up(cs, 4), n), dim3(256), 0, st, pooled, w1, b1, w2, gate, dgate, c, cs, ws);
    MC_LAUNCH_CHECK();
    const long long total = 2LL * c * cs + c + cs;
    const int dpool_blocks = mc_div_up(c, 256) * n;
    hipLaunchKernelGGL(se_bwd_tail_k, dim3(dpool_blocks + mc_div_up(total, 256)), dim3(256), cs * sizeof(float), st, pooled, w1, ws, n, c,
                       cs, dpool_blocks, dpooled, dw1, db1, dw2, db2);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
