// BERT (BioClinicalBERT = BERT-base shape) non-GEMM pieces for gfx950: embeddings + LayerNorm, residual +
// dropout + LayerNorm, masked softmax with dropout, erf-GELU, eos pooling -- forward and backward.
// [ref: model/modules/text_encoder.py:47-49 -> transformers BertModel (BertEmbeddings, BertSelfAttention,
//       BertSelfOutput, BertIntermediate, BertOutput); model/clip.py:65-68 (eos pooling)]
// One 64-lane wave owns one row (hidden <= 1024, keys <= 512); row reductions are wavefront shuffles;
// dropout masks are Philox functions of (seed, stream id, element index) and are regenerated in backward.
#include "common_hip.h"
#include "../../include/mammoclip_hip.h"

namespace {

constexpr int MAXV = 2;    // 8-element vectors per lane: hidden <= 64*8*2 = 1024

__device__ __forceinline__ void ln_stats(const float (*x)[8], int nvec, int lane, int h, float eps, float* mean,
                                         float* rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + i * 64 < nvec)
#pragma unroll
            for (int q = 0; q < 8; ++q) s += x[i][q];
    float m = wave_sum(s) / (float)h;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (lane + i * 64 < nvec)
#pragma unroll
            for (int q = 0; q < 8; ++q) { float d = x[i][q] - m; v += d * d; }
    v = wave_sum(v) / (float)h;
    *mean = m;
    *rstd = rsqrtf(v + eps);
}

__global__ __launch_bounds__(256) void embed_fwd_k(const long long* __restrict__ ids, const long long* __restrict__ tt,
                                                   const float* __restrict__ word, const float* __restrict__ pos,
                                                   const float* __restrict__ type, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float eps, long long rows, int t,
                                                   int h, float p, unsigned long long seed, unsigned int sid,
                                                   bf16_t* __restrict__ y, float* __restrict__ mean,
                                                   float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = h / 8;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
        const long long id = ids[row];
        const long long ty = tt ? tt[row] : 0;
        const int tp = (int)(row % t);
        float x[MAXV][8];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int v = lane + i * 64;
            if (v < nvec) {
                float a[8], b[8], c[8];
                load8f(word + id * h + v * 8, a);
                load8f(pos + (long long)tp * h + v * 8, b);
                load8f(type + ty * h + v * 8, c);
#pragma unroll
                for (int q = 0; q < 8; ++q) x[i][q] = a[q] + c[q] + b[q];
            }
        }
        float m, rs;
        ln_stats(x, nvec, lane, h, eps, &m, &rs);
        if (lane == 0) { mean[row] = m; rstd[row] = rs; }
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int v = lane + i * 64;
            if (v < nvec) {
                float g[8], b[8], o[8], ds[8];
                load8f(gamma + v * 8, g);
                load8f(beta + v * 8, b);
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = (x[i][q] - m) * rs * g[q] + b[q];
                if (p > 0.f) {
                    dropout_scale8(seed, sid, (unsigned long long)(row * nvec + v), p, ds);
#pragma unroll
                    for (int q = 0; q < 8; ++q) o[q] *= ds[q];
                }
                *reinterpret_cast<uint4*>(y + row * h + v * 8) = pack8(o);
            }
        }
    }
}

// wave w owns position tp = w % t and batch rows bi = w / t, w / t + nbw, ...
__global__ __launch_bounds__(256) void embed_bwd_k(const bf16_t* __restrict__ dy, const long long* __restrict__ ids,
                                                   const long long* __restrict__ tt, const float* __restrict__ word,
                                                   const float* __restrict__ pos, const float* __restrict__ type,
                                                   const float* __restrict__ gamma, const float* __restrict__ mean,
                                                   const float* __restrict__ rstd, int b, int t, int h, int nbw, float p,
                                                   unsigned long long seed, unsigned int sid, float* __restrict__ dword,
                                                   float* __restrict__ dpos, float* __restrict__ dtype,
                                                   float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = h / 8;
    const long long gw = (long long)blockIdx.x * 4 + wave;
    if (gw >= (long long)t * nbw) return;
    const int tp = (int)(gw % t);
    const int bg = (int)(gw / t);
    float aps[MAXV][8], agm[MAXV][8], abt[MAXV][8], at0[MAXV][8], at1[MAXV][8], g[MAXV][8];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { aps[i][q] = agm[i][q] = abt[i][q] = at0[i][q] = at1[i][q] = 0.f; g[i][q] = 0.f; }
        int v = lane + i * 64;
        if (v < nvec) load8f(gamma + v * 8, g[i]);
    }
    for (int bi = bg; bi < b; bi += nbw) {
        const long long row = (long long)bi * t + tp;
        const long long id = ids[row];
        const long long ty = tt ? tt[row] : 0;
        const float m = mean[row], rs = rstd[row];
        float xh[MAXV][8], dg[MAXV][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int v = lane + i * 64;
            if (v < nvec) {
                float a[8], bb[8], c[8], d[8];
                load8f(word + id * h + v * 8, a);
                load8f(pos + (long long)tp * h + v * 8, bb);
                load8f(type + ty * h + v * 8, c);
                unpack8(*reinterpret_cast<const uint4*>(dy + row * h + v * 8), d);
                if (p > 0.f) {
                    float ds[8];
                    dropout_scale8(seed, sid, (unsigned long long)(row * nvec + v), p, ds);
#pragma unroll
                    for (int q = 0; q < 8; ++q) d[q] *= ds[q];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float xv = (a[q] + c[q] + bb[q] - m) * rs;
                    xh[i][q] = xv;
                    agm[i][q] += d[q] * xv;
                    abt[i][q] += d[q];
                    float dgv = d[q] * g[i][q];
                    dg[i][q] = dgv;
                    s1 += dgv;
                    s2 += dgv * xv;
                }
            }
        }
        s1 = wave_sum(s1) / (float)h;
        s2 = wave_sum(s2) / (float)h;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int v = lane + i * 64;
            if (v < nvec) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float dx = rs * (dg[i][q] - s1 - xh[i][q] * s2);
                    aps[i][q] += dx;
                    if (ty == 0) at0[i][q] += dx; else at1[i][q] += dx;
                    atomicAdd(dword + id * h + v * 8 + q, dx);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int v = lane + i * 64;
        if (v < nvec) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                atomicAdd(dpos + (long long)tp * h + v * 8 + q, aps[i][q]);
                atomicAdd(dgamma + v * 8 + q, agm[i][q]);
                atomicAdd(dbeta + v * 8 + q, abt[i][q]);
                atomicAdd(dtype + v * 8 + q, at0[i][q]);
                if (at1[i][q] != 0.f) atomicAdd(dtype + h + v * 8 + q, at1[i][q]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void add_ln_fwd_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float eps, long long rows, int h, float p, unsigned long long seed,
                                                    unsigned int sid, bf16_t* __restrict__ y, float* __restrict__ mean,
                                                    float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = h / 8;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
        float s[MAXV][8];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int v = lane + i * 64;
            if (v < nvec) {
                float a[8], r[8];
                unpack8(*reinterpret_cast<const uint4*>(x + row * h + v * 8), a);
                unpack8(*reinterpret_cast<const uint4*>(res + row * h + v * 8), r);
                if (p > 0.f) {
                    float ds[8];
                    dropout_scale8(seed, sid, (unsigned long long)(row * nvec + v), p, ds);
#pragma unroll
                    for (int q = 0; q < 8; ++q) a[q] *= ds[q];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) s[i][q] = a[q] + r[q];
            }
        }
        float m, rs;
        ln_stats(s, nvec, lane, h, eps, &m, &rs);
        if (lane == 0) { mean[row] = m; rstd[row] = rs; }
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int v = lane + i * 64;
            if (v < nvec) {
                float g[8], b[8], o[8];
                load8f(gamma + v * 8, g);
                load8f(beta + v * 8, b);
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = (s[i][q] - m) * rs * g[q] + b[q];
                *reinterpret_cast<uint4*>(y + row * h + v * 8) = pack8(o);
            }
        }
    }
}

__global__ __launch_bounds__(256) void add_ln_bwd_k(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                    const bf16_t* __restrict__ res, const float* __restrict__ gamma,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    long long rows, int h, float p, unsigned long long seed,
                                                    unsigned int sid, bf16_t* __restrict__ dx, bf16_t* __restrict__ dres,
                                                    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nvec = h / 8;
    float agm[MAXV][8], abt[MAXV][8], g[MAXV][8];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { agm[i][q] = abt[i][q] = 0.f; g[i][q] = 0.f; }
        int v = lane + i * 64;
        if (v < nvec) load8f(gamma + v * 8, g[i]);
    }
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
        const float m = mean[row], rs = rstd[row];
        float xh[MAXV][8], dg[MAXV][8], ds[MAXV][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int v = lane + i * 64;
#pragma unroll
            for (int q = 0; q < 8; ++q) ds[i][q] = 1.f;
            if (v < nvec) {
                float a[8], r[8], d[8];
                unpack8(*reinterpret_cast<const uint4*>(x + row * h + v * 8), a);
                unpack8(*reinterpret_cast<const uint4*>(res + row * h + v * 8), r);
                unpack8(*reinterpret_cast<const uint4*>(dy + row * h + v * 8), d);
                if (p > 0.f) dropout_scale8(seed, sid, (unsigned long long)(row * nvec + v), p, ds[i]);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float xv = (a[q] * ds[i][q] + r[q] - m) * rs;
                    xh[i][q] = xv;
                    agm[i][q] += d[q] * xv;
                    abt[i][q] += d[q];
                    float dgv = d[q] * g[i][q];
                    dg[i][q] = dgv;
                    s1 += dgv;
                    s2 += dgv * xv;
                }
            }
        }
        s1 = wave_sum(s1) / (float)h;
        s2 = wave_sum(s2) / (float)h;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int v = lane + i * 64;
            if (v < nvec) {
                float o[8], o2[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    o[q] = rs * (dg[i][q] - s1 - xh[i][q] * s2);
                    o2[q] = o[q] * ds[i][q];
                }
                *reinterpret_cast<uint4*>(dres + row * h + v * 8) = pack8(o);
                *reinterpret_cast<uint4*>(dx + row * h + v * 8) = pack8(o2);
            }
        }
    }
    // block reduce of dgamma / dbeta over the 4 waves, then one atomic per element per workgroup
    __shared__ float red[4][MAXV * 64 * 16];
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            red[wave][((i * 64 + lane) * 16) + q] = agm[i][q];
            red[wave][((i * 64 + lane) * 16) + 8 + q] = abt[i][q];
        }
    __syncthreads();
    for (int e = threadIdx.x; e < nvec * 16; e += 256) {
        int v = e / 16, q = e % 16;
        int i = v / 64, ln = v % 64;
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += red[w][((i * 64 + ln) * 16) + q];
        if (q < 8) atomicAdd(dgamma + v * 8 + q, s);
        else atomicAdd(dbeta + v * 8 + (q - 8), s);
    }
}

constexpr int MAXT = 8;     // keys per lane: t <= 512

__global__ __launch_bounds__(256) void softmax_fwd_k(const float* __restrict__ sc, long long rows, int t, float p,
                                                     unsigned long long seed, unsigned int sid,
                                                     bf16_t* __restrict__ probs, bf16_t* __restrict__ pd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
        float v[MAXT];
        float mx = -3.4e38f;
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
            int k = lane + i * 64;
            v[i] = (k < t) ? sc[row * t + k] : -3.4e38f;
            mx = fmaxf(mx, v[i]);
        }
        mx = wave_max(mx);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
            int k = lane + i * 64;
            v[i] = (k < t) ? __expf(v[i] - mx) : 0.f;
            s += v[i];
        }
        s = 1.f / wave_sum(s);
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
            int k = lane + i * 64;
            if (k < t) {
                float pr = v[i] * s;
                long long e = row * t + k;
                probs[e] = f2bf(pr);
                if (p > 0.f) {
                    float ds[8];
                    dropout_scale8(seed, sid, (unsigned long long)(e >> 3), p, ds);
                    pd[e] = f2bf(pr * ds[e & 7]);
                } else if (pd != probs) {
                    pd[e] = f2bf(pr);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void softmax_bwd_k(const bf16_t* __restrict__ probs, const float* __restrict__ dpd,
                                                     long long rows, int t, float p, unsigned long long seed,
                                                     unsigned int sid, float alpha, bf16_t* __restrict__ ds_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
        float pr[MAXT], dp[MAXT];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
            int k = lane + i * 64;
            pr[i] = 0.f; dp[i] = 0.f;
            if (k < t) {
                long long e = row * t + k;
                pr[i] = bf2f(probs[e]);
                float d = dpd[e];
                if (p > 0.f) {
                    float dsc[8];
                    dropout_scale8(seed, sid, (unsigned long long)(e >> 3), p, dsc);
                    d *= dsc[e & 7];
                }
                dp[i] = d;
                dot += pr[i] * d;
            }
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {
            int k = lane + i * 64;
            if (k < t) ds_out[row * t + k] = f2bf(pr[i] * (dp[i] - dot) * alpha);
        }
    }
}

// 8 keys per lane (t % 8 == 0, t / 8 a power of two <= 64): a row is G = t/8 adjacent lanes, a wave holds 64/G rows.
// 32-byte score loads, 16-byte probability stores, ONE dropout draw (8 factors) per lane instead of one per element.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
template <int G>
__global__ __launch_bounds__(256) void softmax_fwd_v8_k(const float* __restrict__ sc, long long rows, float p,
                                                        unsigned long long seed, unsigned int sid,
                                                        bf16_t* __restrict__ probs, bf16_t* __restrict__ pd) {
    constexpr int RPW = 64 / G;                                  // rows per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / G, gl = lane % G;
    for (long long row0 = ((long long)blockIdx.x * 4 + wave) * RPW; row0 < rows; row0 += (long long)gridDim.x * 4 * RPW) {
        const long long row = row0 + sub;
        const bool ok = row < rows;
        const long long e8 = (ok ? row : 0) * G + gl;            // index of this lane's 8-element group
        float v[8];
        load8f(sc + e8 * 8, v);
        float mx = v[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) mx = fmaxf(mx, v[q]);
        mx = group_max<G>(mx);
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) { v[q] = __expf(v[q] - mx); s += v[q]; }
        s = 1.f / group_sum<G>(s);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] *= s;
        if (!ok) continue;
        *reinterpret_cast<uint4*>(probs + e8 * 8) = pack8(v);
        if (p > 0.f) {
            float ds[8];
            dropout_scale8(seed, sid, (unsigned long long)e8, p, ds);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] *= ds[q];
            *reinterpret_cast<uint4*>(pd + e8 * 8) = pack8(v);
        } else if (pd != probs) {
            *reinterpret_cast<uint4*>(pd + e8 * 8) = pack8(v);
        }
    }
}
template <int G>
__global__ __launch_bounds__(256) void softmax_bwd_v8_k(const bf16_t* __restrict__ probs, const float* __restrict__ dpd,
                                                        long long rows, float p, unsigned long long seed,
                                                        unsigned int sid, float alpha, bf16_t* __restrict__ ds_out) {
    constexpr int RPW = 64 / G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / G, gl = lane % G;
    for (long long row0 = ((long long)blockIdx.x * 4 + wave) * RPW; row0 < rows; row0 += (long long)gridDim.x * 4 * RPW) {
        const long long row = row0 + sub;
        const bool ok = row < rows;
        const long long e8 = (ok ? row : 0) * G + gl;
        float pr[8], d[8];
        unpack8(*reinterpret_cast<const uint4*>(probs + e8 * 8), pr);
        load8f(dpd + e8 * 8, d);
        if (p > 0.f) {
            float dsc[8];
            dropout_scale8(seed, sid, (unsigned long long)e8, p, dsc);
#pragma unroll
            for (int q = 0; q < 8; ++q) d[q] *= dsc[q];
        }
        float dot = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) dot += pr[q] * d[q];
        dot = group_sum<G>(dot);
        if (!ok) continue;
#pragma unroll
        for (int q = 0; q < 8; ++q) d[q] = pr[q] * (d[q] - dot) * alpha;
        *reinterpret_cast<uint4*>(ds_out + e8 * 8) = pack8(d);
    }
}

__global__ void gelu_fwd_k(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long long n8) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(x + i * 8), f);
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = gelu_f(f[q]);
        *reinterpret_cast<uint4*>(y + i * 8) = pack8(f);
    }
}
__global__ void gelu_bwd_k(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, bf16_t* __restrict__ dx, long long n8) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        float f[8], g[8];
        unpack8(*reinterpret_cast<const uint4*>(x + i * 8), f);
        unpack8(*reinterpret_cast<const uint4*>(dy + i * 8), g);
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = g[q] * gelu_grad_f(f[q]);
        *reinterpret_cast<uint4*>(dx + i * 8) = pack8(f);
    }
}
__global__ void mask_bias_k(const long long* __restrict__ mask, float* __restrict__ out, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mask[i] ? 0.f : -3.4028234663852886e38f;
}
__global__ void eos_gather_k(const bf16_t* __restrict__ hid, const long long* __restrict__ mask, int t, int h,
                             float* __restrict__ out) {
    const int bi = blockIdx.x;
    __shared__ int idx;
    if (threadIdx.x == 0) {
        long long s = 0;
        for (int k = 0; k < t; ++k) s += mask[(long long)bi * t + k];
        int e = (int)s - 1;
        if (e < 0) e += t;                       // python negative index semantics of clip.py:67-68
        idx = e;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < h; i += blockDim.x) out[(long long)bi * h + i] = bf2f(hid[((long long)bi * t + idx) * h + i]);
}
__global__ void eos_scatter_k(const float* __restrict__ dout, const long long* __restrict__ mask, int t, int h,
                              bf16_t* __restrict__ dhid) {
    const int bi = blockIdx.x;
    __shared__ int idx;
    if (threadIdx.x == 0) {
        long long s = 0;
        for (int k = 0; k < t; ++k) s += mask[(long long)bi * t + k];
        int e = (int)s - 1;
        if (e < 0) e += t;
        idx = e;
    }
    __syncthreads();
    for (long long i = threadIdx.x; i < (long long)t * h; i += blockDim.x) {
        int k = (int)(i / h), c = (int)(i % h);
        dhid[(long long)bi * t * h + i] = (k == idx) ? f2bf(dout[(long long)bi * h + c]) : (bf16_t)0;
    }
}

int row_blocks(long long rows) {
    long long b = (rows + 3) / 4;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int mc_bert_embed_fwd(const long long* ids, const long long* tt, const float* word, const float* pos,
                                 const float* type, const float* gamma, const float* beta, float eps, int b, int t, int h,
                                 float p, unsigned long long seed, unsigned int stream_id, mc_bf16* y, float* mean,
                                 float* rstd, void* stream) {
    MC_CHECK(ids && word && pos && type && gamma && beta && y && mean && rstd, "embed_fwd: null arg");
    MC_CHECK(b > 0 && t > 0 && h > 0 && h % 8 == 0 && h <= 1024, "embed_fwd: hidden must be a multiple of 8 and <= 1024");
    long long rows = (long long)b * t;
    hipLaunchKernelGGL(embed_fwd_k, dim3(row_blocks(rows)), dim3(256), 0, (hipStream_t)stream, ids, tt, word, pos, type,
                       gamma, beta, eps, rows, t, h, p, seed, stream_id, y, mean, rstd);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_bert_embed_bwd(const mc_bf16* dy, const long long* ids, const long long* tt, const float* word,
                                 const float* pos, const float* type, const float* gamma, const float* mean,
                                 const float* rstd, int b, int t, int h, float p, unsigned long long seed,
                                 unsigned int stream_id, float* dword, float* dpos, float* dtype, float* dgamma,
                                 float* dbeta, void* stream) {
    MC_CHECK(dy && ids && word && pos && type && gamma && mean && rstd && dword && dpos && dtype && dgamma && dbeta, "embed_bwd: null arg");
    MC_CHECK(b > 0 && t > 0 && h > 0 && h % 8 == 0 && h <= 1024, "embed_bwd: bad shape");
    int nbw = 2048 / t;
    if (nbw < 1) nbw = 1;
    if (nbw > b) nbw = b;
    long long waves = (long long)t * nbw;
    hipLaunchKernelGGL(embed_bwd_k, dim3(mc_div_up(waves, 4)), dim3(256), 0, (hipStream_t)stream, dy, ids, tt, word, pos,
                       type, gamma, mean, rstd, b, t, h, nbw, p, seed, stream_id, dword, dpos, dtype, dgamma, dbeta);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_add_ln_fwd(const mc_bf16* x, const mc_bf16* res, const float* gamma, const float* beta, float eps,
                             long long rows, int h, float p, unsigned long long seed, unsigned int stream_id, mc_bf16* y,
                             float* mean, float* rstd, void* stream) {
    MC_CHECK(x && res && gamma && beta && y && mean && rstd, "add_ln_fwd: null arg");
    MC_CHECK(rows > 0 && h > 0 && h % 8 == 0 && h <= 1024, "add_ln_fwd: bad shape");
    hipLaunchKernelGGL(add_ln_fwd_k, dim3(row_blocks(rows)), dim3(256), 0, (hipStream_t)stream, x, res, gamma, beta, eps, rows,
                       h, p, seed, stream_id, y, mean, rstd);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_add_ln_bwd(const mc_bf16* dy, const mc_bf16* x, const mc_bf16* res, const float* gamma,
                             const float* mean, const float* rstd, long long rows, int h, float p,
                             unsigned long long seed, unsigned int stream_id, mc_bf16* dx, mc_bf16* dres, float* dgamma,
                             float* dbeta, void* stream) {
    MC_CHECK(dy && x && res && gamma && mean && rstd && dx && dres && dgamma && dbeta, "add_ln_bwd: null arg");
    MC_CHECK(rows > 0 && h > 0 && h % 8 == 0 && h <= 1024, "add_ln_bwd: bad shape");
    long long blocks = (rows + 3) / 4;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(add_ln_bwd_k, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, dy, x, res, gamma, mean, rstd, rows,
                       h, p, seed, stream_id, dx, dres, dgamma, dbeta);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_softmax_fwd(const float* scores, long long rows, int t, float p, unsigned long long seed,
                              unsigned int stream_id, mc_bf16* probs, mc_bf16* probs_drop, void* stream) {
    MC_CHECK(scores && probs && probs_drop && rows > 0 && t > 0 && t <= 512, "softmax_fwd: bad args (t <= 512)");
    MC_CHECK(p == 0.f || t % 8 == 0, "softmax_fwd: dropout needs t % 8 == 0");
    const int g = t / 8;
    const bool v8 = t % 8 == 0 && g >= 2 && g <= 64 && (g & (g - 1)) == 0 && mc_aligned16(scores) && mc_aligned16(probs) && mc_aligned16(probs_drop);
    if (v8) {
        const int blocks = row_blocks(mc_div_up(rows, 64 / g)) * 2;
#define SM_FWD(G) hipLaunchKernelGGL(softmax_fwd_v8_k<G>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, scores, rows, p, seed, stream_id, probs, probs_drop)
        switch (g) { case 2: SM_FWD(2); break; case 4: SM_FWD(4); break; case 8: SM_FWD(8); break; case 16: SM_FWD(16); break;
                     case 32: SM_FWD(32); break; default: SM_FWD(64); break; }
#undef SM_FWD
    } else {
        hipLaunchKernelGGL(softmax_fwd_k, dim3(row_blocks(rows) * 2), dim3(256), 0, (hipStream_t)stream, scores, rows, t, p, seed,
                           stream_id, probs, probs_drop);
    }
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_softmax_bwd(const mc_bf16* probs, const float* dprobs_drop, long long rows, int t, float p,
                              unsigned long long seed, unsigned int stream_id, float alpha, mc_bf16* dscores,
                              void* stream) {
    MC_CHECK(probs && dprobs_drop && dscores && rows > 0 && t > 0 && t <= 512, "softmax_bwd: bad args (t <= 512)");
    const int g = t / 8;
    const bool v8 = t % 8 == 0 && g >= 2 && g <= 64 && (g & (g - 1)) == 0 && mc_aligned16(probs) && mc_aligned16(dprobs_drop) && mc_aligned16(dscores);
    if (v8) {
        const int blocks = row_blocks(mc_div_up(rows, 64 / g)) * 2;
#define SM_BWD(G) hipLaunchKernelGGL(softmax_bwd_v8_k<G>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, probs, dprobs_drop, rows, p, seed, stream_id, alpha, dscores)
        switch (g) { case 2: SM_BWD(2); break; case 4: SM_BWD(4); break; case 8: SM_BWD(8); break; case 16: SM_BWD(16); break;
                     case 32: SM_BWD(32); break; default: SM_BWD(64); break; }
#undef SM_BWD
    } else {
        hipLaunchKernelGGL(softmax_bwd_k, dim3(row_blocks(rows) * 2), dim3(256), 0, (hipStream_t)stream, probs, dprobs_drop, rows,
                           t, p, seed, stream_id, alpha, dscores);
    }
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_gelu_fwd(const mc_bf16* x, mc_bf16* y, long long n, void* stream) {
    MC_CHECK(x && y && n > 0 && n % 8 == 0, "gelu_fwd: n must be a positive multiple of 8");
    long long n8 = n / 8;
    int blocks = mc_div_up(n8, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gelu_fwd_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, n8);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_gelu_bwd(const mc_bf16* dy, const mc_bf16* x, mc_bf16* dx, long long n, void* stream) {
    MC_CHECK(dy && x && dx && n > 0 && n % 8 == 0, "gelu_bwd: n must be a positive multiple of 8");
    long long n8 = n / 8;
    int blocks = mc_div_up(n8, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gelu_bwd_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, dx, n8);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_mask_bias(const long long* mask, float* out, long long n, void* stream) {
    MC_CHECK(mask && out && n > 0, "mask_bias: bad args");
    hipLaunchKernelGGL(mask_bias_k, dim3(mc_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, mask, out, n);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_eos_gather(const mc_bf16* hid, const long long* mask, int b, int t, int h, float* out, void* stream) {
    MC_CHECK(hid && mask && out && b > 0 && t > 0 && h > 0, "eos_gather: bad args");
    hipLaunchKernelGGL(eos_gather_k, dim3(b), dim3(256), 0, (hipStream_t)stream, hid, mask, t, h, out);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
extern "C" int mc_eos_scatter(const float* dout, const long long* mask, int b, int t, int h, mc_bf16* dhid, void* stream) {
    MC_CHECK(dout && mask && dhid && b > 0 && t > 0 && h > 0, "eos_scatter: bad args");
    hipLaunchKernelGGL(eos_scatter_k, dim3(b), dim3(256), 0, (hipStream_t)stream, dout, mask, t, h, dhid);
    MC_LAUNCH_CHECK();
    return MC_OK;
}
