// Multi-tensor AdamW update for the hot loop's optimizer step [ref: breastclip/optimizer/__init__.py:28-29 builds
// torch.optim.AdamW over ALL parameters; trainer_ddp.py:300-303 steps it once per iteration].
// HBM-bound: 16 B read + 12 B written per element (param, grad, exp_avg, exp_avg_sq), ~3.9 GB for the 138 M-parameter
// B5 + BERT model; a parameter that is consumed as a bf16 matrix gets its bf16 image rewritten by the same pass (+2 B)
// instead of by a cast kernel of its own in the next forward.  Up to PACK tensors go into one launch (pointers travel as kernel arguments, no device
// table to keep in sync); a workgroup owns one CHUNK of one tensor, found by a scan over the pack's chunk prefix.
#include "common_hip.h"
#include "../../include/mammoclip_hip.h"

namespace {

constexpr int PACK = 40;             // tensors per launch: 40 * 44 B + scalars stays far below the 4 KB argument limit
constexpr int CHUNK = 16384;         // elements per workgroup: 64 KB per stream in flight across its 256 lanes

struct AdamPack {
    bf16_t* img[PACK];               // optional bf16 image of the updated parameter (nullptr: none)
    float* p[PACK];
    const float* g[PACK];
    float* m[PACK];
    float* v[PACK];
    int first_chunk[PACK + 1];       // prefix of chunk counts
    long long n[PACK];
};

struct AdamScalars {
    float lr_wd;        // lr * weight_decay
    float beta1, beta2;
    float one_m_beta1, one_m_beta2;
    float step_size;    // lr / (1 - beta1^t)
    float inv_bc2_sqrt; // 1 / sqrt(1 - beta2^t)
    float eps;
};

__device__ __forceinline__ void adamw1(float& p, float g, float& m, float& v, const AdamScalars& s) {
    // same operation order as torch's AdamW: decoupled decay first, moments, then the bias-corrected step
    p -= s.lr_wd * p;
    m = m + s.one_m_beta1 * (g - m);
    v = s.beta2 * v + s.one_m_beta2 * g * g;
    const float denom = sqrtf(v) * s.inv_bc2_sqrt + s.eps;
    p -= s.step_size * (m / denom);
}

// Loss-scaled step without a host sync (f16 storage build): the non-finite flag of the gradient unscale and the count of
// steps skipped so far stay on the device.  A set flag makes the launch a no-op (GradScaler.step() skips optimizer.step());
// the bias corrections use the number of APPLIED steps, host step - skipped, like an optimizer that was never called.
struct AdamLs {
    const float* found_inf;    // nullptr: plain step with the host-derived scalars
    const float* skipped;      // steps skipped so far (float counter)
    double lr, beta1, beta2;
    long long step;
};

__global__ void __launch_bounds__(256) adamw_multi_k(AdamPack pk, int count, AdamScalars s, AdamLs ls) {
    if (ls.found_inf) {
        if (*ls.found_inf != 0.f) return;                              // (uniform: the whole grid leaves)
        __shared__ float s_dev[2];
        if (threadIdx.x == 0) {
            long long te = ls.step - (long long)(*ls.skipped);
            if (te < 1) te = 1;
            s_dev[0] = (float)(ls.lr / (1.0 - pow(ls.beta1, (double)te)));
            s_dev[1] = (float)(1.0 / sqrt(1.0 - pow(ls.beta2, (double)te)));
        }
        __syncthreads();
        s.step_size = s_dev[0];
        s.inv_bc2_sqrt = s_dev[1];
    }
    int t = 0;
    const int blk = blockIdx.x;
    while (t + 1 < count && pk.first_chunk[t + 1] <= blk) ++t;       // uniform scan, <= PACK scalar compares
    const long long n = pk.n[t];
    const long long base = (long long)(blk - pk.first_chunk[t]) * CHUNK;
    float* __restrict__ p = pk.p[t] + base;
    const float* __restrict__ g = pk.g[t] + base;
    float* __restrict__ m = pk.m[t] + base;
    float* __restrict__ v = pk.v[t] + base;
    bf16_t* __restrict__ img = pk.img[t] ? pk.img[t] + base : nullptr;      // base is a multiple of 16384: alignment kept
    const long long left = n - base;
    const int len = left < CHUNK ? (int)left : CHUNK;
    const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0 &&
                     (((uintptr_t)img) & 7) == 0;
    int i = threadIdx.x * 4;
    if (vec) {
        // 2 float4 per array in flight per lane (8 loads) before the first use
        for (; i + 1024 + 3 < len; i += 2048) {
            float4 P0 = *(const float4*)(p + i), P1 = *(const float4*)(p + i + 1024);
            float4 G0 = *(const float4*)(g + i), G1 = *(const float4*)(g + i + 1024);
            float4 M0 = *(const float4*)(m + i), M1 = *(const float4*)(m + i + 1024);
            float4 V0 = *(const float4*)(v + i), V1 = *(const float4*)(v + i + 1024);
            adamw1(P0.x, G0.x, M0.x, V0.x, s); adamw1(P0.y, G0.y, M0.y, V0.y, s);
            adamw1(P0.z, G0.z, M0.z, V0.z, s); adamw1(P0.w, G0.w, M0.w, V0.w, s);
            adamw1(P1.x, G1.x, M1.x, V1.x, s); adamw1(P1.y, G1.y, M1.y, V1.y, s);
            adamw1(P1.z, G1.z, M1.z, V1.z, s); adamw1(P1.w, G1.w, M1.w, V1.w, s);
            *(float4*)(p + i) = P0; *(float4*)(p + i + 1024) = P1;
            *(float4*)(m + i) = M0; *(float4*)(m + i + 1024) = M1;
            *(float4*)(v + i) = V0; *(float4*)(v + i + 1024) = V1;
            if (img) {
                *(uint2*)(img + i) = make_uint2(pack_bf2(P0.x, P0.y), pack_bf2(P0.z, P0.w));
                *(uint2*)(img + i + 1024) = make_uint2(pack_bf2(P1.x, P1.y), pack_bf2(P1.z, P1.w));
            }
        }
        for (; i + 3 < len; i += 1024) {
            float4 P0 = *(const float4*)(p + i), G0 = *(const float4*)(g + i);
            float4 M0 = *(const float4*)(m + i), V0 = *(const float4*)(v + i);
            adamw1(P0.x, G0.x, M0.x, V0.x, s); adamw1(P0.y, G0.y, M0.y, V0.y, s);
            adamw1(P0.z, G0.z, M0.z, V0.z, s); adamw1(P0.w, G0.w, M0.w, V0.w, s);
            *(float4*)(p + i) = P0; *(float4*)(m + i) = M0; *(float4*)(v + i) = V0;
            if (img) *(uint2*)(img + i) = make_uint2(pack_bf2(P0.x, P0.y), pack_bf2(P0.z, P0.w));
        }
        // ragged tail of the chunk: the (< 4) elements after the last whole float4
        const int done = len & ~3;
        const int j = done + threadIdx.x;
        if (j < len) {
            float P = p[j], M = m[j], V = v[j];
            adamw1(P, g[j], M, V, s);
            p[j] = P; m[j] = M; v[j] = V;
            if (img) img[j] = f2bf(P);
        }
    } else {
        for (int j = threadIdx.x; j < len; j += 256) {
            float P = p[j], M = m[j], V = v[j];
            adamw1(P, g[j], M, V, s);
            p[j] = P; m[j] = M; v[j] = V;
            if (img) img[j] = f2bf(P);
        }
    }
}

// ---- gradient unscale + non-finite check of the loss-scaled (f16 storage) step [ref: trainer.py:271-278 runs the backward
// under torch.cuda.amp.GradScaler: scaler.step() = unscale the gradients, skip the update if any is inf / nan].
// Same packing as the update kernel: a workgroup owns one CHUNK of one gradient tensor.
struct UnscalePack {
    float* g[PACK];
    int first_chunk[PACK + 1];
    long long n[PACK];
};

__global__ void __launch_bounds__(256) grads_unscale_k(UnscalePack pk, int count, float inv_scale, const float* __restrict__ scale_dev,
                                                       float* __restrict__ found_inf) {
    if (scale_dev) inv_scale = 1.0f / *scale_dev;                       // the dynamic scale lives on the device (no host sync)
    int t = 0;
    const int blk = blockIdx.x;
    while (t + 1 < count && pk.first_chunk[t + 1] <= blk) ++t;
    const long long base = (long long)(blk - pk.first_chunk[t]) * CHUNK;
    float* __restrict__ g = pk.g[t] + base;
    const long long left = pk.n[t] - base;
    const int len = left < CHUNK ? (int)left : CHUNK;
    bool bad = false;
    int i = threadIdx.x * 4;
    if ((((uintptr_t)g) & 15) == 0) {
        for (; i + 3 < len; i += 1024) {
            float4 v = *(const float4*)(g + i);
            v.x *= inv_scale; v.y *= inv_scale; v.z *= inv_scale; v.w *= inv_scale;
            bad |= !(fabsf(v.x) <= 3.4e38f) || !(fabsf(v.y) <= 3.4e38f) || !(fabsf(v.z) <= 3.4e38f) || !(fabsf(v.w) <= 3.4e38f);
            *(float4*)(g + i) = v;
        }
        const int j = (len & ~3) + threadIdx.x;
        if (j < len) { const float v = g[j] * inv_scale; bad |= !(fabsf(v) <= 3.4e38f); g[j] = v; }
    } else {
        for (int j = threadIdx.x; j < len; j += 256) { const float v = g[j] * inv_scale; bad |= !(fabsf(v) <= 3.4e38f); g[j] = v; }
    }
    if (bad) *found_inf = 1.0f;          // (every writer stores the same value)
}

// GradScaler.update() on the device [ref: trainer_ddp.py:303]: state = {scale, clean steps in a row, found_inf flag of the
// step, steps skipped (this scaler), _}.  Consumes and clears the flag; the optimizer's own skipped-step counter follows.
__global__ void loss_scale_update_k(float* __restrict__ st, float* __restrict__ opt_skipped, float growth, float backoff, int interval, int dynamic) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool bad = st[2] != 0.f;
    if (bad) { st[3] += 1.f; if (opt_skipped) *opt_skipped += 1.f; }
    if (dynamic) {
        if (bad) { st[0] *= backoff; st[1] = 0.f; }
        else {
            st[1] += 1.f;
            if (st[1] >= (float)interval) { st[0] *= growth; st[1] = 0.f; }
        }
    }
    st[4] = bad ? 1.f : 0.f;        // what happened to the step just finished (for whoever looks, later)
    st[2] = 0.f;
}

int grads_unscale_impl(const mc_adamw_tensor* tensors, int n_tensors, float inv_scale, const float* scale_dev, float* found_inf, void* stream);

}  // namespace

extern "C" int mc_grads_unscale(const mc_adamw_tensor* tensors, int n_tensors, float inv_scale, float* found_inf, void* stream) {
    return grads_unscale_impl(tensors, n_tensors, inv_scale, nullptr, found_inf, stream);
}

extern "C" int mc_grads_unscale_dev(const mc_adamw_tensor* tensors, int n_tensors, const float* scale_dev, float* found_inf, void* stream) {
    MC_CHECK(scale_dev, "grads_unscale_dev: null scale");
    return grads_unscale_impl(tensors, n_tensors, 1.0f, scale_dev, found_inf, stream);
}

extern "C" int mc_loss_scale_update(float* state, float* opt_skipped, float growth_factor, float backoff_factor, int growth_interval,
                                    int dynamic, void* stream) {
    MC_CHECK(state && growth_factor > 0.f && backoff_factor > 0.f && growth_interval >= 1, "loss_scale_update: bad arguments");
    hipLaunchKernelGGL(loss_scale_update_k, dim3(1), dim3(64), 0, (hipStream_t)stream, state, opt_skipped, growth_factor, backoff_factor,
                       growth_interval, dynamic);
    MC_LAUNCH_CHECK();
    return MC_OK;
}

namespace {
int grads_unscale_impl(const mc_adamw_tensor* tensors, int n_tensors, float inv_scale, const float* scale_dev, float* found_inf, void* stream) {
    MC_CHECK(n_tensors >= 0 && (tensors || n_tensors == 0) && found_inf, "grads_unscale: bad arguments");
    UnscalePack pk;
    int cnt = 0, chunks = 0;
    auto flush = [&]() -> int {
        if (cnt == 0) return MC_OK;
        pk.first_chunk[cnt] = chunks;
        hipLaunchKernelGGL(grads_unscale_k, dim3(chunks), dim3(256), 0, (hipStream_t)stream, pk, cnt, inv_scale, scale_dev, found_inf);
        MC_LAUNCH_CHECK();
        cnt = 0; chunks = 0;
        return MC_OK;
    };
    for (int i = 0; i < n_tensors; ++i) {
        const mc_adamw_tensor& t = tensors[i];
        if (t.numel == 0) continue;
        MC_CHECK(t.grad && t.numel > 0, "grads_unscale: null gradient pointer");
        const long long nch = (t.numel + CHUNK - 1) / CHUNK;
        MC_CHECK(nch < (1ll << 30), "grads_unscale: tensor too large");
        if (cnt == PACK || (long long)chunks + nch > (1ll << 30)) {
            int r = flush();
            if (r != MC_OK) return r;
        }
        pk.g[cnt] = const_cast<float*>(t.grad); pk.n[cnt] = t.numel; pk.first_chunk[cnt] = chunks;
        chunks += (int)nch;
        ++cnt;
    }
    return flush();
}

int adamw_impl(const mc_adamw_tensor* tensors, int n_tensors, double lr, double beta1, double beta2, double eps, double weight_decay,
               long long step, const float* found_inf, const float* skipped, void* stream);
}  // namespace

extern "C" int mc_adamw_step(const mc_adamw_tensor* tensors, int n_tensors, double lr, double beta1, double beta2,
                             double eps, double weight_decay, long long step, void* stream) {
    return adamw_impl(tensors, n_tensors, lr, beta1, beta2, eps, weight_decay, step, nullptr, nullptr, stream);
}

extern "C" int mc_adamw_step_ls(const mc_adamw_tensor* tensors, int n_tensors, double lr, double beta1, double beta2,
                                double eps, double weight_decay, long long step, const float* found_inf, const float* skipped,
                                void* stream) {
    MC_CHECK(found_inf && skipped, "adamw_step_ls: null loss-scale state");
    return adamw_impl(tensors, n_tensors, lr, beta1, beta2, eps, weight_decay, step, found_inf, skipped, stream);
}

namespace {
int adamw_impl(const mc_adamw_tensor* tensors, int n_tensors, double lr, double beta1, double beta2, double eps, double weight_decay,
               long long step, const float* found_inf, const float* skipped, void* stream) {
    MC_CHECK(n_tensors >= 0 && (tensors || n_tensors == 0), "adamw: bad tensor list");
    MC_CHECK(step >= 1 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, "adamw: bad hyper-parameters");
    // scalars are derived in double like torch does on the host (1 - 0.999 in fp32 is already 1.3e-5 off)
    AdamScalars s;
    s.lr_wd = (float)(lr * weight_decay);
    s.beta1 = (float)beta1; s.beta2 = (float)beta2;
    s.one_m_beta1 = (float)(1.0 - beta1); s.one_m_beta2 = (float)(1.0 - beta2);
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    s.step_size = (float)(lr / bc1);
    s.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    s.eps = (float)eps;
    AdamLs ls;
    ls.found_inf = found_inf; ls.skipped = skipped; ls.lr = lr; ls.beta1 = beta1; ls.beta2 = beta2; ls.step = step;
    AdamPack pk;
    int cnt = 0, chunks = 0;
    auto flush = [&]() -> int {
        if (cnt == 0) return MC_OK;
        pk.first_chunk[cnt] = chunks;
        hipLaunchKernelGGL(adamw_multi_k, dim3(chunks), dim3(256), 0, (hipStream_t)stream, pk, cnt, s, ls);
        MC_LAUNCH_CHECK();
        cnt = 0; chunks = 0;
        return MC_OK;
    };
    for (int i = 0; i < n_tensors; ++i) {
        const mc_adamw_tensor& t = tensors[i];
        if (t.numel == 0) continue;
        MC_CHECK(t.param && t.grad && t.exp_avg && t.exp_avg_sq && t.numel > 0, "adamw: null tensor pointer");
        const long long nch = (t.numel + CHUNK - 1) / CHUNK;
        MC_CHECK(nch < (1ll << 30), "adamw: tensor too large");
        if (cnt == PACK || (long long)chunks + nch > (1ll << 30)) {
            int r = flush();
            if (r != MC_OK) return r;
        }
        pk.p[cnt] = t.param; pk.g[cnt] = t.grad; pk.m[cnt] = t.exp_avg; pk.v[cnt] = t.exp_avg_sq;
        pk.img[cnt] = t.bf16_image;
        pk.n[cnt] = t.numel; pk.first_chunk[cnt] = chunks;
        chunks += (int)nch;
        ++cnt;
    }
    return flush();
}
}  // namespace
