// One-launch AdamW over ALL parameter tensors of the network.
//
// Replaces torch.optim.AdamW(model.parameters(), lr, weight_decay) as the reference example builds it
// (examples/train_unet_neurodata.py:257-262): 70 parameter tensors -> one HBM-bound kernel (28 B per element: p, g, m, v
// read, p, m, v written) instead of ~10 ATen launches per tensor.  The per-step scalars (step count, bias corrections)
// live on the device so that a torch GradScaler's found_inf can veto the step without a host sync
// (training/trainer.py:539-542 scaler.step(optimizer)).
//
// Tensors keep their own allocations (checkpoint / deepcopy / SWA p.data.copy_ compatibility): the kernel gets the
// pointer pairs through its argument block; moment buffers are flat, each tensor's slice padded to whole chunks.
#include "kernels.h"

namespace {

constexpr int OPT_CHUNK = 1024;          // elements per workgroup (float4 per lane)
constexpr int OPT_MAX_T = 128;           // tensors per launch (argument block ~3.1 KB, below the 4 KB kernarg limit)

struct AdamWArgs {
    float* p[OPT_MAX_T];
    const float* g[OPT_MAX_T];
    int cstart[OPT_MAX_T + 1];            // first chunk of tensor t (relative to this launch); cstart[nt] = grid size
    int numel[OPT_MAX_T];
    int nt;
    long long chunk0;                     // global chunk index of this launch's first chunk (offset into the moment buffers)
    float* m; float* v;
    const float* coef;                    // device: {step_size, sqrt(bias_correction2), decay, inv_scale, skip}
    float beta2, w1, w2, eps;             // w1 = 1 - beta1, w2 = 1 - beta2 rounded from double (1.f - 0.999f is off by 1.3e-5)
};

// coef[0] = lr / (1 - beta1^t), coef[1] = sqrt(1 - beta2^t), coef[2] = 1 - lr*wd, coef[3] = 1 / grad_scale, coef[4] = skip flag
__global__ void adamw_prepare_kernel(float* step, const float* grad_scale, const float* found_inf,
                                     double lr, double beta1, double beta2, double wd, float* coef) {
    const bool skip = found_inf && found_inf[0] != 0.f;
    float t = step[0];
    if (!skip) { t += 1.f; step[0] = t; }
    const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
    coef[0] = (float)(lr / bc1);
    coef[1] = (float)sqrt(bc2);
    coef[2] = (float)(1.0 - lr * wd);
    coef[3] = grad_scale ? 1.f / grad_scale[0] : 1.f;
    coef[4] = skip ? 1.f : 0.f;
}

// B16: parameters and gradients are bfloat16 (a module cast with .to(torch.bfloat16)); the moments and all arithmetic stay fp32, the
// parameter is rounded to bf16 once per step.
template <bool B16>
__global__ __launch_bounds__(256) void adamw_kernel(const AdamWArgs a) {
    if (a.coef[4] != 0.f) return;                                    // GradScaler found inf/nan: parameters and moments stay
    const int b = blockIdx.x;
    int lo = 0, hi = a.nt;                                           // tensor of this chunk: largest t with cstart[t] <= b
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.cstart[mid] <= b) lo = mid; else hi = mid; }
    const int t = lo;
    const int e0 = (b - a.cstart[t]) * OPT_CHUNK + 4 * (int)threadIdx.x;
    const int n = a.numel[t];
    if (e0 >= n || a.g[t] == nullptr) return;                        // no gradient this step: tensor untouched (as torch)
    typedef unsigned short u16;
    float* __restrict__ p = B16 ? nullptr : a.p[t] + e0;
    const float* __restrict__ g = B16 ? nullptr : a.g[t] + e0;
    u16* __restrict__ p16 = B16 ? reinterpret_cast<u16*>(a.p[t]) + e0 : nullptr;
    const u16* __restrict__ g16 = B16 ? reinterpret_cast<const u16*>(a.g[t]) + e0 : nullptr;
    const size_t so = (size_t)(a.chunk0 + b) * OPT_CHUNK + 4 * threadIdx.x;
    float* __restrict__ m = a.m + so; float* __restrict__ v = a.v + so;
    const float step_size = a.coef[0], bc2s = a.coef[1], decay = a.coef[2], inv_scale = a.coef[3];
    const float b2 = a.beta2, w1 = a.w1, w2 = a.w2;
    const bool vec = !B16 && e0 + 4 <= n && ((((size_t)p | (size_t)g) & 15) == 0);
    float pv[4], gv[4], mv[4], vv[4];
    if (B16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = e0 + e < n;
            pv[e] = ok ? __uint_as_float((unsigned)p16[e] << 16) : 0.f; gv[e] = ok ? __uint_as_float((unsigned)g16[e] << 16) : 0.f;
            mv[e] = ok ? m[e] : 0.f; vv[e] = ok ? v[e] : 0.f;
        }
    } else if (vec) {
        const f32x4 P = *reinterpret_cast<const f32x4*>(p), G = *reinterpret_cast<const f32x4*>(g);
        const f32x4 M = *reinterpret_cast<const f32x4*>(m), V = *reinterpret_cast<const f32x4*>(v);
#pragma unroll
        for (int e = 0; e < 4; ++e) { pv[e] = P[e]; gv[e] = G[e]; mv[e] = M[e]; vv[e] = V[e]; }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = e0 + e < n;
            pv[e] = ok ? p[e] : 0.f; gv[e] = ok ? g[e] : 0.f; mv[e] = ok ? m[e] : 0.f; vv[e] = ok ? v[e] : 0.f;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float gr = gv[e] * inv_scale;
        pv[e] *= decay;                                              // decoupled weight decay: p *= 1 - lr*wd
        mv[e] = mv[e] + w1 * (gr - mv[e]);                           // exp_avg.lerp_(grad, 1 - beta1)
        vv[e] = vv[e] * b2 + w2 * gr * gr;                           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float denom = __builtin_sqrtf(vv[e]) / bc2s + a.eps;    // sqrt(v) / sqrt(bias_correction2) + eps
        pv[e] = pv[e] - step_size * (mv[e] / denom);
    }
    if (B16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e0 + e < n) { p16[e] = __builtin_bit_cast(unsigned short, (__bf16)pv[e]); m[e] = mv[e]; v[e] = vv[e]; }
    } else if (vec) {
        *reinterpret_cast<f32x4*>(p) = f32x4{pv[0], pv[1], pv[2], pv[3]};
        *reinterpret_cast<f32x4*>(m) = f32x4{mv[0], mv[1], mv[2], mv[3]};
        *reinterpret_cast<f32x4*>(v) = f32x4{vv[0], vv[1], vv[2], vv[3]};
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e0 + e < n) { p[e] = pv[e]; m[e] = mv[e]; v[e] = vv[e]; }
    }
}

// Stochastic weight averaging over all parameter tensors in one launch: the running average the reference's SWA wrapper keeps per
// parameter (training/swa.py:145-176 update_swa_group: buf += (p - buf) * (1 / (n_avg + 1)), two rounded fp32 operations -- kept
// un-fused here so that the averages are bit-identical) and the exchange of parameters and averages (swa.py:184-202 swap_swa_sgd).
struct SwaArgs {
    float* p[OPT_MAX_T];
    float* b[OPT_MAX_T];
    int cstart[OPT_MAX_T + 1];
    int numel[OPT_MAX_T];
    int nt;
    float decay;                          // 1 / (n_avg + 1)
    int swap;                             // 0: update the average, 1: exchange p and buf
};

__global__ __launch_bounds__(256) void swa_kernel(const SwaArgs a) {
#pragma clang fp contract(off)            // (hipcc contracts a*b+c into an fma by default, also through __fmul_rn/__fadd_rn)
    const int b = blockIdx.x;
    int lo = 0, hi = a.nt;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.cstart[mid] <= b) lo = mid; else hi = mid; }
    const int t = lo, n = a.numel[t];
    float* __restrict__ p = a.p[t];
    float* __restrict__ buf = a.b[t];
    const int e0 = (b - a.cstart[t]) * OPT_CHUNK + 4 * (int)threadIdx.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = e0 + e;
        if (i >= n) break;
        const float pv = p[i], bv = buf[i];
        if (a.swap) { p[i] = bv; buf[i] = pv; }
        else { const float diff = (pv - bv) * a.decay; buf[i] = bv + diff; }
    }
}

}  // namespace

int launch_swa(int n_tensors, void* const* params, void* const* bufs, const long long* numels, double decay, int swap, hipStream_t s) {
    for (int t0 = 0; t0 < n_tensors; t0 += OPT_MAX_T) {
        SwaArgs a;
        a.nt = n_tensors - t0 < OPT_MAX_T ? n_tensors - t0 : OPT_MAX_T;
        int c = 0;
        for (int t = 0; t < a.nt; ++t) {
            E3_REQUIRE(numels[t0 + t] >= 0 && numels[t0 + t] < (1ll << 31), E3_ERR_UNSUPPORTED, "parameter tensor too large");
            E3_REQUIRE(params[t0 + t] && bufs[t0 + t], E3_ERR_INVALID, "swa: NULL tensor pointer");
            a.p[t] = (float*)params[t0 + t]; a.b[t] = (float*)bufs[t0 + t];
            a.cstart[t] = c; a.numel[t] = (int)numels[t0 + t];
            c += (int)((numels[t0 + t] + OPT_CHUNK - 1) / OPT_CHUNK);
        }
        a.cstart[a.nt] = c; a.decay = (float)decay; a.swap = swap;
        if (c > 0) hipLaunchKernelGGL(swa_kernel, dim3(c), dim3(256), 0, s, a);
        E3_CHECK_HIP(hipGetLastError());
    }
    return E3_OK;
}

size_t adamw_state_floats(int n_tensors, const long long* numels) {
    size_t chunks = 0;
    for (int t = 0; t < n_tensors; ++t) chunks += ((size_t)numels[t] + OPT_CHUNK - 1) / OPT_CHUNK;
    return chunks * OPT_CHUNK;
}

size_t adamw_state_offset(int n_tensors, const long long* numels, int tensor) {
    size_t chunks = 0;
    for (int t = 0; t < tensor && t < n_tensors; ++t) chunks += ((size_t)numels[t] + OPT_CHUNK - 1) / OPT_CHUNK;
    return chunks * OPT_CHUNK;
}

int launch_adamw(int n_tensors, void* const* params, void* const* grads, const long long* numels, float* exp_avg, float* exp_avg_sq,
                 float* step, float* coef, double lr, double beta1, double beta2, double eps, double weight_decay,
                 const float* grad_scale, const float* found_inf, hipStream_t s, int bf16) {
    hipLaunchKernelGGL(adamw_prepare_kernel, dim3(1), dim3(1), 0, s, step, grad_scale, found_inf, lr, beta1, beta2, weight_decay, coef);
    E3_CHECK_HIP(hipGetLastError());
    long long chunk0 = 0;
    for (int t0 = 0; t0 < n_tensors; t0 += OPT_MAX_T) {
        AdamWArgs a;
        a.nt = n_tensors - t0 < OPT_MAX_T ? n_tensors - t0 : OPT_MAX_T;
        int c = 0;
        for (int t = 0; t < a.nt; ++t) {
            E3_REQUIRE(numels[t0 + t] >= 0 && numels[t0 + t] < (1ll << 31), E3_ERR_UNSUPPORTED, "parameter tensor too large");
            a.p[t] = (float*)params[t0 + t]; a.g[t] = (const float*)grads[t0 + t];
            a.cstart[t] = c; a.numel[t] = (int)numels[t0 + t];
            c += (int)((numels[t0 + t] + OPT_CHUNK - 1) / OPT_CHUNK);
        }
        a.cstart[a.nt] = c;
        a.chunk0 = chunk0; a.m = exp_avg; a.v = exp_avg_sq; a.coef = coef;
        a.beta2 = (float)beta2; a.w1 = (float)(1.0 - beta1); a.w2 = (float)(1.0 - beta2); a.eps = (float)eps;
        if (c > 0) {
            if (bf16) hipLaunchKernelGGL(adamw_kernel<true>, dim3(c), dim3(256), 0, s, a);
            else hipLaunchKernelGGL(adamw_kernel<false>, dim3(c), dim3(256), 0, s, a);
        }
        E3_CHECK_HIP(hipGetLastError());
        chunk0 += c;
    }
    return E3_OK;
}
