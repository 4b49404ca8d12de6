// Weighted cross-entropy + Dice on device, fused over one read of the logits.
//
// Replaces the criterion of the reference's training example,
//   CombinedLoss([CrossEntropyLoss(weight=w), DiceLoss(apply_softmax=True, weight=w)], weight=(a, b))
// (elektronn3/modules/loss.py:19-49, 158-234; examples/train_unet_neurodata.py:294-296), which runs as ~20 ATen kernels
// with three full-resolution round trips of the logits.  Here:
//   forward  = one pass over (logits NCDHW, target int64): per-block partial sums of
//                A = sum_v w[t_v] * (-log p[t_v]),  Ws = sum_v w[t_v],  I_c = sum p_c [t=c],  P_c = sum p_c,  T_c = sum [t=c]
//              + a one-block finaliser in fp64 that writes the scalar loss and the coefficients of the backward;
//   backward = one pass that recomputes the softmax and writes dL/dlogits
//                dz_c = g * ( a/Ws * w[t] (p_c - [t=c])  +  p_c (G_c - sum_k p_k G_k) ),
//                G_k = (b/C) w_k ( num_k/den_k^2 - 2 [t=k]/den_k ),  num_k = 2 I_k + smooth,  den_k = P_k + T_k + smooth + eps.
// All reductions have a fixed order (no atomics): bit-reproducible.
#include "kernels.h"

namespace {

constexpr int LOSS_MAXC = 16;
constexpr int LOSS_BLOCKS = 1024;

template <int C>
__device__ __forceinline__ void softmax_c(const float* __restrict__ z, size_t cstride, float (&p)[C], float& lse) {
    float m = z[0];
    float v[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { v[c] = z[c * cstride]; m = fmaxf(m, v[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { p[c] = __expf(v[c] - m); s += p[c]; }
    const float inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < C; ++c) p[c] *= inv;
    lse = m + __logf(s);
}

// partial[block][2 + 3C]
template <int C>
__global__ __launch_bounds__(256) void ce_dice_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                          const float* __restrict__ w, int N, size_t vps, float* __restrict__ partial) {
    constexpr int NV = 2 + 3 * C;
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    float wc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) wc[c] = w ? w[c] : 1.f;
    const size_t total = (size_t)N * vps;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / vps, v = i - n * vps;
        const float* z = logits + n * C * vps + v;
        float p[C], lse;
        softmax_c<C>(z, vps, p, lse);
        const int t = (int)target[i];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const bool is = t == c;
            acc[2 + c] += is ? p[c] : 0.f;
            acc[2 + C + c] += p[c];
            acc[2 + 2 * C + c] += is ? 1.f : 0.f;
            if (is) { acc[0] += wc[c] * (lse - z[c * vps]); acc[1] += wc[c]; }
        }
    }
    __shared__ float red[4][NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) partial[(size_t)blockIdx.x * NV + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// Sums over the block partials (fixed order, fp64): one wave per value (16 waves take the NV <= 50 values in turn), 64 lanes stride
// over the blocks, fp64 butterfly.
__device__ __forceinline__ void ce_dice_reduce(const float* __restrict__ partial, int blocks, int NV, double* s) {
    const int lane = threadIdx.x & 63;
    for (int v = threadIdx.x >> 6; v < NV; v += (int)(blockDim.x >> 6)) {
        double t = 0.0;
        for (int k = lane; k < blocks; k += 64) t += (double)partial[(size_t)k * NV + v];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (lane == 0) s[v] = t;
    }
}

// out[0] = loss; coef: [0] = a / Ws, [1 + k] = (b/C) w_k num_k / den_k^2, [1 + C + k] = (b/C) w_k 2 / den_k
__device__ __forceinline__ void ce_dice_coefs(const double* s, int C, const float* __restrict__ w, float a, float b, float eps, float smooth,
                                              float* __restrict__ out, float* __restrict__ coef) {
    const double ce = s[1] > 0.0 ? s[0] / s[1] : 0.0;
    double dice = 0.0;
    for (int k = 0; k < C; ++k) {
        const double wk = w ? (double)w[k] : 1.0;
        const double num = 2.0 * s[2 + k] + smooth, den = s[2 + C + k] + s[2 + 2 * C + k] + smooth + eps;
        dice += wk * (1.0 - num / den);
        coef[1 + k] = (float)((double)b / C * wk * num / (den * den));
        coef[1 + C + k] = (float)((double)b / C * wk * 2.0 / den);
    }
    dice /= C;
    out[0] = (float)((double)a * ce + (double)b * dice);
    coef[0] = s[1] > 0.0 ? (float)((double)a / s[1]) : 0.f;
}

__global__ void ce_dice_finalize_kernel(const float* __restrict__ partial, int blocks, int C, const float* __restrict__ w,
                                        float a, float b, float eps, float smooth, float* __restrict__ out, float* __restrict__ coef) {
    __shared__ double s[2 + 3 * LOSS_MAXC];
    ce_dice_reduce(partial, blocks, 2 + 3 * C, s);
    __syncthreads();
    if (threadIdx.x == 0) ce_dice_coefs(s, C, w, a, b, eps, smooth, out, coef);
}

// The two halves of the finaliser as separate launches, for a criterion over a minibatch that is sharded over ranks: the 2 + 3C sums
// are what the ranks exchange (one all-reduce of <= 26 doubles) between them.
__global__ void ce_dice_sums_kernel(const float* __restrict__ partial, int blocks, int C, double* __restrict__ sums) {
    __shared__ double s[2 + 3 * LOSS_MAXC];
    ce_dice_reduce(partial, blocks, 2 + 3 * C, s);
    __syncthreads();
    if ((int)threadIdx.x < 2 + 3 * C) sums[threadIdx.x] = s[threadIdx.x];
}

__global__ void ce_dice_from_sums_kernel(const double* __restrict__ sums, int C, const float* __restrict__ w,
                                         float a, float b, float eps, float smooth, float* __restrict__ out, float* __restrict__ coef) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ce_dice_coefs(sums, C, w, a, b, eps, smooth, out, coef);
}

template <int C>
__global__ __launch_bounds__(256) void ce_dice_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                          const float* __restrict__ w, const float* __restrict__ coef,
                                                          const float* __restrict__ gout, int N, size_t vps, float* __restrict__ dlogits) {
    float wc[C], gn[C], gd[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { wc[c] = w ? w[c] : 1.f; gn[c] = coef[1 + c]; gd[c] = coef[1 + C + c]; }
    const float aw = coef[0], g = gout ? gout[0] : 1.f;
    const size_t total = (size_t)N * vps;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / vps, v = i - n * vps;
        const float* z = logits + n * C * vps + v;
        float p[C], lse;
        softmax_c<C>(z, vps, p, lse);
        const int t = (int)target[i];
        float G[C], dot = 0.f, wt = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const bool is = t == c;
            G[c] = gn[c] - (is ? gd[c] : 0.f);
            dot += p[c] * G[c];
            wt += is ? wc[c] : 0.f;
        }
        float* dz = dlogits + n * C * vps + v;
#pragma unroll
        for (int c = 0; c < C; ++c) dz[c * vps] = g * (aw * wt * (p[c] - (t == c ? 1.f : 0.f)) + p[c] * (G[c] - dot));
    }
}

}  // namespace

static_assert(LOSS_BLOCKS <= CE_DICE_MAX_ROWS, "workspace rows");
size_t ce_dice_workspace_floats(int C) { return (size_t)CE_DICE_MAX_ROWS * (2 + 3 * C) + 1 + 2 * C; }

#define LOSS_DISPATCH(KERNEL, ...)                                                                                        \
    switch (C) {                                                                                                          \
        case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(LOSS_BLOCKS), dim3(256), 0, s, __VA_ARGS__); break;                    \
        case 3: hipLaunchKernelGGL(KERNEL<3>, dim3(LOSS_BLOCKS), dim3(256), 0, s, __VA_ARGS__); break;                    \
        case 4: hipLaunchKernelGGL(KERNEL<4>, dim3(LOSS_BLOCKS), dim3(256), 0, s, __VA_ARGS__); break;                    \
        case 5: hipLaunchKernelGGL(KERNEL<5>, dim3(LOSS_BLOCKS), dim3(256), 0, s, __VA_ARGS__); break;                    \
        case 6: hipLaunchKernelGGL(KERNEL<6>, dim3(LOSS_BLOCKS), dim3(256), 0, s, __VA_ARGS__); break;                    \
        case 7: hipLaunchKernelGGL(KERNEL<7>, dim3(LOSS_BLOCKS), dim3(256), 0, s, __VA_ARGS__); break;                    \
        case 8: hipLaunchKernelGGL(KERNEL<8>, dim3(LOSS_BLOCKS), dim3(256), 0, s, __VA_ARGS__); break;                    \
        case 9: hipLaunchKernelGGL(KERNEL<9>, dim3(LOSS_BLOCKS), dim3(256), 0, s, __VA_ARGS__); break;                   \
        case 10: hipLaunchKernelGGL(KERNEL<10>, dim3(LOSS_BLOCKS), dim3(256), 0, s, __VA_ARGS__); break;                   \
        case 11: hipLaunchKernelGGL(KERNEL<11>, dim3(LOSS_BLOCKS), dim3(256), 0, s, __VA_ARGS__); break;                   \
        case 12: hipLaunchKernelGGL(KERNEL<12>, dim3(LOSS_BLOCKS), dim3(256), 0, s, __VA_ARGS__); break;                   \
        case 13: hipLaunchKernelGGL(KERNEL<13>, dim3(LOSS_BLOCKS), dim3(256), 0, s, __VA_ARGS__); break;                   \
        case 14: hipLaunchKernelGGL(KERNEL<14>, dim3(LOSS_BLOCKS), dim3(256), 0, s, __VA_ARGS__); break;                   \
        case 15: hipLaunchKernelGGL(KERNEL<15>, dim3(LOSS_BLOCKS), dim3(256), 0, s, __VA_ARGS__); break;                   \
        case 16: hipLaunchKernelGGL(KERNEL<16>, dim3(LOSS_BLOCKS), dim3(256), 0, s, __VA_ARGS__); break;                   \
        default: e3_set_error("ce_dice: 2 <= C <= 16 classes supported"); return E3_ERR_UNSUPPORTED;                       \
    }

// workspace: [LOSS_BLOCKS][2+3C] partial sums, then the 1 + 2C backward coefficients (kept for the backward call)
int launch_ce_dice_fwd(const float* logits, const long long* target, const float* w, int C, int N, size_t vps, float a, float b,
                       float eps, float smooth, float* workspace, float* loss_out, hipStream_t s) {
    float* partial = workspace;
    float* coef = workspace + (size_t)CE_DICE_MAX_ROWS * (2 + 3 * C);
    LOSS_DISPATCH(ce_dice_fwd_kernel, logits, target, w, N, vps, partial)
    hipLaunchKernelGGL(ce_dice_finalize_kernel, dim3(1), dim3(1024), 0, s, partial, LOSS_BLOCKS, C, w, a, b, eps, smooth, loss_out, coef);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_ce_dice_finalize(const float* w, int C, int rows, float a, float b, float eps, float smooth, float* workspace, float* loss_out, hipStream_t s) {
    if (C < 2 || C > LOSS_MAXC || rows < 1 || rows > CE_DICE_MAX_ROWS) { e3_set_error("ce_dice_finalize: bad class / row count"); return E3_ERR_INVALID; }
    float* coef = workspace + (size_t)CE_DICE_MAX_ROWS * (2 + 3 * C);
    hipLaunchKernelGGL(ce_dice_finalize_kernel, dim3(1), dim3(1024), 0, s, workspace, rows, C, w, a, b, eps, smooth, loss_out, coef);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

// the 2 + 3C sums of `rows` partial rows already in the workspace (a fused head wrote them): the first half of a sharded criterion
int launch_ce_dice_sums_rows(int C, int rows, const float* workspace, double* sums, hipStream_t s) {
    if (C < 2 || C > LOSS_MAXC || rows < 1 || rows > CE_DICE_MAX_ROWS) { e3_set_error("ce_dice_sums: bad class / row count"); return E3_ERR_INVALID; }
    hipLaunchKernelGGL(ce_dice_sums_kernel, dim3(1), dim3(1024), 0, s, workspace, rows, C, sums);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_ce_dice_sums(const float* logits, const long long* target, const float* w, int C, int N, size_t vps, float* workspace,
                        double* sums, hipStream_t s) {
    float* partial = workspace;
    LOSS_DISPATCH(ce_dice_fwd_kernel, logits, target, w, N, vps, partial)
    hipLaunchKernelGGL(ce_dice_sums_kernel, dim3(1), dim3(1024), 0, s, partial, LOSS_BLOCKS, C, sums);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_ce_dice_from_sums(const double* sums, const float* w, int C, float a, float b, float eps, float smooth, float* workspace,
                             float* loss_out, hipStream_t s) {
    if (C < 2 || C > LOSS_MAXC) { e3_set_error("ce_dice: 2 <= C <= 16 classes supported"); return E3_ERR_UNSUPPORTED; }
    float* coef = workspace + (size_t)CE_DICE_MAX_ROWS * (2 + 3 * C);
    hipLaunchKernelGGL(ce_dice_from_sums_kernel, dim3(1), dim3(64), 0, s, sums, C, w, a, b, eps, smooth, loss_out, coef);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_ce_dice_bwd(const float* logits, const long long* target, const float* w, int C, int N, size_t vps,
                       const float* workspace, const float* gout, float* dlogits, hipStream_t s) {
    const float* coef = workspace + (size_t)CE_DICE_MAX_ROWS * (2 + 3 * C);
    LOSS_DISPATCH(ce_dice_bwd_kernel, logits, target, w, coef, gout, N, vps, dlogits)
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
