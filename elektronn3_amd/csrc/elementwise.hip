// HBM-bound kernels of the U-Net hot path: weight packing, BatchNorm statistics/finalise/apply, ReLU,
// ceil-mode max-pool, their backward, and layout helpers.  All activations are fp32 NDHWC.
//
//   BatchNorm3d (train / eval)  unet.py:77-105   (torch: eps 1e-5, momentum 0.1, biased var for y, unbiased for running_var)
//   ReLU                        unet.py:183-186
//   MaxPool3d(k=2|(1,2,2), ceil_mode=True)  unet.py:67-74,225-230
//
// Each of these is a pure streaming pass: float4 (4 channels) per lane, consecutive lanes walk consecutive
// channel quads and then consecutive voxels, so every wave issues full 1 KiB coalesced requests.
#include "kernels.h"

namespace {

constexpr int EW_BLOCK = 256;
constexpr int EW_MAX_GRID = 256 * 8;   // ~8 resident workgroups per CU, grid-stride beyond that

// tensors above this size are read with non-temporal loads (they cannot stay in the 256 MB Infinity Cache anyway)
inline size_t ew_nt_bytes() {
    static const size_t v = []() { const char* e = getenv("E3_EW_NT_MB"); return (size_t)(e ? atol(e) : 200) << 20; }();
    return v;
}

inline int ew_grid(size_t items) {
    size_t g = (items + EW_BLOCK - 1) / EW_BLOCK;
    if (g > (size_t)EW_MAX_GRID) g = EW_MAX_GRID;
    if (g == 0) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------ weight packing
__global__ void pack_weights_kernel(int mode, const float* __restrict__ w, float* __restrict__ out,
                                    int Cout, int Cin, int T, int NPad) {
    // out index space: [G][Tin][NPad][K]
    size_t total;
    int K, G = 1, Tin = T;
    if (mode == PACK_CONV_FWD) { K = Cin; }
    else if (mode == PACK_CONV_DGRAD) { K = Cout; }
    else if (mode == PACK_UP_FWD) { K = Cin; Tin = 1; }
    else { K = Cout; G = T; Tin = 1; }
    total = (size_t)G * Tin * NPad * K;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = i % K; size_t r = i / K;
        const int n = r % NPad; r /= NPad;
        const int t = r % Tin; const int g = r / Tin;
        float v = 0.f;
        if (mode == PACK_CONV_FWD) {            // w[co=n][ci=k][t]
            if (n < Cout) v = w[((size_t)n * Cin + k) * T + t];
        } else if (mode == PACK_CONV_DGRAD) {   // column n = ci, k = co, flipped tap
            if (n < Cin) v = w[((size_t)k * Cin + n) * T + (T - 1 - t)];
        } else if (mode == PACK_UP_FWD) {       // w[ci=k][co][tap], column n = tap*Cout + co
            if (n < T * Cout) { const int tap = n / Cout, co = n % Cout; v = w[((size_t)k * Cout + co) * T + tap]; }
        } else {                                // PACK_UP_DGRAD: gather tap g, column n = ci, k = co
            if (n < Cin) v = w[((size_t)n * Cout + k) * T + g];
        }
        out[i] = v;
    }
}

// ------------------------------------------------------------------ BN statistics: coalesced pre-merge of many records
// in [parts][C][3] -> out [BN_PRERED][C][3].  Block b merges records [b*chunk, (b+1)*chunk): thread = (channel c, group g)
// Chan-merges its strided share (adjacent threads read adjacent channels of one record row: coalesced), then the groups of
// a channel are merged through LDS.  (One block per channel, as in bn_finalize_kernel, reads every 384-B record row
// 32 times from L2 when there are thousands of records.)
__global__ __launch_bounds__(1024) void bn_premerge_kernel(const float* __restrict__ in, int parts, int C, float* __restrict__ out) {
    __shared__ float sh[1024][3];
    const int chunk = (parts + gridDim.x - 1) / gridDim.x;
    const int p0 = blockIdx.x * chunk, p1 = p0 + chunk < parts ? p0 + chunk : parts;
    const int groups = 1024 / C;                       // C <= 1024
    const int c = threadIdx.x % C, g = threadIdx.x / C;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    if (g < groups)
        for (int p = p0 + g; p < p1; p += groups) {
            const float* r = in + ((size_t)p * C + c) * 3;
            welford_merge(n, mean, m2, r[0], r[1], r[2]);
        }
    sh[threadIdx.x][0] = n; sh[threadIdx.x][1] = mean; sh[threadIdx.x][2] = m2;
    __syncthreads();
    if (threadIdx.x < C) {
        for (int k = 1; k < groups; ++k) welford_merge(n, mean, m2, sh[k * C + c][0], sh[k * C + c][1], sh[k * C + c][2]);
        float* o = out + ((size_t)blockIdx.x * C + c) * 3;
        o[0] = n; o[1] = mean; o[2] = m2;
    }
}

// ------------------------------------------------------------------ BN finalise (one 1024-thread workgroup per channel)
// Merges the per-tile (count, mean, M2) records of a channel in fp64, two division-free passes over the (L2-resident)
// records:  N = sum n_i,  mu = sum n_i mean_i / N;  M2 = sum [ M2_i + n_i (mean_i - mu)^2 ].  Fixed order: deterministic.
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const BnFinalizeArgs a) {
    // 256 threads for up to 256 records, 1024 beyond (launcher).  The kernel sits between a conv and the pass that applies its BatchNorm: all of its
    // 5 - 7 us are latency on the step's critical path, 17 times per step.  Round 6: the thread's first record stays in registers for the second
    // reduction (one trip to memory instead of two), and a reduction costs ONE barrier: every thread adds the wave sums itself, in wave order.
    __shared__ double sh[2][16][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int c = blockIdx.x;
    auto block_sum2 = [&](double& x, double& y, int which) {          // sums x and y over the workgroup, result in every thread
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { x += __shfl_xor(x, off); y += __shfl_xor(y, off); }
        if (lane == 0) { sh[which][wave][0] = x; sh[which][wave][1] = y; }
        __syncthreads();
        double sx = 0.0, sy = 0.0;
        for (int w = 0; w < nwaves; ++w) { sx += sh[which][w][0]; sy += sh[which][w][1]; }
        x = sx; y = sy;
    };
    // group = 1: BatchNorm (statistics of channel c).  group > 1: GroupNorm -- the records of all `group` channels of c's group are
    // merged (nn.GroupNorm normalises over (C/G, D, H, W) of one sample; the caller passes one sample's records)
    const int gs = a.group > 1 ? a.group : 1;
    const int c_first = (c / gs) * gs;
    const int items = a.parts * gs;
    auto record = [&](int i) { return a.stats + ((size_t)(i / gs) * a.C + c_first + i % gs) * 3; };
    double nb0 = 0.0, mean0 = 0.0, m20 = 0.0;                          // this thread's first record
    if ((int)threadIdx.x < items) { const float* r = record(threadIdx.x); nb0 = r[0]; mean0 = r[1]; m20 = r[2]; }
    double n = nb0, s1 = nb0 * mean0;
    for (int i = threadIdx.x + nthreads; i < items; i += nthreads) {
        const float* r = record(i);
        const double nb = r[0];
        n += nb; s1 += nb * (double)r[1];
    }
    block_sum2(n, s1, 0);
    const double mean = n > 0.0 ? s1 / n : 0.0;
    const double d0 = mean0 - mean;
    double m2 = nb0 > 0.0 ? m20 + nb0 * d0 * d0 : 0.0, unused = 0.0;
    for (int i = threadIdx.x + nthreads; i < items; i += nthreads) {
        const float* r = record(i);
        const double nb = r[0], d = (double)r[1] - mean;
        m2 += nb > 0.0 ? (double)r[2] + nb * d * d : 0.0;
    }
    block_sum2(m2, unused, 1);
    if (threadIdx.x == 0) {
        const double var = n > 0.0 ? m2 / n : 0.0;
        const double invstd = 1.0 / sqrt(var + (double)a.eps);
        const double g = a.gamma ? (double)a.gamma[c] : 1.0, b = a.beta ? (double)a.beta[c] : 0.0;
        a.mean[c] = (float)mean;
        a.invstd[c] = (float)invstd;
        a.scale[c] = (float)(g * invstd);
        a.shift[c] = (float)(b - mean * g * invstd);
        if (a.running_mean) a.running_mean[c] = (float)((1.0 - a.momentum) * a.running_mean[c] + a.momentum * mean);
        if (a.running_var) {
            const double unb = n > 1.0 ? m2 / (n - 1.0) : var;
            a.running_var[c] = (float)((1.0 - a.momentum) * a.running_var[c] + a.momentum * unb);
        }
    }
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                               const float* conv_bias, float eps, float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double invstd = 1.0 / sqrt((double)rv[c] + (double)eps);
    const double s = (gamma ? (double)gamma[c] : 1.0) * invstd;
    scale[c] = (float)s;
    shift[c] = (float)((beta ? (double)beta[c] : 0.0) + ((conv_bias ? (double)conv_bias[c] : 0.0) - (double)rm[c]) * s);
}

// every conv unit's epilogue constants in one launch (eval mode folds ~17 BatchNorms per forward; one 5-us launch each otherwise).
// gamma == nullptr marks a unit without a norm (nn.Identity): scale = 1, shift = conv bias.
struct FoldMultiArgs {
    const float* gamma[FOLD_MAX_JOBS]; const float* beta[FOLD_MAX_JOBS]; const float* rm[FOLD_MAX_JOBS]; const float* rv[FOLD_MAX_JOBS];
    const float* bias[FOLD_MAX_JOBS]; float* scale[FOLD_MAX_JOBS]; float* shift[FOLD_MAX_JOBS];
    int C[FOLD_MAX_JOBS];
    int n; float eps;
};
__global__ void bn_fold_multi_kernel(const FoldMultiArgs a) {
    const int j = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.n || c >= a.C[j]) return;
    const double bias = a.bias[j] ? (double)a.bias[j][c] : 0.0;
    if (!a.gamma[j]) { a.scale[j][c] = 1.f; a.shift[j][c] = (float)bias; return; }
    const double invstd = 1.0 / sqrt((double)a.rv[j][c] + (double)a.eps);          // same expressions as bn_fold_kernel
    const double s = (double)a.gamma[j][c] * invstd;
    a.scale[j][c] = (float)s;
    a.shift[j][c] = (float)((double)a.beta[j][c] + (bias - (double)a.rm[j][c]) * s);
}

// ------------------------------------------------------------------ BN apply + ReLU (+ max-pool)
__global__ void bn_relu_apply_kernel(const float* __restrict__ x, int x_ldc, const float* __restrict__ scale,
                                     const float* __restrict__ shift, float* __restrict__ a, int a_ldc,
                                     size_t voxels, int C, size_t nt_bytes, ActArg act) {
    const float slope = act.get();
    const int Q = C >> 2;
    const size_t total = voxels * Q;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const bool big = voxels * (size_t)C * 4 > nt_bytes;   // larger than the Infinity Cache: stream past it
    // 4 independent 16-byte loads in flight per lane (memory-level parallelism is what an HBM-bound pass needs)
    for (size_t i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i0 < total; i0 += 4 * stride) {
        f32x4 xv[4]; size_t off[4]; int qq[4]; bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = i0 + u * stride;
            ok[u] = i < total;
            const size_t v = i / Q; qq[u] = (int)(i - v * Q);
            off[u] = v;
            if (big) xv[u] = ok[u] ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + v * x_ldc + 4 * qq[u])) : f32x4{0.f, 0.f, 0.f, 0.f};
            else xv[u] = ok[u] ? *reinterpret_cast<const f32x4*>(x + v * x_ldc + 4 * qq[u]) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + 4 * qq[u]);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + 4 * qq[u]);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = act_fwd(__builtin_fmaf(xv[u][e], sc[e], sh[e]), act_slope_at(act, slope, (unsigned)(off[u] * C + 4 * qq[u] + e)));
            if (ok[u]) *reinterpret_cast<f32x4*>(a + off[u] * a_ldc + 4 * qq[u]) = o;
        }
    }
}

// one lane = one pooling window x 4 channels: applies BN+ReLU to the (up to) kd*2*2 voxels of the window,
// writes them to `a` (the skip connection, possibly a concat-buffer half) and their max to `pooled`.
// APPLY=false: `x` already holds activations (plain max-pool).
template <bool APPLY>
__global__ void bn_relu_pool_kernel(const float* __restrict__ x, int x_ldc, const float* __restrict__ scale,
                                    const float* __restrict__ shift, float* __restrict__ a, int a_ldc,
                                    float* __restrict__ pooled, int kd, int N, int D, int H, int W, int C, ActArg act) {
    const float slope = act.get();
    const int Q = C >> 2;
    const int Dp = (D + kd - 1) / kd, Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;
    const size_t total = (size_t)N * Dp * Hp * Wp * Q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = i % Q; size_t r = i / Q;
        const int pw = r % Wp; r /= Wp; const int ph = r % Hp; r /= Hp; const int pd = r % Dp; const int n = r / Dp;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (APPLY) { sc = *reinterpret_cast<const f32x4*>(scale + 4 * q); sh = *reinterpret_cast<const f32x4*>(shift + 4 * q); }
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int dz = 0; dz < kd; ++dz) {
            const int d = pd * kd + dz; if (d >= D) break;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int h = ph * 2 + dy; if (h >= H) break;
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int w = pw * 2 + dx; if (w >= W) break;
                    const size_t v = (((size_t)n * D + d) * H + h) * W + w;
                    f32x4 o = *reinterpret_cast<const f32x4*>(x + v * x_ldc + 4 * q);
                    if (APPLY) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = act_fwd(__builtin_fmaf(o[e], sc[e], sh[e]), act_slope_at(act, slope, (unsigned)(v * C + 4 * q + e)));
                        *reinterpret_cast<f32x4*>(a + v * a_ldc + 4 * q) = o;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) best[e] = (o[e] > best[e] || o[e] != o[e]) ? o[e] : best[e];
                }
            }
        }
        *reinterpret_cast<f32x4*>(pooled + ((((size_t)n * Dp + pd) * Hp + ph) * Wp + pw) * C + 4 * q) = best;
    }
}

// ------------------------------------------------------------------ BN + ReLU (+ pool) backward
// dA(v) = g1(v) + [v is the first arg-max of its pooling window] * gpool(window)
// dz = dA * (z > 0),  z = x*scale + shift ;  xhat = (x - mean) * invstd
// pass 1 (REDUCE): per-channel sum dz, sum dz*xhat.   pass 2 (APPLY): dx = gamma*invstd*(dz - c1 - xhat*c2), sum dx.
// HL > 0 (head form only): the head gradient comes from the criterion (hl_* fields) for HL classes, and the REDUCE pass also takes the head's own
// weight / bias gradient sums -- a separate instantiation so that the plain forms keep their register count (64: eight waves per SIMD)
// items in flight per thread in the criterion form of the head's REDUCE pass: 2 (106 registers, four workgroups per CU = the 1024-workgroup grid in one
// residency round: 103 us at cfg 2); 4 needs 162 registers -- three per CU, 174 us (138 us on a 768-workgroup grid)
#ifndef E3_HEADRED_NU
#define E3_HEADRED_NU 2
#endif
template <bool POOL, bool APPLYPASS, bool HEAD = false, int HL = 0>
__global__ __launch_bounds__(256) void bn_bwd_kernel(const BnBwdArgs a) {
    __shared__ float red[3][256][4];
    const int Q = a.C >> 2;
    const int kd = a.kd;
    const int Dp = POOL ? (a.D + kd - 1) / kd : a.D, Hp = POOL ? (a.H + 1) >> 1 : a.H, Wp = POOL ? (a.W + 1) >> 1 : a.W;
    // work item = (window or voxel, channel quad); a block always works on channel quad (tid % Q') so that the
    // in-block reduction is a fixed pattern: items are laid out [unit][Q]
    const size_t units = (size_t)a.N * Dp * Hp * Wp;
    const size_t total = units * Q;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, s3 = s1;
    const float slope = a.act.get();
    const bool prelu = a.act.ptr != nullptr;       // REDUCE pass: also sum dA * min(z, 0) = d(activation)/d(slope) contributions
    // Only BT = floor(256/Q)*Q threads work, so that a thread's channel quad (tid % Q) never changes across its
    // grid-stride iterations and the in-block reduction below is a fixed pattern.
    const int BT = (256 / Q) * Q;
    const size_t stride = (size_t)gridDim.x * BT;
    if (!POOL) {
        // plain (voxel, channel quad) items: the thread's channel quad never changes (stride is a multiple of Q), so the
        // per-channel constants are loaded once, and 4 independent items are in flight per iteration (HBM-bound pass)
        const size_t i00 = (size_t)blockIdx.x * BT + threadIdx.x;
        const int q = (int)(i00 % Q);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + 4 * q);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + 4 * q);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(a.mean + 4 * q);
        const f32x4 is = *reinterpret_cast<const f32x4*>(a.invstd + 4 * q);
        f32x4 c1 = s1, c2 = s1, gi = s1, k1 = s1, k2 = s1;
        if (APPLYPASS) {
            c1 = *reinterpret_cast<const f32x4*>(a.coef + 4 * q);
            c2 = *reinterpret_cast<const f32x4*>(a.coef + a.C + 4 * q);
            k1 = *reinterpret_cast<const f32x4*>(a.coef + 2 * a.C + 4 * q);     // GroupNorm: group-mean terms that do not carry the
            k2 = *reinterpret_cast<const f32x4*>(a.coef + 3 * a.C + 4 * q);     // channel's gamma (zero for BatchNorm: x - 0 is exact)
            const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) gi[e] = gm[e] * is[e];
        }
        const size_t vstride = stride / Q;                  // voxels between consecutive items of this thread
        constexpr int NU = (HEAD && HL > 0 && !APPLYPASS) ? E3_HEADRED_NU : 4;     // items in flight per thread (8 for the HEAD form: 96 -> 168 us, measured)
        const bool big = units * (size_t)a.C * 4 > a.nt_bytes;
        // HEAD: the head's weights of this thread's channel quad stay in registers (the first four outputs: every usual head), and the sample
        // index of a voxel is carried along instead of divided out per item (the division cost more than the rest of the item)
        f32x4 hw[HL > 0 ? HL : 4];
        size_t hn = 0, hbase = 0;
        // criterion form of the head gradient: per-thread copies of the class constants (<= 4 classes)
        constexpr bool hloss = HEAD && HL > 0;
        constexpr int HC = HL > 0 ? HL : 1;
        float lw[HC], lgn[HC], lgd[HC], law = 0.f, lg = 1.f;
        // head gradients (REDUCE pass): dW of this thread's channel quad, db
        constexpr bool hgrad = HEAD && HL > 0 && !APPLYPASS;
        f32x4 dwacc[HC]; float dbacc[HC];
#pragma unroll
        for (int co = 0; co < HC; ++co) { dwacc[co] = f32x4{0.f, 0.f, 0.f, 0.f}; dbacc[co] = 0.f; lw[co] = 1.f; lgn[co] = 0.f; lgd[co] = 0.f; }
        if (HEAD) {
#pragma unroll
            for (int co = 0; co < (HL > 0 ? HL : 4); ++co) hw[co] = co < a.head_cout ? *reinterpret_cast<const f32x4*>(a.head_w + co * a.C + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            hn = (i00 / Q) / a.head_S; hbase = hn * a.head_S;
            if (hloss) {
#pragma unroll
                for (int co = 0; co < HC; ++co) { lw[co] = a.hl_cw ? a.hl_cw[co] : 1.f; lgn[co] = a.hl_coef[1 + co]; lgd[co] = a.hl_coef[1 + HC + co]; }
                law = a.hl_coef[0]; lg = a.hl_gout ? a.hl_gout[0] : 1.f;
            }
        }
        float gys[hloss ? NU : 1][HC];         // the head's logits gradient of the items in flight (kept for dW / db)
        for (size_t v0 = i00 / Q; threadIdx.x < BT && v0 < units; v0 += NU * vstride) {
            f32x4 xv[NU], g[NU]; bool ok[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const size_t v = v0 + u * vstride;
                ok[u] = v < units;
                // streaming tensors larger than the 256 MB Infinity Cache: non-temporal loads (measured 5.2 -> 6.2 TB/s on the
                // forward apply); smaller ones were just written by the previous kernel and still sit on-die
                if (HEAD) {
                    xv[u] = ok[u] ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.x + v * a.x_ldc + 4 * q)) : f32x4{0.f, 0.f, 0.f, 0.f};
                    g[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (hloss) {
#pragma unroll
                        for (int co = 0; co < HC; ++co) gys[u][co] = 0.f;
                    }
                    if (ok[u]) {
                        while (v >= hbase + a.head_S) { hbase += a.head_S; ++hn; }       // (v only grows along a thread's items)
                        const size_t n = hn, sp = v - hbase;
                        if (hloss) {
                            // dL/dlogits of this voxel: the expressions of ce_dice_bwd_kernel (loss.hip) on the logits the forward wrote
                            float z[HC], pr[HC], m = -3.4e38f;
#pragma unroll
                            for (int co = 0; co < HC; ++co) { z[co] = a.hl_logits[(n * HC + co) * a.head_S + sp]; m = fmaxf(m, z[co]); }
                            float sum = 0.f;
#pragma unroll
                            for (int co = 0; co < HC; ++co) { pr[co] = __expf(z[co] - m); sum += pr[co]; }
                            const float inv = 1.f / sum;
                            const int t = (int)a.hl_target[n * a.head_S + sp];
                            float G[HC], dot = 0.f, wt = 0.f;
#pragma unroll
                            for (int co = 0; co < HC; ++co) {
                                pr[co] *= inv;
                                const bool is = t == co;
                                G[co] = lgn[co] - (is ? lgd[co] : 0.f);
                                dot += pr[co] * G[co];
                                wt += is ? lw[co] : 0.f;
                            }
#pragma unroll
                            for (int co = 0; co < HC; ++co) {
                                gys[u][co] = lg * (law * wt * (pr[co] - (t == co ? 1.f : 0.f)) + pr[co] * (G[co] - dot));
#pragma unroll
                                for (int e = 0; e < 4; ++e) g[u][e] = __builtin_fmaf(gys[u][co], hw[co][e], g[u][e]);      // same fma order over co as conv_final_bwd_kernel
                            }
                        } else {
#pragma unroll
                            for (int co = 0; co < 4; ++co)                    // same fma order over co as conv_final_bwd_kernel
                                if (co < a.head_cout) {
                                    const float gy = a.head_dy[(n * a.head_cout + co) * a.head_S + sp];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) g[u][e] = __builtin_fmaf(gy, hw[co][e], g[u][e]);
                                }
                            for (int co = 4; co < a.head_cout; ++co) {
                                const float gy = a.head_dy[(n * a.head_cout + co) * a.head_S + sp];
                                const f32x4 wv = *reinterpret_cast<const f32x4*>(a.head_w + co * a.C + 4 * q);
#pragma unroll
                                for (int e = 0; e < 4; ++e) g[u][e] = __builtin_fmaf(gy, wv[e], g[u][e]);
                            }
                        }
                    }
                } else if (big) {
                    xv[u] = ok[u] ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.x + v * a.x_ldc + 4 * q)) : f32x4{0.f, 0.f, 0.f, 0.f};
                    g[u] = ok[u] ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.g1 + v * a.g1_ldc + 4 * q)) : f32x4{0.f, 0.f, 0.f, 0.f};
                } else {
                    xv[u] = ok[u] ? *reinterpret_cast<const f32x4*>(a.x + v * a.x_ldc + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
                    g[u] = ok[u] ? *reinterpret_cast<const f32x4*>(a.g1 + v * a.g1_ldc + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float z = __builtin_fmaf(xv[u][e], sc[e], sh[e]);   // same expression as the forward apply
                    const float sl_e = act_slope_at(a.act, slope, (unsigned)((v0 + u * vstride) * a.C + 4 * q + e));
                    const float dz = act_bwd(z, g[u][e], sl_e);
                    if (hgrad && ok[u]) {       // head weight gradient: the activation the head saw (same expression as its prologue)
                        const float av = act_fwd(z, sl_e);
#pragma unroll
                        for (int co = 0; co < HC; ++co) dwacc[co][e] = __builtin_fmaf(gys[u][co], av, dwacc[co][e]);
                    }
                    if (!APPLYPASS && prelu) s3[e] += g[u][e] * fminf(z, 0.f);
                    const float xh = (xv[u][e] - mu[e]) * is[e];
                    if (APPLYPASS) { o[e] = ok[u] ? gi[e] * (dz - c1[e] - xh * c2[e]) - (k1[e] + xh * k2[e]) : 0.f; s3[e] += o[e]; }
                    else { s1[e] += dz; s2[e] += dz * xh; }
                }
                if (APPLYPASS && ok[u]) *reinterpret_cast<f32x4*>(a.dx_chunk ? a.dx + (size_t)(q >> 1) * a.dx_chunk + (v0 + u * vstride) * 8 + 4 * (q & 1) : a.dx + (v0 + u * vstride) * a.dx_ldc + 4 * q) = o;
                if (hgrad && ok[u] && q == 0) {
#pragma unroll
                    for (int co = 0; co < HC; ++co) dbacc[co] += gys[u][co];
                }
            }
        }
        if (hgrad) {        // (uniform) block partials of the head's gradients, one output row at a time: the layout / order of conv_final_bwd_kernel
            const int pstride = HC * a.C + HC;
            const int tid = threadIdx.x;
#pragma unroll
            for (int co = 0; co < HC; ++co) {
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 4; ++e) red[0][tid][e] = tid < BT ? dwacc[co][e] : 0.f;
                red[1][tid][0] = (tid < BT && q == 0) ? dbacc[co] : 0.f;
                __syncthreads();
                for (int t = tid; t < Q * 4; t += 256) {
                    const int e = t & 3, qq = t >> 2;
                    float acc = 0.f;
                    for (int k = qq; k < BT; k += Q) acc += red[0][k][e];
                    a.head_part[(size_t)blockIdx.x * pstride + co * a.C + 4 * qq + e] = acc;
                }
                if (tid == 0) {
                    float acc = 0.f;
                    for (int k = 0; k < BT; k += Q) acc += red[1][k][0];
                    a.head_part[(size_t)blockIdx.x * pstride + HC * a.C + co] = acc;
                }
            }
            __syncthreads();
        }
    }
    for (size_t i = (size_t)blockIdx.x * BT + threadIdx.x; POOL && threadIdx.x < BT && i < total; i += stride) {
        const int q = i % Q; size_t r = i / Q;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(a.mean + 4 * q);
        const f32x4 is = *reinterpret_cast<const f32x4*>(a.invstd + 4 * q);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + 4 * q);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + 4 * q);
        f32x4 c1 = {0.f, 0.f, 0.f, 0.f}, c2 = c1, gi = c1, k1 = c1, k2 = c1;
        if (APPLYPASS) {
            c1 = *reinterpret_cast<const f32x4*>(a.coef + 4 * q);
            c2 = *reinterpret_cast<const f32x4*>(a.coef + a.C + 4 * q);
            k1 = *reinterpret_cast<const f32x4*>(a.coef + 2 * a.C + 4 * q);
            k2 = *reinterpret_cast<const f32x4*>(a.coef + 3 * a.C + 4 * q);
            const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) gi[e] = gm[e] * is[e];
        }
        {
            const int pw = r % Wp; r /= Wp; const int ph = r % Hp; r /= Hp; const int pd = r % Dp; const int n = r / Dp;
            const size_t pidx = ((((size_t)n * Dp + pd) * Hp + ph) * Wp + pw) * a.C + 4 * q;
            const f32x4 gp = *reinterpret_cast<const f32x4*>(a.gpool + pidx);
            const f32x4 pm = *reinterpret_cast<const f32x4*>(a.pooled + pidx);
            bool taken[4] = {false, false, false, false};
            for (int dz_ = 0; dz_ < kd; ++dz_) {
                const int d = pd * kd + dz_; if (d >= a.D) break;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const int h = ph * 2 + dy; if (h >= a.H) break;
#pragma unroll
                    for (int dx_ = 0; dx_ < 2; ++dx_) {
                        const int w = pw * 2 + dx_; if (w >= a.W) break;
                        const size_t v = (((size_t)n * a.D + d) * a.H + h) * a.W + w;
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(a.x + v * a.x_ldc + 4 * q);
                        // the activation is recomputed, not re-read (a quarter of this pass' HBM traffic): the same expression as the
                        // forward apply, so it is bit-identical to the stored tensor the pooled maxima were taken from
                        f32x4 av;
                        f32x4 zv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { zv[e] = __builtin_fmaf(xv[e], sc[e], sh[e]); av[e] = act_fwd(zv[e], act_slope_at(a.act, slope, (unsigned)(v * a.C + 4 * q + e))); }
                        f32x4 g = {0.f, 0.f, 0.f, 0.f};
                        if (a.g1) g = *reinterpret_cast<const f32x4*>(a.g1 + v * a.g1_ldc + 4 * q);
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float dA = g[e];
                            if (!taken[e] && av[e] == pm[e]) { dA += gp[e]; taken[e] = true; }   // first arg-max wins (ATen)
                            const float dz = act_bwd(zv[e], dA, act_slope_at(a.act, slope, (unsigned)(v * a.C + 4 * q + e)));
                            if (!APPLYPASS && prelu) s3[e] += dA * fminf(zv[e], 0.f);
                            const float xh = (xv[e] - mu[e]) * is[e];
                            if (APPLYPASS) { o[e] = gi[e] * (dz - c1[e] - xh * c2[e]) - (k1[e] + xh * k2[e]); s3[e] += o[e]; }
                            else { s1[e] += dz; s2[e] += dz * xh; }
                        }
                        if (APPLYPASS) *reinterpret_cast<f32x4*>(a.dx_chunk ? a.dx + (size_t)(q >> 1) * a.dx_chunk + v * 8 + 4 * (q & 1) : a.dx + v * a.dx_ldc + 4 * q) = o;
                    }
                }
            }
        }
    }
    // ---- block reduction: threads with equal (tid % Q) own the same channel quad
    const int tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][tid][e] = APPLYPASS ? s3[e] : s1[e]; red[1][tid][e] = s2[e]; red[2][tid][e] = s3[e]; }
    __syncthreads();
    const int rows = APPLYPASS ? 1 : (prelu ? 3 : 2);
    for (int t = tid; t < Q * 4 * rows; t += 256) {
        const int e = t & 3, q = (t >> 2) % Q, which = (t >> 2) / Q;
        float acc = 0.f;
        for (int k = q; k < BT; k += Q) acc += red[which][k][e];
        a.part[((size_t)blockIdx.x * 3 + (APPLYPASS ? 2 : which)) * a.C + 4 * q + e] = acc;   // part layout [parts][3][C]
    }
}

// sums part[p][row][c] over p: one WAVE per channel (lanes stride over the partial rows, butterfly in double)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// one WORKGROUP per channel: 256 lanes stride over the partial rows (4 independent loads each for 1024 rows instead of a
// 16-deep chain per wave), butterfly in double, 4 wave results combined through LDS in a fixed order
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int parts, int C, float inv_n,
                                                              float* dgamma, float* dbeta, float* coef) {
    __shared__ double red[2][4];
    const int c = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double s1 = 0.0, s2 = 0.0;
    for (int p = tid; p < parts; p += 256) { s1 += part[((size_t)p * 3 + 0) * C + c]; s2 += part[((size_t)p * 3 + 1) * C + c]; }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) { red[0][wv] = s1; red[1][wv] = s2; }
    __syncthreads();
    if (tid == 0) {
        s1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        s2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        if (dbeta) dbeta[c] = (float)s1;
        if (dgamma) dgamma[c] = (float)s2;
        coef[c] = (float)(s1 * inv_n);
        coef[C + c] = (float)(s2 * inv_n);
        coef[2 * C + c] = 0.f;
        coef[3 * C + c] = 0.f;
    }
}

// GroupNorm backward coefficients (after bn_bwd_finalize wrote dbeta = sum dz, dgamma = sum dz*xhat per channel):
//   dx_c = invstd_g * [ gamma_c dz - mean_g(gamma dz) - xhat mean_g(gamma dz xhat) ]   (means over the group's channels and voxels)
// in the apply pass' form  gi*(dz - c1 - xhat*c2) - (k1 + xhat*k2):  c1 = c2 = 0,  k1 = invstd_g * sum_{c' in g} gamma_c' dbeta_c' * inv_n / gs.
__global__ void gn_bwd_coef_kernel(const float* __restrict__ dgamma, const float* __restrict__ dbeta, const float* __restrict__ gamma,
                                   const float* __restrict__ invstd, int C, int gs, float inv_n, float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int c0 = (c / gs) * gs;
    double m1 = 0.0, m2 = 0.0;
    for (int k = 0; k < gs; ++k) { m1 += (double)gamma[c0 + k] * (double)dbeta[c0 + k]; m2 += (double)gamma[c0 + k] * (double)dgamma[c0 + k]; }
    const double f = (double)invstd[c] * (double)inv_n / (double)gs;
    coef[c] = 0.f; coef[C + c] = 0.f;
    coef[2 * C + c] = (float)(m1 * f);
    coef[3 * C + c] = (float)(m2 * f);
}

__global__ void colsum_finalize_kernel(const float* __restrict__ part, int parts, int part_stride, int offset, int C, float* out) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= C) return;
    double s = 0.0;
    for (int p = lane; p < parts; p += 64) s += part[(size_t)p * part_stride + offset + c];
    s = wave_sum(s);
    if (lane == 0) out[c] = (float)s;
}

__global__ void fill_kernel(float* __restrict__ p, float v, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
// merge_mode='add' (unet.py:400-401): out[v][c] = a[v][c] + b[v][c] over channel views
__global__ void add_views_kernel(const float* __restrict__ a, int a_ldc, const float* __restrict__ b, int b_ldc,
                                 float* __restrict__ out, int out_ldc, size_t vox, int C) {
    const int Q = C >> 2;
    const size_t total = vox * Q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % Q); const size_t v = i / Q;
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + v * a_ldc + 4 * q);
        const f32x4 y = *reinterpret_cast<const f32x4*>(b + v * b_ldc + 4 * q);
        *reinterpret_cast<f32x4*>(out + v * out_ldc + 4 * q) = f32x4{x[0] + y[0], x[1] + y[1], x[2] + y[2], x[3] + y[3]};
    }
}
// conv -> nn.Identity -> ReLU (normalization='none' / full_norm=False): the "folded norm" of the conv epilogue is y = relu(1*acc + bias)
__global__ void bias_fold_kernel(const float* __restrict__ bias, float* __restrict__ scale, float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { scale[c] = 1.f; shift[c] = bias ? bias[c] : 0.f; }
}

// many column sums in one launch (the conv-bias gradients of all layers: 17 tiny latency-bound launches per step otherwise)
struct ColsumMultiArgs {
    const float* part[COLSUM_MAX_JOBS]; float* out[COLSUM_MAX_JOBS];
    int parts[COLSUM_MAX_JOBS], stride[COLSUM_MAX_JOBS], offset[COLSUM_MAX_JOBS], C[COLSUM_MAX_JOBS];
    int bstart[COLSUM_MAX_JOBS + 1];
    int n;
};
__global__ __launch_bounds__(256) void colsum_multi_kernel(const ColsumMultiArgs a) {
    const int b = blockIdx.x;
    int lo = 0, hi = a.n;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.bstart[mid] <= b) lo = mid; else hi = mid; }
    const int j = lo;
    const int lane = threadIdx.x & 63;
    const int c = (b - a.bstart[j]) * 4 + (threadIdx.x >> 6);
    if (c >= a.C[j]) return;
    const float* __restrict__ part = a.part[j];
    const int stride = a.stride[j], off = a.offset[j] + c, parts = a.parts[j];
    double s = 0.0;
    for (int p = lane; p < parts; p += 64) s += part[(size_t)p * stride + off];   // same order as colsum_finalize_kernel
    s = wave_sum(s);
    if (lane == 0) a.out[j][c] = (float)s;
}

// ------------------------------------------------------------------ up_mode='resizeconv_nearest' (ResizeConv, unet.py:411-449)
// nn.Upsample(scale_factor=(sd,2,2), mode='nearest'): out[n, d, h, w] = x[n, d / sd, h >> 1, w >> 1]
__global__ void upsample_nearest_kernel(const float* __restrict__ x, int x_ldc, float* __restrict__ out, int C, int N, int Di, int Hi, int Wi, int sd) {
    const int Q = C >> 2, Do = Di * sd, Ho = Hi * 2, Wo = Wi * 2;
    const size_t total = (size_t)N * Do * Ho * Wo * Q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % Q); size_t r = i / Q;
        const int w = (int)(r % Wo); r /= Wo; const int h = (int)(r % Ho); r /= Ho; const int d = (int)(r % Do); const int n = (int)(r / Do);
        const size_t vi = (((size_t)n * Di + d / sd) * Hi + (h >> 1)) * Wi + (w >> 1);
        *reinterpret_cast<f32x4*>(out + (i / Q) * C + 4 * q) = *reinterpret_cast<const f32x4*>(x + vi * x_ldc + 4 * q);
    }
}
// its backward: dx[n, d, h, w] = sum of g over the (sd x 2 x 2) block it was copied to, in a fixed order
__global__ void downsample_sum_kernel(const float* __restrict__ g, float* __restrict__ dx, int dx_ldc, int C, int N, int Di, int Hi, int Wi, int sd) {
    const int Q = C >> 2, Ho = Hi * 2, Wo = Wi * 2;
    const size_t total = (size_t)N * Di * Hi * Wi * Q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % Q); size_t r = i / Q;
        const int w = (int)(r % Wi); r /= Wi; const int h = (int)(r % Hi); r /= Hi; const int d = (int)(r % Di); const int n = (int)(r / Di);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < sd; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const size_t vo = (((size_t)n * (Di * sd) + d * sd + a) * Ho + 2 * h + b) * Wo + 2 * w + c;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(g + vo * C + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += v[e];
                }
        *reinterpret_cast<f32x4*>(dx + (i / Q) * dx_ldc + 4 * q) = acc;
    }
}
// nn.Upsample(scale_factor=(sd,2,2), mode='trilinear' / 'bilinear', align_corners=False): per axis with scale 2 the source coordinate of
// output o is max((o + 0.5)/2 - 0.5, 0); i0 = floor, i1 = min(i0 + 1, n - 1), weights (1 - l, l): out[0] = x[0], out[2i+1] = .75 x[i] + .25 x[i+1],
// out[2i] = .25 x[i-1] + .75 x[i], clamped at the high end.  An axis with scale 1 is the identity.
__device__ __forceinline__ void lin_src(int o, int n, int scale, int& i0, int& i1, float& l) {
    if (scale == 1) { i0 = i1 = o; l = 0.f; return; }
    const float src = fmaxf(((float)o + 0.5f) * 0.5f - 0.5f, 0.f);
    i0 = (int)src; i1 = i0 + 1 < n ? i0 + 1 : n - 1; l = src - (float)i0;
}
__global__ void upsample_linear_kernel(const float* __restrict__ x, int x_ldc, float* __restrict__ out, int C, int N, int Di, int Hi, int Wi, int sd) {
    const int Q = C >> 2, Do = Di * sd, Ho = Hi * 2, Wo = Wi * 2;
    const size_t total = (size_t)N * Do * Ho * Wo * Q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % Q); size_t r = i / Q;
        const int w = (int)(r % Wo); r /= Wo; const int h = (int)(r % Ho); r /= Ho; const int d = (int)(r % Do); const int n = (int)(r / Do);
        int d0, d1, h0, h1, w0, w1; float ld, lh, lw;
        lin_src(d, Di, sd, d0, d1, ld); lin_src(h, Hi, 2, h0, h1, lh); lin_src(w, Wi, 2, w0, w1, lw);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float wt = (a ? ld : 1.f - ld) * (b ? lh : 1.f - lh) * (c ? lw : 1.f - lw);
                    const size_t vi = (((size_t)n * Di + (a ? d1 : d0)) * Hi + (b ? h1 : h0)) * Wi + (c ? w1 : w0);
                    const f32x4 v = *reinterpret_cast<const f32x4*>(x + vi * x_ldc + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(wt, v[e], acc[e]);
                }
        *reinterpret_cast<f32x4*>(out + (i / Q) * C + 4 * q) = acc;
    }
}
// its backward as a GATHER (fixed order, no atomics): input voxel i is referenced only by outputs 2i-1 .. 2i+2 of an axis
__global__ void downsample_linear_kernel(const float* __restrict__ g, float* __restrict__ dx, int dx_ldc, int C, int N, int Di, int Hi, int Wi, int sd) {
    const int Q = C >> 2, Do = Di * sd, Ho = Hi * 2, Wo = Wi * 2;
    const size_t total = (size_t)N * Di * Hi * Wi * Q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % Q); size_t r = i / Q;
        const int w = (int)(r % Wi); r /= Wi; const int h = (int)(r % Hi); r /= Hi; const int d = (int)(r % Di); const int n = (int)(r / Di);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int nd = sd == 1 ? 1 : 4;
        for (int a = 0; a < nd; ++a) {
            const int od = sd == 1 ? d : 2 * d - 1 + a;
            if (od < 0 || od >= Do) continue;
            int i0, i1; float l; lin_src(od, Di, sd, i0, i1, l);
            const float wd = (i0 == d ? 1.f - l : 0.f) + (i1 == d ? l : 0.f);
            if (wd == 0.f) continue;
            for (int b = 0; b < 4; ++b) {
                const int oh = 2 * h - 1 + b;
                if (oh < 0 || oh >= Ho) continue;
                lin_src(oh, Hi, 2, i0, i1, l);
                const float wh = (i0 == h ? 1.f - l : 0.f) + (i1 == h ? l : 0.f);
                if (wh == 0.f) continue;
                for (int c = 0; c < 4; ++c) {
                    const int ow = 2 * w - 1 + c;
                    if (ow < 0 || ow >= Wo) continue;
                    lin_src(ow, Wi, 2, i0, i1, l);
                    const float ww = (i0 == w ? 1.f - l : 0.f) + (i1 == w ? l : 0.f);
                    if (ww == 0.f) continue;
                    const size_t vo = (((size_t)n * Do + od) * Ho + oh) * Wo + ow;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(g + vo * C + 4 * q);
                    const float wt = wd * wh * ww;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(wt, v[e], acc[e]);
                }
            }
        }
        *reinterpret_cast<f32x4*>(dx + (i / Q) * dx_ldc + 4 * q) = acc;
    }
}
// ResizeConv(kernel_size=1) ('resizeconv_*1'): the 1x1x1 weights as the centre tap of an otherwise zero T-tap kernel, so that the
// layer runs on the 3x3x3 / 1x3x3 conv kernels (correct, 27x / 9x more multiplies than a pointwise GEMM needs), and back
__global__ void embed_center_tap_kernel(const float* __restrict__ w1, float* __restrict__ wT, size_t pairs, int T) {
    const size_t total = pairs * T;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        wT[i] = (int)(i % T) == T / 2 ? w1[i / T] : 0.f;
}
__global__ void extract_center_tap_kernel(const float* __restrict__ gT, float* __restrict__ g1, size_t pairs, int T) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) g1[i] = gT[i * T + T / 2];
}
// autocrop of the up-convolved tensor (unet.py:289-299: one voxel at the high end where the skip has an odd size), together with the
// BatchNorm statistics of the CROPPED tensor: src [N, Ds, Hs, Ws, C] -> dst [N, Dd, Hd, Wd, C] (leading box) + one (count, mean, M2)
// record per workgroup and channel.  Same fixed-pattern block reduction as bn_bwd_kernel.
__global__ __launch_bounds__(256) void crop_stats_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int N,
                                                         int Ds, int Hs, int Ws, int Dd, int Hd, int Wd, float* __restrict__ stats,
                                                         int od, int oh, int ow) {   // (od, oh, ow): position of the box inside src
    __shared__ float red[256][3][4];
    const int Q = C >> 2;
    const int BT = (256 / Q) * Q;
    const size_t units = (size_t)N * Dd * Hd * Wd, total = units * Q;
    const size_t stride = (size_t)gridDim.x * BT;
    f32x4 cn = {0.f, 0.f, 0.f, 0.f}, mean = cn, m2 = cn;
    for (size_t i = (size_t)blockIdx.x * BT + threadIdx.x; threadIdx.x < BT && i < total; i += stride) {
        const int q = (int)(i % Q); size_t r = i / Q;
        const int w = (int)(r % Wd); r /= Wd; const int h = (int)(r % Hd); r /= Hd; const int d = (int)(r % Dd); const int n = (int)(r / Dd);
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + ((((size_t)n * Ds + d + od) * Hs + h + oh) * Ws + w + ow) * C + 4 * q);
        *reinterpret_cast<f32x4*>(dst + (i / Q) * C + 4 * q) = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { cn[e] += 1.f; const float dl = v[e] - mean[e]; mean[e] += dl / cn[e]; m2[e] += dl * (v[e] - mean[e]); }
    }
    const int tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[tid][0][e] = cn[e]; red[tid][1][e] = mean[e]; red[tid][2][e] = m2[e]; }
    __syncthreads();
    for (int t = tid; t < Q * 4; t += 256) {
        const int e = t & 3, q = t >> 2;
        float n = 0.f, mu = 0.f, s = 0.f;
        for (int k = q; k < BT; k += Q) welford_merge(n, mu, s, red[k][0][e], red[k][1][e], red[k][2][e]);
        float* o = stats + ((size_t)blockIdx.x * C + 4 * q + e) * 3;
        o[0] = n; o[1] = mu; o[2] = s;
    }
}
// Reduction of a split-K convolution (conv_wino_splitk): dst[u][c] = src_0[u][c] + src_1[u][c] (+ ...) + bias[c] in that fixed order, plus
// (optionally) the BatchNorm statistics of the result -- one (count, mean, M2) record per workgroup and channel, as crop_stats_kernel.
template <bool STATS>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ src, int nsrc, size_t src_stride, const float* __restrict__ bias,
                                                            float* __restrict__ dst, int dst_ldc, int C, size_t units, float* __restrict__ stats) {
    __shared__ float red[STATS ? 256 : 1][3][4];
    const int Q = C >> 2;
    const int BT = (256 / Q) * Q;
    const size_t total = units * Q, stride = (size_t)gridDim.x * BT;
    f32x4 cn = {0.f, 0.f, 0.f, 0.f}, mean = cn, m2 = cn;
    for (size_t i = (size_t)blockIdx.x * BT + threadIdx.x; threadIdx.x < BT && i < total; i += stride) {
        const int q = (int)(i % Q); const size_t u = i / Q;
        f32x4 v = *reinterpret_cast<const f32x4*>(src + u * C + 4 * q);
        for (int k = 1; k < nsrc; ++k) v += *reinterpret_cast<const f32x4*>(src + k * src_stride + u * C + 4 * q);
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + 4 * q);
        *reinterpret_cast<f32x4*>(dst + u * dst_ldc + 4 * q) = v;
        if (STATS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { cn[e] += 1.f; const float dl = v[e] - mean[e]; mean[e] += dl / cn[e]; m2[e] += dl * (v[e] - mean[e]); }
        }
    }
    if (STATS) {
        const int tid = threadIdx.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[tid][0][e] = cn[e]; red[tid][1][e] = mean[e]; red[tid][2][e] = m2[e]; }
        __syncthreads();
        for (int t = tid; t < Q * 4; t += 256) {
            const int e = t & 3, q = t >> 2;
            float n = 0.f, mu = 0.f, sq = 0.f;
            for (int k = q; k < BT; k += Q) welford_merge(n, mu, sq, red[k][0][e], red[k][1][e], red[k][2][e]);
            float* o = stats + ((size_t)blockIdx.x * C + 4 * q + e) * 3;
            o[0] = n; o[1] = mu; o[2] = sq;
        }
    }
}
// centre crop of the skip connection in conv_mode='valid' (autocrop, unet.py:300-325): dst view (ldc) [N, Dd, Hd, Wd] = src box at (od, oh, ow)
__global__ void crop_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int dst_ldc, int C, int N, int Ds, int Hs, int Ws,
                                 int Dd, int Hd, int Wd, int od, int oh, int ow) {
    const int Q = C >> 2;
    const size_t total = (size_t)N * Dd * Hd * Wd * Q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % Q); size_t r = i / Q;
        const int w = (int)(r % Wd); r /= Wd; const int h = (int)(r % Hd); r /= Hd; const int d = (int)(r % Dd); const int n = (int)(r / Dd);
        *reinterpret_cast<f32x4*>(dst + (i / Q) * dst_ldc + 4 * q) =
            *reinterpret_cast<const f32x4*>(src + ((((size_t)n * Ds + d + od) * Hs + h + oh) * Ws + w + ow) * C + 4 * q);
    }
}
// backward of the crop: dst [N, Dd, Hd, Wd, C] = src inside the leading box [Ds, Hs, Ws], zero elsewhere
__global__ void pad_box_kernel(const float* __restrict__ src, int src_ldc, float* __restrict__ dst, int C, int N, int Ds, int Hs, int Ws, int Dd, int Hd, int Wd,
                               int od, int oh, int ow) {   // src sits at (od, oh, ow) inside dst
    const int Q = C >> 2;
    const size_t total = (size_t)N * Dd * Hd * Wd * Q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % Q); size_t r = i / Q;
        const int w = (int)(r % Wd); r /= Wd; const int h = (int)(r % Hd); r /= Hd; const int d = (int)(r % Dd); const int n = (int)(r / Dd);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int sd_ = d - od, sh_ = h - oh, sw_ = w - ow;
        if (sd_ >= 0 && sd_ < Ds && sh_ >= 0 && sh_ < Hs && sw_ >= 0 && sw_ < Ws)
            v = *reinterpret_cast<const f32x4*>(src + ((((size_t)n * Ds + sd_) * Hs + sh_) * Ws + sw_) * src_ldc + 4 * q);
        *reinterpret_cast<f32x4*>(dst + (i / Q) * C + 4 * q) = v;
    }
}

// ------------------------------------------------------------------ layout helpers (module boundary only)
__global__ void ncdhw_to_ndhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, size_t S) {
    const size_t total = (size_t)N * C * S;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % C; size_t r = i / C; const size_t sp = r % S; const int n = r / S;
        dst[i] = src[((size_t)n * C + c) * S + sp];
    }
}
__global__ void ndhwc_to_ncdhw_kernel(const float* __restrict__ src, int ldc, float* __restrict__ dst, int N, int C, size_t S) {
    const size_t total = (size_t)N * C * S;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t sp = i % S; size_t r = i / S; const int c = r % C; const int n = r / C;
        dst[i] = src[((size_t)n * S + sp) * ldc + c];
    }
}

}  // namespace

int launch_pack_weights(PackMode mode, const float* w, float* out, int Cout, int Cin, int T, int NPad, hipStream_t s) {
    size_t total;
    if (mode == PACK_CONV_FWD) total = (size_t)T * NPad * Cin;
    else if (mode == PACK_CONV_DGRAD) total = (size_t)T * NPad * Cout;
    else if (mode == PACK_UP_FWD) total = (size_t)NPad * Cin;
    else total = (size_t)T * NPad * Cout;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, s, (int)mode, w, out, Cout, Cin, T, NPad);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_bn_finalize(BnFinalizeArgs a, hipStream_t s) {
    // (up to 1024 records -- the persistent kernels' one-per-workgroup records, the 512-brick levels -- the finaliser's 1024 threads take one record each:
    // the pre-merge launch would cost more than it saves; thousands of per-brick records are merged to BN_PRERED coalesced partials first)
    if (a.scratch && a.parts > 1024 && a.C <= 1024) {
        hipLaunchKernelGGL(bn_premerge_kernel, dim3(BN_PRERED), dim3(1024), 0, s, a.stats, a.parts, a.C, a.scratch);
        a.stats = a.scratch; a.parts = BN_PRERED;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(a.C), dim3(a.parts * (a.group > 1 ? a.group : 1) <= 256 ? 256 : 1024), 0, s, a);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_fill(float* p, float v, size_t n, hipStream_t s) {
    if (n == 0) return E3_OK;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0, s, p, v, n);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_add_views(const float* a, int a_ldc, const float* b, int b_ldc, float* out, int out_ldc, size_t vox, int C, hipStream_t s) {
    E3_REQUIRE(C % 4 == 0 && a_ldc % 4 == 0 && b_ldc % 4 == 0 && out_ldc % 4 == 0, E3_ERR_UNSUPPORTED, "channels must be a multiple of 4");
    hipLaunchKernelGGL(add_views_kernel, dim3(ew_grid(vox * (C / 4))), dim3(EW_BLOCK), 0, s, a, a_ldc, b, b_ldc, out, out_ldc, vox, C);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_bias_fold(const float* conv_bias, float* scale, float* shift, int C, hipStream_t s) {
    hipLaunchKernelGGL(bias_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, conv_bias, scale, shift, C);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

// frozen statistics (a forward in eval mode that a backward will follow): the normalisation uses the RUNNING statistics, written in the
// four per-channel vectors the train-mode passes read (mean, invstd, scale, shift); nothing is updated
__global__ void bn_frozen_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                 float* mean, float* invstd, float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double is = 1.0 / sqrt((double)rv[c] + (double)eps), g = gamma ? (double)gamma[c] : 1.0, b = beta ? (double)beta[c] : 0.0;
    mean[c] = rm[c]; invstd[c] = (float)is; scale[c] = (float)(g * is); shift[c] = (float)(b - (double)rm[c] * g * is);
}
int launch_bn_frozen(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                     float* mean, float* invstd, float* scale, float* shift, int C, hipStream_t s) {
    hipLaunchKernelGGL(bn_frozen_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, gamma, beta, rm, rv, eps, mean, invstd, scale, shift, C);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_bn_fold(const float* gamma, const float* beta, const float* rm, const float* rv, const float* conv_bias,
                   float eps, float* scale, float* shift, int C, hipStream_t s) {
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, gamma, beta, rm, rv, conv_bias, eps, scale, shift, C);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_fold_multi(const FoldJob* jobs, int njobs, float eps, hipStream_t s) {
    for (int j0 = 0; j0 < njobs; j0 += FOLD_MAX_JOBS) {
        FoldMultiArgs a;
        a.n = njobs - j0 < FOLD_MAX_JOBS ? njobs - j0 : FOLD_MAX_JOBS; a.eps = eps;
        int cmax = 1;
        for (int j = 0; j < a.n; ++j) {
            const FoldJob& q = jobs[j0 + j];
            a.gamma[j] = q.gamma; a.beta[j] = q.beta; a.rm[j] = q.rm; a.rv[j] = q.rv; a.bias[j] = q.bias; a.scale[j] = q.scale; a.shift[j] = q.shift; a.C[j] = q.C;
            if (q.C > cmax) cmax = q.C;
        }
        hipLaunchKernelGGL(bn_fold_multi_kernel, dim3(cdiv(cmax, 256), a.n), dim3(256), 0, s, a);
        E3_CHECK_HIP(hipGetLastError());
    }
    return E3_OK;
}

int launch_bn_relu_apply(const float* x, int x_ldc, const float* scale, const float* shift, float* a, int a_ldc,
                         float* pooled, int kd, int N, int D, int H, int W, int C, hipStream_t s, ActArg slope) {
    E3_REQUIRE(C % 4 == 0 && x_ldc % 4 == 0 && a_ldc % 4 == 0, E3_ERR_UNSUPPORTED, "channels must be a multiple of 4");
    if (!pooled) {
        const size_t vox = (size_t)N * D * H * W;
        hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(ew_grid(vox * (C / 4))), dim3(EW_BLOCK), 0, s, x, x_ldc, scale, shift, a, a_ldc, vox, C, ew_nt_bytes(), slope);
    } else {
        const size_t items = (size_t)N * cdiv(D, kd) * cdiv(H, 2) * cdiv(W, 2) * (C / 4);
        hipLaunchKernelGGL(bn_relu_pool_kernel<true>, dim3(ew_grid(items)), dim3(EW_BLOCK), 0, s, x, x_ldc, scale, shift, a, a_ldc, pooled, kd, N, D, H, W, C, slope);
    }
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_maxpool(const float* a, int a_ldc, float* pooled, int kd, int N, int D, int H, int W, int C, hipStream_t s) {
    E3_REQUIRE(C % 4 == 0 && a_ldc % 4 == 0, E3_ERR_UNSUPPORTED, "channels must be a multiple of 4");
    const size_t items = (size_t)N * cdiv(D, kd) * cdiv(H, 2) * cdiv(W, 2) * (C / 4);
    hipLaunchKernelGGL(bn_relu_pool_kernel<false>, dim3(ew_grid(items)), dim3(EW_BLOCK), 0, s, a, a_ldc,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, 0, pooled, kd, N, D, H, W, C, ActArg(0.f));
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

// number of partial rows == grid size of the backward kernels
int bn_bwd_parts(size_t voxels, int C) {
    const int Q = C / 4;
    size_t g = (voxels * Q + 255) / 256;
    if (g > 1024) g = 1024;
    if (g == 0) g = 1;
    (void)Q;
    return (int)g;
}

static int bn_bwd_launch(BnBwdArgs a, bool apply, hipStream_t s) {
    a.nt_bytes = ew_nt_bytes();
    const int Q = a.C / 4;
    E3_REQUIRE(a.C % 4 == 0, E3_ERR_UNSUPPORTED, "channels must be a multiple of 4");
    E3_REQUIRE(Q >= 1 && Q <= 256, E3_ERR_UNSUPPORTED, "BN backward supports up to 1024 channels");
    const bool pool = a.gpool != nullptr;
    const dim3 grid(a.parts), block(256);
    if (pool) {
        if (apply) hipLaunchKernelGGL((bn_bwd_kernel<true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((bn_bwd_kernel<true, false>), grid, block, 0, s, a);
    } else if (a.g1 == nullptr) {
        E3_REQUIRE((a.head_dy || a.hl_logits) && a.head_w && a.head_cout > 0 && a.head_S > 0, E3_ERR_INVALID, "BN backward without an incoming gradient");
        E3_REQUIRE(!a.hl_logits || (a.head_cout >= 2 && a.head_cout <= 4 && a.hl_target && a.hl_coef), E3_ERR_UNSUPPORTED,
                   "BN backward, head form with the criterion: 2..4 classes");
        E3_REQUIRE(!a.head_part || a.hl_logits, E3_ERR_INVALID, "BN backward: the head's gradients come with the criterion form only");
        if (a.hl_logits) {      // criterion form (+ the head's own gradients in the REDUCE pass)
            E3_REQUIRE(apply || a.head_part, E3_ERR_INVALID, "BN backward, criterion form: the reduce pass takes the head's gradients");
            switch (a.head_cout * 2 + (apply ? 1 : 0)) {
                case 4: hipLaunchKernelGGL((bn_bwd_kernel<false, false, true, 2>), grid, block, 0, s, a); break;
                case 5: hipLaunchKernelGGL((bn_bwd_kernel<false, true, true, 2>), grid, block, 0, s, a); break;
                case 6: hipLaunchKernelGGL((bn_bwd_kernel<false, false, true, 3>), grid, block, 0, s, a); break;
                case 7: hipLaunchKernelGGL((bn_bwd_kernel<false, true, true, 3>), grid, block, 0, s, a); break;
                case 8: hipLaunchKernelGGL((bn_bwd_kernel<false, false, true, 4>), grid, block, 0, s, a); break;
                default: hipLaunchKernelGGL((bn_bwd_kernel<false, true, true, 4>), grid, block, 0, s, a); break;
            }
        } else if (apply) hipLaunchKernelGGL((bn_bwd_kernel<false, true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((bn_bwd_kernel<false, false, true>), grid, block, 0, s, a);
    } else {
        if (apply) hipLaunchKernelGGL((bn_bwd_kernel<false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((bn_bwd_kernel<false, false>), grid, block, 0, s, a);
    }
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
int launch_bn_bwd_reduce(BnBwdArgs a, hipStream_t s) { return bn_bwd_launch(a, false, s); }
int launch_bn_bwd_apply(BnBwdArgs a, hipStream_t s) { return bn_bwd_launch(a, true, s); }

int launch_bn_bwd_finalize(const float* part, int parts, int C, float inv_n, float* dgamma, float* dbeta, float* coef, hipStream_t s) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, s, part, parts, C, inv_n, dgamma, dbeta, coef);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

// PReLU slope gradient: sum of row 2 of the REDUCE pass' partials over rows and channels, fixed order
__global__ __launch_bounds__(256) void prelu_dslope_kernel(const float* __restrict__ part, int parts, int C, float* __restrict__ tmp, float* __restrict__ dslope) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c = wv; c < C; c += 4) {                    // one wave per channel at a time
        double s = 0.0;
        for (int p = lane; p < parts; p += 64) s += part[((size_t)p * 3 + 2) * C + c];
        s = wave_sum(s);
        if (lane == 0) tmp[c] = (float)s;
    }
    __syncthreads();
    double t = 0.0;
    for (int c = threadIdx.x; c < C; c += 256) t += (double)tmp[c];
    t = wave_sum(t);
    if (lane == 0) red[wv] = t;
    __syncthreads();
    if (threadIdx.x == 0) dslope[0] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}
int launch_prelu_dslope(const float* part, int parts, int C, float* tmp, float* dslope, hipStream_t s) {
    hipLaunchKernelGGL(prelu_dslope_kernel, dim3(1), dim3(256), 0, s, part, parts, C, tmp, dslope);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_gn_bwd_coef(const float* dgamma, const float* dbeta, const float* gamma, const float* invstd, int C, int group, float inv_n,
                       float* coef, hipStream_t s) {
    E3_REQUIRE(group >= 1 && C % group == 0, E3_ERR_INVALID, "group size must divide the channel count");
    hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, dgamma, dbeta, gamma, invstd, C, group, inv_n, coef);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_colsum_finalize(const float* part, int parts, int part_stride, int offset, int C, float* out, hipStream_t s) {
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, s, part, parts, part_stride, offset, C, out);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_colsum_multi(const ColsumJob* jobs, int njobs, hipStream_t s) {
    for (int j0 = 0; j0 < njobs; j0 += COLSUM_MAX_JOBS) {
        ColsumMultiArgs a;
        a.n = njobs - j0 < COLSUM_MAX_JOBS ? njobs - j0 : COLSUM_MAX_JOBS;
        int b = 0;
        for (int j = 0; j < a.n; ++j) {
            const ColsumJob& q = jobs[j0 + j];
            a.part[j] = q.part; a.out[j] = q.out; a.parts[j] = q.parts; a.stride[j] = q.stride; a.offset[j] = q.offset; a.C[j] = q.C;
            a.bstart[j] = b;
            b += cdiv(q.C, 4);
        }
        a.bstart[a.n] = b;
        if (b > 0) hipLaunchKernelGGL(colsum_multi_kernel, dim3(b), dim3(256), 0, s, a);
        E3_CHECK_HIP(hipGetLastError());
    }
    return E3_OK;
}

int launch_upsample_nearest(const float* x, int x_ldc, float* out, int C, int N, int Di, int Hi, int Wi, int sd, hipStream_t s, int linear) {
    E3_REQUIRE(C % 4 == 0 && x_ldc % 4 == 0, E3_ERR_UNSUPPORTED, "channels must be a multiple of 4");
    if (linear) {
        hipLaunchKernelGGL(upsample_linear_kernel, dim3(ew_grid((size_t)N * Di * sd * Hi * 2 * Wi * 2 * (C / 4))), dim3(EW_BLOCK), 0, s, x, x_ldc, out, C, N, Di, Hi, Wi, sd);
        E3_CHECK_HIP(hipGetLastError());
        return E3_OK;
    }
    hipLaunchKernelGGL(upsample_nearest_kernel, dim3(ew_grid((size_t)N * Di * sd * Hi * 2 * Wi * 2 * (C / 4))), dim3(EW_BLOCK), 0, s, x, x_ldc, out, C, N, Di, Hi, Wi, sd);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
int launch_downsample_sum(const float* g, float* dx, int dx_ldc, int C, int N, int Di, int Hi, int Wi, int sd, hipStream_t s, int linear) {
    E3_REQUIRE(C % 4 == 0 && dx_ldc % 4 == 0, E3_ERR_UNSUPPORTED, "channels must be a multiple of 4");
    if (linear) {
        hipLaunchKernelGGL(downsample_linear_kernel, dim3(ew_grid((size_t)N * Di * Hi * Wi * (C / 4))), dim3(EW_BLOCK), 0, s, g, dx, dx_ldc, C, N, Di, Hi, Wi, sd);
        E3_CHECK_HIP(hipGetLastError());
        return E3_OK;
    }
    hipLaunchKernelGGL(downsample_sum_kernel, dim3(ew_grid((size_t)N * Di * Hi * Wi * (C / 4))), dim3(EW_BLOCK), 0, s, g, dx, dx_ldc, C, N, Di, Hi, Wi, sd);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
int launch_embed_center_tap(const float* w1, float* wT, size_t pairs, int T, hipStream_t s) {
    hipLaunchKernelGGL(embed_center_tap_kernel, dim3(ew_grid(pairs * T)), dim3(EW_BLOCK), 0, s, w1, wT, pairs, T);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
int launch_extract_center_tap(const float* gT, float* g1, size_t pairs, int T, hipStream_t s) {
    hipLaunchKernelGGL(extract_center_tap_kernel, dim3(ew_grid(pairs)), dim3(EW_BLOCK), 0, s, gT, g1, pairs, T);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
int crop_stats_parts(size_t voxels, int C) { return bn_bwd_parts(voxels, C); }
int launch_crop_stats(const float* src, float* dst, int C, int N, int Ds, int Hs, int Ws, int Dd, int Hd, int Wd, float* stats, hipStream_t s,
                      int od, int oh, int ow) {
    E3_REQUIRE(C % 4 == 0 && C <= 1024, E3_ERR_UNSUPPORTED, "channels must be a multiple of 4, at most 1024");
    E3_REQUIRE(od >= 0 && oh >= 0 && ow >= 0 && Dd + od <= Ds && Hd + oh <= Hs && Wd + ow <= Ws, E3_ERR_INVALID, "crop box outside the source");
    const int parts = crop_stats_parts((size_t)N * Dd * Hd * Wd, C);
    hipLaunchKernelGGL(crop_stats_kernel, dim3(parts), dim3(256), 0, s, src, dst, C, N, Ds, Hs, Ws, Dd, Hd, Wd, stats, od, oh, ow);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
int launch_splitk_reduce(const float* src, int nsrc, size_t src_stride, const float* bias, float* dst, int dst_ldc, int C, size_t units,
                         float* stats, hipStream_t s) {
    E3_REQUIRE(C % 4 == 0 && C <= 1024 && dst_ldc % 4 == 0 && nsrc >= 1, E3_ERR_UNSUPPORTED, "channels must be a multiple of 4, at most 1024");
    const int parts = crop_stats_parts(units, C);
    if (stats) hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(parts), dim3(256), 0, s, src, nsrc, src_stride, bias, dst, dst_ldc, C, units, stats);
    else hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(parts), dim3(256), 0, s, src, nsrc, src_stride, bias, dst, dst_ldc, C, units, stats);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
int launch_crop_copy(const float* src, float* dst, int dst_ldc, int C, int N, int Ds, int Hs, int Ws, int Dd, int Hd, int Wd, int od, int oh, int ow, hipStream_t s) {
    E3_REQUIRE(C % 4 == 0 && dst_ldc % 4 == 0, E3_ERR_UNSUPPORTED, "channels must be a multiple of 4");
    E3_REQUIRE(od >= 0 && oh >= 0 && ow >= 0 && Dd + od <= Ds && Hd + oh <= Hs && Wd + ow <= Ws, E3_ERR_INVALID, "crop box outside the source");
    hipLaunchKernelGGL(crop_copy_kernel, dim3(ew_grid((size_t)N * Dd * Hd * Wd * (C / 4))), dim3(EW_BLOCK), 0, s, src, dst, dst_ldc, C, N, Ds, Hs, Ws, Dd, Hd, Wd, od, oh, ow);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
int launch_pad_box(const float* src, float* dst, int C, int N, int Ds, int Hs, int Ws, int Dd, int Hd, int Wd, hipStream_t s, int od, int oh, int ow, int src_ldc) {
    E3_REQUIRE(C % 4 == 0, E3_ERR_UNSUPPORTED, "channels must be a multiple of 4");
    if (src_ldc == 0) src_ldc = C;
    hipLaunchKernelGGL(pad_box_kernel, dim3(ew_grid((size_t)N * Dd * Hd * Wd * (C / 4))), dim3(EW_BLOCK), 0, s, src, src_ldc, dst, C, N, Ds, Hs, Ws, Dd, Hd, Wd, od, oh, ow);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_ncdhw_to_ndhwc(const float* src, float* dst, int N, int C, size_t S, hipStream_t s) {
    hipLaunchKernelGGL(ncdhw_to_ndhwc_kernel, dim3(ew_grid((size_t)N * C * S)), dim3(EW_BLOCK), 0, s, src, dst, N, C, S);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
int launch_ndhwc_to_ncdhw(const float* src, int src_ldc, float* dst, int N, int C, size_t S, hipStream_t s) {
    hipLaunchKernelGGL(ndhwc_to_ncdhw_kernel, dim3(ew_grid((size_t)N * C * S)), dim3(EW_BLOCK), 0, s, src, src_ldc, dst, N, C, S);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
