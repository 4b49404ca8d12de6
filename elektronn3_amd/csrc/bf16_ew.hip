// HBM-bound passes of the native bf16 path: BatchNorm apply + ReLU (+ MaxPool3d ceil_mode) forward and backward, the first conv
// (in_channels < 8) and the 1x1x1 head, for bf16 NDHWC tensors.  Same roles as the fp32 kernels of elementwise.hip / conv_small.hip
// (reference ops: nn.BatchNorm3d unet.py:77-105, nn.ReLU :183-186, nn.MaxPool3d(2, ceil_mode=True) :225-230, conv1() :178-180);
// a lane moves 8 channels = 16 bytes per access, all arithmetic is fp32, every stored tensor is rounded to bf16 once.
#include "bf16.h"

namespace {

constexpr int EW_BLOCK = 256;
constexpr int EW_MAX_GRID = 256 * 8;

struct f8 { float v[8]; };
__device__ __forceinline__ f8 ld8(const bf16_t* p) {
    const u16x8 r = *reinterpret_cast<const u16x8*>(p);
    f8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.v[e] = bf2f(r[e]);
    return o;
}
__device__ __forceinline__ f8 ld8nt(const bf16_t* p) {
    const u16x8 r = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(p));
    f8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.v[e] = bf2f(r[e]);
    return o;
}
__device__ __forceinline__ void st8(bf16_t* p, const f8& x) {
    u16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = f2bf(x.v[e]);
    *reinterpret_cast<u16x8*>(p) = r;
}
__device__ __forceinline__ f8 ldf8(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    return f8{{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}};
}

// ------------------------------------------------------------------ BN apply + ReLU
__global__ void bn_relu_apply_b16_kernel(const bf16_t* __restrict__ x, int x_ldc, const float* __restrict__ scale, const float* __restrict__ shift,
                                         bf16_t* __restrict__ a, int a_ldc, size_t voxels, int C) {
    const int Q = C >> 3;
    const size_t total = voxels * Q;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i0 < total; i0 += 4 * stride) {
        f8 xv[4]; size_t vv[4]; int qq[4]; bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = i0 + u * stride;
            ok[u] = i < total;
            const size_t v = ok[u] ? i / Q : 0; qq[u] = ok[u] ? (int)(i - v * Q) : 0; vv[u] = v;
            xv[u] = ld8nt(x + v * x_ldc + 8 * qq[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f8 sc = ldf8(scale + 8 * qq[u]), sh = ldf8(shift + 8 * qq[u]);
            f8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.v[e] = fmaxf(__builtin_fmaf(xv[u].v[e], sc.v[e], sh.v[e]), 0.f);
            if (ok[u]) st8(a + vv[u] * a_ldc + 8 * qq[u], o);
        }
    }
}

// one lane = one pooling window x 8 channels (APPLY = false: `x` already holds activations: plain max-pool)
template <bool APPLY>
__global__ void bn_relu_pool_b16_kernel(const bf16_t* __restrict__ x, int x_ldc, const float* __restrict__ scale, const float* __restrict__ shift,
                                        bf16_t* __restrict__ a, int a_ldc, bf16_t* __restrict__ pooled, int kd, int N, int D, int H, int W, int C) {
    const int Q = C >> 3;
    const int Dp = (D + kd - 1) / kd, Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;
    const size_t total = (size_t)N * Dp * Hp * Wp * Q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = i % Q; size_t r = i / Q;
        const int pw = r % Wp; r /= Wp; const int ph = r % Hp; r /= Hp; const int pd = r % Dp; const int n = r / Dp;
        f8 sc, sh;
        if (APPLY) { sc = ldf8(scale + 8 * q); sh = ldf8(shift + 8 * q); }
        f8 best;
#pragma unroll
        for (int e = 0; e < 8; ++e) best.v[e] = -INFINITY;
        for (int dz = 0; dz < kd; ++dz) {
            const int d = pd * kd + dz; if (d >= D) break;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int h = ph * 2 + dy; if (h >= H) break;
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int w = pw * 2 + dx; if (w >= W) break;
                    const size_t v = (((size_t)n * D + d) * H + h) * W + w;
                    f8 o = ld8(x + v * x_ldc + 8 * q);
                    if (APPLY) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o.v[e] = round_bf(fmaxf(__builtin_fmaf(o.v[e], sc.v[e], sh.v[e]), 0.f));
                        st8(a + v * a_ldc + 8 * q, o);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) best.v[e] = (o.v[e] > best.v[e] || o.v[e] != o.v[e]) ? o.v[e] : best.v[e];
                }
            }
        }
        st8(pooled + ((((size_t)n * Dp + pd) * Hp + ph) * Wp + pw) * C + 8 * q, best);
    }
}

// the same with a window's two w-columns on two lanes (lane ^ Q; Q a power of two <= 32): 2Q consecutive lanes read / write 2 voxels x C
// channels as one contiguous run (whole 128-byte lines at C = 32, see bn_bwd_b16_kernel); the pair's maxima meet through one shuffle
template <bool APPLY>
__global__ __launch_bounds__(256) void bn_relu_pool2_b16_kernel(const bf16_t* __restrict__ x, int x_ldc, const float* __restrict__ scale, const float* __restrict__ shift,
                                                                bf16_t* __restrict__ a, int a_ldc, bf16_t* __restrict__ pooled, int kd, int N, int D, int H, int W, int C) {
    const int Q = C >> 3;
    const int Dp = (D + kd - 1) / kd, Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;
    const size_t units = (size_t)N * Dp * Hp * Wp, total = units * 2 * Q;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (total + stride - 1) / stride * stride; i += stride) {     // (pairs iterate together)
        const bool uok = i < total;
        const int q = (int)(i % Q), dxl = (int)((i / Q) & 1);
        size_t r = uok ? i / (2 * Q) : 0;
        const int pw = r % Wp; r /= Wp; const int ph = r % Hp; r /= Hp; const int pd = r % Dp; const int n = r / Dp;
        const int w = pw * 2 + dxl;
        f8 sc, sh;
        if (APPLY) { sc = ldf8(scale + 8 * q); sh = ldf8(shift + 8 * q); }
        f8 o[4]; bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int d = pd * kd + (k >> 1), h = ph * 2 + (k & 1);
            ok[k] = uok && (k >> 1) < kd && d < D && h < H && w < W;
            const size_t v = ok[k] ? (((size_t)n * D + d) * H + h) * W + w : 0;
            o[k] = ld8(x + v * x_ldc + 8 * q);
        }
        f8 best;
#pragma unroll
        for (int e = 0; e < 8; ++e) best.v[e] = -INFINITY;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ok[k]) continue;
            const int d = pd * kd + (k >> 1), h = ph * 2 + (k & 1);
            const size_t v = (((size_t)n * D + d) * H + h) * W + w;
            if (APPLY) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[k].v[e] = round_bf(fmaxf(__builtin_fmaf(o[k].v[e], sc.v[e], sh.v[e]), 0.f));
                st8(a + v * a_ldc + 8 * q, o[k]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) best.v[e] = (o[k].v[e] > best.v[e] || o[k].v[e] != o[k].v[e]) ? o[k].v[e] : best.v[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float other = __shfl_xor(best.v[e], Q);           // the other column's maximum (-inf when that column is outside the tensor)
            best.v[e] = (other > best.v[e] || other != other) ? other : best.v[e];     // NaN propagates, as in nn.MaxPool3d
        }
        if (uok && dxl == 0) st8(pooled + ((((size_t)n * Dp + pd) * Hp + ph) * Wp + pw) * C + 8 * q, best);
    }
}

// ------------------------------------------------------------------ BN + ReLU (+ pool, + skip) backward
// dA(v) = g1(v) + [v is the first arg-max of its window] * gpool(window);  dz = dA * (z > 0), z = x*scale + shift
// REDUCE: per-channel sum dz, sum dz*xhat.  APPLY: dx = bf16(gamma*invstd*(dz - c1 - xhat*c2)), sum dx (conv-bias gradient).
// HL > 0 (head form): criterion variant for HL classes (BnBwdB16Args::hl_*), see elementwise.hip's bn_bwd_kernel
template <bool POOL, bool APPLYPASS, bool HEAD, int HL = 0>
__global__ __launch_bounds__(256) void bn_bwd_b16_kernel(const BnBwdB16Args a) {
    const bool g_pool_rows = a.pool_one_lane != 0;        // the one-lane-per-window form (channel counts the lane-pair form does not cover)
    __shared__ float red[2][256][8];
    const int Q = a.C >> 3;
    const int kd = a.kd;
    const int Dp = POOL ? (a.D + kd - 1) / kd : a.D, Hp = POOL ? (a.H + 1) >> 1 : a.H, Wp = POOL ? (a.W + 1) >> 1 : a.W;
    const size_t units = (size_t)a.N * Dp * Hp * Wp;
    const int BT = (256 / Q) * Q;             // active threads: a thread's channel octet never changes across its iterations
    const size_t stride = (size_t)gridDim.x * BT;
    const size_t i00 = (size_t)blockIdx.x * BT + threadIdx.x;
    const int q = (int)(i00 % Q);
    const bool active = threadIdx.x < BT;
    f8 s1, s2;
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1.v[e] = 0.f; s2.v[e] = 0.f; }
    const f8 sc = ldf8(a.scale + 8 * q), sh = ldf8(a.shift + 8 * q), mu = ldf8(a.mean + 8 * q), is = ldf8(a.invstd + 8 * q);
    f8 c1 = s1, c2 = s1, gi = s1;
    if (APPLYPASS) {
        c1 = ldf8(a.coef + 8 * q); c2 = ldf8(a.coef + a.C + 8 * q);
        const f8 gm = ldf8(a.gamma + 8 * q);
#pragma unroll
        for (int e = 0; e < 8; ++e) gi.v[e] = gm.v[e] * is.v[e];
    }
    const size_t vstride = stride / Q;
    if (!POOL) {
        constexpr int U = HEAD ? 2 : 4;          // independent (x, g) pairs in flight per lane
        constexpr bool hloss = HEAD && HL > 0, hgrad = hloss && !APPLYPASS;
        constexpr int HC = HL > 0 ? HL : 1;
        float lw[HC], lgn[HC], lgd[HC], law = 0.f, lg = 1.f, dbacc[HC];
        f8 dwacc[HC], hw[HC];
#pragma unroll
        for (int co = 0; co < HC; ++co) {
            lw[co] = 1.f; lgn[co] = 0.f; lgd[co] = 0.f; dbacc[co] = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { dwacc[co].v[e] = 0.f; hw[co].v[e] = 0.f; }
        }
        if (hloss) {
#pragma unroll
            for (int co = 0; co < HC; ++co) { lw[co] = a.hl_cw ? a.hl_cw[co] : 1.f; lgn[co] = a.hl_coef[1 + co]; lgd[co] = a.hl_coef[1 + HC + co]; hw[co] = ldf8(a.head_w + co * a.C + 8 * q); }
            law = a.hl_coef[0]; lg = a.hl_gout ? a.hl_gout[0] : 1.f;
        }
        float gys[hloss ? U : 1][HC];
        for (size_t v0 = i00 / Q; active && v0 < units; v0 += U * vstride) {
            f8 xv[U], g[U]; bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t v = v0 + u * vstride;
                ok[u] = v < units;
                const size_t vs = ok[u] ? v : 0;
                xv[u] = ld8nt(a.x + vs * a.x_ldc + 8 * q);
                if (HEAD) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[u].v[e] = 0.f;
                    const size_t n = vs / a.head_S, sp = vs - n * a.head_S;
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
                        float Gc[HC], dot = 0.f, wt = 0.f;
#pragma unroll
                        for (int co = 0; co < HC; ++co) {
                            pr[co] *= inv;
                            const bool is = t == co;
                            Gc[co] = lgn[co] - (is ? lgd[co] : 0.f);
                            dot += pr[co] * Gc[co];
                            wt += is ? lw[co] : 0.f;
                        }
#pragma unroll
                        for (int co = 0; co < HC; ++co) {
                            gys[u][co] = ok[u] ? lg * (law * wt * (pr[co] - (t == co ? 1.f : 0.f)) + pr[co] * (Gc[co] - dot)) : 0.f;
#pragma unroll
                            for (int e = 0; e < 8; ++e) g[u].v[e] = __builtin_fmaf(gys[u][co], hw[co].v[e], g[u].v[e]);
                        }
                    } else
                    for (int co = 0; co < a.head_cout; ++co) {
                        const float gy = a.head_dy[(n * a.head_cout + co) * a.head_S + sp];
                        const f8 wv = ldf8(a.head_w + co * a.C + 8 * q);
#pragma unroll
                        for (int e = 0; e < 8; ++e) g[u].v[e] = __builtin_fmaf(gy, wv.v[e], g[u].v[e]);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[u].v[e] = round_bf(g[u].v[e]);     // (autograd hands a bf16 gradient tensor to the norm)
                } else {
                    g[u] = ld8nt(a.g1 + vs * a.g1_ldc + 8 * q);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                f8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float z = __builtin_fmaf(xv[u].v[e], sc.v[e], sh.v[e]);
                    const float dz = (ok[u] && z > 0.f) ? g[u].v[e] : 0.f;
                    const float xh = (xv[u].v[e] - mu.v[e]) * is.v[e];
                    if (APPLYPASS) { o.v[e] = ok[u] ? round_bf(gi.v[e] * (dz - c1.v[e] - xh * c2.v[e])) : 0.f; s1.v[e] += o.v[e]; }
                    else { s1.v[e] += dz; s2.v[e] = __builtin_fmaf(dz, xh, s2.v[e]); }
                    if (hgrad) {       // head weight gradient: the activation the head saw (the expression of conv_final_b16_bwd_kernel's prologue)
                        const float av = round_bf(fmaxf(z, 0.f));
#pragma unroll
                        for (int co = 0; co < HC; ++co) dwacc[co].v[e] = __builtin_fmaf(gys[u][co], av, dwacc[co].v[e]);
                    }
                }
                if (APPLYPASS && ok[u]) st8(a.dx + (v0 + u * vstride) * a.dx_ldc + 8 * q, o);
                if (hgrad && q == 0) {
#pragma unroll
                    for (int co = 0; co < HC; ++co) dbacc[co] += gys[u][co];
                }
            }
        }
        if (hgrad) {        // (uniform) block partials of the head's gradients, one output row at a time: the layout of conv_final_b16_bwd_kernel
            const int pstride = HC * a.C + HC;
            const int tid = threadIdx.x;
#pragma unroll
            for (int co = 0; co < HC; ++co) {
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 8; ++e) red[0][tid][e] = active ? dwacc[co].v[e] : 0.f;
                red[1][tid][0] = (active && q == 0) ? dbacc[co] : 0.f;
                __syncthreads();
                for (int t = tid; t < Q * 8; t += 256) {
                    const int e = t & 7, qq = t >> 3;
                    float acc = 0.f;
                    for (int k = qq; k < BT; k += Q) acc += red[0][k][e];
                    a.head_part[(size_t)blockIdx.x * pstride + co * a.C + 8 * qq + e] = acc;
                }
                if (tid == 0) {
                    float acc = 0.f;
                    for (int k = 0; k < BT; k += Q) acc += red[1][k][0];
                    a.head_part[(size_t)blockIdx.x * pstride + HC * a.C + co] = acc;
                }
            }
            __syncthreads();
        }
    } else if ((Q & (Q - 1)) == 0 && Q <= 32 && !g_pool_rows) {
        // A window's two w-columns go to two lanes (lane ^ Q): 2Q consecutive lanes then read 2 voxels x C channels = ONE contiguous run
        // (128 bytes at C = 32) per (dz, dy), where the one-lane-per-window form asks for half cache lines at a 128-byte stride -- the
        // fp32 kernel, whose voxels are whole lines, runs the same loop at 5 TB/s, this one ran at 1.7-2.2.  "First arg-max wins" across
        // the pair: each lane finds its first match (dz, dy), the pair compares 2 * k + dx through one shuffle.
        // Software-pipelined: the ten 16-byte loads of the lane's NEXT window are in flight -- kept as raw bf16 registers, 40 instead of 80 -- while the current
        // one is computed.  (The kernel holds 220 registers = two waves per SIMD; without the look-ahead each wave alternated between a load phase and ~1100
        // VALU instructions with nothing in flight: 2.9 TB/s, VALU 11 % busy, SQ_WAIT_ANY 0.6 -- tools/pmc_valu.sh.  With the look-ahead: APPLY 139 -> 123 us,
        // REDUCE 125 -> 104 us on the level-0 tensor; forcing three waves per SIMD instead spills 46 registers and is slower.)
        const int dxl = (int)((i00 / Q) & 1);
        const size_t step = vstride / 2;
        const size_t uend = (units + step - 1) / step * step;      // (all lanes of a pair iterate together)
        struct Win { int pw, ph, pd, n; bool uok; };
        struct Raw { u16x8 x[4], g[4], gp, pm; };
        auto decode = [&](size_t u0) {
            Win w; w.uok = active && u0 < units;
            size_t r = w.uok ? u0 : 0;
            w.pw = (int)(r % Wp); r /= Wp; w.ph = (int)(r % Hp); r /= Hp; w.pd = (int)(r % Dp); w.n = (int)(r / Dp);
            return w;
        };
        auto okk = [&](const Win& w, int k) { return w.uok && (k >> 1) < kd && w.pd * kd + (k >> 1) < a.D && w.ph * 2 + (k & 1) < a.H && w.pw * 2 + dxl < a.W; };
        auto vox = [&](const Win& w, int k) { return (((size_t)w.n * a.D + (w.pd * kd + (k >> 1))) * a.H + (w.ph * 2 + (k & 1))) * a.W + (w.pw * 2 + dxl); };
        auto load = [&](const Win& w, Raw& R) {
            const size_t pidx = ((((size_t)w.n * Dp + w.pd) * Hp + w.ph) * Wp + w.pw) * a.C + 8 * q;
            R.gp = *reinterpret_cast<const u16x8*>(a.gpool + pidx); R.pm = *reinterpret_cast<const u16x8*>(a.pooled + pidx);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t v = okk(w, k) ? vox(w, k) : 0;
                R.x[k] = *reinterpret_cast<const u16x8*>(a.x + v * a.x_ldc + 8 * q);
                if (a.g1) R.g[k] = *reinterpret_cast<const u16x8*>(a.g1 + v * a.g1_ldc + 8 * q);
                else R.g[k] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        };
        Win wc = decode(i00 / (2 * Q));
        Raw Rc; load(wc, Rc);
        for (size_t u0 = i00 / (2 * Q); u0 < uend; u0 += step) {
            Win wn = wc; Raw Rn = Rc;
            if (u0 + step < uend) { wn = decode(u0 + step); load(wn, Rn); }
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) ok[k] = okk(wc, k);
            unsigned code = 0;                       // per channel: 2 * (first matching k) + dx, 15 = no match in this column
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                unsigned c = 15u;
                const float pme = bf2f(Rc.pm[e]);
#pragma unroll
                for (int k = 3; k >= 0; --k) {
                    const float av = round_bf(fmaxf(__builtin_fmaf(bf2f(Rc.x[k][e]), sc.v[e], sh.v[e]), 0.f));
                    if (ok[k] && av == pme) c = (unsigned)(2 * k + dxl);
                }
                code |= c << (4 * e);
            }
            const unsigned other = (unsigned)__shfl_xor((int)code, Q);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!ok[k]) continue;
                const size_t v = vox(wc, k);
                f8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xe = bf2f(Rc.x[k][e]);
                    const float z = __builtin_fmaf(xe, sc.v[e], sh.v[e]);
                    const unsigned mine = (code >> (4 * e)) & 15u, theirs = (other >> (4 * e)) & 15u;
                    float dA = bf2f(Rc.g[k][e]);
                    if (mine == (unsigned)(2 * k + dxl) && mine < theirs) dA += bf2f(Rc.gp[e]);     // first arg-max of the window wins (ATen)
                    const float dz = z > 0.f ? dA : 0.f;
                    const float xh = (xe - mu.v[e]) * is.v[e];
                    if (APPLYPASS) { o.v[e] = round_bf(gi.v[e] * (dz - c1.v[e] - xh * c2.v[e])); s1.v[e] += o.v[e]; }
                    else { s1.v[e] += dz; s2.v[e] = __builtin_fmaf(dz, xh, s2.v[e]); }
                }
                if (APPLYPASS) st8(a.dx + v * a.dx_ldc + 8 * q, o);
            }
            wc = wn; Rc = Rn;
        }
    } else {
        for (size_t u0 = i00 / Q; active && u0 < units; u0 += vstride) {
            size_t r = u0;
            const int pw = r % Wp; r /= Wp; const int ph = r % Hp; r /= Hp; const int pd = r % Dp; const int n = r / Dp;
            const size_t pidx = ((((size_t)n * Dp + pd) * Hp + ph) * Wp + pw) * a.C + 8 * q;
            const f8 gp = ld8(a.gpool + pidx), pm = ld8(a.pooled + pidx);
            bool taken[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) taken[e] = false;
            for (int dz_ = 0; dz_ < kd; ++dz_) {
                const int d = pd * kd + dz_; if (d >= a.D) break;
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    const int h = ph * 2 + dy; if (h >= a.H) break;
#pragma unroll
                    for (int dx_ = 0; dx_ < 2; ++dx_) {
                        const int w = pw * 2 + dx_; if (w >= a.W) break;
                        const size_t v = (((size_t)n * a.D + d) * a.H + h) * a.W + w;
                        const f8 xv = ld8(a.x + v * a.x_ldc + 8 * q);
                        f8 g;
                        if (a.g1) g = ld8(a.g1 + v * a.g1_ldc + 8 * q);
                        else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) g.v[e] = 0.f;
                        }
                        f8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            // the activation is recomputed with the forward's expression: bit-identical to the tensor the maxima were taken from
                            const float z = __builtin_fmaf(xv.v[e], sc.v[e], sh.v[e]);
                            const float av = round_bf(fmaxf(z, 0.f));
                            float dA = g.v[e];
                            if (!taken[e] && av == pm.v[e]) { dA += gp.v[e]; taken[e] = true; }     // first arg-max wins (ATen)
                            const float dz = z > 0.f ? dA : 0.f;
                            const float xh = (xv.v[e] - mu.v[e]) * is.v[e];
                            if (APPLYPASS) { o.v[e] = round_bf(gi.v[e] * (dz - c1.v[e] - xh * c2.v[e])); s1.v[e] += o.v[e]; }
                            else { s1.v[e] += dz; s2.v[e] = __builtin_fmaf(dz, xh, s2.v[e]); }
                        }
                        if (APPLYPASS) st8(a.dx + v * a.dx_ldc + 8 * q, o);
                    }
                }
            }
        }
    }
    // ---- block reduction: threads with equal (tid % Q) own the same channel octet
    const int tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][tid][e] = active ? s1.v[e] : 0.f; red[1][tid][e] = active ? s2.v[e] : 0.f; }
    __syncthreads();
    const int rows = APPLYPASS ? 1 : 2;
    for (int t = tid; t < Q * 8 * rows; t += 256) {
        const int e = t & 7, qq = (t >> 3) % Q, which = (t >> 3) / Q;
        float acc = 0.f;
        for (int k = qq; k < BT; k += Q) acc += red[which][k][e];
        a.part[((size_t)blockIdx.x * 3 + (APPLYPASS ? 2 : which)) * a.C + 8 * qq + e] = acc;     // part layout [parts][3][C]
    }
}

// ------------------------------------------------------------------ first conv (Cin < 8), forward: direct VALU conv
// brick = 2x8x16 (planar 1x16x16) voxels; thread = (channel quad q = tid % 8, voxel group tid / 8) computes 8 voxels x 4 channels
template <int KD, int TD, int TH>
__global__ __launch_bounds__(256) void conv_small_b16_fwd_kernel(const bf16_t* __restrict__ x, int Cin, const float* __restrict__ wgt,
                                                                 const float* __restrict__ bias, bf16_t* __restrict__ y, int y_ldc, int N, int D, int H, int W,
                                                                 int Cout, const float* __restrict__ epi_scale, const float* __restrict__ epi_shift,
                                                                 float* __restrict__ stats, int tilesD, int tilesH, int tilesW) {
    constexpr int TW = 16, PD = KD / 2, LD = TD + 2 * PD, LH = TH + 2, LW = TW + 2, NV = LD * LH * LW, T = KD * 9;
    extern __shared__ __attribute__((aligned(16))) float smemf[];
    float* xs = smemf;
    float* ws = smemf + ((Cin * NV + 3) & ~3);
    const int tid = threadIdx.x, q = tid & 7, g = tid >> 3;
    unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int tw_ = L % tilesW; L /= tilesW; const int th_ = L % tilesH; L /= tilesH; const int td_ = L % tilesD; const int nb = L / tilesD;
    const int d0 = td_ * TD, h0 = th_ * TH, w0 = tw_ * TW;
    for (int idx = tid; idx < Cin * NV; idx += 256) {
        const int ci = idx / NV, v = idx % NV;
        const int zw = v % LW, zh = (v / LW) % LH, zd = v / (LW * LH);
        const int gd = d0 + zd - PD, gh = h0 + zh - 1, gw = w0 + zw - 1;
        float val = 0.f;
        if (gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W) val = bf2f(x[((((size_t)nb * D + gd) * H + gh) * W + gw) * Cin + ci]);
        xs[ci * NV + v] = val;
    }
    int vbase[8]; bool vok[8]; size_t voff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int v = g + 32 * i;
        const int ww = v & 15, hh = (v >> 4) % TH, dd = (v >> 4) / TH;
        vbase[i] = (dd * LH + hh) * LW + ww;
        const int gd = d0 + dd, gh = h0 + hh, gw = w0 + ww;
        vok[i] = gd < D && gh < H && gw < W;
        voff[i] = ((((size_t)nb * D + gd) * H + gh) * W + gw) * y_ldc;
    }
    for (int pass = 0; pass * 32 < Cout; ++pass) {
        __syncthreads();
        for (int idx = tid; idx < Cin * T * 32; idx += 256) {
            const int c = idx & 31, t = (idx >> 5) % T, ci = (idx >> 5) / T;
            const int co = pass * 32 + c;
            ws[idx] = co < Cout ? round_bf(wgt[((size_t)co * Cin + ci) * T + t]) : 0.f;     // bf16 weights, like every other layer
        }
        __syncthreads();
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ci = 0; ci < Cin; ++ci) {
            f32x4 wr[T];
#pragma unroll
            for (int t = 0; t < T; ++t) wr[t] = *reinterpret_cast<const f32x4*>(ws + (ci * T + t) * 32 + 4 * q);
            const float* xc = xs + ci * NV;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
                    const float xv = xc[vbase[i] + (kd * LH + kh) * LW + kw];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][e] = __builtin_fmaf(xv, wr[t][e], acc[i][e]);
                }
        }
        const int co0 = pass * 32 + 4 * q;
        const bool cok = co0 < Cout;
        f32x4 bs = {0.f, 0.f, 0.f, 0.f}, es = {1.f, 1.f, 1.f, 1.f}, eh = bs;
        if (cok && bias) bs = *reinterpret_cast<const f32x4*>(bias + co0);
        const bool aff = epi_scale != nullptr;
        if (cok && aff) { es = *reinterpret_cast<const f32x4*>(epi_scale + co0); eh = *reinterpret_cast<const f32x4*>(epi_shift + co0); }
        f32x4 cnt = {0.f, 0.f, 0.f, 0.f}, sum = cnt;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            u16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[i][e];
                v = aff ? fmaxf(__builtin_fmaf(v, es[e], eh[e]), 0.f) : v + bs[e];
                o[e] = f2bf(v);
                acc[i][e] = bf2f(o[e]);
            }
            if (vok[i] && cok) {
                *reinterpret_cast<u16x4*>(y + voff[i] + co0) = o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { cnt[e] += 1.f; sum[e] += acc[i][e]; }
            }
        }
        if (stats) {
            f32x4 mean, m2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) mean[e] = cnt[e] > 0.f ? sum[e] / cnt[e] : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (vok[i] && cok)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float dd_ = acc[i][e] - mean[e]; m2[e] += dd_ * dd_; }
#pragma unroll
            for (int off = 8; off <= 32; off <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float c1 = cnt[e], mn1 = mean[e], s1 = m2[e];
                    const float c2 = __shfl_xor(c1, off), mn2 = __shfl_xor(mn1, off), s2 = __shfl_xor(s1, off);
                    welford_merge(c1, mn1, s1, c2, mn2, s2);
                    cnt[e] = c1; mean[e] = mn1; m2[e] = s1;
                }
            __syncthreads();
            const int wave = tid >> 6, lane = tid & 63;
            if (lane < 8)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float* scp = ws + ((wave * 32) + 4 * lane + e) * 3;
                    scp[0] = cnt[e]; scp[1] = mean[e]; scp[2] = m2[e];
                }
            __syncthreads();
            if (tid < 32 && pass * 32 + tid < Cout) {
                float c = 0.f, mn = 0.f, s = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) { const float* scp = ws + (w * 32 + tid) * 3; welford_merge(c, mn, s, scp[0], scp[1], scp[2]); }
                float* o = stats + ((size_t)blockIdx.x * Cout + pass * 32 + tid) * 3;
                o[0] = c; o[1] = mn; o[2] = s;
            }
        }
    }
}

// first conv, weight gradient: thread = (tap slot t = tid / 8, channel quad cq = tid % 8); part [splits][T][Cout][Cin]
template <int KD, int TD, int TH>
__global__ __launch_bounds__(256) void conv_small_b16_wgrad_kernel(const bf16_t* __restrict__ x, int Cin, const bf16_t* __restrict__ dy, int dy_ldc,
                                                                   float* __restrict__ part, int N, int D, int H, int W, int Cout,
                                                                   int tilesD, int tilesH, int tilesW, int tiles_per_split) {
    constexpr int TW = 16, PD = KD / 2, LD = TD + 2 * PD, LH = TH + 2, LW = TW + 2, NV = LD * LH * LW, T = KD * 9;
    extern __shared__ __attribute__((aligned(16))) float smemf[];
    float* xs = smemf;
    float* gs = smemf + ((NV + 3) & ~3);
    const int tid = threadIdx.x, cq = tid & 7, t = tid >> 3;
    const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
    const int toff = (kd * LH + kh) * LW + kw;
    const int ntiles = N * tilesD * tilesH * tilesW;
    const int tile0 = blockIdx.x * tiles_per_split;
    for (int pass = 0; pass * 32 < Cout; ++pass) {
        for (int ci = 0; ci < Cin; ++ci) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int tile = tile0; tile < tile0 + tiles_per_split && tile < ntiles; ++tile) {
                int L = tile;
                const int tw_ = L % tilesW; L /= tilesW; const int th_ = L % tilesH; L /= tilesH; const int td_ = L % tilesD; const int nb = L / tilesD;
                const int d0 = td_ * TD, h0 = th_ * TH, w0 = tw_ * TW;
                __syncthreads();
                for (int v = tid; v < NV; v += 256) {
                    const int zw = v % LW, zh = (v / LW) % LH, zd = v / (LW * LH);
                    const int gd = d0 + zd - PD, gh = h0 + zh - 1, gw = w0 + zw - 1;
                    float val = 0.f;
                    if (gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W) val = bf2f(x[((((size_t)nb * D + gd) * H + gh) * W + gw) * Cin + ci]);
                    xs[v] = val;
                }
                for (int idx = tid; idx < 256 * 8; idx += 256) {
                    const int v = idx >> 3, qq = idx & 7;
                    const int ww = v & 15, hh = (v >> 4) % TH, dd = (v >> 4) / TH;
                    const int gd = d0 + dd, gh = h0 + hh, gw = w0 + ww;
                    f32x4 val = {0.f, 0.f, 0.f, 0.f};
                    if (gd < D && gh < H && gw < W && pass * 32 + 4 * qq < Cout) {
                        const size_t vox = (((size_t)nb * D + gd) * H + gh) * W + gw;
                        const u16x4 r = *reinterpret_cast<const u16x4*>(dy + vox * dy_ldc + pass * 32 + 4 * qq);
#pragma unroll
                        for (int e = 0; e < 4; ++e) val[e] = bf2f(r[e]);
                    }
                    *reinterpret_cast<f32x4*>(gs + v * 32 + 4 * qq) = val;
                }
                __syncthreads();
                if (t < T) {
#pragma unroll 8
                    for (int v = 0; v < 256; ++v) {
                        const int ww = v & 15, hh = (v >> 4) % TH, dd = (v >> 4) / TH;
                        const float xv = xs[(dd * LH + hh) * LW + ww + toff];
                        const f32x4 gv = *reinterpret_cast<const f32x4*>(gs + v * 32 + 4 * cq);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(xv, gv[e], acc[e]);
                    }
                }
            }
            if (t < T)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = pass * 32 + 4 * cq + e;
                    if (co < Cout) part[(((size_t)blockIdx.x * T + t) * Cout + co) * Cin + ci] = acc[e];
                }
        }
    }
}

// ------------------------------------------------------------------ 1x1x1 head on a bf16 activation (fp32 NCDHW logits)
template <int COUT>
__global__ void conv_final_b16_fwd_kernel(const bf16_t* __restrict__ a, int a_ldc, int C, const float* __restrict__ w, const float* __restrict__ bias,
                                          float* __restrict__ y, size_t S, int N, int lpv, int softmax,
                                          const float* __restrict__ pro_scale, const float* __restrict__ pro_shift, size_t ychan) {
    const int Q = C >> 3;
    const size_t total = (size_t)N * S;
    const int sub = threadIdx.x % lpv;
    const size_t vpb = blockDim.x / lpv;
    // U voxels per lane group and iteration, their loads issued together (one load in flight per lane left the kernel latency-bound: 58 us for
    // 151 MB); a workgroup's U x vpb voxels of an iteration are one contiguous run.  Same summation order per voxel.
    constexpr int U = 4;
    const size_t chunk = vpb * U;
    for (size_t v0 = blockIdx.x * chunk + threadIdx.x / lpv; v0 < (total + chunk - 1) / chunk * chunk; v0 += (size_t)gridDim.x * chunk) {
        float acc[U][COUT];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int co = 0; co < COUT; ++co) acc[u][co] = 0.f;
        for (int q = sub; q < Q; q += lpv) {
            f8 av[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t v = v0 + u * vpb;
                av[u] = ld8(a + (v < total ? v : 0) * a_ldc + 8 * q);
            }
            f8 sc, sh;
            if (pro_scale) { sc = ldf8(pro_scale + 8 * q); sh = ldf8(pro_shift + 8 * q); }
            f8 wv[COUT];
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                wv[co] = ldf8(w + co * C + 8 * q);
#pragma unroll
                for (int e = 0; e < 8; ++e) wv[co].v[e] = round_bf(wv[co].v[e]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (pro_scale) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) av[u].v[e] = round_bf(fmaxf(__builtin_fmaf(av[u].v[e], sc.v[e], sh.v[e]), 0.f));
                }
#pragma unroll
                for (int co = 0; co < COUT; ++co)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[u][co] = __builtin_fmaf(av[u].v[e], wv[co].v[e], acc[u][co]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            for (int off = 1; off < lpv; off <<= 1)
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[u][co] += __shfl_xor(acc[u][co], off);
            const size_t v = v0 + u * vpb;
            if (v < total && sub == 0) {
                const size_t n = v / S, sp = v % S;
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[u][co] = round_bf(acc[u][co] + (bias ? bias[co] : 0.f));     // the module returns bf16 logits
                if (softmax) {
                    float m = acc[u][0];
#pragma unroll
                    for (int co = 1; co < COUT; ++co) m = fmaxf(m, acc[u][co]);
                    float s = 0.f;
#pragma unroll
                    for (int co = 0; co < COUT; ++co) { acc[u][co] = __expf(acc[u][co] - m); s += acc[u][co]; }
                    const float inv = 1.f / s;
#pragma unroll
                    for (int co = 0; co < COUT; ++co) acc[u][co] *= inv;
                }
#pragma unroll
                for (int co = 0; co < COUT; ++co) y[(n * COUT + co) * ychan + sp] = acc[u][co];      // (ychan: voxels between the channel planes of y; = S unless y is a range of d-planes)
            }
        }
    }
}

template <int COUT>
__global__ __launch_bounds__(256) void conv_final_b16_bwd_kernel(const bf16_t* __restrict__ a, int a_ldc, int C, const float* __restrict__ dy,
                                                                 float* __restrict__ part, size_t S, int N,
                                                                 const float* __restrict__ pro_scale, const float* __restrict__ pro_shift) {
    __shared__ float red[256][8];
    const int Q = C >> 3;
    const int BT = (256 / Q) * Q;
    const size_t total = (size_t)N * S * Q;
    const int tid = threadIdx.x;
    const int q = tid % Q;
    f8 dwacc[COUT];
    float dbacc[COUT];
    f8 sc, sh;
    if (pro_scale) { sc = ldf8(pro_scale + 8 * q); sh = ldf8(pro_shift + 8 * q); }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
#pragma unroll
        for (int e = 0; e < 8; ++e) dwacc[co].v[e] = 0.f;
        dbacc[co] = 0.f;
    }
    for (size_t i = (size_t)blockIdx.x * BT + tid; tid < BT && i < total; i += (size_t)gridDim.x * BT) {
        const size_t v = i / Q;
        const size_t n = v / S, sp = v % S;
        f8 av = ld8(a + v * a_ldc + 8 * q);
        if (pro_scale) {
#pragma unroll
            for (int e = 0; e < 8; ++e) av.v[e] = round_bf(fmaxf(__builtin_fmaf(av.v[e], sc.v[e], sh.v[e]), 0.f));
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float g = dy[(n * COUT + co) * S + sp];
#pragma unroll
            for (int e = 0; e < 8; ++e) dwacc[co].v[e] = __builtin_fmaf(g, av.v[e], dwacc[co].v[e]);
            if (q == 0) dbacc[co] += g;
        }
    }
    const int pstride = COUT * C + COUT;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid][e] = tid < BT ? dwacc[co].v[e] : 0.f;
        __syncthreads();
        for (int t = tid; t < Q * 8; t += 256) {
            const int e = t & 7, qq = t >> 3;
            float acc = 0.f;
            for (int k = qq; k < BT; k += Q) acc += red[k][e];
            part[(size_t)blockIdx.x * pstride + co * C + 8 * qq + e] = acc;
        }
        __syncthreads();
        red[tid][0] = (tid < BT && q == 0) ? dbacc[co] : 0.f;
        __syncthreads();
        if (tid == 0) {
            float acc = 0.f;
            for (int k = 0; k < BT; k += Q) acc += red[k][0];
            part[(size_t)blockIdx.x * pstride + COUT * C + co] = acc;
        }
    }
}

__global__ void ncdhw_to_ndhwc_b16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int N, int C, size_t S) {
    const size_t total = (size_t)N * C * S;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % C; const size_t r = i / C; const size_t sp = r % S; const size_t n = r / S;
        dst[i] = src[(n * C + c) * S + sp];
    }
}
__global__ void ndhwc_to_ncdhw_b16_kernel(const bf16_t* __restrict__ src, int ldc, bf16_t* __restrict__ dst, int N, int C, size_t S) {
    const size_t total = (size_t)N * C * S;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t sp = i % S; const size_t r = i / S; const int c = r % C; const size_t n = r / C;
        dst[i] = src[(n * S + sp) * ldc + c];
    }
}
__global__ void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = f2bf(src[i]);
}

unsigned ew_grid(size_t items) {
    size_t g = (items + EW_BLOCK - 1) / EW_BLOCK;
    if (g > EW_MAX_GRID) g = EW_MAX_GRID;
    if (g == 0) g = 1;
    return (unsigned)g;
}

int final_lpv8(int C) {
    const int Q = C / 8;
    int l = 1;
    while (l < 8 && Q % (l * 2) == 0) l *= 2;
    return l;
}

// a pooling window's two w-columns on two lanes: needs lane ^ Q inside the wave (other channel counts take the one-lane-per-window form)
bool pool_pairs(int C) {
    const int Q = C / 8;
    return (Q & (Q - 1)) == 0 && Q <= 32;
}

}  // namespace

int launch_bn_relu_apply_b16(const bf16_t* x, int x_ldc, const float* scale, const float* shift, bf16_t* a, int a_ldc,
                             bf16_t* pooled, int kd, int N, int D, int H, int W, int C, hipStream_t s) {
    E3_REQUIRE(C % 8 == 0 && x_ldc % 8 == 0 && a_ldc % 8 == 0, E3_ERR_UNSUPPORTED, "bf16 passes need channel counts that are multiples of 8");
    const size_t vox = (size_t)N * D * H * W;
    if (pooled) {
        const size_t items = (size_t)N * cdiv(D, kd) * cdiv(H, 2) * cdiv(W, 2) * (C / 8);
        if (pool_pairs(C)) hipLaunchKernelGGL(bn_relu_pool2_b16_kernel<true>, dim3(ew_grid(2 * items)), dim3(EW_BLOCK), 0, s, x, x_ldc, scale, shift, a, a_ldc, pooled, kd, N, D, H, W, C);
        else
        hipLaunchKernelGGL(bn_relu_pool_b16_kernel<true>, dim3(ew_grid(items)), dim3(EW_BLOCK), 0, s, x, x_ldc, scale, shift, a, a_ldc, pooled, kd, N, D, H, W, C);
    } else {
        hipLaunchKernelGGL(bn_relu_apply_b16_kernel, dim3(ew_grid((vox * (C / 8) + 3) / 4)), dim3(EW_BLOCK), 0, s, x, x_ldc, scale, shift, a, a_ldc, vox, C);
    }
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_maxpool_b16(const bf16_t* a, int a_ldc, bf16_t* pooled, int kd, int N, int D, int H, int W, int C, hipStream_t s) {
    E3_REQUIRE(C % 8 == 0 && a_ldc % 8 == 0, E3_ERR_UNSUPPORTED, "bf16 passes need channel counts that are multiples of 8");
    const size_t items = (size_t)N * cdiv(D, kd) * cdiv(H, 2) * cdiv(W, 2) * (C / 8);
    if (pool_pairs(C)) hipLaunchKernelGGL(bn_relu_pool2_b16_kernel<false>, dim3(ew_grid(2 * items)), dim3(EW_BLOCK), 0, s, a, a_ldc, nullptr, nullptr, nullptr, 0, pooled, kd, N, D, H, W, C);
    else
    hipLaunchKernelGGL(bn_relu_pool_b16_kernel<false>, dim3(ew_grid(items)), dim3(EW_BLOCK), 0, s, a, a_ldc, nullptr, nullptr, nullptr, 0, pooled, kd, N, D, H, W, C);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int bn_bwd_b16_parts(size_t voxels, int C) {
    size_t g = (voxels * (size_t)(C / 8) + 255) / 256 / 8;
    if (g > 1024) g = 1024;
    if (g == 0) g = 1;
    return (int)g;
}

static int bn_bwd_b16_launch(BnBwdB16Args a, bool apply, hipStream_t s) {
    a.pool_one_lane = pool_pairs(a.C) ? 0 : 1;
    E3_REQUIRE(a.C % 8 == 0 && a.C <= 2048 && a.x_ldc % 8 == 0, E3_ERR_UNSUPPORTED, "bf16 passes need channel counts that are multiples of 8");
    const dim3 grid(a.parts), block(256);
    const bool pool = a.gpool != nullptr, head = a.g1 == nullptr && !pool;
    if (head) E3_REQUIRE((a.head_dy || a.hl_logits) && a.head_w, E3_ERR_INVALID, "bn backward: no incoming gradient");
    if (head && a.hl_logits) {      // criterion variant (+ the head's own gradients in the REDUCE pass)
        E3_REQUIRE(a.head_cout >= 2 && a.head_cout <= 4 && a.hl_target && a.hl_coef && (apply || a.head_part), E3_ERR_UNSUPPORTED, "bn backward, criterion form: 2..4 classes");
        switch (a.head_cout * 2 + (apply ? 1 : 0)) {
            case 4: hipLaunchKernelGGL((bn_bwd_b16_kernel<false, false, true, 2>), grid, block, 0, s, a); break;
            case 5: hipLaunchKernelGGL((bn_bwd_b16_kernel<false, true, true, 2>), grid, block, 0, s, a); break;
            case 6: hipLaunchKernelGGL((bn_bwd_b16_kernel<false, false, true, 3>), grid, block, 0, s, a); break;
            case 7: hipLaunchKernelGGL((bn_bwd_b16_kernel<false, true, true, 3>), grid, block, 0, s, a); break;
            case 8: hipLaunchKernelGGL((bn_bwd_b16_kernel<false, false, true, 4>), grid, block, 0, s, a); break;
            default: hipLaunchKernelGGL((bn_bwd_b16_kernel<false, true, true, 4>), grid, block, 0, s, a); break;
        }
        E3_CHECK_HIP(hipGetLastError());
        return E3_OK;
    }
    if (pool) {
        if (apply) hipLaunchKernelGGL((bn_bwd_b16_kernel<true, true, false>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((bn_bwd_b16_kernel<true, false, false>), grid, block, 0, s, a);
    } else if (head) {
        if (apply) hipLaunchKernelGGL((bn_bwd_b16_kernel<false, true, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((bn_bwd_b16_kernel<false, false, true>), grid, block, 0, s, a);
    } else {
        if (apply) hipLaunchKernelGGL((bn_bwd_b16_kernel<false, true, false>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((bn_bwd_b16_kernel<false, false, false>), grid, block, 0, s, a);
    }
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
int launch_bn_bwd_b16_reduce(BnBwdB16Args a, hipStream_t s) { return bn_bwd_b16_launch(a, false, s); }
int launch_bn_bwd_b16_apply(BnBwdB16Args a, hipStream_t s) { return bn_bwd_b16_launch(a, true, s); }

// brick of the first conv: 2 x 8 x 16 voxels, planar (1x3x3 taps, unet.py:114-128) 1 x 16 x 16
int conv_small_b16_stats_parts(int N, int D, int H, int W, int planar) { return planar ? N * D * cdiv(H, 16) * cdiv(W, 16) : N * cdiv(D, 2) * cdiv(H, 8) * cdiv(W, 16); }
int conv_small_b16_stats_parts2(int N, int D, int H, int W, int planar, int Cin, int Cout) {
    const int p = conv_first_b16_supported(Cin, Cout, planar) ? conv_first_b16_stats_parts(N, D, H, W, Cout) : 0;
    return p > 0 ? p : conv_small_b16_stats_parts(N, D, H, W, planar);
}

int launch_conv_small_b16_fwd(const bf16_t* x, int Cin, const float* w, const float* bias, bf16_t* y, int y_ldc,
                              int N, int D, int H, int W, int Cout, int planar, const float* epi_scale, const float* epi_shift, float* stats, hipStream_t s) {
    E3_REQUIRE(Cin >= 1 && Cin < 8, E3_ERR_UNSUPPORTED, "bf16 first conv: 1..7 input channels");
    E3_REQUIRE(Cout % 4 == 0 && y_ldc % 4 == 0, E3_ERR_UNSUPPORTED, "output channels must be a multiple of 4");
    if (conv_first_b16_supported(Cin, Cout, planar)) return launch_conv_first_b16_fwd(x, w, bias, y, y_ldc, N, D, H, W, Cout, epi_scale, epi_shift, stats, s);
    const int T = planar ? 9 : 27;
    const int tD = planar ? D : cdiv(D, 2), tH = cdiv(H, planar ? 16 : 8), tW = cdiv(W, 16);
    const int NV = planar ? 1 * 18 * 18 : 4 * 10 * 18;
    const int wslab = Cin * T * 32 > 4 * 32 * 3 ? Cin * T * 32 : 4 * 32 * 3;
    const size_t lds = (size_t)(((Cin * NV + 3) & ~3) + wslab) * 4;
    const dim3 grid((unsigned)((size_t)N * tD * tH * tW));
    if (planar) hipLaunchKernelGGL((conv_small_b16_fwd_kernel<1, 1, 16>), grid, dim3(256), lds, s, x, Cin, w, bias, y, y_ldc, N, D, H, W, Cout, epi_scale, epi_shift, stats, tD, tH, tW);
    else hipLaunchKernelGGL((conv_small_b16_fwd_kernel<3, 2, 8>), grid, dim3(256), lds, s, x, Cin, w, bias, y, y_ldc, N, D, H, W, Cout, epi_scale, epi_shift, stats, tD, tH, tW);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

static int small_b16_tps(int ntiles) { return cdiv(ntiles, 1024); }
int conv_small_b16_wgrad_splits(int N, int D, int H, int W, int planar) {
    const int ntiles = conv_small_b16_stats_parts(N, D, H, W, planar);
    return cdiv(ntiles, small_b16_tps(ntiles));
}

int launch_conv_small_b16_wgrad(const bf16_t* x, int Cin, const bf16_t* dy, int dy_ldc, float* part,
                                int N, int D, int H, int W, int Cout, int planar, hipStream_t s) {
    E3_REQUIRE(Cin >= 1 && Cin < 8, E3_ERR_UNSUPPORTED, "bf16 first conv: 1..7 input channels");
    const int tD = planar ? D : cdiv(D, 2), tH = cdiv(H, planar ? 16 : 8), tW = cdiv(W, 16);
    const int ntiles = N * tD * tH * tW;
    const int tps = small_b16_tps(ntiles);
    const int splits = cdiv(ntiles, tps);
    if (conv_first_b16_supported(Cin, Cout, planar)) return launch_conv_first_b16_wgrad(x, dy, dy_ldc, part, N, D, H, W, Cout, tps, splits, s);
    const int NV = planar ? 1 * 18 * 18 : 4 * 10 * 18;
    const size_t lds = (size_t)(((NV + 3) & ~3) + 256 * 32) * 4;
    if (planar) hipLaunchKernelGGL((conv_small_b16_wgrad_kernel<1, 1, 16>), dim3(splits), dim3(256), lds, s, x, Cin, dy, dy_ldc, part, N, D, H, W, Cout, tD, tH, tW, tps);
    else hipLaunchKernelGGL((conv_small_b16_wgrad_kernel<3, 2, 8>), dim3(splits), dim3(256), lds, s, x, Cin, dy, dy_ldc, part, N, D, H, W, Cout, tD, tH, tW, tps);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

#define E3_COUT_SWITCH_B16(COUT, ...)                         \
    switch (COUT) {                                       \
        case 1: { constexpr int CO = 1; __VA_ARGS__; break; } \
        case 2: { constexpr int CO = 2; __VA_ARGS__; break; } \
        case 3: { constexpr int CO = 3; __VA_ARGS__; break; } \
        case 4: { constexpr int CO = 4; __VA_ARGS__; break; } \
        case 5: { constexpr int CO = 5; __VA_ARGS__; break; } \
        case 6: { constexpr int CO = 6; __VA_ARGS__; break; } \
        case 7: { constexpr int CO = 7; __VA_ARGS__; break; } \
        case 8: { constexpr int CO = 8; __VA_ARGS__; break; } \
        default: e3_set_error("bf16 1x1x1 head supports 1..8 output channels"); return E3_ERR_UNSUPPORTED; \
    }

int launch_conv_final_b16_fwd(const bf16_t* a, int a_ldc, int C, const float* w, const float* bias, float* y, int Cout,
                              size_t S, int N, int softmax, const float* pro_scale, const float* pro_shift, hipStream_t s, size_t ychan) {
    E3_REQUIRE(C % 8 == 0 && a_ldc % 8 == 0, E3_ERR_UNSUPPORTED, "channels must be a multiple of 8");
    if (ychan == 0) ychan = S;
    const int lpv = final_lpv8(C);
    const size_t vox = (size_t)N * S;
    size_t g = (vox * lpv + 255) / 256; if (g > 4096) g = 4096; if (g == 0) g = 1;
    E3_COUT_SWITCH_B16(Cout, hipLaunchKernelGGL((conv_final_b16_fwd_kernel<CO>), dim3((unsigned)g), dim3(256), 0, s, a, a_ldc, C, w, bias, y, S, N, lpv, softmax, pro_scale, pro_shift, ychan));
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int conv_final_b16_bwd_parts(size_t total_voxels) {
    size_t g = (total_voxels + 31) / 32; if (g > 1024) g = 1024; if (g == 0) g = 1;
    return (int)g;
}

int launch_conv_final_b16_bwd(const bf16_t* a, int a_ldc, int C, const float* w, const float* dy, bf16_t* da, int da_ldc,
                              float* part, int Cout, size_t S, int N, const float* pro_scale, const float* pro_shift, hipStream_t s) {
    (void)w; (void)da_ldc;
    E3_REQUIRE(C % 8 == 0 && C <= 2048, E3_ERR_UNSUPPORTED, "channels must be a multiple of 8 and <= 2048");
    E3_REQUIRE(da == nullptr, E3_ERR_UNSUPPORTED, "bf16 head backward: the consumer recomputes the data gradient (BnBwdB16Args::head_dy)");
    const int parts = conv_final_b16_bwd_parts((size_t)N * S);
    E3_COUT_SWITCH_B16(Cout, hipLaunchKernelGGL((conv_final_b16_bwd_kernel<CO>), dim3(parts), dim3(256), 0, s, a, a_ldc, C, dy, part, S, N, pro_scale, pro_shift));
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_ncdhw_to_ndhwc_b16(const bf16_t* src, bf16_t* dst, int N, int C, size_t S, hipStream_t s) {
    hipLaunchKernelGGL(ncdhw_to_ndhwc_b16_kernel, dim3(ew_grid((size_t)N * C * S)), dim3(EW_BLOCK), 0, s, src, dst, N, C, S);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
int launch_ndhwc_to_ncdhw_b16(const bf16_t* src, int src_ldc, bf16_t* dst, int N, int C, size_t S, hipStream_t s) {
    hipLaunchKernelGGL(ndhwc_to_ncdhw_b16_kernel, dim3(ew_grid((size_t)N * C * S)), dim3(EW_BLOCK), 0, s, src, src_ldc, dst, N, C, S);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
int launch_f32_to_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, s, src, dst, n);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
