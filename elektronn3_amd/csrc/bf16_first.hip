// The network's first 3x3x3 conv (in_channels = 1, elektronn3/models/unet.py:131-149 as DownConv.conv1 of the first block) on the bf16
// matrix cores: forward and weight gradient.  With one input channel the GEMM-K is the 27 taps (padded to 32): far too thin for the
// staged-image kernels of bf16_conv.hip, and as a VALU kernel (bf16_ew.hip, conv_small_*) it is FMA-bound: 2 M voxels x 32 channels x 27
// taps = 1.8 G FMAs ~ 100 us of the whole chip's fp32 VALU for a layer that moves 134 MB.  Here the taps are the K of two
// v_mfma_f32_32x32x16_bf16 per 32 voxels x 32 channels, the operands are gathered from a 1.4 KB LDS image of the brick.
//   forward:  Y^T[co][v]   = sum_tap W[co][tap] * X[v + tap]         A = weights (rows co), B = patches (columns = voxels)
//   wgrad:    dW[co][tap]  = sum_v  dY[v][co]   * X[v + tap]         A = dY^T (transposing LDS reads), B = patches (columns = taps)
// Same brick (2 x 8 x 16 voxels), statistics records and slab layout as the VALU kernels they replace.
#include "bf16.h"

namespace {

constexpr int TD = 2, TH = 8, TW = 16, LD = 4, LH = 10, LW = 18, NV = LD * LH * LW, T = 27;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_frag(unsigned addr) {       // 8 k-values: rows 0..3 (addr) and 4..7 (addr + 4 rows of 64 B)
    typedef s16x4 __attribute__((address_space(3))) * lp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)addr);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(addr + 256));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ float dpp_sum32(float x) {             // sum over the lanes of each half-wave; valid in lanes 16-31 / 48-63
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xb1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4e, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));    // row_half_mirror
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true));    // row_mirror
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xa, 0xf, true));    // row_bcast:15 into rows 1, 3
    return x;
}
__device__ __forceinline__ int tap_offset(int k) {               // image offset of tap k (taps 27..31 of the padded K read tap 26: their weights are 0)
    const int t = k < T ? k : T - 1;
    return ((t / 9) * LH + (t / 3) % 3) * LW + t % 3;
}
__device__ __forceinline__ void stage_image(unsigned short* xs, const bf16_t* __restrict__ x, int nb, int d0, int h0, int w0, int D, int H, int W, int tid) {
    constexpr int IT = (NV + 255) / 256;
    unsigned short val[IT];                  // all loads in flight before the first LDS write (a rolled loop waits a memory round trip per pass)
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int v = tid + 256 * i;
        const int zw = v % LW, zh = (v / LW) % LH, zd = v / (LW * LH);
        const int gd = d0 + zd - 1, gh = h0 + zh - 1, gw = w0 + zw - 1;
        val[i] = (v < NV && gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W) ? x[(((size_t)nb * D + gd) * H + gh) * W + gw] : (unsigned short)0;
    }
#pragma unroll
    for (int i = 0; i < IT; ++i)
        if (tid + 256 * i < NV) xs[tid + 256 * i] = val[i];
}

__global__ __launch_bounds__(256) void conv_first_b16_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ bias,
                                                                 bf16_t* __restrict__ y, int y_ldc, int N, int D, int H, int W, int Cout,
                                                                 const float* __restrict__ epi_scale, const float* __restrict__ epi_shift,
                                                                 float* __restrict__ stats, int tilesD, int tilesH, int tilesW) {
    __shared__ unsigned short xs[NV + 8];
    __shared__ __attribute__((aligned(16))) unsigned short wl[32 * 32];
    __shared__ float S[4 * 2 * 32];                                  // the waves' channel sums: [wave][quantity][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, g = lane >> 5;
    unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned brick = L;
    const int tw_ = L % tilesW; L /= tilesW; const int th_ = L % tilesH; L /= tilesH; const int td_ = L % tilesD; const int nb = L / tilesD;
    const int d0 = td_ * TD, h0 = th_ * TH, w0 = tw_ * TW;
    stage_image(xs, x, nb, d0, h0, w0, D, H, W, tid);
    // tiles of 2 rows x 16 voxels; wave w owns tiles 2w, 2w + 1 (d-plane w >> 1, rows 4 (w & 1) + 2 t + r); lane = voxel (r = j >> 4, c = j & 15)
    const int dd = wave >> 1, hh0 = 4 * (wave & 1), r = j >> 4, c = j & 15;
    const int lanebase = (dd * LH + hh0 + r) * LW + c;
    int toff[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) toff[ks][e] = tap_offset(16 * ks + 8 * g + e);
    __syncthreads();
    bf16x8 bfr[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = xs[lanebase + t * 2 * LW + toff[ks][e]];
            bfr[t][ks] = __builtin_bit_cast(bf16x8, v);
        }
    const int gd = d0 + dd, gw = w0 + c;
    const int nd = D - d0 < TD ? D - d0 : TD, nh = H - h0 < TH ? H - h0 : TH, nw = W - w0 < TW ? W - w0 : TW;
    const float cnt = (float)(nd * nh * nw);
    const bool want_stats = stats != nullptr;
    for (int pass = 0; pass * 32 < Cout; ++pass) {
        // weights of this channel tile, rounded to bf16 like every other layer's: [co][32 taps] in LDS (coalesced), lane (row co = j, k-group g)
        if (pass) __syncthreads();
        for (int idx = tid; idx < 32 * 32; idx += 256) {
            const int co = idx >> 5, k = idx & 31;
            wl[idx] = k < T ? f2bf(wgt[(size_t)(pass * 32 + co) * T + k]) : (unsigned short)0;
        }
        __syncthreads();
        bf16x8 af[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(wl + j * 32 + 16 * ks + 8 * g);
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
            acc[t] = E3_MFMA16(af[0], bfr[t][0], acc[t], 0, 0, 0);
            acc[t] = E3_MFMA16(af[1], bfr[t][1], acc[t], 0, 0, 0);
        }
        // lane (voxel j, half g) holds channels (e&3) + 8*(e>>2) + 4*g of its voxel: bias, rounding, 8-byte stores, lane sums
        float ssum[16], ssq[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
        f32x4 bq[4], sq[4], hq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cb = pass * 32 + 8 * q + 4 * g;
            bq[q] = bias ? *reinterpret_cast<const f32x4*>(bias + cb) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (epi_scale) { sq[q] = *reinterpret_cast<const f32x4*>(epi_scale + cb); hq[q] = *reinterpret_cast<const f32x4*>(epi_shift + cb); }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int gh = h0 + hh0 + 2 * t + r;
            const bool valid = gd < D && gh < H && gw < W;
            bf16_t* yrow = y + ((((size_t)nb * D + gd) * H + gh) * W + gw) * y_ldc + pass * 32 + 4 * g;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float val = acc[t][4 * q + e];
                    if (epi_scale) val = fmaxf(__builtin_fmaf(val, sq[q][e], hq[q][e]), 0.f);
                    else val += bq[q][e];
                    const bf16_t rb = f2bf(val);
                    o[e] = rb;
                    const float dv = valid ? bf2f(rb) - bq[q][e] : 0.f;
                    ssum[4 * q + e] += dv; ssq[4 * q + e] = __builtin_fmaf(dv, dv, ssq[4 * q + e]);
                }
                if (valid) *reinterpret_cast<u16x4*>(yrow + 8 * q) = o;
            }
        }
        if (!want_stats) continue;
        // sums over the 32 voxel lanes of each half-wave in registers (5 DPP adds per value: no 34 KB exchange buffer, twice the
        // workgroups per CU), the four waves through 1 KB of LDS -> one (n, mean, M2) record per brick and channel
        if (pass) __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float s = dpp_sum32(ssum[e]), q2 = dpp_sum32(ssq[e]);
            const int col = (e & 3) + 8 * (e >> 2) + 4 * g;
            if (j == 31) { S[(wave * 2 + 0) * 32 + col] = s; S[(wave * 2 + 1) * 32 + col] = q2; }
        }
        __syncthreads();
        if (tid < 32) {
            const float s = (S[0 * 64 + tid] + S[1 * 64 + tid]) + (S[2 * 64 + tid] + S[3 * 64 + tid]);
            const float q2 = (S[0 * 64 + 32 + tid] + S[1 * 64 + 32 + tid]) + (S[2 * 64 + 32 + tid] + S[3 * 64 + 32 + tid]);
            const int co = pass * 32 + tid;
            const float b = bias ? bias[co] : 0.f;
            const float m = s / cnt;
            float* rec = stats + ((size_t)brick * Cout + co) * 3;
            rec[0] = cnt; rec[1] = b + m; rec[2] = fmaxf(q2 - s * m, 0.f);
        }
    }
}

// slab part[split][tap][Cout] (Cin = 1); a workgroup sums `tiles_per_split` bricks; wave w takes the k-steps (rows of 16 voxels) 4w..4w+3
__global__ __launch_bounds__(256) void conv_first_b16_wgrad_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, int dy_ldc, float* __restrict__ part,
                                                                   int N, int D, int H, int W, int Cout, int tilesD, int tilesH, int tilesW, int tiles_per_split) {
    __shared__ unsigned short xs[NV + 8];
    __shared__ __attribute__((aligned(16))) unsigned char gs[256 * 64];          // dY tile [voxel][32 channels] bf16; later the waves' partial sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, g = lane >> 5;
    const int ntiles = N * tilesD * tilesH * tilesW;
    const int tile0 = blockIdx.x * tiles_per_split;
    // dY fragment (A, rows = channels): transposing read of 16 voxels x 32 channels
    const int G = lane >> 4, tt = lane & 15;
    const int krow = 8 * (G >> 1) + (tt >> 2), chb = (16 * (G & 1) + 4 * (tt & 3)) * 2;
    const unsigned gbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)gs + (unsigned)(krow * 64 + chb);                 // + k-step * 1024
    const int toffj = tap_offset(j);                                                             // this lane's tap column
    for (int pass = 0; pass * 32 < Cout; ++pass) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        for (int tile = tile0; tile < tile0 + tiles_per_split && tile < ntiles; ++tile) {
            int L = tile;
            const int tw_ = L % tilesW; L /= tilesW; const int th_ = L % tilesH; L /= tilesH; const int td_ = L % tilesD; const int nb = L / tilesD;
            const int d0 = td_ * TD, h0 = th_ * TH, w0 = tw_ * TW;
            __syncthreads();
            stage_image(xs, x, nb, d0, h0, w0, D, H, W, tid);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * i;
                const int v = idx >> 2, q = idx & 3;
                const int ww = v & 15, hh = (v >> 4) & 7, dd = v >> 7;
                const int gd = d0 + dd, gh = h0 + hh, gw = w0 + ww;
                u16x8 val = {0, 0, 0, 0, 0, 0, 0, 0};
                if (gd < D && gh < H && gw < W) val = *reinterpret_cast<const u16x8*>(dy + ((((size_t)nb * D + gd) * H + gh) * W + gw) * dy_ldc + pass * 32 + 8 * q);
                *reinterpret_cast<u16x8*>(gs + v * 64 + q * 16) = val;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int s = 4 * wave + i, dd = s >> 3, hh = s & 7;
                const bf16x8 af = tr_frag(gbase + s * 1024);
                u16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = xs[(dd * LH + hh) * LW + 8 * g + e + toffj];
                acc = E3_MFMA16(af, __builtin_bit_cast(bf16x8, v), acc, 0, 0, 0);
            }
        }
        // the four waves' partial sums -> slab.  lane holds column tap = j, rows co = (e&3) + 8*(e>>2) + 4*g
        __syncthreads();
        float* red = reinterpret_cast<float*>(gs);                                               // [wave][co][tap] = 16 KB
#pragma unroll
        for (int e = 0; e < 16; ++e) red[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * g) * 32 + j] = acc[e];
        __syncthreads();
        for (int idx = tid; idx < 32 * 32; idx += 256) {
            const int co = idx >> 5, tap = idx & 31;
            if (tap < T) part[((size_t)blockIdx.x * T + tap) * Cout + pass * 32 + co] = (red[idx] + red[1024 + idx]) + (red[2048 + idx] + red[3072 + idx]);
        }
    }
}

}  // namespace

bool conv_first_b16_supported(int Cin, int Cout, int planar) {
    return Cin == 1 && Cout % 32 == 0 && !planar;
}

int launch_conv_first_b16_fwd(const bf16_t* x, const float* w, const float* bias, bf16_t* y, int y_ldc, int N, int D, int H, int W, int Cout,
                              const float* epi_scale, const float* epi_shift, float* stats, hipStream_t s) {
    const int tD = cdiv(D, TD), tH = cdiv(H, TH), tW = cdiv(W, TW);
    hipLaunchKernelGGL(conv_first_b16_fwd_kernel, dim3((unsigned)((size_t)N * tD * tH * tW)), dim3(256), 0, s, x, w, bias, y, y_ldc, N, D, H, W, Cout,
                       epi_scale, epi_shift, stats, tD, tH, tW);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_conv_first_b16_wgrad(const bf16_t* x, const bf16_t* dy, int dy_ldc, float* part, int N, int D, int H, int W, int Cout, int tiles_per_split, int splits,
                                hipStream_t s) {
    const int tD = cdiv(D, TD), tH = cdiv(H, TH), tW = cdiv(W, TW);
    hipLaunchKernelGGL(conv_first_b16_wgrad_kernel, dim3(splits), dim3(256), 0, s, x, dy, dy_ldc, part, N, D, H, W, Cout, tD, tH, tW, tiles_per_split);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
