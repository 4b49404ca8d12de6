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

// ---------------------------------------------------------------- persistent form of the forward (conv_first_b16_pkernel)
// The layer moves 134 MB (25 us of HBM time at cfg 2's size) and needs 3 us of the matrix cores; as one workgroup per 256 voxels it took 82 us: 8192
// workgroups that each fetch and round the weights, decode an index, fill an image element by element and reduce their statistics through two barriers.
// Here 1024 workgroups walk bricks of 4 x 8 x 32 voxels (wave = d-slice, tile = a row of 32 voxels), as csrc/conv_small.hip's conv_first_mfma_kernel does
// in fp32: the weight fragments (two per lane) are built ONCE per workgroup (row r of the tile = channel swap23(r): a lane owns the channels 16 k + 8 g +
// 0..7 -- 16-byte stores), the next brick's halo (4 KB) is requested before the current brick's MFMAs and written to the other LDS image behind them,
// loads and stores are buffer accesses whose range check does the zero padding / the masking, statistics are per-lane running sums and ONE record per
// workgroup (conv_small_b16_stats_parts2).
constexpr int PB_D = 4, PB_H = 8, PB_W = 32, PH_H = PB_H + 2, PH_W = PB_W + 2, PNV = (PB_D + 2) * PH_H * PH_W, PGRID1 = 1024;
__host__ __device__ __forceinline__ int swap23f(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }
typedef unsigned u32x4f __attribute__((ext_vector_type(4)));

template <bool AFF, bool STATS>
__global__ __launch_bounds__(256, 2) void conv_first_b16_pkernel(const bf16_t* __restrict__ x, const float* __restrict__ wgt, const float* __restrict__ bias,
                                                                  bf16_t* __restrict__ y, int y_ldc, int N, int D, int H, int W, int Cout,
                                                                  const float* __restrict__ epi_scale, const float* __restrict__ epi_shift,
                                                                  float* __restrict__ stats, int tilesD, int tilesH, int tilesW, int npass, unsigned nitems) {
    __shared__ __attribute__((aligned(16))) unsigned short xs[2][2048];
    __shared__ float red[4][32][3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, g = lane >> 5;
    const unsigned G = gridDim.x;
    const int pass = (int)(blockIdx.x % (unsigned)npass);      // (G is a multiple of npass: every item of this workgroup has this pass)
    const int cbase = pass * 32;
    constexpr unsigned OOB = 0x80000000u;
    // ---- weight fragments (A operand): row j = channel cbase + swap23(j), k = 16 ks + 8 g + e, rounded to the 16-bit type like every other layer's
    bf16x8 af[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        u16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * ks + 8 * g + e;
            v[e] = k < T ? f2bf(wgt[(size_t)(cbase + swap23f(j)) * T + k]) : (unsigned short)0;
        }
        af[ks] = __builtin_bit_cast(bf16x8, v);
    }
    // ---- patch addresses (B operand): element index of tap k in the image for this lane's voxel of row 0 of the wave's slice
    int pa[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * ks + 8 * g + e, t = k < T ? k : T - 1;
            pa[ks][e] = ((wave + t / 9) * PH_H + (t / 3) % 3) * PH_W + j + t % 3;
        }
    // ---- epilogue constants: registers 8 k + 4 m + e of a tile = channel cbase + 16 k + 8 g + 4 m + e
    f32x4 bq[4], sq[4], hq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = cbase + 16 * (q >> 1) + 8 * g + 4 * (q & 1);
        bq[q] = bias ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (AFF) { sq[q] = *reinterpret_cast<const f32x4*>(epi_scale + c); hq[q] = *reinterpret_cast<const f32x4*>(epi_shift + c); }
    }
    struct Item { int d0, h0, w0, nb; };
    auto decode = [&](unsigned item) {
        unsigned L = item / (unsigned)npass;
        Item it;
        it.w0 = (int)(L % (unsigned)tilesW) * PB_W; L /= (unsigned)tilesW;
        it.h0 = (int)(L % (unsigned)tilesH) * PB_H; L /= (unsigned)tilesH;
        it.d0 = (int)(L % (unsigned)tilesD) * PB_D; it.nb = (int)(L / (unsigned)tilesD);
        return it;
    };
    unsigned short hv[8];
    auto load_halo = [&](const Item& it) {
        // descriptor at the brick's halo origin (possibly in front of the tensor: only valid lanes form addresses from it); zero padding = range check
        const bf16_t* base = x + (((long long)it.nb * D + (it.d0 - 1)) * H + (it.h0 - 1)) * (long long)W + (it.w0 - 1);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, 0x7fffffff, 0x00020000);
        int tv = tid;
        asm volatile("" : "+v"(tv));        // (the 8 x 3 coordinates are recomputed per brick instead of living in 24 registers)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = tv + 256 * k;
            const int zw = idx % PH_W, zh = (idx / PH_W) % PH_H, zd = idx / (PH_W * PH_H);
            const unsigned gd = (unsigned)(it.d0 - 1 + zd), gh = (unsigned)(it.h0 - 1 + zh), gw = (unsigned)(it.w0 - 1 + zw);
            const bool ok = idx < PNV && gd < (unsigned)D && gh < (unsigned)H && gw < (unsigned)W;
            hv[k] = __builtin_amdgcn_raw_buffer_load_b16(rs, ok ? (unsigned)(((zd * H + zh) * W + zw) * 2) : OOB, 0, 0);
        }
    };
    auto store_halo = [&](unsigned short* dst) {
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[tid + 256 * k] = hv[k];
    };
    f32x4 s1[4], s2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { s1[q] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[q] = s1[q]; }
    float cnt = 0.f;

    unsigned item = blockIdx.x;
    int cur = 0;
    if (item < nitems) {
        Item it = decode(item);
        load_halo(it);
        store_halo(xs[0]);
        __syncthreads();
        while (true) {
            const unsigned nxt = item + G;
            const bool more = nxt < nitems;
            Item itn = it;
            if (more) { itn = decode(nxt); load_halo(itn); }       // in flight during this brick's MFMAs
            const unsigned short* img = xs[cur];
            const int d = it.d0 + wave;
            // output descriptor at the brick's first voxel (launcher: four d-planes of the output view < 2^31 bytes); voxels outside the tensor: out-of-range offset
            const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(y + ((((size_t)it.nb * D + it.d0) * H + it.h0) * W + it.w0) * y_ldc + cbase, 0, 0x7fffffff, 0x00020000);
            u16x8 bb[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) bb[0][ks][e] = img[pa[ks][e]];
#pragma unroll
            for (int r = 0; r < PB_H; ++r) {
                if (r + 1 < PB_H) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int e = 0; e < 8; ++e) bb[(r + 1) & 1][ks][e] = img[pa[ks][e] + (r + 1) * PH_W];
                }
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                acc = E3_MFMA16(af[0], __builtin_bit_cast(bf16x8, bb[r & 1][0]), acc, 0, 0, 0);
                acc = E3_MFMA16(af[1], __builtin_bit_cast(bf16x8, bb[r & 1][1]), acc, 0, 0, 0);
                const int h = it.h0 + r, w = it.w0 + j;
                const bool valid = d < D && h < H && w < W;
                const unsigned yoff = valid ? (unsigned)(((((wave * H) + r) * W + j) * y_ldc + 8 * g) * 2) : OOB;
                bf16x4 rprev;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                    if (AFF) {
                        v = v * sq[q] + hq[q];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    } else v = v + bq[q];
                    const bf16x4 rb = __builtin_convertvector(v, bf16x4);            // round to nearest even
                    if (q & 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4f, __builtin_shufflevector(rprev, rb, 0, 1, 2, 3, 4, 5, 6, 7)), y_rs, yoff, 16 * (q >> 1) * 2, 0);
                    else rprev = rb;
                    if (STATS) {
                        f32x4 dv = __builtin_convertvector(rb, f32x4) - bq[q];
#pragma unroll
                        for (int e = 0; e < 4; ++e) dv[e] = valid ? dv[e] : 0.f;
                        s1[q] += dv; s2[q] += dv * dv;
                    }
                }
                if (STATS) cnt += valid ? 1.f : 0.f;
            }
            if (!more) break;
            store_halo(xs[cur ^ 1]);
            __syncthreads();
            cur ^= 1; item = nxt; it = itn;
        }
    }
    if (!STATS) return;
    // ---- statistics: plain sums over the 32 lanes that hold the same channels give the wave's count and mean; every lane then takes its own sum of
    // squares about THAT mean, summed over the lanes; Chan merges over the 4 waves
    float cn = cnt, mn[16], m2[16];
#pragma unroll
    for (int off = 1; off <= 16; off <<= 1) {
        cn += __shfl_xor(cn, off);
#pragma unroll
        for (int e = 0; e < 16; ++e) mn[e] = (off == 1 ? s1[e >> 2][e & 3] : mn[e]) + __shfl_xor(off == 1 ? s1[e >> 2][e & 3] : mn[e], off);
    }
    const float rcn = cn > 0.f ? 1.f / cn : 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        mn[e] *= rcn;
        const float sl = s1[e >> 2][e & 3], ql = s2[e >> 2][e & 3];
        m2[e] = __builtin_fmaf(mn[e], __builtin_fmaf(mn[e], cnt, -2.f * sl), ql);
    }
#pragma unroll
    for (int off = 1; off <= 16; off <<= 1)
#pragma unroll
        for (int e = 0; e < 16; ++e) m2[e] += __shfl_xor(m2[e], off);
    if (j == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = 16 * (e >> 3) + 8 * g + (e & 7);
            red[wave][c][0] = cn; red[wave][c][1] = mn[e]; red[wave][c][2] = fmaxf(m2[e], 0.f);
        }
    }
    __syncthreads();
    if (tid < 32) {
        float c = 0.f, m = 0.f, q = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) welford_merge(c, m, q, red[wv][tid][0], red[wv][tid][1], red[wv][tid][2]);
        const float b = bias ? bias[cbase + tid] : 0.f;
        float* o = stats + ((size_t)(blockIdx.x / (unsigned)npass) * Cout + cbase + tid) * 3;
        o[0] = c; o[1] = b + m; o[2] = q;
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

// enough bricks of 4 x 8 x 32 voxels: the persistent kernel (E3_B16_FIRST_NO_PERSIST=1: A/B switch)
static bool first_b16_persist(int N, int D, int H, int W, int Cout) {
    static const bool off = getenv("E3_B16_FIRST_NO_PERSIST") != nullptr;
    if (off || PGRID1 % (Cout / 32) != 0) return false;
    const long long items = (long long)N * cdiv(D, PB_D) * cdiv(H, PB_H) * cdiv(W, PB_W) * (Cout / 32);
    static const long long min_items = getenv("E3_B16_FIRST_PERSIST_MIN") ? atoll(getenv("E3_B16_FIRST_PERSIST_MIN")) : PGRID1;      // (tests: 1 = every shape the kernel can take)
    return items >= min_items && items < (1ll << 31) && (long long)H * W * 6 * 2 < 0x7fffffffll;
}
int conv_first_b16_stats_parts(int N, int D, int H, int W, int Cout) {      // 0: the brick records of the conv_small_* kernels
    return first_b16_persist(N, D, H, W, Cout) ? PGRID1 / (Cout / 32) : 0;
}

int launch_conv_first_b16_fwd(const bf16_t* x, const float* w, const float* bias, bf16_t* y, int y_ldc, int N, int D, int H, int W, int Cout,
                              const float* epi_scale, const float* epi_shift, float* stats, hipStream_t s) {
    if (first_b16_persist(N, D, H, W, Cout) && (long long)H * W * y_ldc * 4 * 2 < 0x7fffffffll && ((uintptr_t)y & 15) == 0 && y_ldc % 8 == 0) {
        const int tD = cdiv(D, PB_D), tH = cdiv(H, PB_H), tW = cdiv(W, PB_W), npass = Cout / 32;
        const unsigned nitems = (unsigned)((size_t)N * tD * tH * tW * npass);
#define E3_PK_LAUNCH(A_, S_) hipLaunchKernelGGL((conv_first_b16_pkernel<A_, S_>), dim3(PGRID1), dim3(256), 0, s, x, w, bias, y, y_ldc, N, D, H, W, Cout, epi_scale, epi_shift, stats, tD, tH, tW, npass, nitems)
        if (epi_scale) E3_PK_LAUNCH(true, false); else if (stats) E3_PK_LAUNCH(false, true); else E3_PK_LAUNCH(false, false);
#undef E3_PK_LAUNCH
        E3_CHECK_HIP(hipGetLastError());
        return E3_OK;
    }
    E3_REQUIRE(!(stats && first_b16_persist(N, D, H, W, Cout)), E3_ERR_INVALID, "first conv: output view not aligned for the persistent kernel whose record count the caller assumed");
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
