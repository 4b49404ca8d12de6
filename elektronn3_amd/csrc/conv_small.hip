// Direct (VALU) kernels for the two HBM-bound convolutions at the ends of the U-Net:
//   * the first 3x3x3 / 1x3x3 conv, whose input has in_channels (1..7) channels   (unet.py:218-220, SURVEY 8d: 13 FLOP/B)
//   * the final 1x1x1 conv C -> out_channels (<= 16)                                (unet.py:178-180,881,912: 0.9 FLOP/B)
// and their gradients.  Neither has a dense contraction (K = 27 or N = 2), so they do not go to the matrix cores.
#include "kernels.h"

namespace {
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ first conv, forward
// brick = 256 voxels (2x8x16, planar 1x16x16); thread = (channel quad q = tid%8, voxel group g = tid/8) and
// computes 8 voxels x 4 channels.  x halo and the [ci][tap][32] weight slab live in LDS; the 27x4 weights of the
// current input channel are held in registers.
template <int KD, int TD, int TH>
__global__ __launch_bounds__(256) void conv_small_fwd_kernel(const ConvSmallArgs a, int tilesD, int tilesH, int tilesW) {
    // (LW = 20: rows of 18 halo voxels padded to 80 bytes, so that a thread's run of 8 voxels + halo is three aligned ds_read_b128)
    constexpr int TW = 16, PD = KD / 2, LD = TD + 2 * PD, LH = TH + 2, LW = TW + 4, NV = LD * LH * LW, T = KD * 9;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                       // [Cin][NV]
    float* ws = smem + ((a.Cin * NV + 3) & ~3);   // [Cin][T][32]  (current pass)
    const int tid = threadIdx.x, q = tid & 7, g = tid >> 3;
    unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int tw_ = L % tilesW; L /= tilesW; const int th_ = L % tilesH; L /= tilesH; const int td_ = L % tilesD; const int nb = L / tilesD;
    const int d0 = td_ * TD, h0 = th_ * TH, w0 = tw_ * TW;
    const int mtile = blockIdx.x;           // any bijection works for the stats records

    // (four loads in flight per thread, then the LDS writes: one exposed memory round trip per 1024 halo values instead of four)
    for (int i0 = 0; i0 < a.Cin * NV; i0 += 1024) {
        float val[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = i0 + u * 256 + tid;
            const int ci = idx / NV, v = idx % NV;
            const int zw = v % LW, zh = (v / LW) % LH, zd = v / (LW * LH);
            const int gd = d0 + zd - PD, gh = h0 + zh - 1, gw = w0 + zw - 1;
            val[u] = 0.f;
            if (idx < a.Cin * NV && zw < TW + 2 && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W)
                val[u] = a.xs_d ? a.x[(size_t)nb * a.xs_n + (size_t)gd * a.xs_d + (size_t)gh * a.xs_h + (size_t)gw * a.Cin + ci]      // a view inside a larger volume (e3_unet_forward_tile)
                                : a.x[((((size_t)nb * a.D + gd) * a.H + gh) * a.W + gw) * a.Cin + ci];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int idx = i0 + u * 256 + tid; if (idx < a.Cin * NV) xs[idx] = val[u]; }
    }

    // thread = channel quad q x a run of 8 consecutive voxels along w: run g = (row g >> 1 of the brick's TD x TH rows, half g & 1)
    bool vok[8]; size_t voff[8];
    const int rrow = g >> 1, ww0 = 8 * (g & 1), rhh = rrow % TH, rdd = rrow / TH;
    const int rbase = (rdd * LH + rhh) * LW + ww0;             // halo voxel (kd = kh = kw = 0) of the run's first output: a multiple of 4 floats
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int gd = d0 + rdd, gh = h0 + rhh, gw = w0 + ww0 + i;
        vok[i] = gd < a.D && gh < a.H && gw < a.W;
        voff[i] = ((((size_t)nb * a.D + gd) * a.H + gh) * a.W + gw) * a.y_ldc;
    }

    for (int pass = 0; pass * 32 < a.Cout; ++pass) {
        __syncthreads();
        for (int i0 = 0; i0 < a.Cin * T * 32; i0 += 1024) {
            float wv4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = i0 + u * 256 + tid;
                const int c = idx & 31, t = (idx >> 5) % T, ci = (idx >> 5) / T;
                const int co = pass * 32 + c;
                wv4[u] = (idx < a.Cin * T * 32 && co < a.Cout) ? a.w[((size_t)co * a.Cin + ci) * T + t] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int idx = i0 + u * 256 + tid; if (idx < a.Cin * T * 32) ws[idx] = wv4[u]; }
        }
        __syncthreads();
        // the 8 x 4 accumulators as pairs: hipcc packs the <2 x float> fma into v_pk_fma_f32 (two FMAs per lane and issue slot: the scalar form of
        // this loop was FMA-issue bound, 1.8 G v_fma_f32 at 4 cycles per wave); a (kd, kh) row of the run = 10 halo values = three ds_read_b128
        // instead of 24 ds_read_b32
        f32x2 alo[8], ahi[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { alo[i] = f32x2{0.f, 0.f}; ahi[i] = f32x2{0.f, 0.f}; }
        for (int ci = 0; ci < a.Cin; ++ci) {
            f32x4 wr[T];
#pragma unroll
            for (int t = 0; t < T; ++t) wr[t] = *reinterpret_cast<const f32x4*>(ws + (ci * T + t) * 32 + 4 * q);
            const float* xc = xs + ci * NV + rbase;
#pragma unroll
            for (int r9 = 0; r9 < KD * 3; ++r9) {
                const int kd = r9 / 3, kh = r9 % 3;
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(xc + (kd * LH + kh) * LW), x1 = *reinterpret_cast<const f32x4*>(xc + (kd * LH + kh) * LW + 4),
                            x2 = *reinterpret_cast<const f32x4*>(xc + (kd * LH + kh) * LW + 8);
                const float xr[12] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3], x2[0], x2[1], x2[2], x2[3]};
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const f32x4 wv = wr[r9 * 3 + kw];
                    const f32x2 wlo = {wv[0], wv[1]}, whi = {wv[2], wv[3]};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const f32x2 xv = {xr[i + kw], xr[i + kw]};
                        alo[i] = __builtin_elementwise_fma(xv, wlo, alo[i]);
                        ahi[i] = __builtin_elementwise_fma(xv, whi, ahi[i]);
                    }
                }
            }
        }
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{alo[i][0], alo[i][1], ahi[i][0], ahi[i][1]};
        const int co0 = pass * 32 + 4 * q;
        const bool cok = co0 < a.Cout;
        f32x4 bias = {0.f, 0.f, 0.f, 0.f}, es = {1.f, 1.f, 1.f, 1.f}, eh = bias;
        if (cok && a.bias) bias = *reinterpret_cast<const f32x4*>(a.bias + co0);
        const bool aff = a.epi_scale != nullptr;
        if (cok && aff) { es = *reinterpret_cast<const f32x4*>(a.epi_scale + co0); eh = *reinterpret_cast<const f32x4*>(a.epi_shift + co0); }
        f32x4 cnt = {0.f, 0.f, 0.f, 0.f}, sum = cnt;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[i][e] + bias[e];
                if (aff) v = fmaxf(__builtin_fmaf(v, es[e], eh[e]), 0.f);
                acc[i][e] = v;
            }
            if (vok[i] && cok) {
                *reinterpret_cast<f32x4*>(a.y + voff[i] + co0) = acc[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) { cnt[e] += 1.f; sum[e] += acc[i][e]; }
            }
        }
        if (a.stats) {
            // two passes over the registers at WAVE level: (count, sum) of a channel over the 8 lanes x 8 voxels that hold it (lanes with equal q:
            // lane bits 3,4,5), the wave's mean, then the squared deviations from it -- plain adds through the butterflies (Welford merges with their
            // divisions at every level were as long as the convolution loop)
            f32x4 mean, m2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int off = 8; off <= 32; off <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) { cnt[e] += __shfl_xor(cnt[e], off); sum[e] += __shfl_xor(sum[e], off); }
#pragma unroll
            for (int e = 0; e < 4; ++e) mean[e] = cnt[e] > 0.f ? sum[e] / cnt[e] : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (vok[i] && cok)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = acc[i][e] - mean[e]; m2[e] = __builtin_fmaf(d, d, m2[e]); }
#pragma unroll
            for (int off = 8; off <= 32; off <<= 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) m2[e] += __shfl_xor(m2[e], off);
            __syncthreads();   // ws is free now: reuse as scratch [wave][32][3]
            const int wave = tid >> 6, lane = tid & 63;
            if (lane < 8)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float* sc = ws + ((wave * 32) + 4 * lane + e) * 3;
                    sc[0] = cnt[e]; sc[1] = mean[e]; sc[2] = m2[e];
                }
            __syncthreads();
            if (tid < 32 && pass * 32 + tid < a.Cout) {
                float c = 0.f, mn = 0.f, s = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) { const float* sc = ws + (w * 32 + tid) * 3; welford_merge(c, mn, s, sc[0], sc[1], sc[2]); }
                float* o = a.stats + ((size_t)mtile * a.Cout + pass * 32 + tid) * 3;
                o[0] = c; o[1] = mn; o[2] = s;
            }
        }
    }
}

// ------------------------------------------------------------------ first conv (ONE input channel, 3x3x3), forward on the fp32 matrix cores
// The 27 taps (padded to 28) are the K of 14 v_mfma_f32_32x32x2_f32 per 32 voxels x 32 output channels:  Y^T[co][v] = sum_tap W[co][tap] * X[v + tap].
//   A operand = the weights, 14 registers per lane, loaded ONCE per workgroup (row r of the tile holds channel swap23(r): a lane then owns the channels
//               16 k + 8 hf + 0..7 in registers 8 k .. 8 k + 7 -- 16-byte stores, two lanes = 64 contiguous bytes of a voxel's row)
//   B operand = patches: lane (voxel n of a 32-voxel row, k-half kk) reads x[v + tap(2 s + kk)] from the LDS image of the brick's halo (one ds_read_b32 with
//               an immediate row offset per MFMA; 8 KB per brick)
// PERSISTENT: 1024 workgroups walk the bricks (4 x 8 x 32 voxels, wave = d-slice, 8 rows of 32 voxels); the halo of the next brick is requested before
// the current brick's MFMAs and written to the other LDS buffer behind them (nothing a wave waits for inside a brick comes through the vector L1 -- its
// returns are in order, csrc/bf16_conv.hip); lane constants and weights are computed / loaded once.  As a VALU kernel with one workgroup per 256 voxels the
// layer ran at a third of the FMA peak (448 us per Predictor tile whose 822 MB of output need 150 us): per-brick weight fetch, index arithmetic per halo
// element, three barriers; an MFMA form with the same one-brick workgroups was slower still (round-4 notes in DESIGN.md).
// Statistics (training): per-lane running sums of (y - bias) and its square over the lane's voxels, Chan merges across lanes and waves at the end: ONE
// (n, mean, M2) record per workgroup (conv_small_stats_parts2).
constexpr int FB_D = 4, FB_H = 8, FB_W = 32, FH_H = FB_H + 2, FH_W = FB_W + 2, FNV = (FB_D + 2) * FH_H * FH_W, FGRID = 1024;
__host__ __device__ __forceinline__ int swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

template <bool AFF, bool STATS>
__global__ __launch_bounds__(256, 2) void conv_first_mfma_kernel(const ConvSmallArgs a, int tilesD, int tilesH, int tilesW, int npass, unsigned nitems) {
    __shared__ __attribute__((aligned(16))) float xs[2][2048];
    __shared__ float red[4][32][3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, kk = lane >> 5;
    const unsigned G = gridDim.x;
    const int pass = (int)(blockIdx.x % (unsigned)npass);      // (G is a multiple of npass: every item of this workgroup has this pass)
    const int cbase = pass * 32;
    // ---- weights (A operand) and the lane's patch addresses (B operand), one per k-step
    float aw[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int tap = 2 * s + kk;
        aw[s] = tap < 27 ? a.w[(size_t)(cbase + swap23(n)) * 27 + tap] : 0.f;
    }
    const unsigned ab0 = (unsigned)(((wave * FH_H) * FH_W + n) * 4);
    // byte offset of tap 2 s + kk inside the image (tap 27 is the zero-weight pad: any address) -- a select between two immediates, not 14 registers
    auto ab = [&](int s) {
        const int t0 = 2 * s, t1 = 2 * s + 1 < 27 ? 2 * s + 1 : 26;
        const unsigned o0 = (unsigned)((((t0 / 9) * FH_H + (t0 / 3) % 3) * FH_W + t0 % 3) * 4), o1 = (unsigned)((((t1 / 9) * FH_H + (t1 / 3) % 3) * FH_W + t1 % 3) * 4);
        return ab0 + (kk ? o1 : o0);
    };
    // ---- epilogue constants: registers 8 k + 4 m + e of a tile = channel cbase + 16 k + 8 kk + 4 m + e
    f32x4 bq[4], sq[4], hq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = cbase + 16 * (q >> 1) + 8 * kk + 4 * (q & 1);
        bq[q] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (AFF) { sq[q] = *reinterpret_cast<const f32x4*>(a.epi_scale + c); hq[q] = *reinterpret_cast<const f32x4*>(a.epi_shift + c); }
    }
    // ---- staging plan: thread -> 8 halo elements idx = tid + 256 k (consecutive threads = consecutive w: coalesced rows)
    const long long sd = a.xs_d ? a.xs_d : (long long)a.H * a.W, sh = a.xs_d ? a.xs_h : (long long)a.W, sn = a.xs_d ? a.xs_n : (long long)a.D * a.H * a.W;
    constexpr unsigned OOB = 0x80000000u;
    struct Item { int d0, h0, w0, nb; };
    auto decode = [&](unsigned item) {
        unsigned L = item / (unsigned)npass;
        Item it;
        it.w0 = (int)(L % (unsigned)tilesW) * FB_W; L /= (unsigned)tilesW;
        it.h0 = (int)(L % (unsigned)tilesH) * FB_H; L /= (unsigned)tilesH;
        it.d0 = (int)(L % (unsigned)tilesD) * FB_D; it.nb = (int)(L / (unsigned)tilesD);
        return it;
    };
    float hv[8];
    auto load_halo = [&](const Item& it) {
        // descriptor at the brick's halo origin (possibly in front of the tensor: only valid lanes form addresses from it); zero padding = range check
        const float* base = a.x + (long long)it.nb * sn + (long long)(it.d0 - 1) * sd + (long long)(it.h0 - 1) * sh + (it.w0 - 1);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
        int tv = tid;
        asm volatile("" : "+v"(tv));        // (keeps the compiler from hoisting the 8 x 3 coordinates out of the brick loop: a few dozen VALU operations per brick against 24 registers)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = tv + 256 * k;
            const int zw = idx % FH_W, zh = (idx / FH_W) % FH_H, zd = idx / (FH_W * FH_H);
            const unsigned gd = (unsigned)(it.d0 - 1 + zd), gh = (unsigned)(it.h0 - 1 + zh), gw = (unsigned)(it.w0 - 1 + zw);
            const bool ok = idx < FNV && gd < (unsigned)a.D && gh < (unsigned)a.H && gw < (unsigned)a.W;
            hv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (unsigned)((zd * sd + zh * sh + zw) * 4) : OOB, 0, 0));
        }
    };
    auto store_halo = [&](float* dst) {
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[tid + 256 * k] = hv[k];
    };
    // running statistics of this lane's 16 channels: sums of (y - bias) and its square over the lane's valid voxels
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
            const char* img = reinterpret_cast<const char*>(xs[cur]);
            const int d = it.d0 + wave;
            // output descriptor at the brick's first voxel (launcher: four d-planes of the output view < 2^31 bytes); voxels outside the tensor: out-of-range offset
            // (channel-chunked output, ConvSmallArgs::y_chunk: a voxel's row is the 8 channels of one chunk plane; channel cbase + 16 (q >> 1) + 8 kk + 4 (q & 1)
            // lives in plane cbase / 8 + 2 (q >> 1) + kk at float 4 (q & 1): the plane offsets are part of the LANE offsets)
            const size_t ychk = a.y_chunk;
            const int ys = ychk ? 8 : a.y_ldc;
            const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(a.y + ((((size_t)it.nb * a.D + it.d0) * a.H + it.h0) * a.W + it.w0) * ys +
                                                                                  (ychk ? (size_t)(cbase >> 3) * ychk : (size_t)cbase), 0, 0x7fffffff, 0x00020000);
            const unsigned qstep = ychk ? (unsigned)(2 * ychk * 4) : 64u, kstep = ychk ? (unsigned)(ychk * 4) : 32u;      // bytes between q >> 1 = 0, 1 / between kk = 0, 1
            // patches of a row: 14 LDS reads, requested one row ahead of the 14 MFMAs that consume them (one accumulator chain: a dependent
            // v_mfma_f32_32x32x2_f32 issues right behind its predecessor's 16 passes)
            float bb[2][14];
#pragma unroll
            for (int s = 0; s < 14; ++s) bb[0][s] = *reinterpret_cast<const float*>(img + ab(s));
#pragma unroll
            for (int r = 0; r < FB_H; ++r) {
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
                if (r + 1 < FB_H) {
#pragma unroll
                    for (int s = 0; s < 14; ++s) bb[(r + 1) & 1][s] = *reinterpret_cast<const float*>(img + ab(s) + (r + 1) * FH_W * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 14; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[s], bb[r & 1][s], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                {
                    const int h = it.h0 + r, w = it.w0 + n;
                    const bool valid = d < a.D && h < a.H && w < a.W;
                    const unsigned yoff = valid ? (unsigned)((((wave * a.H) + r) * a.W + n) * ys * 4) + (unsigned)kk * kstep : OOB;       // relative to the brick's first voxel
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                        v = v + bq[q];
                        if (AFF) {
                            v = v * sq[q] + hq[q];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
#if !defined(E3_FIRST_SOFFSET)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), y_rs, valid ? yoff + (unsigned)(q >> 1) * qstep : OOB, (4 * (q & 1)) * 4, 0);
#elif E3_FIRST_SOFFSET == 1      // developer builds (tools/repro_first_soffset.sh): the wave-uniform part (q >> 1) * qstep in the SCALAR offset operand
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), y_rs, valid ? yoff : OOB, (unsigned)(q >> 1) * qstep + (4 * (q & 1)) * 4, 0);
#else                            // ... and the lane-dependent plane offset kk * kstep as well (round 5's dropped form: the compiler wraps the store in a waterfall loop)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), y_rs, valid ? yoff - (unsigned)kk * kstep : OOB,
                                                               (unsigned)kk * kstep + (unsigned)(q >> 1) * qstep + (4 * (q & 1)) * 4, 0);
#endif
                        if (STATS) {
                            f32x4 dv = v - bq[q];
#pragma unroll
                            for (int e = 0; e < 4; ++e) dv[e] = valid ? dv[e] : 0.f;
                            s1[q] += dv; s2[q] += dv * dv;
                        }
                    }
                    if (STATS) cnt += valid ? 1.f : 0.f;
                }
            }
            if (!more) break;
            store_halo(xs[cur ^ 1]);
            __syncthreads();
            cur ^= 1; item = nxt; it = itn;
        }
    }
    if (!STATS) return;
    // ---- statistics: plain sums over the 32 lanes that hold the same channels give the wave's count and mean; every lane then takes its own sum of
    // squares about THAT mean (its sums cover a few dozen voxels: no cancellation to speak of), summed over the lanes; Chan merges over the 4 waves
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
        m2[e] = __builtin_fmaf(mn[e], __builtin_fmaf(mn[e], cnt, -2.f * sl), ql);        // sum (dv - mean)^2 over the lane's voxels
    }
#pragma unroll
    for (int off = 1; off <= 16; off <<= 1)
#pragma unroll
        for (int e = 0; e < 16; ++e) m2[e] += __shfl_xor(m2[e], off);
    if (n == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = 16 * (e >> 3) + 8 * kk + (e & 7);
            red[wave][c][0] = cn; red[wave][c][1] = mn[e]; red[wave][c][2] = fmaxf(m2[e], 0.f);
        }
    }
    __syncthreads();
    if (tid < 32) {
        float c = 0.f, m = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) welford_merge(c, m, q, red[w][tid][0], red[w][tid][1], red[w][tid][2]);
        const float b = a.bias ? a.bias[cbase + tid] : 0.f;
        float* o = a.stats + ((size_t)(blockIdx.x / (unsigned)npass) * a.Cout + cbase + tid) * 3;
        o[0] = c; o[1] = b + m; o[2] = q;
    }
}

// ------------------------------------------------------------------ first conv, weight gradient
// thread = (tap slot t = tid/8 (27 of 32 used), channel quad cq = tid%8): dW[co][ci][t] partial over the voxels of
// the bricks owned by this block.  part layout [splits][T][Cout][Cin].
// MFMA (one input channel, 3x3x3, Cout % 32 == 0): the sum over the brick's voxels is the K of v_mfma_f32_32x32x2_f32 -- dW[co][tap] =
// sum_v dY[v][co] * X[v + tap], A = dY^T (lane = channel, straight LDS rows), B = patches (lane = tap); wave w takes the rows 4w..4w+3 of
// the brick, the four partial tiles meet in LDS.  The VALU form spends 2 LDS reads per 4 FMAs: ~90 us of a 180 us kernel whose traffic
// (the fused BN backward reads the raw output and the incoming gradient) needs ~100 us.
template <int KD, int TD, int TH, bool MFMA>
__global__ __launch_bounds__(256) void conv_small_wgrad_kernel(const float* __restrict__ x, int Cin, const float* __restrict__ dy, int dy_ldc,
                                                               float* __restrict__ part, int N, int D, int H, int W, int Cout,
                                                               int tilesD, int tilesH, int tilesW, int tiles_per_split, const SmallWgradFuse f) {
    // f.x1 != nullptr: `dy` is not given; it is the BN + ReLU backward of this conv's own output, computed while the brick is staged
    //   dy = gamma*invstd * (dz - c1 - xhat*c2),  dz = g * (x1*scale + shift > 0),  xhat = (x1 - mean)*invstd
    // (the APPLY pass of bn_bwd_kernel, same expressions) from the raw conv output x1 and the incoming gradient g: the first conv has
    // no data gradient, so its dy has no other consumer and is never written; the column sums of dy (= conv-bias gradient) go to
    // f.biaspart [blocks][Cout].
    constexpr int TW = 16, PD = KD / 2, LD = TD + 2 * PD, LH = TH + 2, LW = TW + 2, NV = LD * LH * LW, T = KD * 9;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                 // [NV]        one input channel
    float* gs = smem + ((NV + 3) & ~3);   // [256][32]   dy brick, one pass of 32 output channels
    const int tid = threadIdx.x, cq = tid & 7, t = tid >> 3;
    const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
    const int toff = (kd * LH + kh) * LW + kw;
    const int ntiles = N * tilesD * tilesH * tilesW;
    const int tile0 = blockIdx.x * tiles_per_split;
    const float fslope = f.act.get();
    for (int pass = 0; pass * 32 < Cout; ++pass) {
        // per-channel constants of the fused BN backward for the channel quad this thread stages (tid & 7 is fixed across its pieces)
        f32x4 fsc = {1.f, 1.f, 1.f, 1.f}, fsh = {0.f, 0.f, 0.f, 0.f}, fmu = fsh, fis = fsc, fgi = fsc, fc1 = fsh, fc2 = fsh, bsum = fsh;
        if (f.x1 && pass * 32 + 4 * (tid & 7) < Cout) {
            const int c0 = pass * 32 + 4 * (tid & 7);
            fsc = *reinterpret_cast<const f32x4*>(f.scale + c0); fsh = *reinterpret_cast<const f32x4*>(f.shift + c0);
            fmu = *reinterpret_cast<const f32x4*>(f.mean + c0); fis = *reinterpret_cast<const f32x4*>(f.invstd + c0);
            fc1 = *reinterpret_cast<const f32x4*>(f.coef + c0); fc2 = *reinterpret_cast<const f32x4*>(f.coef + Cout + c0);
            const f32x4 gm = *reinterpret_cast<const f32x4*>(f.gamma + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) fgi[e] = gm[e] * fis[e];
        }
        for (int ci = 0; ci < Cin; ++ci) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            f32x16 macc;
            if constexpr (MFMA) {
#pragma unroll
                for (int e = 0; e < 16; ++e) macc[e] = 0.f;
            }
            // The brick's operands travel global -> registers -> LDS; the loads of brick b + 1 are issued BEHIND the barrier that publishes brick b's LDS image
            // and fly during its MFMAs (as one load / wait / compute chain per brick the kernel moved 3.6 GB/s per workgroup: 146 us for 545 MB).
            constexpr int XI = (NV + 255) / 256;
            float xh[XI];
            f32x4 xv[8], gv[8]; bool ok[8]; unsigned eidx[8];
            auto issue = [&](int tile) {
                int L = tile;
                const int tw_ = L % tilesW; L /= tilesW; const int th_ = L % tilesH; L /= tilesH; const int td_ = L % tilesD; const int nb = L / tilesD;
                const int d0 = td_ * TD, h0 = th_ * TH, w0 = tw_ * TW;
#pragma unroll
                for (int i = 0; i < XI; ++i) {
                    const int v = tid + 256 * i;
                    const int zw = v % LW, zh = (v / LW) % LH, zd = v / (LW * LH);
                    const int gd = d0 + zd - PD, gh = h0 + zh - 1, gw = w0 + zw - 1;
                    xh[i] = 0.f;
                    if (v < NV && gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W)
                        xh[i] = x[((((size_t)nb * D + gd) * H + gh) * W + gw) * Cin + ci];
                }
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int idx = tid + 256 * it;
                    const int v = idx >> 3, qq = idx & 7;
                    const int ww = v & 15, hh = (v >> 4) % TH, dd = (v >> 4) / TH;
                    const int gd = d0 + dd, gh = h0 + hh, gw = w0 + ww;
                    ok[it] = gd < D && gh < H && gw < W && pass * 32 + 4 * qq < Cout;
                    const size_t vox = ok[it] ? (((size_t)nb * D + gd) * H + gh) * W + gw : 0;
                    const int c0 = ok[it] ? pass * 32 + 4 * qq : 0;
                    eidx[it] = (unsigned)(vox * Cout + c0);
                    if (f.x1) { xv[it] = *reinterpret_cast<const f32x4*>(f.x1 + vox * f.x1_ldc + c0); gv[it] = *reinterpret_cast<const f32x4*>(f.g + vox * f.g_ldc + c0); }
                    else gv[it] = *reinterpret_cast<const f32x4*>(dy + vox * dy_ldc + c0);
                }
            };
            const int tile_end = tile0 + tiles_per_split < ntiles ? tile0 + tiles_per_split : ntiles;
            if (tile0 < tile_end) issue(tile0);
            for (int tile = tile0; tile < tile_end; ++tile) {
                __syncthreads();                      // the previous brick's MFMAs are done with the LDS images
#pragma unroll
                for (int i = 0; i < XI; ++i)
                    if (tid + 256 * i < NV) xs[tid + 256 * i] = xh[i];
                {
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int idx = tid + 256 * it;
                        const int v = idx >> 3, qq = idx & 7;
                        f32x4 val = {0.f, 0.f, 0.f, 0.f};
                        if (ok[it]) {
                            if (f.x1) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float z = __builtin_fmaf(xv[it][e], fsc[e], fsh[e]);
                                    const float dz = act_bwd(z, gv[it][e], act_slope_at(f.act, fslope, eidx[it] + e));
                                    const float xh2 = (xv[it][e] - fmu[e]) * fis[e];
                                    val[e] = fgi[e] * (dz - fc1[e] - xh2 * fc2[e]);
                                    if (ci == 0) bsum[e] += val[e];
                                }
                            } else val = gv[it];
                        }
                        *reinterpret_cast<f32x4*>(gs + v * 32 + 4 * qq) = val;
                    }
                }
                __syncthreads();
                if (tile + 1 < tile_end) issue(tile + 1);      // in flight during this brick's MFMAs
                if constexpr (MFMA) {
                    const int lane = tid & 63, wave = tid >> 6, jj = lane & 31, kk = lane >> 5;
                    const int tj = jj < T ? jj : T - 1;                                  // (columns 27..31 of the tap tile are never stored)
                    const int tofs = ((tj / 9) * LH + (tj / 3) % 3) * LW + tj % 3 + kk;
#pragma unroll
                    for (int rw = 0; rw < 4; ++rw) {
                        const int row = 4 * wave + rw, dd = row / TH, hh = row % TH;     // 16 voxels of one w-row = 8 k-steps
                        const float* xr = xs + (dd * LH + hh) * LW + tofs;
                        const float* gr = gs + (row * 16 + kk) * 32 + jj;
#pragma unroll
                        for (int st = 0; st < 8; ++st) macc = __builtin_amdgcn_mfma_f32_32x32x2f32(gr[st * 64], xr[2 * st], macc, 0, 0, 0);
                    }
                } else if (t < T) {
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
            if constexpr (MFMA) {     // lane holds column tap = lane & 31, rows co = (e&3) + 8*(e>>2) + 4*(lane>>5); the waves' tiles meet in `gs`
                const int lane = tid & 63, wave = tid >> 6;
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 16; ++e) gs[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = macc[e];
                __syncthreads();
                for (int idx = tid; idx < 32 * 32; idx += 256) {
                    const int co = idx >> 5, tap = idx & 31;
                    if (tap < T) part[(((size_t)blockIdx.x * T + tap) * Cout + pass * 32 + co) * Cin + ci] = (gs[idx] + gs[1024 + idx]) + (gs[2048 + idx] + gs[3072 + idx]);
                }
            } else if (t < T)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = pass * 32 + 4 * cq + e;
                    if (co < Cout) part[(((size_t)blockIdx.x * T + t) * Cout + co) * Cin + ci] = acc[e];
                }
        }
        if (f.x1) {   // conv-bias gradient partial of this block: threads with equal (tid & 7) staged the same channel quad
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) gs[tid * 4 + e] = bsum[e];
            __syncthreads();
            if (tid < 32 && pass * 32 + tid < Cout) {
                const int qq = tid >> 2, e = tid & 3;
                float sum = 0.f;
                for (int m = 0; m < 32; ++m) sum += gs[(qq + 8 * m) * 4 + e];
                f.biaspart[(size_t)blockIdx.x * Cout + pass * 32 + tid] = sum;
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ final 1x1x1 conv, forward (+ optional softmax)
// LOSS: the criterion of the training example (weighted CE + Dice, loss.hip) is evaluated on the logits while they are in registers --
// the voxel's softmax, -log p[target], and the per-class Dice sums go into per-thread accumulators and leave as ONE partial row per
// workgroup in the layout of ce_dice_fwd_kernel, so ce_dice_finalize_kernel / e3_ce_dice_bwd work unchanged ("loss on
// device fused with conv_final": the logits are not re-read and the separate pass over logits + target disappears).
struct HeadLossArgs { const long long* target; const float* w; float* partial; };
// Box form (e3_unet_forward_tile): only the voxels [d0, d0 + bd) x [h0, h0 + bh) x [w0, w0 + bw) of the (D, H, W) grid are read and their
// results go to a view of a larger NCDHW volume: voxel (bz, by, bx) of the box, channel co, sample n -> y + n ys_n + co ys_c + bz ys_d + by ys_h + bx
struct HeadBox { int on, bd, bh, bw, d0, h0, w0, D, H, W; long long ys_n, ys_c, ys_d, ys_h; };
template <int COUT, bool LOSS>
__global__ void conv_final_fwd_kernel(const float* __restrict__ a, int a_ldc, int C, const float* __restrict__ w,
                                      const float* __restrict__ bias, float* __restrict__ y, size_t S, int N, int lpv, int softmax,
                                      const float* __restrict__ pro_scale, const float* __restrict__ pro_shift, ActArg pro_act, HeadLossArgs la, HeadBox box) {
    constexpr int NV = 2 + 3 * COUT;
    float lacc[LOSS ? NV : 1];
#pragma unroll
    for (int i = 0; i < (LOSS ? NV : 1); ++i) lacc[i] = 0.f;
    const float pro_slope = pro_act.get();
    // pro_scale/pro_shift: `a` is the RAW output of the last conv; its BatchNorm + ReLU, a := relu(a*scale + shift), is applied while
    // loading (same expression as bn_relu_apply_kernel) -- the last activation of the network is never written or re-read
    // lpv (1,2,4,8) consecutive lanes share one voxel; each walks every lpv-th channel quad
    const int Q = C >> 2;
    const size_t total = box.on ? (size_t)N * box.bd * box.bh * box.bw : (size_t)N * S;
    // loop index -> voxel of the input grid (the identity without a box)
    auto src_voxel = [&](size_t i) -> size_t {
        if (!box.on) return i;
        unsigned r = (unsigned)i;
        const unsigned bx = r % (unsigned)box.bw; r /= (unsigned)box.bw;
        const unsigned by = r % (unsigned)box.bh; r /= (unsigned)box.bh;
        const unsigned bz = r % (unsigned)box.bd; const unsigned n = r / (unsigned)box.bd;
        return (((size_t)n * box.D + box.d0 + bz) * box.H + box.h0 + by) * box.W + box.w0 + bx;
    };
    const int sub = threadIdx.x % lpv;
    const size_t vpb = blockDim.x / lpv;
    // U voxels per thread and iteration, their loads issued together; a workgroup's U x vpb voxels of an iteration are one contiguous run.
    // `a` is streamed once: non-temporal.  Same summation order per voxel.
    constexpr int U = 4;
    const bool one = Q <= lpv;          // at most one channel quad per lane (the usual head: C = 32, eight lanes per voxel)
    const size_t ustride = one ? vpb : 0, chunk = one ? vpb * U : vpb;
    const size_t vend = (total + chunk - 1) / chunk * chunk;
    for (size_t v0 = blockIdx.x * chunk + threadIdx.x / lpv; v0 < vend; v0 += gridDim.x * chunk) {
        float acc[U][COUT];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int co = 0; co < COUT; ++co) acc[u][co] = 0.f;
        if (one) {
            f32x4 av[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t v = v0 + u * ustride;
                av[u] = (v < total && sub < Q) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a + src_voxel(v) * a_ldc + 4 * sub)) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (sub < Q) {
                f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                if (pro_scale) { sc = *reinterpret_cast<const f32x4*>(pro_scale + 4 * sub); sh = *reinterpret_cast<const f32x4*>(pro_shift + 4 * sub); }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const size_t v = v0 + u * ustride;
                    if (v >= total) continue;
                    if (pro_scale) {
                        const size_t vs = src_voxel(v);
#pragma unroll
                        for (int e = 0; e < 4; ++e) av[u][e] = act_fwd(__builtin_fmaf(av[u][e], sc[e], sh[e]), act_slope_at(pro_act, pro_slope, (unsigned)(vs * C + 4 * sub + e)));
                    }
#pragma unroll
                    for (int co = 0; co < COUT; ++co) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + co * C + 4 * sub);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[u][co] = __builtin_fmaf(av[u][e], wv[e], acc[u][co]);
                    }
                }
            }
        } else if (v0 < total) {
            const size_t vs0 = src_voxel(v0);
            for (int q = sub; q < Q; q += lpv) {
                f32x4 xv = *reinterpret_cast<const f32x4*>(a + vs0 * a_ldc + 4 * q);
                if (pro_scale) {
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(pro_scale + 4 * q), sh = *reinterpret_cast<const f32x4*>(pro_shift + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[e] = act_fwd(__builtin_fmaf(xv[e], sc[e], sh[e]), act_slope_at(pro_act, pro_slope, (unsigned)(vs0 * C + 4 * q + e)));
                }
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + co * C + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[0][co] = __builtin_fmaf(xv[e], wv[e], acc[0][co]);
                }
            }
        }
        // per-voxel epilogue: bias, [criterion sums], [softmax], NCDHW stores.  `lg`: the voxel's logits, v: its index.
        auto finish = [&](float (&lg)[COUT], size_t v) {
            const size_t n = v / S, sp = v % S;
#pragma unroll
            for (int co = 0; co < COUT; ++co) lg[co] += bias ? bias[co] : 0.f;
            if (LOSS) {          // same expressions as ce_dice_fwd_kernel (softmax_c) on the values that are stored below
                float m = lg[0];
#pragma unroll
                for (int co = 1; co < COUT; ++co) m = fmaxf(m, lg[co]);
                float pr[COUT], sm = 0.f;
#pragma unroll
                for (int co = 0; co < COUT; ++co) { pr[co] = __expf(lg[co] - m); sm += pr[co]; }
                const float inv = 1.f / sm, lse = m + __logf(sm);
                const int t = (int)la.target[v];
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    const float pc = pr[co] * inv;
                    const bool is = t == co;
                    lacc[2 + co] += is ? pc : 0.f;
                    lacc[2 + COUT + co] += pc;
                    lacc[2 + 2 * COUT + co] += is ? 1.f : 0.f;
                    if (is) { const float wc = la.w ? la.w[co] : 1.f; lacc[0] += wc * (lse - lg[co]); lacc[1] += wc; }
                }
            }
            if (softmax) {
                float m = lg[0];
#pragma unroll
                for (int co = 1; co < COUT; ++co) m = fmaxf(m, lg[co]);
                float sm = 0.f;
#pragma unroll
                for (int co = 0; co < COUT; ++co) { lg[co] = __expf(lg[co] - m); sm += lg[co]; }
                const float inv = 1.f / sm;
#pragma unroll
                for (int co = 0; co < COUT; ++co) lg[co] *= inv;
            }
            if (box.on) {
                unsigned r = (unsigned)v;
                const unsigned bx = r % (unsigned)box.bw; r /= (unsigned)box.bw;
                const unsigned by = r % (unsigned)box.bh; r /= (unsigned)box.bh;
                const unsigned bz = r % (unsigned)box.bd; const unsigned nn = r / (unsigned)box.bd;
                float* const yo = y + (long long)nn * box.ys_n + (long long)bz * box.ys_d + (long long)by * box.ys_h + bx;
#pragma unroll
                for (int co = 0; co < COUT; ++co) yo[(long long)co * box.ys_c] = lg[co];
                return;
            }
#pragma unroll
            for (int co = 0; co < COUT; ++co) y[(n * COUT + co) * S + sp] = lg[co];
        };
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u > 0 && !one) break;
            for (int off = 1; off < lpv; off <<= 1)
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[u][co] += __shfl_xor(acc[u][co], off);
        }
        if (one && lpv >= U) {
            // after the butterflies every lane of a voxel group holds all U voxels' logits: lane `sub` finishes voxel u = sub -- ONE instruction
            // stream for the U voxels (the serial form below issues it U times with an eighth of the lanes active, which made the head with the
            // criterion instruction-bound: 125 us instead of 77)
            float lg[COUT];
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                lg[co] = acc[0][co];
#pragma unroll
                for (int u = 1; u < U; ++u) lg[co] = sub == u ? acc[u][co] : lg[co];
            }
            const size_t v = v0 + (size_t)sub * ustride;
            if (sub < U && v < total) finish(lg, v);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u > 0 && !one) break;
                const size_t v = v0 + u * ustride;
                if (v < total && sub == 0) finish(acc[u], v);
            }
        }
    }
    if (LOSS) {              // one partial row per workgroup, fixed order (wave butterflies, then the 4 waves in turn)
        __shared__ float red[4][NV];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float sv = lacc[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sv += __shfl_xor(sv, o);
            if (lane == 0) red[wave][i] = sv;
        }
        __syncthreads();
        if ((int)threadIdx.x < NV) la.partial[(size_t)blockIdx.x * NV + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

// ------------------------------------------------------------------ final conv, backward: da, dW/db partials
template <int COUT>
__global__ __launch_bounds__(256) void conv_final_bwd_kernel(const float* __restrict__ a, int a_ldc, int C, const float* __restrict__ w,
                                                             const float* __restrict__ dy, float* __restrict__ da, int da_ldc,
                                                             float* __restrict__ part, size_t S, int N,
                                                             const float* __restrict__ pro_scale, const float* __restrict__ pro_shift, ActArg pro_act) {
    __shared__ float red[256][4];
    const float pro_slope = pro_act.get();
    const int Q = C >> 2;
    const int BT = (256 / Q) * Q;
    const size_t total = (size_t)N * S * Q;
    const int tid = threadIdx.x;
    const int q = tid % Q;
    f32x4 wv[COUT], dwacc[COUT];
    float dbacc[COUT];
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (pro_scale) { sc = *reinterpret_cast<const f32x4*>(pro_scale + 4 * q); sh = *reinterpret_cast<const f32x4*>(pro_shift + 4 * q); }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        wv[co] = *reinterpret_cast<const f32x4*>(w + co * C + 4 * q);
        dwacc[co] = f32x4{0.f, 0.f, 0.f, 0.f}; dbacc[co] = 0.f;
    }
    // two items per iteration with all their loads issued first (`a` streamed once: non-temporal)
    const size_t istride = (size_t)gridDim.x * BT;
    for (size_t i0 = (size_t)blockIdx.x * BT + tid; tid < BT && i0 < total; i0 += 2 * istride) {
        f32x4 av[2]; float g[2][COUT]; size_t vv[2]; bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t i = i0 + u * istride;
            ok[u] = i < total;
            const size_t v = ok[u] ? i / Q : 0;
            vv[u] = v;
            const size_t n = v / S, sp = v % S;
            av[u] = ok[u] ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a + v * a_ldc + 4 * q)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int co = 0; co < COUT; ++co) g[u][co] = ok[u] ? dy[(n * COUT + co) * S + sp] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!ok[u]) continue;
            const size_t v = vv[u];
            if (pro_scale) {
#pragma unroll
                for (int e = 0; e < 4; ++e) av[u][e] = act_fwd(__builtin_fmaf(av[u][e], sc[e], sh[e]), act_slope_at(pro_act, pro_slope, (unsigned)(v * C + 4 * q + e)));
            }
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = __builtin_fmaf(g[u][co], wv[co][e], o[e]); dwacc[co][e] = __builtin_fmaf(g[u][co], av[u][e], dwacc[co][e]); }
                if (q == 0) dbacc[co] += g[u][co];
            }
            if (da) *reinterpret_cast<f32x4*>(da + v * da_ldc + 4 * q) = o;     // da == nullptr: the consumer recomputes it (BnBwdArgs::head_dy)
        }
    }
    // block reduce, one output row (co) at a time
    const int pstride = COUT * C + COUT;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) red[tid][e] = tid < BT ? dwacc[co][e] : 0.f;
        __syncthreads();
        for (int t = tid; t < Q * 4; t += 256) {
            const int e = t & 3, qq = t >> 2;
            float acc = 0.f;
            for (int k = qq; k < BT; k += Q) acc += red[k][e];
            part[(size_t)blockIdx.x * pstride + co * C + 4 * qq + e] = acc;
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

}  // namespace

static void small_brick(int planar, int& TD, int& TH) { if (planar) { TD = 1; TH = 16; } else { TD = 2; TH = 8; } }

int conv_small_stats_parts(int N, int D, int H, int W, int planar) {
    int TD, TH; small_brick(planar, TD, TH);
    return N * cdiv(D, TD) * cdiv(H, TH) * cdiv(W, 16);
}
// one input channel, 3x3x3, 32-channel output tiles, enough bricks of 4 x 8 x 32 voxels: the persistent matrix-core kernel (E3_FIRST_NO_MFMA=1: A/B switch)
static bool first_mfma(int N, int D, int H, int W, int planar, int Cin, int Cout) {
    static const bool off = getenv("E3_FIRST_NO_MFMA") != nullptr;
    if (off || planar || Cin != 1 || Cout % 32 != 0 || FGRID % (Cout / 32) != 0) return false;
    const long long items = (long long)N * cdiv(D, FB_D) * cdiv(H, FB_H) * cdiv(W, FB_W) * (Cout / 32);
    static const long long min_items = getenv("E3_FIRST_MFMA_MIN") ? atoll(getenv("E3_FIRST_MFMA_MIN")) : FGRID;      // (tests: 1 = every shape the kernel can take)
    // (32-bit buffer offsets: four d-planes of a packed output and six of the input must stay below 2^31 bytes -- decided HERE, from the shape alone, so that
    // the statistics sizing and the launcher agree; what is left to the launcher are properties of a caller's VIEW: alignment, a larger channel stride)
    const bool planes_fit = (long long)H * W * Cout * 4 * 4 < 0x7fffffffll && (long long)H * W * 6 * 4 < 0x7fffffffll;
    return items >= min_items && items < (1ll << 31) && planes_fit;
}
bool conv_first_chunk_ok(int N, int D, int H, int W, int planar, int Cin, int Cout) {
    return first_mfma(N, D, H, W, planar, Cin, Cout) && chunked_layout_ok((size_t)N * D * H * W, Cout);
}
int conv_small_stats_parts2(int N, int D, int H, int W, int planar, int Cin, int Cout) {
    return first_mfma(N, D, H, W, planar, Cin, Cout) ? FGRID / (Cout / 32) : conv_small_stats_parts(N, D, H, W, planar);
}

int launch_conv_small_fwd(ConvSmallArgs a, hipStream_t s) {
    E3_REQUIRE(a.Cin >= 1 && a.Cin < 8, E3_ERR_UNSUPPORTED, "direct conv handles 1..7 input channels");
    E3_REQUIRE(a.Cout % 4 == 0 && a.y_ldc % 4 == 0, E3_ERR_UNSUPPORTED, "output channels must be a multiple of 4");
    // (32-bit buffer offsets: four d-planes of the output view and six of the input view must stay below 2^31 bytes; larger views -- and misaligned outputs --
    // take the one-brick kernel below, except with statistics, whose record count the caller has already sized for this kernel)
    const bool fits = ((uintptr_t)a.y & 15) == 0 && (long long)a.H * a.W * a.y_ldc * 4 * 4 < 0x7fffffffll && (a.xs_d ? a.xs_d : (long long)a.H * a.W) * 6 * 4 < 0x7fffffffll;
    E3_REQUIRE(!a.y_chunk || (fits && !a.stats && a.y_chunk == (size_t)a.N * a.D * a.H * a.W * 8 && conv_first_chunk_ok(a.N, a.D, a.H, a.W, a.planar, a.Cin, a.Cout)), E3_ERR_INVALID,
               "first conv: bad channel-chunked output (persistent matrix-core kernel without statistics only)");
    if (first_mfma(a.N, a.D, a.H, a.W, a.planar, a.Cin, a.Cout) && (fits || a.stats)) {
        E3_REQUIRE(fits, E3_ERR_UNSUPPORTED, "first conv with statistics: view too large or misaligned for the persistent kernel (32-bit buffer offsets)");
        const int tD = cdiv(a.D, FB_D), tH = cdiv(a.H, FB_H), tW = cdiv(a.W, FB_W), npass = a.Cout / 32;
        const unsigned nitems = (unsigned)((size_t)a.N * tD * tH * tW * npass);
        if (a.epi_scale) hipLaunchKernelGGL((conv_first_mfma_kernel<true, false>), dim3(FGRID), dim3(256), 0, s, a, tD, tH, tW, npass, nitems);
        else if (a.stats) hipLaunchKernelGGL((conv_first_mfma_kernel<false, true>), dim3(FGRID), dim3(256), 0, s, a, tD, tH, tW, npass, nitems);
        else hipLaunchKernelGGL((conv_first_mfma_kernel<false, false>), dim3(FGRID), dim3(256), 0, s, a, tD, tH, tW, npass, nitems);
        E3_CHECK_HIP(hipGetLastError());
        return E3_OK;
    }
    int TD, TH; small_brick(a.planar, TD, TH);
    const int tD = cdiv(a.D, TD), tH = cdiv(a.H, TH), tW = cdiv(a.W, 16);
    const int KD = a.planar ? 1 : 3;
    const int NV = (TD + (a.planar ? 0 : 2)) * (TH + 2) * 20;
    // the weight slab is reused as the statistics scratch [4 waves][32][3]: planar convs with one input channel have only
    // 288 floats of weights (this under-allocation corrupted the BN statistics of a planar first conv with > 16 channels)
    const int wslab = a.Cin * KD * 9 * 32 > 4 * 32 * 3 ? a.Cin * KD * 9 * 32 : 4 * 32 * 3;
    const size_t lds = (size_t)(((a.Cin * NV + 3) & ~3) + wslab) * 4;
    const dim3 grid((unsigned)((size_t)a.N * tD * tH * tW)), block(256);
    if (a.planar) hipLaunchKernelGGL((conv_small_fwd_kernel<1, 1, 16>), grid, block, lds, s, a, tD, tH, tW);
    else hipLaunchKernelGGL((conv_small_fwd_kernel<3, 2, 8>), grid, block, lds, s, a, tD, tH, tW);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

static int small_wgrad_tiles_per_split(int ntiles) { return cdiv(ntiles, 1024); }

int conv_small_wgrad_splits(int N, int D, int H, int W, int planar) {
    const int ntiles = conv_small_stats_parts(N, D, H, W, planar);
    return cdiv(ntiles, small_wgrad_tiles_per_split(ntiles));
}

int launch_conv_small_wgrad(const float* x, int Cin, const float* dy, int dy_ldc, float* part,
                            int N, int D, int H, int W, int Cout, int planar, hipStream_t s, const SmallWgradFuse* fuse) {
    SmallWgradFuse f{};
    if (fuse) { f = *fuse; E3_REQUIRE(Cout % 4 == 0 && f.x1 && f.g && f.biaspart, E3_ERR_INVALID, "bad fused BN-backward arguments"); }
    E3_REQUIRE(Cin >= 1 && Cin < 8, E3_ERR_UNSUPPORTED, "direct conv handles 1..7 input channels");
    int TD, TH; small_brick(planar, TD, TH);
    const int tD = cdiv(D, TD), tH = cdiv(H, TH), tW = cdiv(W, 16);
    const int ntiles = N * tD * tH * tW;
    const int tps = small_wgrad_tiles_per_split(ntiles);
    const int splits = cdiv(ntiles, tps);
    const int NV = (TD + (planar ? 0 : 2)) * (TH + 2) * 18;
    const size_t lds = (size_t)(((NV + 3) & ~3) + 256 * 32) * 4;
    static const bool planar_valu = getenv("E3_FIRST_WGRAD_PLANAR_VALU") != nullptr;      // A/B switch
    if (planar && Cin == 1 && Cout % 32 == 0 && !planar_valu)       // (the 9 taps are the MFMA tile's columns, as the 27 of the 3x3x3 form)
        hipLaunchKernelGGL((conv_small_wgrad_kernel<1, 1, 16, true>), dim3(splits), dim3(256), lds, s, x, Cin, dy, dy_ldc, part, N, D, H, W, Cout, tD, tH, tW, tps, f);
    else if (planar) hipLaunchKernelGGL((conv_small_wgrad_kernel<1, 1, 16, false>), dim3(splits), dim3(256), lds, s, x, Cin, dy, dy_ldc, part, N, D, H, W, Cout, tD, tH, tW, tps, f);
    else if (Cin == 1 && Cout % 32 == 0)
        hipLaunchKernelGGL((conv_small_wgrad_kernel<3, 2, 8, true>), dim3(splits), dim3(256), lds, s, x, Cin, dy, dy_ldc, part, N, D, H, W, Cout, tD, tH, tW, tps, f);
    else hipLaunchKernelGGL((conv_small_wgrad_kernel<3, 2, 8, false>), dim3(splits), dim3(256), lds, s, x, Cin, dy, dy_ldc, part, N, D, H, W, Cout, tD, tH, tW, tps, f);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

static int final_lpv(int C) {
    const int Q = C / 4;
    int l = 1;
    while (l < 8 && Q % (l * 2) == 0) l *= 2;
    return l;
}

#define E3_COUT_SWITCH(COUT, ...)                         \
    switch (COUT) {                                       \
        case 1: { constexpr int CO = 1; __VA_ARGS__; break; } \
        case 2: { constexpr int CO = 2; __VA_ARGS__; break; } \
        case 3: { constexpr int CO = 3; __VA_ARGS__; break; } \
        case 4: { constexpr int CO = 4; __VA_ARGS__; break; } \
        case 5: { constexpr int CO = 5; __VA_ARGS__; break; } \
        case 6: { constexpr int CO = 6; __VA_ARGS__; break; } \
        case 7: { constexpr int CO = 7; __VA_ARGS__; break; } \
        case 8: { constexpr int CO = 8; __VA_ARGS__; break; } \
        case 9: { constexpr int CO = 9; __VA_ARGS__; break; } \
        case 10: { constexpr int CO = 10; __VA_ARGS__; break; } \
        case 11: { constexpr int CO = 11; __VA_ARGS__; break; } \
        case 12: { constexpr int CO = 12; __VA_ARGS__; break; } \
        case 13: { constexpr int CO = 13; __VA_ARGS__; break; } \
        case 14: { constexpr int CO = 14; __VA_ARGS__; break; } \
        case 15: { constexpr int CO = 15; __VA_ARGS__; break; } \
        case 16: { constexpr int CO = 16; __VA_ARGS__; break; } \
        default: e3_set_error("final 1x1x1 conv supports 1..16 output channels"); return E3_ERR_UNSUPPORTED; \
    }

int launch_conv_final_fwd(const float* a, int a_ldc, int C, const float* w, const float* bias, float* y,
                          int Cout, size_t S, int N, int softmax, hipStream_t s, const float* pro_scale, const float* pro_shift, ActArg pro_slope) {
    E3_REQUIRE(C % 4 == 0 && a_ldc % 4 == 0, E3_ERR_UNSUPPORTED, "channels must be a multiple of 4");
    const int lpv = final_lpv(C);
    const size_t vox = (size_t)N * S;
    size_t g = (vox * lpv + 255) / 256; if (g > 4096) g = 4096; if (g == 0) g = 1;
    E3_COUT_SWITCH(Cout, hipLaunchKernelGGL((conv_final_fwd_kernel<CO, false>), dim3((unsigned)g), dim3(256), 0, s, a, a_ldc, C, w, bias, y, S, N, lpv, softmax, pro_scale, pro_shift, pro_slope, HeadLossArgs{}, HeadBox{}));
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

// box form: the voxels of `box` (lo, size; on the (D, H, W) grid of `a`) only, written into a view of a larger NCDHW volume (strides in elements)
int launch_conv_final_fwd_box(const float* a, int a_ldc, int C, const float* w, const float* bias, float* y, int Cout, int N, int D, int H, int W,
                              const int lo[3], const int size[3], const long long ystride[4], int softmax, hipStream_t s,
                              const float* pro_scale, const float* pro_shift, ActArg pro_slope) {
    E3_REQUIRE(C % 4 == 0 && a_ldc % 4 == 0, E3_ERR_UNSUPPORTED, "channels must be a multiple of 4");
    const size_t vox = (size_t)N * size[0] * size[1] * size[2];
    E3_REQUIRE(vox > 0 && vox < 0xffffffffull, E3_ERR_INVALID, "head box: empty or too large");
    for (int i = 0; i < 3; ++i) E3_REQUIRE(lo[i] >= 0 && size[i] > 0, E3_ERR_INVALID, "head box: bad extent");
    E3_REQUIRE(lo[0] + size[0] <= D && lo[1] + size[1] <= H && lo[2] + size[2] <= W, E3_ERR_INVALID, "head box: outside the grid");
    const int lpv = final_lpv(C);
    size_t g = (vox * lpv + 255) / 256; if (g > 4096) g = 4096; if (g == 0) g = 1;
    const HeadBox box{1, size[0], size[1], size[2], lo[0], lo[1], lo[2], D, H, W, ystride[0], ystride[1], ystride[2], ystride[3]};
    E3_COUT_SWITCH(Cout, hipLaunchKernelGGL((conv_final_fwd_kernel<CO, false>), dim3((unsigned)g), dim3(256), 0, s, a, a_ldc, C, w, bias, y, (size_t)D * H * W, N, lpv, softmax, pro_scale, pro_shift, pro_slope, HeadLossArgs{}, box));
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

// head + criterion partial sums; returns the number of partial rows (= workgroups) written to `partial` ([rows][2 + 3 Cout]) through *rows
int launch_conv_final_fwd_loss(const float* a, int a_ldc, int C, const float* w, const float* bias, float* y, int Cout, size_t S, int N, hipStream_t s,
                               const float* pro_scale, const float* pro_shift, ActArg pro_slope, const long long* target, const float* class_w,
                               float* partial, int max_rows, int* rows) {
    E3_REQUIRE(C % 4 == 0 && a_ldc % 4 == 0, E3_ERR_UNSUPPORTED, "channels must be a multiple of 4");
    E3_REQUIRE(Cout >= 2, E3_ERR_INVALID, "the criterion needs at least two classes");
    const int lpv = final_lpv(C);
    const size_t vox = (size_t)N * S;
    // (1024 workgroups: the grid-stride loop does not care, and the criterion's finaliser walks the rows with one wave per value)
    size_t g = (vox * lpv + 255) / 256; if (g > 1024) g = 1024; if (g > (size_t)max_rows) g = (size_t)max_rows; if (g == 0) g = 1;
    const HeadLossArgs la{target, class_w, partial};
    E3_COUT_SWITCH(Cout, hipLaunchKernelGGL((conv_final_fwd_kernel<CO, true>), dim3((unsigned)g), dim3(256), 0, s, a, a_ldc, C, w, bias, y, S, N, lpv, 0, pro_scale, pro_shift, pro_slope, la, HeadBox{}));
    E3_CHECK_HIP(hipGetLastError());
    *rows = (int)g;
    return E3_OK;
}

int conv_final_bwd_parts(size_t total_voxels) {
    size_t g = (total_voxels + 31) / 32; if (g > 2048) g = 2048; if (g == 0) g = 1;
    return (int)g;
}

int launch_conv_final_bwd(const float* a, int a_ldc, int C, const float* w, const float* dy, float* da, int da_ldc,
                          float* part, int Cout, size_t S, int N, hipStream_t s, const float* pro_scale, const float* pro_shift, ActArg pro_slope) {
    E3_REQUIRE(C % 4 == 0 && C <= 1024, E3_ERR_UNSUPPORTED, "channels must be a multiple of 4 and <= 1024");
    const int parts = conv_final_bwd_parts((size_t)N * S);
    E3_COUT_SWITCH(Cout, hipLaunchKernelGGL((conv_final_bwd_kernel<CO>), dim3(parts), dim3(256), 0, s, a, a_ldc, C, w, dy, da, da_ldc, part, S, N, pro_scale, pro_shift, pro_slope));
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
