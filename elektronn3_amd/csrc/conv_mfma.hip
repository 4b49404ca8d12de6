// Implicit-GEMM 3D convolution on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces the ATen/MIOpen kernels behind elektronn3's `conv3()` (nn.Conv3d k=3 / (1,3,3), unet.py:131-149)
// and, in POINT mode, the GEMM inside `upconv2('transpose')` (nn.ConvTranspose3d k=s=2, unet.py:152-165).
// The same kernel computes the data gradient (dgrad = conv with flipped, role-swapped weights).
//
// GEMM view:  rows  = output voxels, cols = output channels (32*NT per workgroup),
//             K     = taps x input channels, walked as  [channel chunk CK] x [tap]
// A operand:  the brick's input halo ((TD+2)x(TH+2)x(TW+2) voxels x CK channels) is staged ONCE per channel
//             chunk into LDS (zero padding and the optional BN+ReLU prologue applied on the way); every tap
//             then reads it at a compile-time LDS offset with ds_read_b128 (4 consecutive channels per lane =
//             the k-slices of 4 consecutive MFMAs).  The 16-B chunks of a voxel are XOR-swizzled with the voxel's W
//             coordinate so that the 16 voxels a ds_read_b128 lane group touches fall on 16 different 16-B bank slots
//             without padding: the brick needs 46 KB instead of 57.6 KB and THREE workgroups fit a CU (12 waves).
// B operand:  packed weights [tap][co][ci] are read straight from global memory (L1/L2 resident: every
//             workgroup walks the same 27*Cin*Cout*4 B), one float4 per lane per 4 MFMAs, kept two taps ahead in a
//             3-deep register ring.  No per-tap barrier: the only barriers bracket the halo staging.
// Two work decompositions (template KS):
//   KS = 1  workgroup = 256-voxel brick (2x8x16, planar 1x16x16); each of the 4 waves owns 64 rows, all K.
//   KS = 4  workgroup =  64-voxel brick (1x4x16); the 4 waves own the SAME 64 rows and split the channel chunks
//           (intra-workgroup split-K), partial tiles are summed through LDS in a fixed order.  Used for the low-resolution
//           levels (8x16x16: 4096 voxels) where 256-voxel bricks would leave most of the 256 CUs idle.
// fp32 MFMA is exact fp32 (an fmaf chain) at 157 TFLOP/s dense; LDS and L1 traffic per MFMA is tiny because a
// 32x32x2 MFMA takes 64 cycles, so the kernel is matrix-pipe bound, not LDS bound.
#include <stdlib.h>

#include "kernels.h"

namespace {

template <int KD, int KHW, int TD, int TH, int TW, int CK, int KS>
struct Geo {
    static constexpr int PD = KD / 2, PH = KHW / 2;
    static constexpr int LD = TD + 2 * PD, LH = TH + 2 * PH, LW = TW + 2 * PH;
    static constexpr int VS = CK;                     // LDS floats per voxel (no padding: 16-B chunks are XOR-swizzled)
    static constexpr int Q = CK / 4;                  // 16-B chunks per voxel
    static constexpr int FSH = Q == 4 ? 2 : (Q == 2 ? 3 : 4);   // chunk q of a voxel at halo-W coordinate zw is stored at q ^ ((zw >> FSH) & (Q-1))
    static constexpr int NVOX = LD * LH * LW;
    static constexpr int T = KD * KHW * KHW;
    static constexpr int PART = NVOX * VS;            // floats of one staged chunk
    static constexpr int LDS_FLOATS = PART * KS;
    static_assert(TD * TH * TW * KS == 256, "brick must hold 256/KS voxels (64 rows per wave)");
    static_assert(TW == 16, "row mapping assumes 16 voxels along W");
};

template <int KD, int KHW, int TD, int TH, int TW, int CK, int NT, int KS>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs a) {
    using G = Geo<KD, KHW, TD, TH, TW, CK, KS>;
    constexpr int VS = G::VS, LH = G::LH, LW = G::LW, PD = G::PD, PH = G::PH, T = G::T, K8 = CK / 8;
    static_assert(CK == 8 || CK == 16, "swizzle is derived for CK in {8,16}");
    static_assert(KS == 1 || KS == 4, "KS in {1,4}");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hf = lane >> 5;
    const int rowbase = KS == 1 ? wave * 64 : 0;      // first GEMM row (voxel of the brick) this wave owns
    float* lds = smem + (KS == 1 ? 0 : wave * G::PART);   // this wave's staged chunk

    // ---- which brick / column tile (XCD-aware: neighbours in (ntile, w, h, d) order share an L2)
    const unsigned nblk = gridDim.x;
    unsigned L = xcd_remap(blockIdx.x, nblk);
    const int ntile = L % a.ntiles; L /= a.ntiles;
    const int tw_ = L % a.tilesW; L /= a.tilesW;
    const int th_ = L % a.tilesH; L /= a.tilesH;
    const int td_ = L % a.tilesD; const int nb = L / a.tilesD;
    const int d0 = td_ * TD, h0 = th_ * TH, w0 = tw_ * TW;
    const int n0 = ntile * 32 * NT;
    const int mtile = ((nb * a.tilesD + td_) * a.tilesH + th_) * a.tilesW + tw_;
    const bool gather = (a.flags & CF_GATHER_UP) != 0;
    const bool scatter = (a.flags & CF_SCATTER_UP) != 0;

    // ---- per-lane LDS offsets of its two 32-row sub-tiles (rows = voxels).  The chunk swizzle key depends only on
    // the voxel's W coordinate inside the halo brick, so for each of the KHW values of kw the (sub-tile, k8) offsets are
    // lane constants and every tap adds a compile-time immediate: no address arithmetic inside the tap loop.
    int aoff[KHW][2][K8];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int m = rowbase + s * 32 + j;
        const int ww = m & 15, hh = (m >> 4) % TH, dd = (m >> 4) / TH;
        const int vox = (dd * LH + hh) * LW + ww;
#pragma unroll
        for (int kw = 0; kw < KHW; ++kw) {
            const int key = ((ww + kw) >> G::FSH) & (G::Q - 1);
#pragma unroll
            for (int k8 = 0; k8 < K8; ++k8) aoff[kw][s][k8] = (vox + kw) * VS + 4 * ((2 * k8 + hf) ^ key);
        }
    }

    f32x16 acc[2][NT];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ns = 0; ns < NT; ++ns)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][ns][r] = 0.f;

    const size_t tapstride = (size_t)a.NPad * a.Cin;
    const bool pro = a.pro_scale != nullptr;

    // ---- staging plan of this thread: item `it` is one 16-B piece (voxel v, channel quad q [, chunk part]) of the halo
    // brick(s).  Its global element offset (without the chunk base), LDS offset and validity are lane constants for the
    // whole workgroup, so the per-chunk staging below is a branch-free burst of loads: out-of-range voxels read a clamped
    // (valid) address and are zeroed by a select; ALL loads of a chunk are in flight together.
    constexpr int Q = CK / 4;
    constexpr int ITEMS = G::NVOX * Q * KS;
    constexpr int XI = (ITEMS + 255) / 256;
    int soff[XI], loff[XI];
    unsigned okbits = 0;

    for (int g = 0; g < a.G; ++g) {
        int gtd = 0, gth = 0, gtw = 0;
        if (gather) { gtw = g & 1; gth = (g >> 1) & 1; gtd = g >> 2; }
        if (g == 0 || gather) {
            okbits = 0;
#pragma unroll
            for (int it = 0; it < XI; ++it) {
                const int idx = tid + it * 256;
                const int part = idx / (G::NVOX * Q);
                const int rem = idx - part * (G::NVOX * Q);
                const int v = rem / Q, q = rem % Q;
                const int zw = v % LW; const int t2 = v / LW; const int zh = t2 % LH; const int zd = t2 / LH;
                int gd = d0 + zd - PD, gh = h0 + zh - PH, gw = w0 + zw - PH;
                bool ok = idx < ITEMS && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
                int ed = a.D, eh = a.H, ew = a.W;
                if (gather) {
                    gd = a.sd * gd + gtd; gh = 2 * gh + gth; gw = 2 * gw + gtw;
                    ok = ok && gd < a.Do && gh < a.Ho && gw < a.Wo;
                    ed = a.Do; eh = a.Ho; ew = a.Wo;
                }
                soff[it] = ok ? ((((nb * ed + gd) * eh + gh) * ew + gw) * a.x_ldc + part * CK + 4 * q) : 0;   // < 2^31 (launcher checks)
                loff[it] = part * G::PART + v * VS + 4 * (q ^ ((zw >> G::FSH) & (Q - 1)));
                okbits |= (ok ? 1u : 0u) << it;
            }
        }
        for (int cb0 = 0; cb0 < a.Cin; cb0 += CK * KS) {
            __syncthreads();
            if (!(E3_DBG_FLAGS(a.flags) & 256)) {   // flag 256: timing ablation (skip staging)
                // every item of a thread is the same channel quad (256 % Q == 0): one scale/shift pair per chunk
                f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
                if (pro) {
                    psc = *reinterpret_cast<const f32x4*>(a.pro_scale + cb0 + 4 * (tid % Q));
                    psh = *reinterpret_cast<const f32x4*>(a.pro_shift + cb0 + 4 * (tid % Q));
                }
#ifndef E3_STAGE_BATCH
#define E3_STAGE_BATCH 6
#endif
                constexpr int SB = E3_STAGE_BATCH < XI ? E3_STAGE_BATCH : XI;   // loads in flight per thread and batch
#pragma unroll
                for (int b0 = 0; b0 < XI; b0 += SB) {
                    f32x4 xr[SB];
#pragma unroll
                    for (int u = 0; u < SB; ++u) {
                        const int it = b0 + u;
                        if (it < XI) {
                            bool ok = (okbits >> it) & 1u;
                            if (KS > 1) ok = ok && (cb0 + (loff[it] / G::PART) * CK < a.Cin);   // ragged last group of chunks
                            const int off = ok ? soff[it] + cb0 : 0;
                            f32x4 val = *reinterpret_cast<const f32x4*>(a.x + off);
                            if (pro) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) val[e] = fmaxf(__builtin_fmaf(val[e], psc[e], psh[e]), 0.f);
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) xr[u][e] = ok ? val[e] : 0.f;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < SB; ++u) {
                        const int it = b0 + u;
                        if (it < XI && tid + it * 256 < ITEMS) *reinterpret_cast<f32x4*>(smem + loff[it]) = xr[u];
                    }
                }
            }
            __syncthreads();

            const int cb = cb0 + (KS == 1 ? 0 : wave * CK);      // this wave's channel chunk
            if (KS == 1 || cb < a.Cin) {
                // ---- walk the taps; B fragments come straight from global memory (L1/L2 resident), prefetched TWO
                // taps ahead into a 3-deep register ring.  sched_barrier pins each prefetch above the MFMAs it hides
                // behind (left alone, the scheduler sinks the loads to ~8 MFMAs before their use, which is less than an
                // L2 round trip when the wave has its SIMD to itself).
                const float* wl = a.wt + ((size_t)g * T * a.NPad + n0 + j) * a.Cin + cb + 4 * hf;
                // keep the 27 per-tap pointers from being hoisted out of the chunk loop (54+ VGPRs live across everything)
                asm volatile("" : "+v"(wl));
                f32x4 bq[3][NT][K8];
#pragma unroll
                for (int pre = 0; pre < 2 && pre < T; ++pre)
#pragma unroll
                    for (int ns = 0; ns < NT; ++ns)
#pragma unroll
                        for (int k8 = 0; k8 < K8; ++k8)
                            bq[pre][ns][k8] = *reinterpret_cast<const f32x4*>(wl + (size_t)pre * tapstride + (size_t)ns * 32 * a.Cin + k8 * 8);
                // flat software pipeline over steps (tap, k8): A fragments of step i+1 and the B fragments of tap+2 are
                // requested BEFORE the 8*NT MFMAs of step i; sched_barrier keeps that order in the emitted code.
                constexpr int S = T * K8;
                f32x4 av[2][2];
#pragma unroll
                for (int s = 0; s < 2; ++s) av[0][s] = *reinterpret_cast<const f32x4*>(lds + aoff[0][s][0]);
#pragma unroll
                for (int i = 0; i < S; ++i) {
                    const int tap = i / K8, k8 = i % K8;
                    if (i + 1 < S) {
                        const int tn = (i + 1) / K8, k8n = (i + 1) % K8;
                        const int kdn = tn / (KHW * KHW), khn = (tn / KHW) % KHW, kwn = tn % KHW;
                        const int tapoffn = (kdn * LH + khn) * LW * VS;   // compile-time: folded into the ds_read immediate
#pragma unroll
                        for (int s = 0; s < 2; ++s)
                            av[(i + 1) & 1][s] = *reinterpret_cast<const f32x4*>(lds + aoff[kwn][s][k8n] + tapoffn);
                    }
                    if (k8 == 0 && tap + 2 < T) {
                        const float* wn = wl + (size_t)(tap + 2) * tapstride;
#pragma unroll
                        for (int ns = 0; ns < NT; ++ns)
#pragma unroll
                            for (int kk = 0; kk < K8; ++kk)
                                bq[(tap + 2) % 3][ns][kk] = *reinterpret_cast<const f32x4*>(wn + (size_t)ns * 32 * a.Cin + kk * 8);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int ns = 0; ns < NT; ++ns)
#pragma unroll
                            for (int s = 0; s < 2; ++s)
                                acc[s][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i & 1][s][e], bq[tap % 3][ns][k8][e], acc[s][ns], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    // ---- KS == 4: sum the four waves' partial tiles through LDS (fixed order); wave t keeps tile t = s*NT + ns
    constexpr int NTILES = 2 * NT;
    bool owns[2][NT];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ns = 0; ns < NT; ++ns) owns[s][ns] = KS == 1 ? true : (wave == s * NT + ns);
    if (KS == 4) {
        __syncthreads();   // every wave is done with its halo: LDS becomes [wave][tile][reg][lane]
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int ns = 0; ns < NT; ++ns)
#pragma unroll
                for (int r = 0; r < 16; ++r) smem[((wave * NTILES + s * NT + ns) * 16 + r) * 64 + lane] = acc[s][ns][r];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int ns = 0; ns < NT; ++ns)
                if (owns[s][ns]) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = 0.f;
#pragma unroll
                        for (int w = 0; w < 4; ++w) v += smem[((w * NTILES + s * NT + ns) * 16 + r) * 64 + lane];
                        acc[s][ns][r] = v;
                    }
                }
    }

    // ---- epilogue: bias (+ folded BN + ReLU in eval mode), store, per-tile channel statistics
    const bool do_stats = a.stats != nullptr;
    const bool aff = a.epi_scale != nullptr;
    // Wide stores (KS == 1): the accumulator layout gives a lane ONE channel of 32 voxels = 32 dword stores per tile, which is
    // store-issue bound (measured: the transposed-conv forward spent 2/3 of its time here).  Transposed through a per-wave
    // 64x32 LDS tile each lane instead writes 4 consecutive channels of 8 voxels (8 dwordx4 stores, whole 128-B voxel rows).
    // Needs a 32-column tile that lies inside one tap's channel range.
    const bool wide = KS == 1 && (a.Cout & 31) == 0 && (a.y_ldc & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0);
    constexpr int EPI_SCR = 4 * NT * 32 * 3;             // statistics scratch floats in front of the store tiles
    if (do_stats || wide) __syncthreads();   // all waves are done reading LDS: it is reused as scratch below
#pragma unroll
    for (int ns = 0; ns < NT; ++ns) {
        const int n = n0 + 32 * ns + j;
        const bool nvalid = n < a.Ncols;
        int co = n, utd = 0, uth = 0, utw = 0, ut = 0;
        if (scatter) { ut = n / a.Cout; co = n - ut * a.Cout; utw = ut & 1; uth = (ut >> 1) & 1; utd = ut >> 2; }
        const float bias = (a.bias && nvalid) ? a.bias[co] : 0.f;
        float es = 1.f, eh = 0.f;
        if (aff && nvalid) { es = a.epi_scale[co]; eh = a.epi_shift[co]; }
        float cnt = 0.f, sum = 0.f;
        unsigned okmask = 0u;
        float* tile = smem + EPI_SCR + wave * (64 * 32);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (!owns[s][ns]) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hf;
                const int m = rowbase + s * 32 + row;
                const int gw = w0 + (m & 15), gh = h0 + (m >> 4) % TH, gd = d0 + (m >> 4) / TH;
                bool ok = nvalid && gd < a.D && gh < a.H && gw < a.W;
                size_t off;
                if (scatter) {
                    const int od = a.sd * gd + utd, oh = 2 * gh + uth, ow = 2 * gw + utw;
                    ok = ok && od < a.Do && oh < a.Ho && ow < a.Wo;
                    off = ((((size_t)nb * a.Do + od) * a.Ho + oh) * a.Wo + ow) * a.y_ldc + co;
                } else {
                    off = ((((size_t)nb * a.D + gd) * a.H + gh) * a.W + gw) * a.y_ldc + co;
                }
                float v = acc[s][ns][r] + bias;
                if (aff) v = fmaxf(__builtin_fmaf(v, es, eh), 0.f);
                acc[s][ns][r] = v;
                if (ok && (E3_DBG_FLAGS(a.flags) & 512)) ok = (v == 12345.678f);   // flag 512: timing ablation (skip the stores)
                if (wide) tile[(s * 32 + row) * 32 + j] = v;
                if (ok) {
                    if (!wide) a.y[off] = v;
                    cnt += 1.f; sum += v;
                    okmask |= 1u << (s * 16 + r);
                }
            }
        }
        if (wide) {      // (same wave wrote the tile: LDS operations of one wave are ordered, no barrier needed)
            const int nt0 = n0 + 32 * ns;                 // first column of the tile (wave-uniform)
            int c0 = nt0, wtd = 0, wth = 0, wtw = 0;
            if (scatter) { const int wt = nt0 / a.Cout; c0 = nt0 - wt * a.Cout; wtw = wt & 1; wth = (wt >> 1) & 1; wtd = wt >> 2; }
            const int c4 = 4 * (lane & 7);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int trow = 8 * p + (lane >> 3);
                const int m = rowbase + trow;
                const int gw = w0 + (m & 15), gh = h0 + (m >> 4) % TH, gd = d0 + (m >> 4) / TH;
                bool ok = nt0 + c4 < a.Ncols && gd < a.D && gh < a.H && gw < a.W && !(E3_DBG_FLAGS(a.flags) & 512);
                size_t off;
                if (scatter) {
                    const int od = a.sd * gd + wtd, oh = 2 * gh + wth, ow = 2 * gw + wtw;
                    ok = ok && od < a.Do && oh < a.Ho && ow < a.Wo;
                    off = ((((size_t)nb * a.Do + od) * a.Ho + oh) * a.Wo + ow) * a.y_ldc + c0 + c4;
                } else {
                    off = ((((size_t)nb * a.D + gd) * a.H + gh) * a.W + gw) * a.y_ldc + c0 + c4;
                }
                const f32x4 v = *reinterpret_cast<const f32x4*>(tile + trow * 32 + c4);
                if (ok) *reinterpret_cast<f32x4*>(a.y + off) = v;
            }
        }
        if (do_stats) {
            float mean = cnt > 0.f ? sum / cnt : 0.f, m2 = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (okmask & (1u << (s * 16 + r))) { const float d = acc[s][ns][r] - mean; m2 += d * d; }
            // other half-wave holds the other 32 rows of the same column
            const float cnt2 = __shfl_xor(cnt, 32), mean2 = __shfl_xor(mean, 32), m22 = __shfl_xor(m2, 32);
            welford_merge(cnt, mean, m2, cnt2, mean2, m22);
            if (hf == 0) {
                float* sc = smem + ((wave * NT + ns) * 32 + j) * 3;
                sc[0] = cnt; sc[1] = mean; sc[2] = m2;
            }
        }
    }
    if (do_stats) {
        __syncthreads();
        if (tid < 32 * NT) {
            const int ns = tid >> 5, jj = tid & 31;
            const int n = n0 + 32 * ns + jj;
            if (n < a.Ncols) {
                float cnt = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float* sc = smem + ((w * NT + ns) * 32 + jj) * 3;
                    welford_merge(cnt, mean, m2, sc[0], sc[1], sc[2]);
                }
                int co = n, pidx = mtile;
                if (scatter) { const int ut = n / a.Cout; co = n - ut * a.Cout; pidx = mtile * (a.Ncols / a.Cout) + ut; }
                float* o = a.stats + ((size_t)pidx * a.Cout + co) * 3;
                o[0] = cnt; o[1] = mean; o[2] = m2;
            }
        }
    }
}

template <int KD, int KHW, int TD, int TH, int TW, int CK, int NT, int KS>
int launch_inst(ConvArgs a, hipStream_t s) {
    using G = Geo<KD, KHW, TD, TH, TW, CK, KS>;
    a.tilesD = cdiv(a.D, TD); a.tilesH = cdiv(a.H, TH); a.tilesW = cdiv(a.W, TW);
    a.ntiles = a.NPad / (32 * NT);
    const size_t nblk = (size_t)a.N * a.tilesD * a.tilesH * a.tilesW * a.ntiles;
    E3_REQUIRE(nblk > 0 && nblk < (1u << 31), E3_ERR_INVALID, "conv grid out of range");
    constexpr int scratch = KS == 4 ? 4 * 2 * NT * 16 * 64 : 4 * NT * 32 * 3 + 4 * 64 * 32;   // KS == 1: statistics scratch + one 64x32 store tile per wave
    constexpr int lds_bytes = (G::LDS_FLOATS > scratch ? G::LDS_FLOATS : scratch) * 4;
    auto kern = conv_mfma_kernel<KD, KHW, TD, TH, TW, CK, NT, KS>;
    static bool attr_set = false;
    if (!attr_set) {
        E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds_bytes, s, a);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

template <int KD, int KHW, int TD, int TH, int TW, int KS>
int dispatch_ck_nt(const ConvArgs& a, int NT, hipStream_t s) {
    const bool ck16 = (a.Cin % 16) == 0;
    if (ck16) return NT == 2 ? launch_inst<KD, KHW, TD, TH, TW, 16, 2, KS>(a, s) : launch_inst<KD, KHW, TD, TH, TW, 16, 1, KS>(a, s);
    return NT == 2 ? launch_inst<KD, KHW, TD, TH, TW, 8, 2, KS>(a, s) : launch_inst<KD, KHW, TD, TH, TW, 8, 1, KS>(a, s);
}

struct Brick { int TD, TH; };
Brick brick_of(ConvKind kind, int ks) {
    if (ks == 4) return {1, 4};
    if (kind == CONV_K3_PLANAR) return {1, 16};
    return {2, 8};
}

size_t grid_of(ConvKind kind, int ks, int nt, int N, int D, int H, int W, int ncols) {
    const Brick b = brick_of(kind, ks);
    return (size_t)N * cdiv(D, b.TD) * cdiv(H, b.TH) * cdiv(W, 16) * cdiv(ncols, 32 * nt);
}

}  // namespace

// Work decomposition: prefer big tiles (A staged once per 64 columns, all K in one wave), but only if the launch still
// fills the chip (256 CUs x 2-3 resident workgroups); otherwise narrower column tiles, then intra-workgroup split-K.
void conv_decomposition(ConvKind kind, int flags, int N, int D, int H, int W, int Cin, int ncols, int* ks, int* nt) {
    const int nt_max = ncols >= 64 ? 2 : 1;
    const bool ks_ok = kind != CONV_POINT && (flags & ~CF_NO_PERSIST) == 0 && Cin >= 64;   // (callers with a BN prologue pass flags |= CF_NO_KSPLIT)
    const int cand[4][2] = {{1, 2}, {1, 1}, {4, 2}, {4, 1}};
    int best_ks = 1, best_nt = nt_max; size_t best_grid = 0;
    for (const auto& c : cand) {
        if (c[1] > nt_max) continue;
        if (c[0] == 4 && !ks_ok) continue;
        const size_t g = grid_of(kind, c[0], c[1], N, D, H, W, ncols);
        if (g >= 256u) { *ks = c[0]; *nt = c[1]; return; }   // at least one workgroup per CU
        if (g > best_grid) { best_grid = g; best_ks = c[0]; best_nt = c[1]; }
    }
    *ks = best_ks; *nt = best_nt;
}

int conv_col_tile(int ncols) { return ncols >= 64 ? 64 : 32; }

int conv_stats_parts(ConvKind kind, int flags, int N, int D, int H, int W, int sd, int Cin, int ncols) {
    if (conv_use_wino(kind, flags, N, D, H, W, Cin, ncols)) return wino_stats_parts(N, D, H, W, Cin, ncols, flags);
    if (conv_use_wino2d(kind, flags, N, D, H, W, Cin, ncols)) return wino2d_bricks(N, D, H, W);
    if (kind == CONV_POINT && (flags & CF_SCATTER_UP) && upconv_gemm_ok(flags, Cin, ncols / (sd * 4), ncols)) return upconv_stats_parts(N, D, H, W, sd, Cin, ncols / (sd * 4));
    int ks, nt; conv_decomposition(kind, flags, N, D, H, W, Cin, ncols, &ks, &nt);
    const Brick b = brick_of(kind, ks);
    int parts = N * cdiv(D, b.TD) * cdiv(H, b.TH) * cdiv(W, 16);
    if (flags & CF_SCATTER_UP) parts *= sd * 4;
    return parts;
}

int launch_conv_mfma(ConvKind kind, ConvArgs a, hipStream_t s) {
    E3_REQUIRE(a.Cin % 8 == 0 && a.Cin >= 8, E3_ERR_UNSUPPORTED, "MFMA conv needs input channels to be a multiple of 8");
    E3_REQUIRE(a.x_ldc % 4 == 0 && ((uintptr_t)a.x % 16) == 0, E3_ERR_INVALID, "conv input view must be 16-byte aligned");
    E3_REQUIRE(a.NPad % conv_col_tile(a.Ncols) == 0 && a.NPad >= a.Ncols, E3_ERR_INVALID, "bad NPad");
    {
        const size_t vin = (a.flags & CF_GATHER_UP) ? (size_t)a.N * a.Do * a.Ho * a.Wo : (size_t)a.N * a.D * a.H * a.W;
        E3_REQUIRE(vin * (size_t)a.x_ldc < ((size_t)1 << 31), E3_ERR_UNSUPPORTED, "conv input view exceeds 2^31 elements (32-bit offsets)");
    }
    if (a.G <= 0) a.G = 1;
    E3_REQUIRE(!a.y_chunk || (kind == CONV_K3 && a.splitk <= 1 && conv_use_wino(kind, a.flags, a.N, a.D, a.H, a.W, a.Cin, a.Ncols)) ||
                   (kind == CONV_POINT && (a.flags & CF_SCATTER_UP) && upconv_gemm_ok(a.flags, a.Cin, a.Cout, a.Ncols)), E3_ERR_UNSUPPORTED,
               "channel-chunked output: F(2x2x4) Winograd kernel and the transposed-conv GEMM kernels only");
    E3_REQUIRE(!a.x_chunk || (kind == CONV_K3 && a.splitk <= 1 && conv_use_wino(kind, a.flags, a.N, a.D, a.H, a.W, a.Cin, a.Ncols)), E3_ERR_UNSUPPORTED,
               "channel-chunked input: F(2x2x4) Winograd kernel only");
    if (kind == CONV_POINT && upconv_gemm_ok(a.flags, a.Cin, a.Cout, a.Ncols)) return launch_upconv_gemm(a, s);   // upconv_gemm.hip
    if (a.splitk > 1 || conv_use_wino(kind, a.flags, a.N, a.D, a.H, a.W, a.Cin, a.Ncols)) return launch_conv3_wino(a, s);   // a.wt packed by launch_pack_conv_auto
    if (conv_use_wino2d(kind, a.flags, a.N, a.D, a.H, a.W, a.Cin, a.Ncols)) return launch_conv2_wino(a, s);
    int ks, nt; conv_decomposition(kind, a.flags, a.N, a.D, a.H, a.W, a.Cin, a.Ncols, &ks, &nt);
    static const bool use_v3 = getenv("E3_CONV_NO_V3") == nullptr;   // debug switch: fall back to the global-B kernel
    if (use_v3 && kind != CONV_POINT && ks == 1 && (a.flags & (CF_SCATTER_UP | CF_GATHER_UP)) == 0)
        return launch_conv3_v3(kind, a, nt, s);    // both operands in LDS, register-prefetched chunks (conv_v3.hip)
    switch (kind) {
        case CONV_K3: return ks == 4 ? dispatch_ck_nt<3, 3, 1, 4, 16, 4>(a, nt, s) : dispatch_ck_nt<3, 3, 2, 8, 16, 1>(a, nt, s);
        case CONV_K3_PLANAR: return ks == 4 ? dispatch_ck_nt<1, 3, 1, 4, 16, 4>(a, nt, s) : dispatch_ck_nt<1, 3, 1, 16, 16, 1>(a, nt, s);
        case CONV_POINT: return dispatch_ck_nt<1, 1, 2, 8, 16, 1>(a, nt, s);
    }
    e3_set_error("unknown conv kind");
    return E3_ERR_INVALID;
}
