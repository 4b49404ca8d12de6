// conv3 "v3": the 3x3x3 / 1x3x3 implicit-GEMM convolution with BOTH operands in LDS and register-prefetched chunks.
//
// Same GEMM view and MFMA (v_mfma_f32_32x32x2_f32) as conv_mfma.hip, different data movement:
//   * K is walked in chunks of 8 input channels.  Per chunk the workgroup stages the brick's halo (720 voxels x 8
//     channels = 23 KB, 16-B pieces XOR-swizzled by the W coordinate) AND the chunk's weights of all taps
//     ([tap][col][8] = 27 KB per 32 columns, halves swizzled by the column) into LDS.
//   * The global loads of chunk c+1 (6 + 7*NT float4 per thread) are issued into registers right after the barrier
//     that starts the MFMAs of chunk c and are written to LDS only after those MFMAs: their HBM/L2 latency is
//     completely hidden, and the tap loop contains NO vector-memory wait at all (weights are no longer an in-order
//     global-load stream that stalls behind slower activation loads).
//   * LDS per workgroup 50 KB (NT=1) / 77 KB (NT=2) -> 3 / 2 workgroups per CU cover each other's barriers, the first
//     chunk's exposed load and the epilogue.
// Epilogue (bias, folded eval-BN + ReLU, per-tile Welford statistics) is identical to conv_mfma.hip.
#include "kernels.h"

namespace {

template <int KD, int TD, int TH>
struct G3 {
    static constexpr int TW = 16, KHW = 3, PD = KD / 2, PH = 1;
    static constexpr int LD = TD + 2 * PD, LH = TH + 2, LW = TW + 2;
    static constexpr int NVOX = LD * LH * LW;
    static constexpr int T = KD * 9;
    static constexpr int CK = 8, VS = 8;
    static constexpr int AI = (NVOX * 2 + 255) / 256;          // 16-B pieces of A per thread
    static constexpr int A_FLOATS = AI * 256 * 4;               // padded so every thread may write its AI pieces
    static_assert(TD * TH * TW == 256, "brick must hold 256 voxels");
};

template <int KD, int TD, int TH, int NT>
__global__ __launch_bounds__(256, NT == 1 ? 3 : 2) void conv3_v3_kernel(const ConvArgs a) {
    using G = G3<KD, TD, TH>;
    constexpr int LH = G::LH, LW = G::LW, PD = G::PD, T = G::T, VS = G::VS, AI = G::AI;
    constexpr int BPIECES = T * 64 * NT;                        // 16-B pieces of B per chunk
    constexpr int BI = (BPIECES + 255) / 256;
    constexpr int TAPS_PER_IT = 256 / (64 * NT);                // taps covered by one pass of the 256 threads
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + G::A_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hf = lane >> 5;

    unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = L % a.ntiles; L /= a.ntiles;
    const int tw_ = L % a.tilesW; L /= a.tilesW;
    const int th_ = L % a.tilesH; L /= a.tilesH;
    const int td_ = L % a.tilesD; const int nb = L / a.tilesD;
    const int d0 = td_ * TD, h0 = th_ * TH, w0 = tw_ * 16;
    const int n0 = ntile * 32 * NT;
    const int mtile = ((nb * a.tilesD + td_) * a.tilesH + th_) * a.tilesW + tw_;

    // ---- A fragment offsets: lane (j, hf) of sub-tile s reads 16 B = channels 4hf..4hf+3 of voxel (row + tap)
    int aoff[3][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int m = wave * 64 + s * 32 + j;
        const int ww = m & 15, hh = (m >> 4) % TH, dd = (m >> 4) / TH;
        const int vox = (dd * LH + hh) * LW + ww;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) aoff[kw][s] = (vox + kw) * VS + 4 * (hf ^ (((ww + kw) >> 3) & 1));
    }
    // ---- B fragment offset: lane (j, hf) reads 16 B = channels 4hf..4hf+3 of column j (+32 ns) of the tap
    const int boff = j * 8 + 4 * (hf ^ ((j >> 3) & 1));

    // ---- staging plans (lane constants)
    int a_src[AI], a_dst[AI];
    unsigned a_ok = 0;
#pragma unroll
    for (int it = 0; it < AI; ++it) {
        const int idx = tid + it * 256;
        const int v = idx >> 1, q = idx & 1;
        const int zw = v % LW; const int t2 = v / LW; const int zh = t2 % LH; const int zd = t2 / LH;
        const int gd = d0 + zd - PD, gh = h0 + zh - 1, gw = w0 + zw - 1;
        const bool ok = v < G::NVOX && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
        a_src[it] = ok ? ((((nb * a.D + gd) * a.H + gh) * a.W + gw) * a.x_ldc + 4 * q) : 0;
        a_dst[it] = v < G::NVOX ? v * VS + 4 * (q ^ ((zw >> 3) & 1)) : idx * 4;   // pieces beyond the brick land in the pad
        a_ok |= (ok ? 1u : 0u) << it;
    }
    // B piece idx = tid + it*256: half = idx & 1, col = (idx >> 1) % (32 NT), tap = idx / (64 NT) = tid/(64NT) + it*TAPS_PER_IT
    const int b_half = tid & 1, b_col = (tid >> 1) % (32 * NT), b_tap0 = tid / (64 * NT);
    const size_t tapstride = (size_t)a.NPad * a.Cin;
    const float* b_src0 = a.wt + ((size_t)b_tap0 * a.NPad + n0 + b_col) * a.Cin + 4 * b_half;
    const int b_dst0 = (b_tap0 * 32 * NT + b_col) * 8 + 4 * (b_half ^ (((b_col & 31) >> 3) & 1));

    f32x16 acc[2][NT];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ns = 0; ns < NT; ++ns)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][ns][r] = 0.f;

    const bool pro = a.pro_scale != nullptr;
    const bool dbg = (E3_DBG_FLAGS(a.flags) & 1024) != 0;
    long long* dbgp = reinterpret_cast<long long*>(a.stats) + (size_t)blockIdx.x * 16;
    int dbgi = 0;
    auto stamp = [&]() { if (dbg && tid == 0 && dbgi < 12) dbgp[dbgi++] = (long long)__builtin_amdgcn_s_memtime(); };
    stamp();
    if (dbg && tid == 0) { dbgp[12] = (long long)__builtin_amdgcn_s_memtime(); dbgp[14] = (long long)__builtin_amdgcn_s_memrealtime(); }
    f32x4 xa[AI], xb[BI];
    auto issue_loads = [&](int cb) {
#pragma unroll
        for (int it = 0; it < AI; ++it) {
            const bool ok = (a_ok >> it) & 1u;
            xa[it] = *reinterpret_cast<const f32x4*>(a.x + (ok ? a_src[it] + cb : 0));
        }
#pragma unroll
        for (int it = 0; it < BI; ++it) {
            const bool ok = b_tap0 + it * TAPS_PER_IT < T;
            xb[it] = *reinterpret_cast<const f32x4*>(ok ? b_src0 + (size_t)it * TAPS_PER_IT * tapstride + cb : a.wt);
        }
    };
    issue_loads(0);

    for (int cb = 0; cb < a.Cin; cb += 8) {
        stamp();
        if (cb > 0) __syncthreads();          // every wave is done reading the previous chunk
        // ---- registers -> LDS (zero padding, optional BN+ReLU prologue on the in-range voxels)
        {
            f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
            if (pro) {
                psc = *reinterpret_cast<const f32x4*>(a.pro_scale + cb + 4 * (tid & 1));
                psh = *reinterpret_cast<const f32x4*>(a.pro_shift + cb + 4 * (tid & 1));
            }
#pragma unroll
            for (int it = 0; it < AI; ++it) {
                const bool ok = (a_ok >> it) & 1u;
                f32x4 v = xa[it];
                if (pro) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(__builtin_fmaf(v[e], psc[e], psh[e]), 0.f);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
                *reinterpret_cast<f32x4*>(As + a_dst[it]) = v;
            }
#pragma unroll
            for (int it = 0; it < BI; ++it)
                if (b_tap0 + it * TAPS_PER_IT < T)
                    *reinterpret_cast<f32x4*>(Bs + b_dst0 + it * TAPS_PER_IT * 32 * NT * 8) = xb[it];
        }
        __syncthreads();
        stamp();
        // in flight during the whole tap loop below (unconditional: the last chunk harmlessly re-loads itself, which
        // keeps the loop body free of a branch/join where the compiler would otherwise drain vmcnt)
        issue_loads(cb + 8 < a.Cin ? cb + 8 : cb);
        __builtin_amdgcn_sched_barrier(0);

        // ---- 27 taps x 8*NT MFMAs; fragments of tap+1 are read from LDS before the MFMAs of tap
        f32x4 av[2][2], bv[2][NT];
#pragma unroll
        for (int s = 0; s < 2; ++s) av[0][s] = *reinterpret_cast<const f32x4*>(As + aoff[0][s]);
#pragma unroll
        for (int ns = 0; ns < NT; ++ns) bv[0][ns] = *reinterpret_cast<const f32x4*>(Bs + boff + ns * 32 * 8);
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
            if (tap + 1 < T) {
                const int tn = tap + 1;
                const int kdn = tn / 9, khn = (tn / 3) % 3, kwn = tn % 3;
                const int tapoffn = (kdn * LH + khn) * LW * VS;
#pragma unroll
                for (int s = 0; s < 2; ++s) av[tn & 1][s] = *reinterpret_cast<const f32x4*>(As + aoff[kwn][s] + tapoffn);
#pragma unroll
                for (int ns = 0; ns < NT; ++ns) bv[tn & 1][ns] = *reinterpret_cast<const f32x4*>(Bs + boff + (tn * 32 * NT + ns * 32) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ns = 0; ns < NT; ++ns)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        acc[s][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tap & 1][s][e], bv[tap & 1][ns][e], acc[s][ns], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: bias (+ folded BN + ReLU in eval mode), per-tile channel statistics, store.
    // Stores go through LDS: the accumulator layout gives every lane ONE channel of 32 different voxels (32 dword stores
    // per lane, store-issue bound: ~36k cycles per workgroup measured); transposed through a per-wave 64x32 LDS tile each
    // lane instead writes 4 consecutive channels of 8 voxels = 8 dwordx4 stores, each wave-instruction covering 1 KB of
    // whole 128-B voxel rows.
    const bool do_stats = a.stats != nullptr && !dbg;
    const bool aff = a.epi_scale != nullptr;
    const bool wide = (a.Ncols & 3) == 0 && (a.y_ldc & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0);
    __syncthreads();                               // every wave is done with As/Bs: LDS is reused below
    stamp();
    float* tile = smem + 4 * NT * 32 * 3 + wave * (64 * 32);   // after the statistics scratch
    // Row geometry of the accumulator layout (TW = 16): row = (r&3) + 8*(r>>2) + 4*hf (+32 s) -> W coordinate
    // (r&3) + 4*hf + 8*((r>>2)&1), line (D,H) index wave*4 + 2*s + (r>>3).  Validity is separable, so the per-element
    // work below is branch-free (the compiler otherwise emits ~6 branches per element).
    bool oklin[2][2], okw[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int line = wave * 4 + 2 * s + u;
            oklin[s][u] = (d0 + line / TH) < a.D && (h0 + line % TH) < a.H;
        }
#pragma unroll
    for (int u = 0; u < 2; ++u) okw[u] = (w0 + 4 * hf + 8 * u + 3) < a.W;   // fast path flag: all 4 voxels (r&3) in range
#pragma unroll
    for (int ns = 0; ns < NT; ++ns) {
        const int n = n0 + 32 * ns + j;
        const bool nvalid = n < a.Ncols;
        const int co = n;
        const float bias = (a.bias && nvalid) ? a.bias[co] : 0.f;
        float es = 1.f, eh = 0.f;
        if (aff && nvalid) { es = a.epi_scale[co]; eh = a.epi_shift[co]; }
        float cnt = 0.f, sum = 0.f;
        unsigned okmask = 0u;
        if (dbg) { asm volatile("" :: "v"(bias)); stamp(); }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hf;
                const bool ok = nvalid && oklin[s][r >> 3] && (okw[(r >> 2) & 1] || (w0 + (row & 15)) < a.W);
                float v = acc[s][ns][r] + bias;
                if (aff) v = fmaxf(__builtin_fmaf(v, es, eh), 0.f);
                acc[s][ns][r] = v;
                tile[(s * 32 + row) * 32 + j] = v;
                cnt += ok ? 1.f : 0.f;
                sum += ok ? v : 0.f;
                okmask |= (ok ? 1u : 0u) << (s * 16 + r);
            }
        // same wave wrote the tile: LDS ops of one wave are ordered, no workgroup barrier needed
        if (dbg) { asm volatile("" :: "v"(sum)); stamp(); }
        if (wide) {
            const int c4 = n0 + 32 * ns + 4 * (lane & 7);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int row = 8 * p + (lane >> 3);
                const int line = wave * 4 + (p >> 1);
                const int gd = d0 + line / TH, gh = h0 + line % TH, gw = w0 + (row & 15);
                const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * 32 + 4 * (lane & 7));
                if (c4 < a.Ncols && gd < a.D && gh < a.H && gw < a.W && !(E3_DBG_FLAGS(a.flags) & 512))
                    *reinterpret_cast<f32x4*>(a.y + (size_t)(((nb * a.D + gd) * a.H + gh) * a.W + gw) * a.y_ldc + c4) = v;
            }
        } else {
            for (int rr = 0; rr < 32; ++rr) {      // rare fallback (unaligned view / channel count not a multiple of 4)
                const int row = 2 * rr + hf;
                const int line = wave * 4 + (row >> 4);
                const int gd = d0 + line / TH, gh = h0 + line % TH, gw = w0 + (row & 15);
                if (nvalid && gd < a.D && gh < a.H && gw < a.W && !(E3_DBG_FLAGS(a.flags) & 512))
                    a.y[(size_t)(((nb * a.D + gd) * a.H + gh) * a.W + gw) * a.y_ldc + co] = tile[row * 32 + j];
            }
        }
        if (do_stats) {
            float mean = cnt > 0.f ? sum / cnt : 0.f, m2 = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = acc[s][ns][r] - mean;
                    m2 += ((okmask >> (s * 16 + r)) & 1u) ? d * d : 0.f;
                }
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
                float* o = a.stats + ((size_t)mtile * a.Cout + n) * 3;
                o[0] = cnt; o[1] = mean; o[2] = m2;
            }
        }
    }
    if (dbg && tid == 0) { dbgp[13] = (long long)__builtin_amdgcn_s_memtime(); dbgp[15] = (long long)__builtin_amdgcn_s_memrealtime(); }
    stamp();   // (no barrier: a workgroup barrier would also wait for the outstanding stores)
}

template <int KD, int TD, int TH, int NT>
int launch_v3(ConvArgs a, hipStream_t s) {
    using G = G3<KD, TD, TH>;
    a.tilesD = cdiv(a.D, TD); a.tilesH = cdiv(a.H, TH); a.tilesW = cdiv(a.W, 16);
    a.ntiles = a.NPad / (32 * NT);
    const size_t nblk = (size_t)a.N * a.tilesD * a.tilesH * a.tilesW * a.ntiles;
    E3_REQUIRE(nblk > 0 && nblk < (1u << 31), E3_ERR_INVALID, "conv grid out of range");
    constexpr int stage_floats = G::A_FLOATS + G::T * 32 * NT * 8;
    constexpr int epi_floats = 4 * NT * 32 * 3 + 4 * 64 * 32;      // statistics scratch + one 64x32 store tile per wave
    constexpr int lds_bytes = (stage_floats > epi_floats ? stage_floats : epi_floats) * 4;
    auto kern = conv3_v3_kernel<KD, TD, TH, NT>;
    static bool attr_set = false;
    if (!attr_set) {
        E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds_bytes, s, a);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

}  // namespace

// kind in {CONV_K3, CONV_K3_PLANAR}, 256-voxel decomposition, nt in {1,2}
int launch_conv3_v3(ConvKind kind, ConvArgs a, int nt, hipStream_t s) {
    if (kind == CONV_K3) return nt == 2 ? launch_v3<3, 2, 8, 2>(a, s) : launch_v3<3, 2, 8, 1>(a, s);
    return nt == 2 ? launch_v3<1, 1, 16, 2>(a, s) : launch_v3<1, 1, 16, 1>(a, s);
}
