// Weight gradient of the 3x3x3 / 1x3x3 convolutions and of the k=s=2 transposed convolution on the fp32 matrix
// cores (autograd twin of conv3()/upconv2()).
//
//   dW[co][ci][tap] = sum_p dY[p][co] * X[p + off(tap)][ci]
//
// GEMM view per tap: D[co][ci] += A[co][k] * B[k][ci] with k = voxel.  A workgroup owns a (32 co x 32 ci) tile of
// ALL taps and a contiguous range of 128-voxel bricks ("split"); its four waves share the LDS-staged dY brick and X
// halo brick and split the TAPS (7/7/7/6), so every wave keeps <= 7 accumulator tiles (112 VGPRs) resident over the
// whole range and the K = N*D*H*W reduction needs no atomics: each split writes one partial slab, a small second
// kernel sums the slabs in a fixed order (deterministic) straight into torch's (Cout,Cin,kd,kh,kw) layout.
// Operand fetch is one ds_read_b32 per MFMA (lanes of a half-wave read 32 consecutive channels of one voxel:
// conflict-free), negligible next to the 64-cycle MFMA.
#include "kernels.h"

namespace {

template <int KD, int TD, int TH>
struct WGeo {
    static constexpr int TW = 16, PD = KD / 2;
    static constexpr int LD = TD + 2 * PD, LH = TH + 2, LW = TW + 2;
    static constexpr int NV = LD * LH * LW;          // halo voxels
    static constexpr int MV = TD * TH * TW;          // brick voxels (128)
    static constexpr int T = KD * 9;
    static constexpr int TPW = (T + 3) / 4;          // taps per wave
    static constexpr int LDS_BYTES = (NV + MV) * 32 * 4;
    static_assert(MV == 128, "brick must hold 128 voxels");
};

template <int KD, int TD, int TH>
__global__ __launch_bounds__(256, 2) void wgrad_conv_kernel(const WgradArgs a, int tilesD, int tilesH, int tilesW,
                                                            int tiles_per_split, int co_tiles, int ci_tiles) {
    using G = WGeo<KD, TD, TH>;
    constexpr int LH = G::LH, LW = G::LW, NV = G::NV, MV = G::MV, T = G::T, TPW = G::TPW, PD = G::PD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;              // [NV][32]  X halo brick, 32 input channels
    float* gs = smem + NV * 32;    // [MV][32]  dY brick, 32 output channels

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: scalar branches, SGPR offsets
    const int j = lane & 31, hf = lane >> 5;
    unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int ci_t = L % ci_tiles; L /= ci_tiles;
    const int co_t = L % co_tiles; const int split = L / co_tiles;
    const int ci0 = ci_t * 32, co0 = co_t * 32;
    const int ntiles = a.N * tilesD * tilesH * tilesW;
    const int tile0 = split * tiles_per_split;

    f32x16 acc[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;

    int tapoff[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        // a wave whose last slot has no tap (wave 3 for 27 taps) recomputes its previous tap into a dummy accumulator:
        // same latency as the other waves, no divergent-looking control flow in the MFMA loop
        const int tap = wave + 4 * u < T ? wave + 4 * u : T - 1;
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        tapoff[u] = ((kd * LH + kh) * LW + kw) * 32;
    }

    for (int tile = tile0; tile < tile0 + tiles_per_split && tile < ntiles; ++tile) {
        int Lt = tile;
        const int tw_ = Lt % tilesW; Lt /= tilesW; const int th_ = Lt % tilesH; Lt /= tilesH; const int td_ = Lt % tilesD; const int nb = Lt / tilesD;
        const int d0 = td_ * TD, h0 = th_ * TH, w0 = tw_ * G::TW;
        // stage X halo (8 float4 per voxel) and dY brick: ALL global loads of the tile are issued back to back into
        // registers (one HBM/L2 round trip instead of one per unrolled group), then written to LDS after the barrier
        // that retires the previous tile's reads.
        constexpr int XI = (NV * 8 + 255) / 256, GI = (MV * 8) / 256;
        f32x4 xr[XI], gr[GI];
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const int idx = tid + it * 256;
            const int v = idx >> 3, q = idx & 7;
            const int zw = v % LW, zh = (v / LW) % LH, zd = v / (LW * LH);
            const int gd = d0 + zd - PD, gh = h0 + zh - 1, gw = w0 + zw - 1;
            xr[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx < NV * 8 && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W && ci0 + 4 * q < a.Cin)
                xr[it] = *reinterpret_cast<const f32x4*>(a.x + ((((size_t)nb * a.D + gd) * a.H + gh) * a.W + gw) * a.x_ldc + ci0 + 4 * q);
        }
#pragma unroll
        for (int it = 0; it < GI; ++it) {
            const int idx = tid + it * 256;
            const int v = idx >> 3, q = idx & 7;
            const int ww = v & 15, hh = (v >> 4) % TH, dd = (v >> 4) / TH;
            const int gd = d0 + dd, gh = h0 + hh, gw = w0 + ww;
            gr[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (gd < a.D && gh < a.H && gw < a.W && co0 + 4 * q < a.Cout)
                gr[it] = *reinterpret_cast<const f32x4*>(a.dy + ((((size_t)nb * a.D + gd) * a.H + gh) * a.W + gw) * a.dy_ldc + co0 + 4 * q);
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const int idx = tid + it * 256;
            if (idx < NV * 8) *reinterpret_cast<f32x4*>(xs + idx * 4) = xr[it];
        }
#pragma unroll
        for (int it = 0; it < GI; ++it) *reinterpret_cast<f32x4*>(gs + (tid + it * 256) * 4) = gr[it];
        __syncthreads();
        // K loop: MFMA #t consumes voxels (2t, 2t+1): lane half hf takes voxel 2t+hf
#pragma unroll 1
        for (int row = 0; row < MV / 16; ++row) {          // one W-row of 16 voxels = 8 MFMA k-steps
            const int hh = row % TH, dd = row / TH;
            const int gbase = row * 16 * 32 + hf * 32 + j;
            const int xbase = ((dd * LH + hh) * LW + hf) * 32 + j;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float av = gs[gbase + t * 64];
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    const float bv = xs[xbase + t * 64 + tapoff[u]];
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[u], 0, 0, 0);
                }
            }
        }
    }
    // write the partial slab: part[split][tap][co][ci]
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const int tap = wave + 4 * u;
        if (tap < T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hf;
                a.part[(((size_t)split * T + tap) * a.CoPad + co0 + row) * a.CiPad + ci0 + j] = acc[u][r];
            }
        }
    }
}

// Transposed conv (k = s): dW[ci][co][tap] = sum_p X[p][ci] * dY[up(p, tap)][co].  One workgroup = one tap, one
// (32 ci x 32 co) tile, a range of 256-voxel bricks; the four waves split the voxels and are summed through LDS.
__global__ __launch_bounds__(256, 2) void wgrad_point_kernel(const WgradArgs a, int tilesD, int tilesH, int tilesW,
                                                             int tiles_per_split, int T, int co_tiles, int ci_tiles) {
    constexpr int TD = 2, TH = 8, MV = 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;              // [256][32] X brick (ci)
    float* gs = smem + MV * 32;    // [256][32] dY gathered at 2p+tap (co)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hf = lane >> 5;
    unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int co_t = L % co_tiles; L /= co_tiles;
    const int ci_t = L % ci_tiles; L /= ci_tiles;
    const int tap = L % T; const int split = L / T;
    const int ci0 = ci_t * 32, co0 = co_t * 32;
    const int utw = tap & 1, uth = (tap >> 1) & 1, utd = tap >> 2;
    const int ntiles = a.N * tilesD * tilesH * tilesW;
    const int tile0 = split * tiles_per_split;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int tile = tile0; tile < tile0 + tiles_per_split && tile < ntiles; ++tile) {
        int Lt = tile;
        const int tw_ = Lt % tilesW; Lt /= tilesW; const int th_ = Lt % tilesH; Lt /= tilesH; const int td_ = Lt % tilesD; const int nb = Lt / tilesD;
        const int d0 = td_ * TD, h0 = th_ * TH, w0 = tw_ * 16;
        __syncthreads();
#pragma unroll 4
        for (int idx = tid; idx < MV * 8; idx += 256) {
            const int v = idx >> 3, q = idx & 7;
            const int gd = d0 + (v >> 4) / TH, gh = h0 + (v >> 4) % TH, gw = w0 + (v & 15);
            const bool ok = gd < a.D && gh < a.H && gw < a.W;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f}, gv = xv;
            if (ok && ci0 + 4 * q < a.Cin)
                xv = *reinterpret_cast<const f32x4*>(a.x + ((((size_t)nb * a.D + gd) * a.H + gh) * a.W + gw) * a.x_ldc + ci0 + 4 * q);
            const int od = a.sd * gd + utd, oh = 2 * gh + uth, ow = 2 * gw + utw;
            if (ok && od < a.Do && oh < a.Ho && ow < a.Wo && co0 + 4 * q < a.Cout)
                gv = *reinterpret_cast<const f32x4*>(a.dy + ((((size_t)nb * a.Do + od) * a.Ho + oh) * a.Wo + ow) * a.dy_ldc + co0 + 4 * q);
            *reinterpret_cast<f32x4*>(xs + v * 32 + 4 * q) = xv;
            *reinterpret_cast<f32x4*>(gs + v * 32 + 4 * q) = gv;
        }
        __syncthreads();
        const int base = (wave * 64 + hf) * 32 + j;
#pragma unroll 8
        for (int t = 0; t < 32; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[base + t * 64], gs[base + t * 64], acc, 0, 0, 0);
    }
    // sum the four waves' tiles through LDS, then write part[split][tap][ci][co]
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hf;
        smem[(wave * 32 + row) * 32 + j] = acc[r];
    }
    __syncthreads();
    for (int idx = tid; idx < 32 * 32; idx += 256) {
        const float v = smem[idx] + smem[1024 + idx] + smem[2048 + idx] + smem[3072 + idx];
        const int row = idx >> 5, col = idx & 31;
        a.part[(((size_t)split * T + tap) * a.CiPad + ci0 + row) * a.CoPad + co0 + col] = v;
    }
}

// out[(r*C + c)*T + t] = sum_s part[((s*T + t)*RPad + r)*CPad + c]
// thread = (output quad of 4 consecutive c, split group g of G): 16-B loads, 8 of them in flight, double accumulation in a fixed
// order (k = g, g + G, ... then the G partial sums in order through LDS) => bit-reproducible.  G (a power of two <= 64) is chosen by
// the launcher so that small layers (27 x 32 x 32 outputs, 256 slabs) still spread over the chip.
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, float* __restrict__ out, int splits, int T,
                                                  int RPad, int CPad, int R, int C, int G, int lgG, unsigned block, double (*red)[4]) {
    const int cq = (C + 3) >> 2;                       // quads per row
    const size_t slab = (size_t)T * RPad * CPad;
    const size_t nquads = (size_t)T * R * cq;
    const int g = threadIdx.x & (G - 1);
    const int qpb = 256 >> lgG;                        // quads per block
    const size_t quad = (size_t)block * qpb + (threadIdx.x >> lgG);
    const bool on = quad < nquads;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    int c0 = 0, r = 0, t = 0;
    if (on) {
        c0 = (int)(quad % cq) * 4; size_t rr = quad / cq; r = (int)(rr % R); t = (int)(rr / R);
        const float* p = part + ((size_t)t * RPad + r) * CPad + c0;
        int k = g;
        for (; k + 7 * G < splits; k += 8 * G) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(k + u * G) * slab);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += (double)v[u][e];
        }
        for (; k < splits; k += G) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + (size_t)k * slab);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += (double)v[e];
        }
    }
    if (G > 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[threadIdx.x][e] = acc[e];
        __syncthreads();
        if (g == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { double tot = 0.0; for (int i = 0; i < G; ++i) tot += red[threadIdx.x + i][e]; acc[e] = tot; }
        }
    }
    if (on && g == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c0 + e < C) out[((size_t)r * C + c0 + e) * T + t] = (float)acc[e];
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int splits, int T,
                                                           int RPad, int CPad, int R, int C, int G, int lgG) {
    __shared__ double red[256][4];
    wgrad_reduce_body(part, out, splits, T, RPad, CPad, R, C, G, lgG, blockIdx.x, red);
}

// the reductions of many layers in ONE launch (each alone is a latency-bound 5..9 us kernel between two big ones: 16 per training step);
// workgroup -> job by binary search over the block prefix.  Same arithmetic and order per output as wgrad_reduce_kernel.
constexpr int WRED_MAX_JOBS = 32;
struct WgradReduceMultiArgs {
    const float* part[WRED_MAX_JOBS]; float* out[WRED_MAX_JOBS];
    int splits[WRED_MAX_JOBS], T[WRED_MAX_JOBS], RPad[WRED_MAX_JOBS], CPad[WRED_MAX_JOBS], R[WRED_MAX_JOBS], C[WRED_MAX_JOBS];
    signed char lg[WRED_MAX_JOBS];
    int bstart[WRED_MAX_JOBS + 1];
    int n;
};
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const WgradReduceMultiArgs a) {
    __shared__ double red[256][4];
    const int b = blockIdx.x;
    int lo = 0, hi = a.n;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.bstart[mid] <= b) lo = mid; else hi = mid; }
    const int j = lo;
    wgrad_reduce_body(a.part[j], a.out[j], a.splits[j], a.T[j], a.RPad[j], a.CPad[j], a.R[j], a.C[j], 1 << a.lg[j], a.lg[j], (unsigned)(b - a.bstart[j]), red);
}

// rows that are not a multiple of 4 floats (first conv: CPad = in_channels): same scheme with one element per thread; a slab is
// one contiguous run of T*RPad*CPad floats, so consecutive threads read consecutive addresses
__global__ __launch_bounds__(256) void wgrad_reduce_scalar_kernel(const float* __restrict__ part, float* __restrict__ out, int splits, int T,
                                                                  int RPad, int CPad, int R, int C, int G, int lgG) {
    __shared__ double red[256];
    const size_t slab = (size_t)T * RPad * CPad;
    const int g = threadIdx.x & (G - 1);
    const size_t i = (size_t)blockIdx.x * (256 >> lgG) + (threadIdx.x >> lgG);     // element of the slab
    const bool on = i < slab;
    double acc = 0.0;
    if (on) {
        const float* p = part + i;
        int k = g;
        for (; k + 7 * G < splits; k += 8 * G) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + u * G) * slab];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (double)v[u];
        }
        for (; k < splits; k += G) acc += (double)p[(size_t)k * slab];
    }
    if (G > 1) {
        red[threadIdx.x] = acc;
        __syncthreads();
        if (g == 0) { double tot = 0.0; for (int j = 0; j < G; ++j) tot += red[threadIdx.x + j]; acc = tot; }
    }
    if (on && g == 0) {
        const int c = (int)(i % CPad); const size_t rr = i / CPad; const int r = (int)(rr % RPad); const int t = (int)(rr / RPad);
        if (c < C && r < R) out[((size_t)r * C + c) * T + t] = (float)acc;
    }
}

void wgeo(ConvKind kind, int& TD, int& TH) { if (kind == CONV_K3_PLANAR) { TD = 1; TH = 8; } else if (kind == CONV_K3) { TD = 2; TH = 4; } else { TD = 2; TH = 8; } }

int tiles_per_split(int ntiles, int other, int resident = 512) {
    // Two workgroups fit a CU (LDS), 256 CUs: aim at exactly 512 workgroups in total so the launch is ONE full
    // residency round (no partially filled tail round) and the partial slabs stay small.  (The Winograd kernel keeps
    // one workgroup per CU: 256.)
    int want = resident / (other > 0 ? other : 1);
    if (want < 1) want = 1;
    return cdiv(ntiles, want);
}

// compute units a one-round weight-gradient launch leaves to a collective's resident workgroups (clamped: at least half the chip works)
int wgrad_reserve(int cu_reserve) { return cu_reserve < 0 ? 0 : (cu_reserve > 128 ? 128 : cu_reserve); }

}  // namespace

int wgrad_splits(ConvKind kind, int N, int D, int H, int W, int Cin, int Cout, int cu_reserve) {
    // (POINT: D, H, W are the INPUT dims of the transposed conv; sd is not known here, the GEMM kernel serves sd = 1 and 2 alike)
    if (kind == CONV_POINT && upconv_wgrad_ok(Cin, Cout, 2)) return upconv_wgrad_splits(N, D, H, W, Cin, Cout);
    if (wgrad_use_wino2d(kind, Cin, Cout)) return wgrad_wino2d_splits(N, D, H, W, Cin, Cout);
    int TD, TH; wgeo(kind, TD, TH);
    const int ntiles = N * cdiv(D, TD) * cdiv(H, TH) * cdiv(W, 16);
    const int other = cdiv(Cout, 32) * cdiv(Cin, 32) * (kind == CONV_POINT ? 8 : 1);   // POINT: one workgroup per tap too
    return cdiv(ntiles, tiles_per_split(ntiles, other, wgrad_use_wino(kind) ? 256 - wgrad_reserve(cu_reserve) : 512));
}

int launch_wgrad_mfma(ConvKind kind, WgradArgs a, hipStream_t s) {
    E3_REQUIRE(a.Cin % 4 == 0 && a.Cout % 4 == 0 && a.x_ldc % 4 == 0 && a.dy_ldc % 4 == 0, E3_ERR_UNSUPPORTED, "wgrad needs channel counts that are multiples of 4");
    E3_REQUIRE(!a.dy_chunk || (kind == CONV_K3 && wgrad_use_wino(kind)), E3_ERR_UNSUPPORTED, "channel-chunked dy: Winograd 3x3x3 weight gradient only");
    if (kind == CONV_POINT && upconv_wgrad_ok(a.Cin, a.Cout, a.sd)) return launch_upconv_wgrad(a, s);   // upconv_gemm.hip
    if (wgrad_use_wino2d(kind, a.Cin, a.Cout)) return launch_wgrad_wino2d(a, s);                        // wgrad_wino2d.hip
    int TD, TH; wgeo(kind, TD, TH);
    const int tD = cdiv(a.D, TD), tH = cdiv(a.H, TH), tW = cdiv(a.W, 16);
    const int ntiles = a.N * tD * tH * tW;
    const int co_tiles = cdiv(a.Cout, 32), ci_tiles = cdiv(a.Cin, 32);
    const int tps = tiles_per_split(ntiles, co_tiles * ci_tiles * (kind == CONV_POINT ? 8 : 1), wgrad_use_wino(kind) ? 256 - wgrad_reserve(a.cu_reserve) : 512);
    const int splits = cdiv(ntiles, tps);
    E3_REQUIRE(splits == a.splits, E3_ERR_INVALID, "wgrad: splits mismatch");
    if (kind == CONV_POINT) {
        const int T = a.sd * 4;
        const size_t lds = 2 * 256 * 32 * 4;
        const dim3 grid((unsigned)((size_t)splits * T * co_tiles * ci_tiles));
        hipLaunchKernelGGL(wgrad_point_kernel, grid, dim3(256), lds, s, a, tD, tH, tW, tps, T, co_tiles, ci_tiles);
    } else {
        const dim3 grid((unsigned)((size_t)splits * co_tiles * ci_tiles));
        if (wgrad_use_wino(kind)) return launch_wgrad_wino(a, tD, tH, tW, tps, co_tiles, ci_tiles, splits, s);   // wgrad_wino.hip
        if (kind == CONV_K3) {
            using G = WGeo<3, 2, 4>;
            auto kern = wgrad_conv_kernel<3, 2, 4>;
            static bool set = false;
            if (!set) { E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES)); set = true; }
            hipLaunchKernelGGL(kern, grid, dim3(256), G::LDS_BYTES, s, a, tD, tH, tW, tps, co_tiles, ci_tiles);
        } else {
            using G = WGeo<1, 1, 8>;
            auto kern = wgrad_conv_kernel<1, 1, 8>;
            static bool set = false;
            if (!set) { E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES)); set = true; }
            hipLaunchKernelGGL(kern, grid, dim3(256), G::LDS_BYTES, s, a, tD, tH, tW, tps, co_tiles, ci_tiles);
        }
    }
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

static void wgrad_reduce_shape(size_t items, int splits, int& lg, size_t& blocks) {
    int G = 1; lg = 0;                                  // threads per item: enough for ~64 K threads, never more than the splits
    while (G < 64 && (size_t)G * items < 65536 && 2 * G <= splits) { G *= 2; ++lg; }
    const size_t ipb = 256 >> lg;
    blocks = (items + ipb - 1) / ipb;
}

int launch_wgrad_reduce_multi(const WgradReduceJob* jobs, int njobs, hipStream_t s) {
    for (int j0 = 0; j0 < njobs; j0 += WRED_MAX_JOBS) {
        WgradReduceMultiArgs a;
        a.n = njobs - j0 < WRED_MAX_JOBS ? njobs - j0 : WRED_MAX_JOBS;
        size_t b = 0;
        for (int j = 0; j < a.n; ++j) {
            const WgradReduceJob& q = jobs[j0 + j];
            E3_REQUIRE(q.CPad % 4 == 0, E3_ERR_INVALID, "wgrad_reduce_multi: rows must be multiples of 4 floats");
            int lg; size_t blocks;
            wgrad_reduce_shape((size_t)q.T * q.R * ((q.C + 3) / 4), q.splits, lg, blocks);
            a.part[j] = q.part; a.out[j] = q.out; a.splits[j] = q.splits; a.T[j] = q.T; a.RPad[j] = q.RPad; a.CPad[j] = q.CPad; a.R[j] = q.R; a.C[j] = q.C;
            a.lg[j] = (signed char)lg; a.bstart[j] = (int)b;
            b += blocks;
        }
        a.bstart[a.n] = (int)b;
        if (b > 0) hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)b), dim3(256), 0, s, a);
        E3_CHECK_HIP(hipGetLastError());
    }
    return E3_OK;
}

int launch_wgrad_reduce(const float* part, float* out, int splits, int T, int RPad, int CPad, int R, int C, hipStream_t s) {
    const bool vec = CPad % 4 == 0;
    const size_t items = vec ? (size_t)T * R * ((C + 3) / 4) : (size_t)T * RPad * CPad;     // output quads / slab elements
    if (items == 0) return E3_OK;
    int G = 1, lg = 0;                                  // threads per item: enough for ~64 K threads, never more than the splits
    while (G < 64 && (size_t)G * items < 65536 && 2 * G <= splits) { G *= 2; ++lg; }
    const size_t ipb = 256 >> lg;
    const dim3 grid((unsigned)((items + ipb - 1) / ipb));
    if (vec) hipLaunchKernelGGL(wgrad_reduce_kernel, grid, dim3(256), 0, s, part, out, splits, T, RPad, CPad, R, C, G, lg);
    else hipLaunchKernelGGL(wgrad_reduce_scalar_kernel, grid, dim3(256), 0, s, part, out, splits, T, RPad, CPad, R, C, G, lg);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
