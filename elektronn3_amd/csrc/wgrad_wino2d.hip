// Weight gradient of the planar 1x3x3 convolution as Winograd F(3x3, 2x2) on the fp32 matrix cores.
//
//   dW[co][ci][kh][kw] = sum over 2x2 output tiles t:  sum_o dY[2t + o][co] * X[2t + o + k - 1][ci]
//
// Same construction as the 3D kernel (wgrad_wino.hip: Xt = B^T x B, Yt = G y G^T with the 1/2 factors moved to the output
// side, dW = A^T M A), one dimension less: 16 multiplies per tile and (ci, co) pair instead of 36.
//
// Work decomposition: a workgroup owns a (64 co x 32 ci) tile of all 16 positions over a contiguous range of 4x16-voxel
// bricks of the d-slices (2 x 8 tiles); wave w owns the 4 positions with ph = w for both co halves = 8 accumulators of
// v_mfma_f32_32x32x2_f32 (K = 2 tiles per instruction: lane half hf supplies tile row hf, k-step s tile column s).
// 128 accumulator registers => two workgroups per CU.  Staging by LDS-DMA into [voxel][32 ci] / [voxel][64 co] images,
// double-buffered; lane (j, hf) transforms channel ci0+j of X and channels co0+j, co0+32+j of dY for its 8 tiles.
// Epilogue: A^T over pw in registers, over ph through LDS (one co half at a time), slab part[split][tap][co][ci].
#include "kernels.h"

namespace {

constexpr int Q_LW = 18, Q_NV = 6 * 18, Q_MV = 64;       // X halo 6x18, dY brick 4x16
constexpr int Q_XW = 14, Q_GW = 16;                      // wave-pieces (1 KB) of X (13.5 used) / dY per brick
constexpr int Q_XS = Q_XW * 256;                         // X image floats incl. the tail of the last piece
constexpr int Q_BUF = Q_XS + Q_GW * 256;                 // one stage: 30 KB
constexpr int Q_EX = 4 * 3 * 4 * 64 * 4;                 // epilogue exchange of ONE co half: [ph][kw][r/4][lane][4] floats (48 KB)
constexpr int Q_LDS_FLOATS = 2 * Q_BUF > Q_EX ? 2 * Q_BUF : Q_EX;
typedef __attribute__((address_space(3))) void* lds_ptr_q;

__global__ __launch_bounds__(256, 2) void wgrad_wino2d_kernel(const WgradArgs a, int tilesH, int tilesW, int bricks_per_split,
                                                              int co_tiles, int ci_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hf = lane >> 5;
    constexpr unsigned OOB = 0x80000000u;
    unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int ci_t = L % ci_tiles; L /= ci_tiles;
    const int co_t = L % co_tiles; const int split = L / co_tiles;
    const int ci0 = ci_t * 32, co0 = co_t * 64;
    const int nbricks = a.N * a.D * tilesH * tilesW;
    const int b0 = split * bricks_per_split;
    const int b1 = b0 + bricks_per_split < nbricks ? b0 + bricks_per_split : nbricks;

    // one descriptor per d-slice (rebuilt per brick, scalar work): offsets stay small whatever the size of the tensor
    __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, 0x7fffffff, 0x00020000);
    const size_t slice_x = (size_t)a.H * a.W * a.x_ldc, slice_g = (size_t)a.H * a.W * a.dy_ldc;
    // lane constants of the DMA pieces (validity of a brick's halo = one scalar mask: 6 h bits | 18 w bits)
    unsigned xpm[4], xrel[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int wp = it * 4 + wave < Q_XW ? it * 4 + wave : Q_XW - 1;
        const int idx = wp * 64 + lane;
        const int v = idx >> 3, q = idx & 7;
        const int zw = v % Q_LW, zh = v / Q_LW;
        const bool ok = v < Q_NV && ci0 + 4 * q < a.Cin;
        xpm[it] = ok ? (1u << zh) | (1u << (6 + zw)) : 0xffffffffu;          // all-ones never matches
        xrel[it] = (unsigned)(((zh * a.W + zw) * a.x_ldc + ci0 + 4 * q) * 4);
    }
    unsigned gpm[4], grel[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = (it * 4 + wave) * 64 + lane;
        const int v = idx >> 4, q = idx & 15;
        const int ww = v & 15, hh = v >> 4;
        const bool ok = co0 + 4 * q < a.Cout;
        gpm[it] = ok ? (1u << hh) | (1u << (6 + ww)) : 0xffffffffu;
        grel[it] = (unsigned)(((hh * a.W + ww) * a.dy_ldc + co0 + 4 * q) * 4);
    }

    // ---- read plan.  H pass of Winograd row ph = wave:  X: 0: r0 - r2, 1: r1 + r2, 2: r2 - r1, 3: r1 - r3;  Y: y0, y0+y1, y0-y1, y1
    const int ha = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int hb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sgn = wave == 1 ? 1.f : -1.f;
    const float ya = wave == 3 ? 0.f : 1.f, yb = wave == 0 ? 0.f : (wave == 2 ? -1.f : 1.f);
    float m1 = -1.f;
    asm volatile("" : "+s"(m1));
    const int xrd_a = ((2 * hf + ha) * Q_LW) * 32 + j;            // + zw * 32
    const int xrd_b = ((2 * hf + hb) * Q_LW) * 32 + j;
    const int yrd = Q_XS + ((2 * hf) * 16) * 64 + j;              // + (oh * 16 + w) * 64 + 32 ct

    f32x16 acc[4][2];     // [pw][co half]
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][c][r] = 0.f;

    auto range_mask = [](int lo, int n, int size) {      // bits z in [0, n) with lo + z in [0, size)
        const int first = lo < 0 ? -lo : 0, last = size - lo < n ? size - lo : n;
        return last > first ? ((1u << last) - 1u) & ~((1u << first) - 1u) : 0u;
    };
    unsigned xmask = 0, gmask = 0, xbase = 0, gbase = 0;
    auto issue_setup = [&](int brick) {
        int Lt = brick < b1 ? brick : b1 - 1;             // (past the end the last brick harmlessly re-stages itself)
        const int tw_ = Lt % tilesW; Lt /= tilesW; const int th_ = Lt % tilesH; Lt /= tilesH; const int d = Lt % a.D; const int nb = Lt / a.D;
        const int h0 = th_ * 4, w0 = tw_ * 16;
        xmask = range_mask(h0 - 1, 6, a.H) | (range_mask(w0 - 1, 18, a.W) << 6);
        gmask = range_mask(h0, 4, a.H) | (range_mask(w0, 16, a.W) << 6);
        x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + ((size_t)nb * a.D + d) * slice_x, 0, 0x7fffffff, 0x00020000);
        g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy) + ((size_t)nb * a.D + d) * slice_g, 0, 0x7fffffff, 0x00020000);
        xbase = (unsigned)((((h0 - 1) * a.W + w0 - 1) * a.x_ldc) * 4);     // relative to the slice; wraps at the borders
        gbase = (unsigned)(((h0 * a.W + w0) * a.dy_ldc) * 4);
    };
    auto issue_part = [&](int part, float* buf) {         // part 0, 1: two X and two dY pieces each (none in front of the last MFMA blocks)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if ((it >> 1) != part) continue;
            const int wp = it * 4 + wave < Q_XW ? it * 4 + wave : Q_XW - 1;
            const bool okx = (xmask & xpm[it]) == xpm[it];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_q)(buf + wp * 256), 16, okx ? xrel[it] + xbase : OOB, 0, 0, 0);
            const bool okg = (gmask & gpm[it]) == gpm[it];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(g_rs, (lds_ptr_q)(buf + Q_XS + (it * 4 + wave) * 256), 16, okg ? grel[it] + gbase : OOB, 0, 0, 0);
        }
    };

    // one brick = 2 halves of 4 tile columns (k-steps 4 hc .. 4 hc + 3)
    auto compute = [&](const float* buf, float* nxt) {
#pragma unroll
        for (int hc = 0; hc < 2; ++hc) {
            // ---- X: H-combined row of this lane's tile row, w window 8 hc + [0, 10) serves tile columns 4 hc .. 4 hc + 3
            float u[10];
#pragma unroll
            for (int w = 0; w < 10; ++w) u[w] = buf[xrd_a + (8 * hc + w) * 32] + sgn * buf[xrd_b + (8 * hc + w) * 32];
            float X[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float* r = &u[2 * t];
                X[t][0] = r[0] + m1 * r[2]; X[t][1] = r[1] + r[2]; X[t][2] = r[2] + m1 * r[1]; X[t][3] = r[1] + m1 * r[3];
            }
            // ---- Y: the two dY rows of the tile row, both co halves, w window 8 hc + [0, 8)
            float Y[2][4][4];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                float g[8];
#pragma unroll
                for (int w = 0; w < 8; ++w)
                    g[w] = ya * buf[yrd + (8 * hc + w) * 64 + 32 * ct] + yb * buf[yrd + (16 + 8 * hc + w) * 64 + 32 * ct];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float g0 = g[2 * t], g1 = g[2 * t + 1];
                    Y[ct][t][0] = g0; Y[ct][t][1] = g0 + g1; Y[ct][t][2] = g0 + m1 * g1; Y[ct][t][3] = g1;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            issue_part(hc, nxt);                          // the next brick's DMA is issued in front of the MFMA blocks
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        acc[p][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(Y[ct][t][p], X[t][p], acc[p][ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (b0 < b1) {
        issue_setup(b0);
        issue_part(0, smem); issue_part(1, smem);
        __syncthreads();                     // (hipcc drains vmcnt before the barrier: the DMA has landed)
        int par = 0;
        for (int b = b0; b < b1; ++b) {
            issue_setup(b + 1);
            compute(smem + par * Q_BUF, smem + (par ^ 1) * Q_BUF);
            __syncthreads();
            par ^= 1;
        }
    }

    // ---- epilogue: A^T rows  M0 + M1/2 + M2/2,  M1/2 - M2/2,  M1/2 + M2/2 - M3  over pw in registers, over ph through LDS
    float hlf = 0.5f;
    asm volatile("" : "+s"(hlf));
    float* ex = smem;     // [ph][kw 3][r/4][lane][4]
    for (int ct = 0; ct < 2; ++ct) {
        __syncthreads();                                 // (staging buffers / the previous half's exchange are free)
        {
            const f32x16 s12 = hlf * (acc[1][ct] + acc[2][ct]), d12 = hlf * (acc[1][ct] + m1 * acc[2][ct]);
            const f32x16 c[3] = {acc[0][ct] + s12, d12, s12 + m1 * acc[3][ct]};
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = c[kw][4 * k4 + e];
                    *reinterpret_cast<f32x4*>(ex + (((wave * 3 + kw) * 4 + k4) * 64 + lane) * 4) = v;
                }
        }
        __syncthreads();
        if (wave < 3) {                                   // wave kw sums the ph axis and writes taps (kh, kw), kh = 0..2
            const int kw = wave;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                f32x4 m[4];
#pragma unroll
                for (int ph = 0; ph < 4; ++ph) m[ph] = *reinterpret_cast<const f32x4*>(ex + (((ph * 3 + kw) * 4 + k4) * 64 + lane) * 4);
                const f32x4 s12 = hlf * (m[1] + m[2]);
                f32x4 w3[3];
                w3[0] = m[0] + s12; w3[1] = hlf * (m[1] + m1 * m[2]); w3[2] = s12 + m1 * m[3];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * k4 + e;
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * hf;
                        a.part[(((size_t)split * 9 + kh * 3 + kw) * a.CoPad + co0 + 32 * ct + row) * a.CiPad + ci0 + j] = w3[kh][e];
                    }
            }
        }
    }
}

int bricks_per_split2d(int nbricks, int pairs) {
    int want = 512 / (pairs > 0 ? pairs : 1);             // two resident workgroups per CU
    if (want < 1) want = 1;
    return cdiv(nbricks, want);
}

}  // namespace

bool wgrad_use_wino2d(ConvKind kind, int Cin, int Cout) {
    static const bool enabled = getenv("E3_WGRAD_NO_WINO") == nullptr && getenv("E3_WGRAD_NO_WINO2D") == nullptr;
    return enabled && kind == CONV_K3_PLANAR && (Cout & 63) == 0 && (Cin & 31) == 0;
}

int wgrad_wino2d_splits(int N, int D, int H, int W, int Cin, int Cout) {
    const int nbricks = N * D * cdiv(H, 4) * cdiv(W, 16);
    return cdiv(nbricks, bricks_per_split2d(nbricks, (Cout / 64) * (Cin / 32)));
}

int launch_wgrad_wino2d(WgradArgs a, hipStream_t s) {
    E3_REQUIRE((a.x_ldc & 3) == 0 && (a.dy_ldc & 3) == 0 && ((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.dy & 15) == 0, E3_ERR_INVALID,
               "planar wgrad views must be 16-byte aligned");
    E3_REQUIRE((size_t)a.H * a.W * (size_t)(a.x_ldc > a.dy_ldc ? a.x_ldc : a.dy_ldc) * 4 < 0x7fffffffu, E3_ERR_UNSUPPORTED,
               "planar Winograd wgrad: a d-slice beyond 2 GiB (32-bit buffer offsets); set E3_WGRAD_NO_WINO2D=1");
    const int tH = cdiv(a.H, 4), tW = cdiv(a.W, 16);
    const int nbricks = a.N * a.D * tH * tW;
    const int co_tiles = a.Cout / 64, ci_tiles = a.Cin / 32;
    const int bps = bricks_per_split2d(nbricks, co_tiles * ci_tiles);
    const int splits = cdiv(nbricks, bps);
    E3_REQUIRE(splits == a.splits, E3_ERR_INVALID, "planar wgrad: splits mismatch");
    constexpr int lds_bytes = Q_LDS_FLOATS * 4;
    static bool set = false;
    if (!set) { E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_wino2d_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); set = true; }
    const dim3 grid((unsigned)((size_t)splits * co_tiles * ci_tiles));
    hipLaunchKernelGGL(wgrad_wino2d_kernel, grid, dim3(256), lds_bytes, s, a, tH, tW, bps, co_tiles, ci_tiles);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
