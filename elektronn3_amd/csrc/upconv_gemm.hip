// Transposed convolution k = s = (sd,2,2) as a plain LDS-tiled fp32-MFMA GEMM (forward with 2x scatter, dgrad with 2x gather).
//
// Replaces nn.ConvTranspose3d(Cin, Cout, kernel_size=2, stride=2) of upconv2() (unet.py:152-165) and its input gradient.
// The taps do not overlap, so forward is   Y[M = voxels][N = (tap, co)] = X[M][K = ci] * W[K][N]   with row p of column
// group `tap` written to voxel 2p + tap, and dgrad is   dX[M][N = ci] = dYg[M][K = (tap, co)] * W'[K][N]   with row p reading
// voxel 2p + tap for the K range of `tap`.  The generic implicit-GEMM kernel (conv_mfma.hip, POINT mode) walks these as
// "1-tap convolutions" in 16-channel chunks with two barriers and an exposed weight fetch per 32 MFMAs: 45 TF.  Here:
//   * workgroup = 128 voxels (1x8x16 brick) x 32*NT columns, each of the 4 waves owns 32 rows x all NT column tiles;
//   * K in chunks of 32: A (128 x 32) and B (32*NT x 32) staged in LDS as 128-B rows whose 16-B pieces are XOR-swizzled
//     with (row >> 1) & 7 -> conflict-free ds_read_b128 fragments (4 k-steps per read, as in conv_v3.hip);
//   * the global loads of chunk c+1 are issued into registers before the MFMAs of chunk c (no vector-memory wait in the
//     MFMA loop), 16*NT MFMAs per wave between barriers;
//   * epilogue: bias / folded eval-BN + ReLU, per-(brick, tap) Welford statistics for the train-mode BN that follows,
//     rows transposed through a per-wave LDS tile so that every lane stores 16 B of whole 128-B voxel rows.
#include "kernels.h"

namespace {

constexpr int U_TH = 8, U_M = 128, U_CK = 32;

__device__ __forceinline__ int swz(int row, int piece) { return row * 32 + 4 * (piece ^ ((row >> 1) & 7)); }

template <bool GATHER, int NT>
__global__ __launch_bounds__(256, 2) void upconv_gemm_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                      // [128][32]
    float* Bs = smem + U_M * 32;           // [32 NT][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hf = lane >> 5;

    unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int ntile = L % a.ntiles; L /= a.ntiles;
    const int tw_ = L % a.tilesW; L /= a.tilesW;
    const int th_ = L % a.tilesH; L /= a.tilesH;
    const int d0 = L % a.D; const int nb = L / a.D;
    const int h0 = th_ * U_TH, w0 = tw_ * 16;
    const int n0 = ntile * 32 * NT;
    const int mtile = ((nb * a.D + d0) * a.tilesH + th_) * a.tilesW + tw_;
    const int Cx = a.Cin;                                  // channels per voxel of x
    const int K = GATHER ? a.G * Cx : Cx;
    const int NCH = K / U_CK;

    // ---- staging plan: A piece idx = tid + 256 it -> (row m, 16-B piece q); B likewise with rows = columns of the tile
    int a_vox[4], a_dst[4];              // x voxel offset (floats, -1 = outside) for SCATTER; (gh, gw) packed for GATHER
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = tid + it * 256;
        const int m = idx >> 3, q = idx & 7;
        const int gh = h0 + (m >> 4), gw = w0 + (m & 15);
        const bool ok = gh < a.H && gw < a.W;
        if (GATHER) a_vox[it] = ok ? (gh << 16) | gw : -1;
        else a_vox[it] = ok ? (((nb * a.D + d0) * a.H + gh) * a.W + gw) * a.x_ldc + 4 * q : -1;
        a_dst[it] = swz(m, q);
    }
    const int aq4 = 4 * (tid & 7);
    int b_src[NT], b_dst[NT];
#pragma unroll
    for (int it = 0; it < NT; ++it) {
        const int idx = tid + it * 256;
        const int nl = idx >> 3, q = idx & 7;
        b_src[it] = (n0 + nl) * Cx + 4 * q;           // + chunk offset (+ tap * NPad * Cx for GATHER)
        b_dst[it] = swz(nl, q);
    }
    // fragment offsets: k-group g (8 channels) -> piece 2g + hf of row (wave*32 + j) / column (32 ns + j)
    const int arow = wave * 32 + j;
    int afrag[4], bfrag[4][NT];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        afrag[g] = swz(arow, 2 * g + hf);
#pragma unroll
        for (int ns = 0; ns < NT; ++ns) bfrag[g][ns] = swz(32 * ns + j, 2 * g + hf);
    }

    f32x16 acc[NT];
#pragma unroll
    for (int ns = 0; ns < NT; ++ns)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ns][r] = 0.f;

    f32x4 xa[4], xb[NT];
    auto issue = [&](int c) {
        int cb = c * U_CK, tap = 0;
        size_t boff = 0;
        if (GATHER) { tap = cb / Cx; cb -= tap * Cx; boff = (size_t)tap * a.NPad * Cx; }
        const int utd = tap >> 2, uth = (tap >> 1) & 1, utw = tap & 1;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int v = a_vox[it];
            xa[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (GATHER) {
                const int od = a.sd * d0 + utd, oh = 2 * (v >> 16) + uth, ow = 2 * (v & 0xffff) + utw;
                if (v >= 0 && od < a.Do && oh < a.Ho && ow < a.Wo)
                    xa[it] = *reinterpret_cast<const f32x4*>(a.x + (size_t)(((nb * a.Do + od) * a.Ho + oh) * a.Wo + ow) * a.x_ldc + cb + aq4);
            } else if (v >= 0) xa[it] = *reinterpret_cast<const f32x4*>(a.x + v + cb);
        }
#pragma unroll
        for (int it = 0; it < NT; ++it) xb[it] = *reinterpret_cast<const f32x4*>(a.wt + boff + b_src[it] + cb);
    };
    issue(0);
    for (int c = 0; c < NCH; ++c) {
        if (c > 0) __syncthreads();                  // every wave is done reading the previous chunk
#pragma unroll
        for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(As + a_dst[it]) = xa[it];
#pragma unroll
        for (int it = 0; it < NT; ++it) *reinterpret_cast<f32x4*>(Bs + b_dst[it]) = xb[it];
        __syncthreads();
        issue(c + 1 < NCH ? c + 1 : c);              // in flight during the MFMAs below (the last chunk re-loads itself: no branch)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(As + afrag[g]);
            f32x4 bv[NT];
#pragma unroll
            for (int ns = 0; ns < NT; ++ns) bv[ns] = *reinterpret_cast<const f32x4*>(Bs + bfrag[g][ns]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ns = 0; ns < NT; ++ns) acc[ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[ns][e], acc[ns], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue
    const bool do_stats = !GATHER && a.stats != nullptr;
    const bool aff = !GATHER && a.epi_scale != nullptr;
    constexpr int SCR = 4 * NT * 32 * 3;
    __syncthreads();                                  // LDS is reused: statistics scratch + one 32x32 store tile per wave
    float* tile = smem + SCR + wave * 1024;
#pragma unroll
    for (int ns = 0; ns < NT; ++ns) {
        const int nt0 = n0 + 32 * ns;                 // first column of this tile (wave-uniform)
        const int n = nt0 + j;
        const bool nvalid = n < a.Ncols;
        int ut = 0, c0 = nt0;
        if (!GATHER) { ut = nt0 / a.Cout; c0 = nt0 - ut * a.Cout; }
        const int utd = ut >> 2, uth = (ut >> 1) & 1, utw = ut & 1;
        const int co = c0 + j;
        const float bias = (a.bias && nvalid) ? a.bias[co] : 0.f;
        float es = 1.f, eh = 0.f;
        if (aff && nvalid) { es = a.epi_scale[co]; eh = a.epi_shift[co]; }
        float cnt = 0.f, sum = 0.f;
        unsigned okmask = 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hf;
            const int m = wave * 32 + row;
            const int gh = h0 + (m >> 4), gw = w0 + (m & 15);
            bool ok = nvalid && gh < a.H && gw < a.W;
            if (!GATHER) ok = ok && a.sd * d0 + utd < a.Do && 2 * gh + uth < a.Ho && 2 * gw + utw < a.Wo;
            float v = acc[ns][r] + bias;
            if (aff) v = fmaxf(__builtin_fmaf(v, es, eh), 0.f);
            acc[ns][r] = v;
            tile[row * 32 + j] = v;
            cnt += ok ? 1.f : 0.f; sum += ok ? v : 0.f; okmask |= (ok ? 1u : 0u) << r;
        }
        // (same wave wrote the tile: LDS operations of one wave are ordered)
        const int c4 = 4 * (lane & 7);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int trow = 8 * p + (lane >> 3);
            const int m = wave * 32 + trow;
            const int gh = h0 + (m >> 4), gw = w0 + (m & 15);
            bool ok = nt0 + c4 < a.Ncols && gh < a.H && gw < a.W;
            size_t off;
            if (GATHER) off = (size_t)(((nb * a.D + d0) * a.H + gh) * a.W + gw) * a.y_ldc + nt0 + c4;
            else {
                const int od = a.sd * d0 + utd, oh = 2 * gh + uth, ow = 2 * gw + utw;
                ok = ok && od < a.Do && oh < a.Ho && ow < a.Wo;
                off = (size_t)(((nb * a.Do + od) * a.Ho + oh) * a.Wo + ow) * a.y_ldc + c0 + c4;
            }
            const f32x4 v = *reinterpret_cast<const f32x4*>(tile + trow * 32 + c4);
            if (ok) *reinterpret_cast<f32x4*>(a.y + off) = v;
        }
        if (do_stats) {
            float mean = cnt > 0.f ? sum / cnt : 0.f, m2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = acc[ns][r] - mean; m2 += ((okmask >> r) & 1u) ? d * d : 0.f; }
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
            const int nt0 = n0 + 32 * ns, n = nt0 + jj;
            if (n < a.Ncols) {
                float cnt = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float* sc = smem + ((w * NT + ns) * 32 + jj) * 3;
                    welford_merge(cnt, mean, m2, sc[0], sc[1], sc[2]);
                }
                const int ut = nt0 / a.Cout, co = n - ut * a.Cout;
                float* o = a.stats + ((size_t)(mtile * (a.Ncols / a.Cout) + ut) * a.Cout + co) * 3;
                o[0] = cnt; o[1] = mean; o[2] = m2;
            }
        }
    }
}

template <bool GATHER, int NT>
int launch_up(ConvArgs a, hipStream_t s) {
    a.tilesD = a.D; a.tilesH = cdiv(a.H, U_TH); a.tilesW = cdiv(a.W, 16);
    a.ntiles = a.NPad / (32 * NT);
    const size_t nblk = (size_t)a.N * a.D * a.tilesH * a.tilesW * a.ntiles;
    E3_REQUIRE(nblk > 0 && nblk < (1u << 31), E3_ERR_INVALID, "upconv grid out of range");
    constexpr int stage = (U_M + 32 * NT) * 32, epi = 4 * NT * 32 * 3 + 4 * 1024;
    constexpr int lds_bytes = (stage > epi ? stage : epi) * 4;
    hipLaunchKernelGGL((upconv_gemm_kernel<GATHER, NT>), dim3((unsigned)nblk), dim3(256), lds_bytes, s, a);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

}  // namespace

// Eligibility of the GEMM kernel for a POINT launch (everything else keeps the generic kernel): 32-channel granularity
// on both sides, 16-byte aligned views, at least 64 GEMM columns.
bool upconv_gemm_ok(int flags, int Cx, int Cout, int ncols) {
    static const bool enabled = getenv("E3_UPCONV_NO_GEMM") == nullptr;
    if (!enabled || !(flags & (CF_SCATTER_UP | CF_GATHER_UP))) return false;
    if ((Cx & 31) != 0 || ncols < 64) return false;
    if ((flags & CF_SCATTER_UP) && (Cout & 31) != 0) return false;
    return true;
}

int upconv_stats_parts(int N, int D, int H, int W, int sd) { return N * D * cdiv(H, U_TH) * cdiv(W, 16) * sd * 4; }

int launch_upconv_gemm(ConvArgs a, hipStream_t s) {
    E3_REQUIRE((a.x_ldc & 3) == 0 && (a.y_ldc & 3) == 0 && ((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.y & 15) == 0, E3_ERR_INVALID,
               "upconv views must be 16-byte aligned");
    const bool gather = (a.flags & CF_GATHER_UP) != 0;
    const bool nt4 = (a.NPad % 128) == 0;
    if (gather) return nt4 ? launch_up<true, 4>(a, s) : launch_up<true, 2>(a, s);
    return nt4 ? launch_up<false, 4>(a, s) : launch_up<false, 2>(a, s);
}
