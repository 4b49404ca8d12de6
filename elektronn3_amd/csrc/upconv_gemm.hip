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
#include <type_traits>

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
    // (needed region of a forward launch, ConvArgs::box_* in INPUT voxels of h and w: the bricks start at the box's low corner and stop at its high one;
    // the launcher sets [0, H) x [0, W) when it is off, so the statistics records keep their order)
    const int bh1 = GATHER ? a.H : a.box_hi[1], bw1 = GATHER ? a.W : a.box_hi[2];
    const int h0 = (GATHER ? 0 : a.box_lo[1]) + th_ * U_TH, w0 = (GATHER ? 0 : a.box_lo[2]) + tw_ * 16;
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
        const bool ok = gh < bh1 && gw < bw1;
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
            bool ok = nvalid && gh < bh1 && gw < bw1;
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
            bool ok = nt0 + c4 < a.Ncols && gh < bh1 && gw < bw1;
            size_t off;
            if (GATHER) off = (size_t)(((nb * a.D + d0) * a.H + gh) * a.W + gw) * a.y_ldc + nt0 + c4;
            else {
                const int od = a.sd * d0 + utd, oh = 2 * gh + uth, ow = 2 * gw + utw;
                ok = ok && od < a.Do && oh < a.Ho && ow < a.Wo;
                const size_t ov = (size_t)(((nb * a.Do + od) * a.Ho + oh) * a.Wo + ow);
                // (channel-chunked output, ConvArgs::y_chunk: [Cout / 8][voxel][8])
                off = a.y_chunk ? (size_t)((c0 + c4) >> 3) * a.y_chunk + ov * 8 + (c4 & 4) : ov * a.y_ldc + c0 + c4;
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
    if (GATHER || a.box_hi[0] <= 0 || a.stats) { a.box_lo[1] = a.box_lo[2] = 0; a.box_hi[1] = a.H; a.box_hi[2] = a.W; }
    else for (int i = 1; i < 3; ++i) {
        const int dim = i == 1 ? a.H : a.W;
        a.box_lo[i] = a.box_lo[i] < 0 ? 0 : a.box_lo[i]; a.box_hi[i] = a.box_hi[i] > dim ? dim : a.box_hi[i];
        E3_REQUIRE(a.box_hi[i] > a.box_lo[i], E3_ERR_INVALID, "upconv with a needed region: empty box");
    }
    a.tilesD = a.D; a.tilesH = cdiv(a.box_hi[1] - a.box_lo[1], U_TH); a.tilesW = cdiv(a.box_hi[2] - a.box_lo[2], 16);
    a.ntiles = a.NPad / (32 * NT);
    const size_t nblk = (size_t)a.N * a.D * a.tilesH * a.tilesW * a.ntiles;
    E3_REQUIRE(nblk > 0 && nblk < (1u << 31), E3_ERR_INVALID, "upconv grid out of range");
    constexpr int stage = (U_M + 32 * NT) * 32, epi = 4 * NT * 32 * 3 + 4 * 1024;
    constexpr int lds_bytes = (stage > epi ? stage : epi) * 4;
    hipLaunchKernelGGL((upconv_gemm_kernel<GATHER, NT>), dim3((unsigned)nblk), dim3(256), lds_bytes, s, a);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}


// ---- forward, the full-resolution shape (Cin = 64: 67 MB in, 268 MB out at L1 -> L0 of a start_filts = 32 network; 140 us = 0.30 of the HBM
// roof with the tiled kernel above, whose two-chunk K loop is all prologue and epilogue).  Persistent workgroups of EIGHT waves, one per CU: the
// weights of a 32-channel output tile (T taps x 32 co x 64 ci = 64 KB) are staged into LDS ONCE, then the workgroup walks 128-voxel tiles of the
// flattened low-resolution grid, each staged by LDS-DMA (256-byte rows, 16-byte pieces XOR-swizzled by row & 15 on the SOURCE side: conflict-free
// ds_read_b128 fragments); waves 0..3 and 4..7 take half of the taps each for the same 4 x 32 voxels (two waves per SIMD cover each other's LDS and
// store latencies); the next tile's DMA is issued after the MFMAs and lands during the epilogue, which is 2/3 of a tile's time.
// Measured (up_convs.2.upconv of cfg 2): 143 -> 105 us.  Tried and dropped: four waves with all taps and a double-buffered X tile (156 us: one wave per
// SIMD exposes every LDS / shuffle / store latency); the two halves one phase apart, MFMAs of one over the epilogue of the other (109 us: the
// epilogue's fp32 VALU work waits for issue slots behind the other wave's MFMAs -- fp32 MFMA and VALU share the FMA lanes).
// A = X (rows = voxels), B = W (columns = co): a lane holds one channel of 16 voxels, so the Welford statistics are per-lane sums (two-pass over
// the 16 registers, merged across taps and tiles per lane) and the results leave through a per-wave transposition tile as whole 128-byte rows.
typedef __attribute__((address_space(3))) void* lds_ptr_p;
constexpr unsigned P_OOB = 0x80000000u;
__device__ __forceinline__ void pdma16(__amdgpu_buffer_rsrc_t rs, lds_ptr_p dst, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, 0, 0, 0);
}

template <int SD>
__global__ __launch_bounds__(512, 1) void upconv_fwd_persist_kernel(const ConvArgs a, unsigned nvox, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pmem[];
    constexpr int T = 4 * SD, TW = T / 2;          // taps; taps per wave (waves 0..3: the first half, 4..7: the second, of the same 4 x 32 voxels)
    constexpr int WB = T * 32 * 256;               // weights: rows (tap, co) of 64 floats
    constexpr int XB = 128 * 256;                  // the X tile: 128 voxels x 64 floats
    constexpr int TP = 40;                         // transposition tile pitch in floats (the two lane halves hit disjoint banks)
    unsigned char* xs = pmem + WB;
    float* tiles = reinterpret_cast<float*>(pmem + WB + XB);           // [8 waves][32][TP]
    float* S = tiles + 8 * 32 * TP;                                    // [8][32][3]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int vw = wave & 3, t0 = (wave >> 2) * TW;
    const int j = lane & 31, g = lane >> 5;
    const int cb = blockIdx.y * 32;
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wt), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0x7fffffff, 0x00020000);
    for (int i = wave; i < T * 8; i += 8) {        // one wave-instruction = 1 KB = 4 rows
        const int R = i * 4 + (lane >> 4), slot = lane & 15;
        const int tap = R >> 5, jj = R & 31;
        pdma16(w_rs, (lds_ptr_p)(pmem + i * 1024), (unsigned)((((tap * a.Cout + cb + jj) * 64) + ((slot ^ (R & 15)) << 2)) * 4));
    }
    // needed region (ConvArgs::box_* in INPUT voxels of h and w; the launcher sets [0, H) x [0, W) when it is off): the flattened index runs over the box,
    // voxel v of it is (n d, h, w) = (v / (bH bW), box_lo[1] + (v / bW) % bH, box_lo[2] + v % bW) of the tensor
    const unsigned bW = (unsigned)(a.box_hi[2] - a.box_lo[2]), bH = (unsigned)(a.box_hi[1] - a.box_lo[1]);
    const bool boxed = bW != (unsigned)a.W || bH != (unsigned)a.H;
    auto box_voxel = [&](unsigned v) -> unsigned {          // index of box voxel v in the tensor
        if (!boxed) return v;
        const unsigned t = v / bW, w = v - t * bW, nd = t / bH, h = t - nd * bH;
        return (nd * (unsigned)a.H + h + (unsigned)a.box_lo[1]) * (unsigned)a.W + w + (unsigned)a.box_lo[2];
    };
    auto stage = [&](int tile) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (i * 8 + wave) * 4 + (lane >> 4), slot = lane & 15;
            const unsigned v = (unsigned)tile * 128u + row;
            pdma16(x_rs, (lds_ptr_p)(xs + (i * 8 + wave) * 1024), v < nvox ? (box_voxel(v) * (unsigned)a.x_ldc + ((slot ^ (row & 15)) << 2)) * 4u : P_OOB);
        }
    };
    const int row = vw * 32 + j;
    const unsigned xrd = (unsigned)(row * 256), wrd = (unsigned)(j * 256), sw = (unsigned)(j & 15);     // (row & 15 == j & 15)
    const float bias = a.bias ? a.bias[cb + j] : 0.f;
    const bool aff = a.epi_scale != nullptr;
    const float es = aff ? a.epi_scale[cb + j] : 1.f, eh = aff ? a.epi_shift[cb + j] : 0.f;
    float rc = 0.f, rm = 0.f, r2 = 0.f;            // running (count, mean, M2) of this lane's channel over its voxels and taps
    float* tl = tiles + wave * 32 * TP;
    const int c4 = 4 * (lane & 7);
    const size_t ych = a.y_chunk ? (size_t)((cb + c4) >> 3) * a.y_chunk + (c4 & 4) : (size_t)(cb + c4);      // the lane's channel offset inside a voxel row / its chunk plane

    if ((int)blockIdx.x < ntiles) stage(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x16 acc[TW];
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            const unsigned pc = ((2 * s4 + g) ^ sw) << 4;
            const f32x4 av = *reinterpret_cast<const f32x4*>(xs + xrd + pc);
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(pmem + (t0 + t) * 8192 + wrd + pc);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc[t], 0, 0, 0);
            }
        }
        __syncthreads();                                        // every wave is done with the X tile:
        if (tile + (int)gridDim.x < ntiles) stage(tile + gridDim.x);       // the next one is in flight during the epilogue
        // this lane's own voxel (row j of the wave's 32): output base and per-tap validity (autocrop, unet.py:289-299)
        const unsigned v = (unsigned)tile * 128u + row;
        const bool vin = v < nvox;
        unsigned r = vin ? v : 0u;
        const int w = a.box_lo[2] + (int)(r % bW); r /= bW; const int h = a.box_lo[1] + (int)(r % bH); r /= bH; const int d = r % a.D; const int n = (int)(r / a.D);
        // (channel-chunked output, ConvArgs::y_chunk: [Cout / 8][voxel][8] -- a voxel's row is 8 floats, this lane's 16 bytes go to plane (cb + c4) / 8)
        const int ys = a.y_chunk ? 8 : a.y_ldc;
        const unsigned obase = (unsigned)(((((size_t)n * a.Do + SD * d) * a.Ho + 2 * h) * a.Wo + 2 * w) * ys);
        unsigned okm = 0;
#pragma unroll
        for (int tap = 0; tap < T; ++tap)
            if (vin && SD * d + (tap >> 2) < a.Do && 2 * h + ((tap >> 1) & 1) < a.Ho && 2 * w + (tap & 1) < a.Wo) okm |= 1u << tap;
        const bool allok = __builtin_amdgcn_ballot_w64(okm == (1u << T) - 1u) == ~0ull;       // wave-uniform: the whole tile is stored
        unsigned obp[4], omp[4];                   // rows this lane stores in the transposed pass: 8 p + (lane >> 3)
#pragma unroll
        for (int p = 0; p < 4; ++p) { obp[p] = (unsigned)__shfl((int)obase, 8 * p + (lane >> 3)); omp[p] = (unsigned)__shfl((int)okm, 8 * p + (lane >> 3)); }
        auto epilogue = [&](auto all_ok) {
            constexpr bool ALL = decltype(all_ok)::value;
            unsigned okr[16];                      // validity of this lane's 16 voxels (rows (e & 3) + 8 (e >> 2) + 4 g): T bits each
            if constexpr (!ALL) {
#pragma unroll
                for (int e = 0; e < 16; ++e) okr[e] = (unsigned)__shfl((int)okm, (e & 3) + 8 * (e >> 2) + 4 * g);
            }
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                const int tap = t0 + t;
                float cnt = ALL ? 16.f : 0.f, sum = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float val = acc[t][e] + bias;
                    if (aff) val = fmaxf(__builtin_fmaf(val, es, eh), 0.f);
                    acc[t][e] = val;
                    tl[((e & 3) + 8 * (e >> 2) + 4 * g) * TP + j] = val;
                    if constexpr (ALL) sum += val;
                    else { const bool ok = ((okr[e] >> tap) & 1u) != 0; cnt += ok ? 1.f : 0.f; sum += ok ? val : 0.f; }
                }
                __builtin_amdgcn_wave_barrier();
                const unsigned toff = (unsigned)((((tap >> 2) * a.Ho + ((tap >> 1) & 1)) * a.Wo + (tap & 1)) * ys);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(tl + (8 * p + (lane >> 3)) * TP + c4);
                    if (ALL || ((omp[p] >> tap) & 1u)) *reinterpret_cast<f32x4*>(a.y + obp[p] + toff + ych) = o;
                }
                __builtin_amdgcn_wave_barrier();
                if (a.stats) {
                    const float mean = cnt > 0.f ? sum / cnt : 0.f;
                    float m2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float dv = acc[t][e] - mean;
                        if constexpr (ALL) m2 = __builtin_fmaf(dv, dv, m2);
                        else m2 += ((okr[e] >> tap) & 1u) ? dv * dv : 0.f;
                    }
                    welford_merge(rc, rm, r2, cnt, mean, m2);
                }
            }
        };
        if (allok) epilogue(std::true_type{}); else epilogue(std::false_type{});
    }
    if (!a.stats) return;
    {
        const float c2 = __shfl_xor(rc, 32), m2_ = __shfl_xor(rm, 32), q2 = __shfl_xor(r2, 32);
        welford_merge(rc, rm, r2, c2, m2_, q2);
    }
    if (g == 0) { float* sc = S + (wave * 32 + j) * 3; sc[0] = rc; sc[1] = rm; sc[2] = r2; }
    __syncthreads();
    if (tid < 32) {
        float cnt = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) { const float* sc = S + (wv * 32 + tid) * 3; welford_merge(cnt, mean, m2, sc[0], sc[1], sc[2]); }
        float* o = a.stats + ((size_t)blockIdx.x * a.Cout + cb + tid) * 3;
        o[0] = cnt; o[1] = mean; o[2] = m2;
    }
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

// the persistent forward above: Cin = 64, 32-channel output tiles, no prologue (E3_UPCONV_NO_PERSIST=1: A/B switch)
bool upconv_fwd_persist_ok(int flags, int Cx, int Cout) {
    static const bool enabled = getenv("E3_UPCONV_NO_PERSIST") == nullptr;
    return enabled && (flags & CF_SCATTER_UP) && Cx == 64 && (Cout & 31) == 0;
}
static int persist_wgs(size_t nvox) { const size_t t = (nvox + 127) / 128; return (int)(t < 256 ? t : 256); }

int upconv_stats_parts(int N, int D, int H, int W, int sd, int Cx, int Cout) {
    if (upconv_fwd_persist_ok(CF_SCATTER_UP, Cx, Cout)) return persist_wgs((size_t)N * D * H * W);      // one record per workgroup
    return N * D * cdiv(H, U_TH) * cdiv(W, 16) * sd * 4;
}

int launch_upconv_gemm(ConvArgs a, hipStream_t s) {
    E3_REQUIRE((a.x_ldc & 3) == 0 && (a.y_ldc & 3) == 0 && ((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.y & 15) == 0, E3_ERR_INVALID,
               "upconv views must be 16-byte aligned");
    const bool gather = (a.flags & CF_GATHER_UP) != 0;
    const size_t nvox = (size_t)a.N * a.D * a.H * a.W;
    E3_REQUIRE(!a.y_chunk || (!gather && !a.stats && (a.Cout & 31) == 0 && a.y_chunk * (size_t)(a.Cout / 8) * 4 < 0x7fffffffu), E3_ERR_INVALID,
               "upconv: bad channel-chunked output (forward without statistics, 32-channel tiles)");
    if (gather || a.box_hi[0] <= 0 || a.stats) { a.box_lo[1] = a.box_lo[2] = 0; a.box_hi[1] = a.H; a.box_hi[2] = a.W; a.box_hi[0] = 0; }
    if (!gather && !a.pro_scale && upconv_fwd_persist_ok(a.flags, a.Cin, a.Cout) && nvox * (size_t)a.x_ldc < (1ull << 29) &&
        (size_t)a.N * a.Do * a.Ho * a.Wo * (a.y_chunk ? 8 : a.y_ldc) < (1ull << 32)) {
        if (a.box_hi[0] > 0) for (int i = 1; i < 3; ++i) {
            const int dim = i == 1 ? a.H : a.W;
            a.box_lo[i] = a.box_lo[i] < 0 ? 0 : a.box_lo[i]; a.box_hi[i] = a.box_hi[i] > dim ? dim : a.box_hi[i];
            E3_REQUIRE(a.box_hi[i] > a.box_lo[i], E3_ERR_INVALID, "upconv with a needed region: empty box");
        }
        const size_t bvox = (size_t)a.N * a.D * (a.box_hi[1] - a.box_lo[1]) * (a.box_hi[2] - a.box_lo[2]);      // voxels the workgroups walk (the box; == nvox without one)
        const int tiles = (int)((bvox + 127) / 128), T = 4 * a.sd;
        const int lds = T * 32 * 256 + 128 * 256 + 8 * 32 * 40 * 4 + 8 * 32 * 3 * 4;
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute((const void*)upconv_fwd_persist_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)upconv_fwd_persist_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            done = true;
        }
        const dim3 grid((unsigned)persist_wgs(a.stats ? nvox : bvox), (unsigned)(a.Cout / 32));
        if (a.sd == 2) hipLaunchKernelGGL(upconv_fwd_persist_kernel<2>, grid, dim3(512), lds, s, a, (unsigned)bvox, tiles);
        else hipLaunchKernelGGL(upconv_fwd_persist_kernel<1>, grid, dim3(512), lds, s, a, (unsigned)bvox, tiles);
        E3_CHECK_HIP(hipGetLastError());
        return E3_OK;
    }
    // four column tiles per workgroup unless that leaves most of the chip idle (the data gradient of the lowest level: 32 bricks x 2 column groups)
    const size_t mt = (size_t)a.N * a.D * cdiv(a.H, U_TH) * cdiv(a.W, 16);
    const bool nt4 = (a.NPad % 128) == 0 && mt * (a.NPad / 128) >= 192;
    if (gather && !nt4 && mt * (a.NPad / 64) < 192 && a.NPad % 32 == 0) return launch_up<true, 1>(a, s);
    if (gather) return nt4 ? launch_up<true, 4>(a, s) : launch_up<true, 2>(a, s);
    return nt4 ? launch_up<false, 4>(a, s) : launch_up<false, 2>(a, s);
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[ci][co][tap] = sum_p X[p][ci] * dY[2p + tap][co]:  per tap a GEMM with K = input voxels.  One workgroup owns a
// (64 ci x 32 co) tile of ALL taps (wave w: taps {w*T/4 ...}, 2 ci tiles each = 4 (2 for planar) accumulators) over a
// contiguous range of 32-voxel bricks (2 h-rows x 16 w), so X is staged once for the 8 taps (the one-tap-per-workgroup
// kernel in wgrad_mfma.hip re-reads it 8 times and spends most of its time on per-element address arithmetic).
// Staging by LDS-DMA into [voxel][64 ci] / [tap][voxel][32 co] images, validity as scalar bit masks, zero fill by the buffer
// range check; single-buffered (40 KB), three workgroups per CU cover each other's staging.  Slab layout as before:
// part[split][tap][ci][co].
namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_u;

// (body as a device function template + two plain __global__ wrappers: hipcc 7.2 fails to emit the host stub of this
// kernel when it is itself a template)
template <int NTAPS>
__device__ __forceinline__ void upconv_wgrad_body(const WgradArgs& a, int tilesH, int tilesW, int bricks_per_split,
                                                  int co_tiles, int ci_tiles) {
    constexpr int TPW = NTAPS / 4;                           // taps per wave
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                                    // [32 voxels][64 ci]
    float* gs = smem + 32 * 64;                          // [NTAPS][32 voxels][32 co]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hf = lane >> 5;
    unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int co_t = L % co_tiles; L /= co_tiles;
    const int ci_t = L % ci_tiles; const int split = L / ci_tiles;
    const int ci0 = ci_t * 64, co0 = co_t * 32;
    const int nbricks = a.N * a.D * tilesH * tilesW;
    const int b0 = split * bricks_per_split;
    const int b1 = b0 + bricks_per_split < nbricks ? b0 + bricks_per_split : nbricks;

    // descriptors are rebuilt per brick for its d-slice (scalar work): offsets stay small whatever the size of the tensors
    const size_t slice_x = (size_t)a.H * a.W * a.x_ldc, slice_g = (size_t)a.Ho * a.Wo * a.dy_ldc;
    // lane constants of the DMA pieces.  X: wave-piece wp = 2*it' + ... (8 per brick, 2 per wave), voxel = idx >> 4, 16-B piece q = idx & 15
    unsigned xrel[2], xpm[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = (it * 4 + wave) * 64 + lane;
        const int v = idx >> 4, q = idx & 15;
        const int vh = v >> 4, vw = v & 15;
        const bool cok = ci0 + 4 * q < a.Cin;
        xrel[it] = (unsigned)(((vh * a.W + vw) * a.x_ldc + ci0 + 4 * q) * 4);
        xpm[it] = cok ? (1u << vh) | (1u << (2 + vw)) : 0xffffffffu;
    }
    // dY: per tap 4 wave-pieces (one per wave): voxel = idx >> 3, piece q = idx & 7 of the voxel 2p + tap
    const int gidx = wave * 64 + lane;
    const int gv = gidx >> 3, gq = gidx & 7, gvh = gv >> 4, gvw = gv & 15;
    const bool gcok = co0 + 4 * gq < a.Cout;
    const unsigned gpm = gcok ? (1u << gvh) | (1u << (2 + gvw)) : 0xffffffffu;
    unsigned grel[NTAPS];
#pragma unroll
    for (int tp = 0; tp < NTAPS; ++tp) {
        const int utw = tp & 1, uth = (tp >> 1) & 1, utd = tp >> 2;
        grel[tp] = (unsigned)((((utd * a.Ho + 2 * gvh + uth) * a.Wo + 2 * gvw + utw) * a.dy_ldc + co0 + 4 * gq) * 4);
    }

    f32x16 acc[TPW][2];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;

    auto range_mask = [](int lo, int n, int size) {          // bits z in [0, n) with lo + z < size
        const int last = size - lo < n ? size - lo : n;
        return last > 0 ? (1u << last) - 1u : 0u;
    };
    for (int b = b0; b < b1; ++b) {
        int Lt = b;
        const int tw_ = Lt % tilesW; Lt /= tilesW; const int th_ = Lt % tilesH; Lt /= tilesH; const int d = Lt % a.D; const int nb = Lt / a.D;
        const int h0 = th_ * 2, w0 = tw_ * 16;
        const unsigned xmask = range_mask(h0, 2, a.H) | (range_mask(w0, 16, a.W) << 2);
        const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + ((size_t)nb * a.D + d) * slice_x, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy) + ((size_t)nb * a.Do + a.sd * d) * slice_g, 0, 0x7fffffff, 0x00020000);
        const unsigned xbase = (unsigned)(((h0 * a.W + w0) * a.x_ldc) * 4);
        __syncthreads();                                  // every wave is done with the previous brick
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const bool ok = (xmask & xpm[it]) == xpm[it];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_u)(xs + (it * 4 + wave) * 256), 16, ok ? xrel[it] + xbase : OOB, 0, 0, 0);
        }
#pragma unroll
        for (int tp = 0; tp < NTAPS; ++tp) {
            const int utw = tp & 1, uth = (tp >> 1) & 1, utd = tp >> 2;
            // output rows 2(h0 + vh) + uth < Ho  <=>  vh < (Ho - uth + 1)/2 - h0; likewise along w
            const unsigned gmask = range_mask(h0, 2, (a.Ho - uth + 1) >> 1) | (range_mask(w0, 16, (a.Wo - utw + 1) >> 1) << 2);
            const bool dok = a.sd * d + utd < a.Do;
            const unsigned gbase = (unsigned)(((2 * h0 * a.Wo + 2 * w0) * a.dy_ldc) * 4);
            const bool ok = dok && (gmask & gpm) == gpm;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(g_rs, (lds_ptr_u)(gs + (tp * 4 + wave) * 256), 16, ok ? grel[tp] + gbase : OOB, 0, 0, 0);
        }
        __syncthreads();                                  // (hipcc drains vmcnt in front of the barrier: the images have landed)
        // K loop: MFMA k-step t consumes voxels (2t, 2t+1): lane half hf takes voxel 2t + hf
#pragma unroll 4
        for (int t = 0; t < 16; ++t) {
            const int v = 2 * t + hf;
            const float a0 = xs[v * 64 + j], a1 = xs[v * 64 + 32 + j];
#pragma unroll
            for (int tp = 0; tp < TPW; ++tp) {
                const float bv = gs[((wave * TPW + tp) * 32 + v) * 32 + j];
                acc[tp][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[tp][0], 0, 0, 0);
                acc[tp][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[tp][1], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hf;
                a.part[(((size_t)split * NTAPS + wave * TPW + tp) * a.CiPad + ci0 + 32 * c + row) * a.CoPad + co0 + j] = acc[tp][c][r];
            }
}

__global__ __launch_bounds__(256, 3) void upconv_wgrad_kernel8(const WgradArgs a, int tilesH, int tilesW, int bps, int co_tiles, int ci_tiles) {
    upconv_wgrad_body<8>(a, tilesH, tilesW, bps, co_tiles, ci_tiles);
}
__global__ __launch_bounds__(256, 3) void upconv_wgrad_kernel4(const WgradArgs a, int tilesH, int tilesW, int bps, int co_tiles, int ci_tiles) {
    upconv_wgrad_body<4>(a, tilesH, tilesW, bps, co_tiles, ci_tiles);
}

}  // namespace

bool upconv_wgrad_ok(int Cin, int Cout, int sd) {
    static const bool enabled = getenv("E3_UPCONV_NO_GEMM") == nullptr;
    return enabled && (Cin & 63) == 0 && (Cout & 31) == 0 && (sd == 1 || sd == 2);
}

static int upconv_wgrad_bricks_per_split(int nbricks, int pairs) {
    int want = 768 / (pairs > 0 ? pairs : 1);              // three resident workgroups per CU
    if (want < 1) want = 1;
    return cdiv(nbricks, want);
}

int upconv_wgrad_splits(int N, int D, int H, int W, int Cin, int Cout) {
    const int nbricks = N * D * cdiv(H, 2) * cdiv(W, 16);
    return cdiv(nbricks, upconv_wgrad_bricks_per_split(nbricks, (Cin / 64) * (Cout / 32)));
}

int launch_upconv_wgrad(WgradArgs a, hipStream_t s) {
    E3_REQUIRE((a.x_ldc & 3) == 0 && (a.dy_ldc & 3) == 0 && ((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.dy & 15) == 0, E3_ERR_INVALID,
               "upconv wgrad views must be 16-byte aligned");
    E3_REQUIRE((size_t)a.H * a.W * a.x_ldc * 4 < 0x7fffffffu && (size_t)2 * a.Ho * a.Wo * a.dy_ldc * 4 < 0x7fffffffu,
               E3_ERR_UNSUPPORTED, "upconv wgrad: a d-slice beyond 2 GiB (32-bit buffer offsets)");
    const int tH = cdiv(a.H, 2), tW = cdiv(a.W, 16);
    const int nbricks = a.N * a.D * tH * tW;
    const int ci_tiles = a.Cin / 64, co_tiles = a.Cout / 32;
    const int bps = upconv_wgrad_bricks_per_split(nbricks, ci_tiles * co_tiles);
    const int splits = cdiv(nbricks, bps);
    E3_REQUIRE(splits == a.splits, E3_ERR_INVALID, "upconv wgrad: splits mismatch");
    const dim3 grid((unsigned)((size_t)splits * ci_tiles * co_tiles));
    const int ntaps = a.sd * 4;
    const size_t lds = (size_t)(32 * 64 + ntaps * 32 * 32) * 4;
    if (ntaps == 8) hipLaunchKernelGGL(upconv_wgrad_kernel8, grid, dim3(256), lds, s, a, tH, tW, bps, co_tiles, ci_tiles);
    else hipLaunchKernelGGL(upconv_wgrad_kernel4, grid, dim3(256), lds, s, a, tH, tW, bps, co_tiles, ci_tiles);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
