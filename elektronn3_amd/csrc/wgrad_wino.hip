// Weight gradient of the 3x3x3 convolution as Winograd F(3x3x3, 2x2x2) on the fp32 matrix cores.
//
//   dW[co][ci][k] = sum over 2x2x2 output tiles t:  sum_o dY[2t + o][co] * X[2t + o + k - 1][ci]        (k in {0,1,2}^3)
//
// Per dimension this is F(3, 2): 3 outputs (the taps), a 2-tap "filter" (the dY tile) and a 4-wide input (the X tile),
// 4 multiplies instead of 6; in 3D 64 instead of 216 per tile and (ci, co) pair, 3.375x fewer matrix FLOPs than the
// direct wgrad of wgrad_mfma.hip.  With the interpolation points (0, 1, -1, inf):
//   Xt = B^T x      B^T rows: x0 - x2,  x1 + x2,  x2 - x1,  x1 - x3           (the forward kernel's input transform)
//   Yt = G y        G rows:   y0,  y0 + y1,  y0 - y1,  y1                     (factors 1/2 moved to the output side)
//   M_p[co][ci] = sum_tiles Yt_p[tile][co] * Xt_p[tile][ci]                   (64 independent GEMMs, K = all tiles)
//   dW = A^T M      A^T rows: M0 + M1/2 + M2/2,  M1/2 - M2/2,  M1/2 + M2/2 - M3
//
// Work decomposition (same bricks, splits and partial-slab layout as wgrad_conv_kernel, so wgrad_reduce_kernel and all
// callers are unchanged): a workgroup owns a (32 co x 32 ci) tile and a contiguous range of 2x4x16-voxel bricks
// (16 tiles each).  Wave w owns the 16 positions with pd = w: 16 accumulators of v_mfma_f32_32x32x2_f32 (K = 2 tiles
// per instruction), resident over the whole range.  Lane (j, hf) transforms channel co0+j of dY and channel ci0+j of X
// for the tiles 4 hf + s it feeds to k-step s, straight from an LDS image stored [d][h][channel][w] (w contiguous:
// one ds_read_b64/b128 serves several overlapping tiles), so the transformed tiles never touch LDS.
// Epilogue: A^T over (ph, pw) in registers, over pd through LDS, partial slab part[split][tap][co][ci].
#include "kernels.h"

namespace {

constexpr int G_LH = 6, G_LW = 18;                       // X halo of a 2x4x16 brick: 4 x 6 x 18
constexpr int G_NV = 4 * G_LH * G_LW;                    // 432 voxels
constexpr int G_MV = 128;                                // dY brick voxels
constexpr int G_XS = G_NV * 32;                          // X image [voxel][32 ci] floats
constexpr int G_GS = G_MV * 32;                          // dY image [voxel][32 co]
constexpr int G_BUF = G_XS + G_GS;                       // one stage (70 KB)
constexpr int G_XW = G_NV * 8 / 64;                      // 54 wave-pieces (1 KB each) of X per brick
constexpr int G_XI = (G_XW + 3) / 4;                     // per wave: 14 (waves 0,1) / 13
constexpr int G_GI = G_MV * 8 / 64 / 4;                  // 4 wave-pieces of dY per wave
constexpr int G_EX = 4 * 9 * 4 * 64 * 4;                 // epilogue exchange: [pd][khkw][r/4][lane][4] floats (144 KB)
constexpr int G_LDS_FLOATS = 2 * G_BUF > G_EX ? 2 * G_BUF : G_EX;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// The SelectionDAG linearisation lets pure VALU code drift across __builtin_amdgcn_sched_barrier (only side-effecting nodes are chained); a
// volatile asm that "modifies" the values is chained with the barriers and pins producers before / consumers after it.  No instructions.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pin4(f32x2* v) { asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])); }

// One unit of work: the (32 co x 32 ci) tile (co0, ci0) of one layer over the bricks [brick0, brick1).  The one-layer kernel runs ONE segment per
// workgroup (its split), the cross-layer stream-K kernel (round 6) a short list of them.  The partial result goes to
// out[tap * tap_stride + row * row_stride + column]  (the layer's slab [split][tap][CoPad][CiPad], resp. a private [27][32][32] tile slab).
struct WSeg {
    const float* x; const float* dy; size_t dy_chunk;
    int x_ldc, dy_ldc, Cin, Cout, N, D, H, W, tilesD, tilesH, tilesW;
    int ci0, co0, brick0, brick1;
    float* out; int tap_stride, row_stride;
};

__device__ __forceinline__ void wgrad_wino_segment(const WSeg& a, float* const smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hf = lane >> 5;
    const int tilesD = a.tilesD, tilesH = a.tilesH, tilesW = a.tilesW;
    const int ci0 = a.ci0, co0 = a.co0;
    const int brick0 = a.brick0, brick1 = a.brick1;
    constexpr unsigned OOB = 0x80000000u;       // buffer offset beyond the descriptors: the DMA writes zeros

    // ---- staging by LDS-DMA (buffer_load_dwordx4 ... lds): a wave-instruction moves 64 x 16 B = 8 voxels x 32 channels
    // into 1 KB of the [voxel][32] image; no staging registers, no ds_write pass, zero padding by the range check.
    // one descriptor per sample (rebuilt per brick, scalar work): offsets stay below 2^31 for any batch size
    __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, 0x7fffffff, 0x00020000);
    const int gl = a.dy_chunk ? 8 : a.dy_ldc;
    const size_t samp_x = (size_t)a.D * a.H * a.W * a.x_ldc, samp_g = (size_t)a.D * a.H * a.W * gl;
    // lane constants: wave-piece (it*4 + wave), lane -> piece idx -> (voxel, 4-channel group).  Validity of a halo voxel is
    // separable, so a brick only needs one 28-bit scalar mask (4 d bits | 6 h bits | 18 w bits) and each piece the
    // constant pattern of its three bits: 4 VALU per piece and brick, no branches.
    unsigned xpm[G_XI], xrel[G_XI];
#pragma unroll
    for (int it = 0; it < G_XI; ++it) {
        const int wp = it * 4 + wave < G_XW ? it * 4 + wave : G_XW - 1;     // (waves 2, 3 repeat the last piece in their 14th slot)
        const int idx = wp * 64 + lane;
        const int v = idx >> 3, q = idx & 7;
        const int zw = v % G_LW, zh = (v / G_LW) % G_LH, zd = v / (G_LW * G_LH);
        const bool cok = ci0 + 4 * q < a.Cin;
        xpm[it] = cok ? (1u << zd) | (1u << (4 + zh)) | (1u << (10 + zw)) : 0xffffffffu;      // all-ones never matches
        xrel[it] = (unsigned)((((zd * a.H + zh) * a.W + zw) * a.x_ldc + ci0 + 4 * q) * 4);
    }
    unsigned gpm[G_GI], grel[G_GI];
#pragma unroll
    for (int it = 0; it < G_GI; ++it) {
        const int idx = (it * 4 + wave) * 64 + lane;
        const int v = idx >> 3, q = idx & 7;
        const int ww = v & 15, hh = (v >> 4) & 3, dd = v >> 6;
        const bool cok = co0 + 4 * q < a.Cout;
        gpm[it] = cok ? (1u << dd) | (1u << (4 + hh)) | (1u << (10 + ww)) : 0xffffffffu;
        // (channel-chunked dy, WgradArgs::dy_chunk: the piece's 4 channels are half of an 8-channel chunk plane's voxel row)
        grel[it] = a.dy_chunk ? (unsigned)(((size_t)((co0 + 4 * q) >> 3) * a.dy_chunk + (size_t)(((dd * a.H + hh) * a.W + ww) * 8 + 4 * (q & 1))) * 4)
                              : (unsigned)((((dd * a.H + hh) * a.W + ww) * a.dy_ldc + co0 + 4 * q) * 4);
    }

    // ---- read plan.  D pass of Winograd row pd = wave: X: 0: x0 - x2, 1: x1 + x2, 2: x2 - x1, 3: x1 - x3;  Y: y0, y0+y1, y0-y1, y1
    const int da = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int db = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sgn = wave == 1 ? 1.f : -1.f;
    const float ya = wave == 3 ? 0.f : 1.f, yb = wave == 0 ? 0.f : (wave == 2 ? -1.f : 1.f);
    float m1 = -1.f;
    asm volatile("" : "+s"(m1));                          // opaque -1 (keeps a + m1*b as one fma)
    const int xrd_a = (da * G_LH * G_LW + 8 * hf) * 32 + j;        // + ((2c + h) * LW + 4 hc + w) * 32
    const int xrd_b = (db * G_LH * G_LW + 8 * hf) * 32 + j;
    const int yrd = G_XS + (8 * hf) * 32 + j;                       // + ((dd * 4 + 2c + oh) * 16 + 4 hc + w) * 32

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // ---- per-brick scalars of the DMA requests (set-up of brick b + 1 runs during the last MFMA block of brick b - 1 ... see below)
    unsigned xmask = 0, gmask = 0;
    auto issue_setup = [&](int brick) {
        int Lt = brick < brick1 ? brick : brick1 - 1;     // (past the end the last brick harmlessly re-stages itself)
        const int tw_ = Lt % tilesW; Lt /= tilesW; const int th_ = Lt % tilesH; Lt /= tilesH; const int td_ = Lt % tilesD; const int nb = Lt / tilesD;
        const int d0 = td_ * 2, h0 = th_ * 4, w0 = tw_ * 16;
        // scalar validity masks of the brick: bit z of a field is set when halo coordinate z is inside the volume
        auto range_mask = [](int lo, int n, int size) {      // z in [0, n): lo + z in [0, size)
            const int first = lo < 0 ? -lo : 0, last = size - lo < n ? size - lo : n;     // valid z in [first, last)
            return last > first ? ((1u << last) - 1u) & ~((1u << first) - 1u) : 0u;
        };
        xmask = range_mask(d0 - 1, 4, a.D) | (range_mask(h0 - 1, 6, a.H) << 4) | (range_mask(w0 - 1, 18, a.W) << 10);
        gmask = range_mask(d0, 2, a.D) | (range_mask(h0, 4, a.H) << 4) | (range_mask(w0, 16, a.W) << 10);
        // the descriptors start at the brick's halo origin (possibly in front of the tensor: only lanes whose voxel is inside the volume
        // ever form an address from them), so a lane's offset is its constant position inside the halo -- no per-brick vector arithmetic
        const long long xorg = (long long)nb * (long long)samp_x + ((((long long)(d0 - 1) * a.H + h0 - 1) * a.W + w0 - 1) * a.x_ldc);
        const long long gorg = (long long)nb * (long long)samp_g + ((((long long)d0 * a.H + h0) * a.W + w0) * gl);
        x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + xorg, 0, 0x7fffffff, 0x00020000);
        g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy) + gorg, 0, 0x7fffffff, 0x00020000);
    };
    // DMA requests of one brick in two parts (issued between the MFMAs of the first two blocks of the brick before): vector offset = the
    // lane's constant offset inside the halo, or OOB for a voxel outside the volume / a channel beyond the tensor: 3 VALU per piece
    // (and-compare-select), none for the address.
    auto issue_part = [&](int part, float* buf) {         // part 0: X pieces 0-7, dY pieces 0-1; part 1: X pieces 8-13, dY pieces 2-3
        const int lo = part == 0 ? 0 : 8, hi = part == 0 ? 8 : G_XI;
        const int glo = part == 0 ? 0 : 2, ghi = part == 0 ? 2 : G_GI;
#pragma unroll
        for (int it = 0; it < G_XI; ++it) {
            if (it < lo || it >= hi) continue;
            const int wp = it * 4 + wave < G_XW ? it * 4 + wave : G_XW - 1;
            const bool ok = (xmask & xpm[it]) == xpm[it];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(buf + wp * 256), 16, ok ? xrel[it] : OOB, 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < G_GI; ++it) {
            if (it < glo || it >= ghi) continue;
            const bool ok = (gmask & gpm[it]) == gpm[it];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(g_rs, (lds_ptr_t)(buf + G_XS + (it * 4 + wave) * 256), 16, ok ? grel[it] : OOB, 0, 0, 0);
        }
    };

    // One brick = 2 tile rows (chunks) x 2 halves of 2 k-steps each = 4 blocks of (transform, 32 MFMAs).  Only the transforms (fp32 VALU:
    // they share the FMA lanes with the fp32 MFMA, nothing can hide them) stay between the MFMA blocks.  Everything else is issued IN
    // BETWEEN the MFMAs (sched_group_barrier pipelines: an MFMA holds the matrix pipe for 64 cycles, a DS / VMEM / SALU instruction issued
    // in its shadow is free; issued in front of the blocks, as in round 2, a brick's 140 LDS reads + 18 DMA requests + ~150 set-up scalars
    // left the pipe idle for ~2-3k of its 12.7k cycles):
    //   block 0, 1: the LDS-DMA requests of brick b + 1 (into the other stage) and the LDS reads of blocks 1, 2
    //   block 2:    the LDS reads of block 3
    //   block 3:    starts with the brick's barrier (DMA of b + 1 landed, every wave done reading stage b); its MFMAs cover the scalar
    //               set-up of brick b + 2 and the LDS reads of block 0 of brick b + 1 -- no read is exposed after the barrier.
    static_assert(G_NV * 128 < 65536 && 2 * G_MV * 128 < 65536, "LDS read offsets must fit the 16-bit immediate");
    // The transforms run PACKED (v_pk_add / v_pk_fma_f32: two lanes of work per VALU slot, ~7.9 vs 2 x 6.1 cycles).  The pairing that needs
    // no register moves and no duplicate loads: a block handles the lane's tiles (hc = 0, tl) and (hc = 1, tl) of a tile row -- w windows
    // 4 hc + 2 tl + [0, 4), disjoint -- as elements 0 / 1 of every pair; every pass is element-wise in the pair and the MFMA operands of
    // k-step s = 2 hc + tl are the halves of the result pairs.
    f32x2 ra[4][4], rb[4][4], ry0[2][2], ry1[2][2];      // [h][w], element hc: column 4 hc + 2 tl + w of the lane's 16-wide row
    // LDS read addresses: one lane base PER WINDOW COLUMN w and array, every read = base + immediate.  (hipcc merges LDS reads that share
    // a base register into ds_read2 forms in order of their offsets; with a base per column the two elements of a pair -- 4 voxels = 128
    // dwords apart -- are offset-neighbours and land in one ds_read2st64_b32 that writes the pair directly.  With a shared base it pairs
    // neighbouring columns instead and assembles the pairs with v_mov + s_waitcnt in the middle of the MFMAs.)
    struct RdBase { int a[4], b[4], y[2]; };
    auto make_bases = [&](int stage_off, RdBase& B) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            B.a[w] = stage_off + xrd_a + w * 32; B.b[w] = stage_off + xrd_b + w * 32;
            asm volatile("" : "+v"(B.a[w]), "+v"(B.b[w]));
        }
#pragma unroll
        for (int w = 0; w < 2; ++w) { B.y[w] = stage_off + yrd + w * 32; asm volatile("" : "+v"(B.y[w])); }
    };
    auto read_raw = [&](const RdBase& B, int c, int tl) {
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int off = ((2 * c + h) * G_LW + 4 * e + 2 * tl) * 32;
                    ra[h][w][e] = smem[B.a[w] + off]; rb[h][w][e] = smem[B.b[w] + off];
                }
#pragma unroll
        for (int oh = 0; oh < 2; ++oh)
#pragma unroll
            for (int w = 0; w < 2; ++w)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int off = ((2 * c + oh) * 16 + 4 * e + 2 * tl) * 32;
                    ry0[oh][w][e] = smem[B.y[w] + off]; ry1[oh][w][e] = smem[B.y[w] + off + 4 * 16 * 32];
                }
    };
    auto compute = [&](int cur_off, int nxt_off, int setup_brick) {
        RdBase Bc, Bn;
        make_bases(cur_off, Bc); make_bases(nxt_off, Bn);
        float* const nxt = smem + nxt_off;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {           // block qd = (tile row c = qd >> 1, tl = qd & 1)
            // (consumers of the block's reads stay behind the previous block)
#pragma unroll
            for (int h = 0; h < 4; ++h) { pin4(ra[h]); pin4(rb[h]); }
            pin4(&ry0[0][0]); pin4(&ry1[0][0]);
            // ---- X: rows h = 0..3 of the tile row, 4-wide w windows of the lane's tiles 4 hf + 2 hc + tl, hc = 0, 1
            f32x2 u[4][4];
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int w = 0; w < 4; ++w) u[h][w] = ra[h][w] + sgn * rb[h][w];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const f32x2 v0 = u[0][w] + m1 * u[2][w], v1 = u[1][w] + u[2][w], v2 = u[2][w] + m1 * u[1][w], v3 = u[1][w] + m1 * u[3][w];
                u[0][w] = v0; u[1][w] = v1; u[2][w] = v2; u[3][w] = v3;
            }
            f32x2 X[4][4];                          // [ph][pw], element hc
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                X[h][0] = u[h][0] + m1 * u[h][2]; X[h][1] = u[h][1] + u[h][2]; X[h][2] = u[h][2] + m1 * u[h][1]; X[h][3] = u[h][1] + m1 * u[h][3];
            }
            // ---- Y: dY rows oh = 0, 1 of the tile row, both d planes
            f32x2 g[2][2];
#pragma unroll
            for (int oh = 0; oh < 2; ++oh)
#pragma unroll
                for (int w = 0; w < 2; ++w) g[oh][w] = ya * ry0[oh][w] + yb * ry1[oh][w];
            f32x2 Y[4][4];
            {
                f32x2 hrow[4][2];
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    hrow[0][w] = g[0][w]; hrow[1][w] = g[0][w] + g[1][w]; hrow[2][w] = g[0][w] + m1 * g[1][w]; hrow[3][w] = g[1][w];
                }
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    Y[h][0] = hrow[h][0]; Y[h][1] = hrow[h][0] + hrow[h][1]; Y[h][2] = hrow[h][0] + m1 * hrow[h][1]; Y[h][3] = hrow[h][1];
                }
            }
            // (producers stay in front of the block)
#pragma unroll
            for (int h = 0; h < 4; ++h) { pin4(X[h]); pin4(Y[h]); }
            __builtin_amdgcn_sched_barrier(0);
            if (qd == 3) {
                __syncthreads();             // (hipcc drains vmcnt in front of the barrier: the next brick's DMA has landed)
                __builtin_amdgcn_sched_barrier(0);
                issue_setup(setup_brick);
                read_raw(Bn, 0, 0);
            } else {
                read_raw(Bc, (qd + 1) >> 1, (qd + 1) & 1);     // (reads first in program order: the scheduler keeps LDS reads and
                if (qd < 2) issue_part(qd, nxt);                        // LDS-DMA writes, which it cannot tell apart, in that order)
            }
            // ---- 16 positions x 2 k-steps (k-step s = 2 hc + tl: lane half hf supplies tile 4 hf + s)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int p = 0; p < 16; ++p)
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(Y[p >> 2][p & 3][e], X[p >> 2][p & 3][e], acc[p], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // one MFMA
                if (qd == 3) __builtin_amdgcn_sched_group_barrier(0x004, 6, 0);      // set-up scalars
                if (qd >= 2 || i < 20) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);   // two LDS reads (early: the next transform waits for them)
                else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);              // one LDS-DMA request (late: it has two blocks to land)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (brick0 < brick1) {
        issue_setup(brick0);
        issue_part(0, smem); issue_part(1, smem);
        issue_setup(brick0 + 1);
        __syncthreads();                     // (hipcc drains vmcnt before the barrier: the DMA has landed)
        { RdBase B0; make_bases(0, B0); read_raw(B0, 0, 0); }
        int par = 0;
        for (int b = brick0; b < brick1; ++b) {
            compute(par * G_BUF, (par ^ 1) * G_BUF, b + 2);
            par ^= 1;
        }
        __syncthreads();                     // the last block's reads of the other stage (unused values) are done: the buffers are free
    }

    // ---- epilogue: A^T over (ph, pw) in registers.  acc[ph*4+pw][r]: row r -> co = (r&3) + 8 (r>>2) + 4 hf, column j = ci.
    float hlf = 0.5f;
    asm volatile("" : "+s"(hlf));
    f32x16 o[3][3];       // [kh][kw]
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        const f32x16 s12 = hlf * (acc[ph * 4 + 1] + acc[ph * 4 + 2]), d12 = hlf * (acc[ph * 4 + 1] + m1 * acc[ph * 4 + 2]);
        const f32x16 c0 = acc[ph * 4 + 0] + s12, c1 = d12, c2 = s12 + m1 * acc[ph * 4 + 3];
        // fold row ph into the kh outputs: kh0 += [1, .5, .5, 0][ph] * c,  kh1 += [0, .5, -.5, 0][ph] * c,  kh2 += [0, .5, .5, -1][ph] * c
        if (ph == 0) { o[0][0] = c0; o[0][1] = c1; o[0][2] = c2; }
        else if (ph == 1) {
            o[0][0] += hlf * c0; o[0][1] += hlf * c1; o[0][2] += hlf * c2;
            o[1][0] = hlf * c0; o[1][1] = hlf * c1; o[1][2] = hlf * c2;
            o[2][0] = hlf * c0; o[2][1] = hlf * c1; o[2][2] = hlf * c2;
        } else if (ph == 2) {
            o[0][0] += hlf * c0; o[0][1] += hlf * c1; o[0][2] += hlf * c2;
            o[1][0] += m1 * (hlf * c0); o[1][1] += m1 * (hlf * c1); o[1][2] += m1 * (hlf * c2);
            o[2][0] += hlf * c0; o[2][1] += hlf * c1; o[2][2] += hlf * c2;
        } else { o[2][0] += m1 * c0; o[2][1] += m1 * c1; o[2][2] += m1 * c2; }
    }
    // (every wave passed the loop's final barrier: the staging buffers are free)
    float* ex = smem;     // [pd][khkw 9][r/4][lane][4]
#pragma unroll
    for (int kk = 0; kk < 9; ++kk)
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = o[kk / 3][kk % 3][4 * k4 + e];
            *reinterpret_cast<f32x4*>(ex + (((wave * 9 + kk) * 4 + k4) * 64 + lane) * 4) = v;
        }
    __syncthreads();
    // wave w sums the pd axis for the (kh, kw) pairs kk = w, w + 4, w + 8 and writes taps (kd, kh, kw), kd = 0..2
    for (int kk = wave; kk < 9; kk += 4) {
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            f32x4 m[4];
#pragma unroll
            for (int pd = 0; pd < 4; ++pd) m[pd] = *reinterpret_cast<const f32x4*>(ex + (((pd * 9 + kk) * 4 + k4) * 64 + lane) * 4);
            const f32x4 s12 = hlf * (m[1] + m[2]);
            f32x4 w3[3];
            w3[0] = m[0] + s12; w3[1] = hlf * (m[1] + m1 * m[2]); w3[2] = s12 + m1 * m[3];
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                const int tap = kd * 9 + kk;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * k4 + e;
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * hf;
                    a.out[(size_t)tap * a.tap_stride + row * a.row_stride + j] = w3[kd][e];
                }
            }
        }
    }
}

__global__ __launch_bounds__(256, 1) void wgrad_wino_kernel(const WgradArgs a, int tilesD, int tilesH, int tilesW,
                                                            int tiles_per_split, int co_tiles, int ci_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int ci_t = L % ci_tiles; L /= ci_tiles;
    const int co_t = L % co_tiles; const int split = L / co_tiles;
    const int nbricks = a.N * tilesD * tilesH * tilesW;
    WSeg g;
    g.x = a.x; g.dy = a.dy; g.dy_chunk = a.dy_chunk; g.x_ldc = a.x_ldc; g.dy_ldc = a.dy_ldc; g.Cin = a.Cin; g.Cout = a.Cout;
    g.N = a.N; g.D = a.D; g.H = a.H; g.W = a.W; g.tilesD = tilesD; g.tilesH = tilesH; g.tilesW = tilesW;
    g.ci0 = ci_t * 32; g.co0 = co_t * 32;
    g.brick0 = split * tiles_per_split;
    g.brick1 = g.brick0 + tiles_per_split < nbricks ? g.brick0 + tiles_per_split : nbricks;
    g.out = a.part + ((size_t)split * 27 * a.CoPad + g.co0) * a.CiPad + g.ci0; g.tap_stride = a.CoPad * a.CiPad; g.row_stride = a.CiPad;
    wgrad_wino_segment(g, smem);
}

// ---------------------------------------------------------------- cross-layer stream-K launch (round 6)
// The Winograd weight gradients of ALL layers of a backward pass in ONE launch: the (layer, tile pair, brick) units of work of the layers form one
// list in which a layer's tile pairs follow each other and a tile pair's bricks are consecutive; workgroup i (logical, XCD-blocked index) takes
// the units [start(i), start(i + 1)) of an equal partition and walks the segments -- maximal runs inside one (block, tile pair) cell, kernels.h WSkPart -- its
// range cuts out.  A segment's partial tile goes to the private slab  i + c  (c = the cell's index over all layers: both indices grow along the list, so no two
// segments share a slab); wgrad_sk_reduce_kernel adds a tile pair's slabs block by block, workgroups ascending.  Fixed partition, fixed
// order: run-to-run identical.  Against one launch per layer (each cut into 256 splits to fill the chip: 28 MB of slabs per layer, 368 MB per cfg-2
// step written and read back) this writes at most 256 + (number of tile pairs) tile slabs -- 49 MB for cfg 2 --, and the bottom levels, whose few
// bricks per workgroup could not amortise a launch's prologue and epilogue, ride along.
struct WSkLayer {      // what the kernel needs of a layer beside the partition (kernels.h: WSkPart)
    const float* x; const float* dy; unsigned long long dy_chunk;
    int x_ldc, dy_ldc, N, D, H, W, tilesD, tilesH, tilesW;
};
struct WSkArgs { WSkPart p; WSkLayer L[WSK_MAX_LAYERS]; };
constexpr int WSK_TILE = 27 * 32 * 32;
constexpr int WSK_CELLS_MAX = 2048;            // blocks of one layer (wgrad_sk_partition checks; a block is about one workgroup's share, so ~ workgroups / tile pairs)
__device__ __forceinline__ unsigned wsk_start(const WSkPart& a, unsigned i) { return i * a.q + (i < a.r ? i : a.r); }
__device__ __forceinline__ unsigned wsk_owner(const WSkPart& a, unsigned g) {      // the workgroup whose range holds unit g
    const unsigned cut = a.r * (a.q + 1);
    return g < cut ? g / (a.q + 1) : a.r + (g - cut) / a.q;
}

__global__ __launch_bounds__(256, 1) void wgrad_wino_sk_kernel(const WSkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    unsigned g = wsk_start(a.p, wg);
    const unsigned gend = wsk_start(a.p, wg + 1);
    int l = 0;
    while (g < gend) {
        while (l + 1 < a.p.n && a.p.L[l + 1].g0 <= g) ++l;
        const WSkPartLayer& Lp = a.p.L[l];
        const WSkLayer& Ly = a.L[l];
        const WSkUnit u = wsk_unit(Lp, g - Lp.g0);
        const unsigned tp = u.tp, b0 = u.brick, want = gend - g, nb = want < u.left ? want : u.left;
        WSeg s;
        s.x = Ly.x; s.dy = Ly.dy; s.dy_chunk = (size_t)Ly.dy_chunk; s.x_ldc = Ly.x_ldc; s.dy_ldc = Ly.dy_ldc; s.Cin = Lp.Cin; s.Cout = Lp.Cout;
        s.N = Ly.N; s.D = Ly.D; s.H = Ly.H; s.W = Ly.W; s.tilesD = Ly.tilesD; s.tilesH = Ly.tilesH; s.tilesW = Ly.tilesW;
        s.ci0 = (int)(tp % (unsigned)Lp.ci_tiles) * 32; s.co0 = (int)(tp / (unsigned)Lp.ci_tiles) * 32;
        s.brick0 = (int)b0; s.brick1 = (int)(b0 + nb);
        s.out = a.p.slab + (size_t)(wg + Lp.c0 + u.block * (unsigned)Lp.tps + tp) * WSK_TILE; s.tap_stride = 1024; s.row_stride = 32;
        wgrad_wino_segment(s, smem);
        __syncthreads();              // the segment's exchange buffer overlays the stage buffers of the next one
        g += nb;
    }
}

// dW (torch layout (Cout, Cin, 27)) of every layer from the tile slabs: one thread per (tile pair, tap, row, 4 columns), the slabs of the tile pair's
// workgroups in ascending order, fp64 accumulation like wgrad_reduce_kernel.  (Shared with the 16-bit path: launch_wgrad_sk_reduce.)
__global__ __launch_bounds__(1024) void wgrad_sk_reduce_kernel(const WSkPart a) {
    const unsigned t = blockIdx.x / 27u, tap = blockIdx.x % 27u;       // one workgroup per (tile pair, tap): 4 groups of 256 threads = 32 rows x 8 column quads each
    int l = 0;
    while (l + 1 < a.n && a.L[l + 1].t0 <= t) ++l;
    const WSkPartLayer& Ly = a.L[l];
    const unsigned tp = t - Ly.t0;
    const int ci0 = (int)(tp % (unsigned)Ly.ci_tiles) * 32, co0 = (int)(tp / (unsigned)Ly.ci_tiles) * 32;
    const int tid = threadIdx.x & 255, grp = threadIdx.x >> 8;
    const int row = tid >> 3, c4 = (tid & 7) * 4;
    // the tile pair's cells, block by block; a cell (at most one workgroup's share of units) was cut by at most ONE workgroup boundary: slabs w0 and, where the
    // cell straddles it, w0 + 1.  The cells' first slab and straddle flag are worked out once per workgroup and read from LDS.  Group g adds the blocks
    // g, g + 4, g + 8, ... (two blocks' loads in flight: a level-0 tile pair of the 16-bit path has 68 blocks -- one thread walking them all was a chain of ~140
    // dependent loads, 130 us for 130 MB), then the four partial sums meet in the order g = 0..3: fixed partition, fixed order.  Absent slabs contribute + 0.0.
    __shared__ unsigned cell[WSK_CELLS_MAX];       // (first slab index << 1) | straddles
    __shared__ double part[3][256][4];
    for (unsigned b = threadIdx.x; b < (unsigned)Ly.nblocks; b += blockDim.x) {
        const unsigned bsz = b + 1 == (unsigned)Ly.nblocks ? (unsigned)Ly.nbricks - b * (unsigned)Ly.B : (unsigned)Ly.B;
        const unsigned g0 = Ly.g0 + b * (unsigned)Ly.B * (unsigned)Ly.tps + tp * bsz, g1 = g0 + bsz;
        const unsigned w0 = wsk_owner(a, g0), w1 = wsk_owner(a, g1 - 1);
        cell[b] = ((w0 + Ly.c0 + b * (unsigned)Ly.tps + tp) << 1) | (w1 > w0 ? 1u : 0u);
    }
    __syncthreads();
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float* const base = a.slab + (size_t)tap * 1024 + row * 32 + c4;
    for (unsigned b = (unsigned)grp; b < (unsigned)Ly.nblocks; b += 8) {
        f32x4 v[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned bb = b + 4 * u;
            const bool on = bb < (unsigned)Ly.nblocks;
            const unsigned cw = cell[on ? bb : 0u];
            const float* p = base + (size_t)(cw >> 1) * WSK_TILE;
            v[u][0] = on ? *reinterpret_cast<const f32x4*>(p) : zero4;
            v[u][1] = (on && (cw & 1u)) ? *reinterpret_cast<const f32x4*>(p + WSK_TILE) : zero4;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += (double)v[u][h][e];
    }
    if (grp > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) part[grp - 1][tid][e] = acc[e];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += part[g][tid][e];
    const int co = co0 + row;
    if (co < Ly.Cout) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (ci0 + c4 + e < Ly.Cin) Ly.dw[((size_t)co * Ly.Cin + ci0 + c4 + e) * 27 + tap] = (float)acc[e];
    }
}

}  // namespace

bool wgrad_use_wino(ConvKind kind) {
    static const bool enabled = getenv("E3_WGRAD_NO_WINO") == nullptr;
    return enabled && kind == CONV_K3;
}

size_t wgrad_sk_slab_floats(int tile_pairs, int workgroups) { return (size_t)(3 * workgroups + tile_pairs) * WSK_TILE; }
size_t wgrad_wino_sk_slab_floats(int tile_pairs) { return wgrad_sk_slab_floats(tile_pairs, 256); }

// the partition of a stream-K launch: layer i has nbricks[i] bricks per tile pair
int wgrad_sk_partition(WSkPart& p, int n, const int* Cin, const int* Cout, const int* nbricks, float* const* dw, int workgroups, float* slab, size_t slab_floats) {
    E3_REQUIRE(n >= 1 && n <= WSK_MAX_LAYERS, E3_ERR_INVALID, "wgrad (stream-K): 1..16 layers per launch");
    p = WSkPart{};
    p.n = n; p.nwg = (unsigned)workgroups; p.slab = slab;
    unsigned g = 0, t = 0;
    for (int i = 0; i < n; ++i) {
        WSkPartLayer& L = p.L[i];
        L.dw = dw[i]; L.Cin = Cin[i]; L.Cout = Cout[i]; L.ci_tiles = cdiv(Cin[i], 32); L.nbricks = nbricks[i];
        L.tps = cdiv(Cout[i], 32) * L.ci_tiles;
        E3_REQUIRE(L.nbricks > 0 && (size_t)g + (size_t)L.tps * L.nbricks < (1u << 31), E3_ERR_INVALID, "wgrad (stream-K): work list out of range");
        L.g0 = g; L.t0 = t;
        g += (unsigned)L.tps * (unsigned)L.nbricks; t += (unsigned)L.tps;
    }
    p.total = g; p.ntp = t; p.q = g / p.nwg; p.r = g % p.nwg;
    unsigned c = 0;
    for (int i = 0; i < n; ++i) {      // blocks of about one workgroup's share
        WSkPartLayer& L = p.L[i];
        const int want = p.q > 0 ? (int)p.q : 1;
        L.B = L.nbricks < want ? L.nbricks : want;
        L.nblocks = cdiv(L.nbricks, L.B);
        E3_REQUIRE(L.nblocks <= WSK_CELLS_MAX, E3_ERR_UNSUPPORTED, "wgrad (stream-K): too many blocks in one layer");
        L.c0 = c;
        c += (unsigned)L.nblocks * (unsigned)L.tps;
    }
    p.ncells = c;      // (a cell holds at most max(q, 1) units <= a workgroup's share: it meets at most two workgroups -- what wgrad_sk_reduce_kernel relies on)
    E3_REQUIRE((size_t)(p.nwg + c) * WSK_TILE <= slab_floats, E3_ERR_WORKSPACE, "wgrad (stream-K): slab too small");
    return E3_OK;
}

int launch_wgrad_sk_reduce(const WSkPart& p, hipStream_t s) {
    hipLaunchKernelGGL(wgrad_sk_reduce_kernel, dim3(p.ntp * 27u), dim3(1024), 0, s, p);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_wgrad_wino_sk(const WgradSkLayer* layers, int n, float* slab, size_t slab_floats, hipStream_t s) {
    constexpr int lds_bytes = G_LDS_FLOATS * 4;
    static bool set = false;
    if (!set) { E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_wino_sk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); set = true; }
    for (int l0 = 0; l0 < n; l0 += WSK_MAX_LAYERS) {      // (more layers than the argument block holds: several launches, each its own partition of the same slab)
        WSkArgs a{};
        const int m = n - l0 < WSK_MAX_LAYERS ? n - l0 : WSK_MAX_LAYERS;
        int Cin[WSK_MAX_LAYERS], Cout[WSK_MAX_LAYERS], nbr[WSK_MAX_LAYERS]; float* dw[WSK_MAX_LAYERS];
        for (int i = 0; i < m; ++i) {
            const WgradSkLayer& q = layers[l0 + i];
            E3_REQUIRE(q.Cin % 4 == 0 && q.Cout % 4 == 0 && q.x_ldc % 4 == 0 && q.dy_ldc % 4 == 0, E3_ERR_UNSUPPORTED, "wgrad needs channel counts that are multiples of 4");
            E3_REQUIRE((size_t)q.D * q.H * q.W * (size_t)(q.x_ldc > q.dy_ldc ? q.x_ldc : q.dy_ldc) * 4 < 0x7fffffffu, E3_ERR_UNSUPPORTED,
                       "Winograd wgrad: one sample beyond 2 GiB (32-bit buffer offsets); set E3_WGRAD_NO_WINO=1");
            E3_REQUIRE(!q.dy_chunk || (q.dy_chunk == (size_t)q.N * q.D * q.H * q.W * 8 && chunked_layout_ok((size_t)q.N * q.D * q.H * q.W, q.Cout)), E3_ERR_INVALID,
                       "Winograd wgrad: bad channel-chunked dy");
            WSkLayer& L = a.L[i];
            L.x = q.x; L.dy = q.dy; L.dy_chunk = q.dy_chunk; L.x_ldc = q.x_ldc; L.dy_ldc = q.dy_ldc;
            L.N = q.N; L.D = q.D; L.H = q.H; L.W = q.W; L.tilesD = cdiv(q.D, 2); L.tilesH = cdiv(q.H, 4); L.tilesW = cdiv(q.W, 16);
            Cin[i] = q.Cin; Cout[i] = q.Cout; nbr[i] = q.N * L.tilesD * L.tilesH * L.tilesW; dw[i] = q.dw;
        }
        const int rc = wgrad_sk_partition(a.p, m, Cin, Cout, nbr, dw, 256, slab, slab_floats);
        if (rc) return rc;
        hipLaunchKernelGGL(wgrad_wino_sk_kernel, dim3(256), dim3(256), lds_bytes, s, a);
        E3_CHECK_HIP(hipGetLastError());
        const int rr = launch_wgrad_sk_reduce(a.p, s);
        if (rr) return rr;
    }
    return E3_OK;
}

int launch_wgrad_wino(WgradArgs a, int tD, int tH, int tW, int tps, int co_tiles, int ci_tiles, int splits, hipStream_t s) {
    E3_REQUIRE((size_t)a.D * a.H * a.W * (size_t)(a.x_ldc > a.dy_ldc ? a.x_ldc : a.dy_ldc) * 4 < 0x7fffffffu, E3_ERR_UNSUPPORTED,
               "Winograd wgrad: one sample beyond 2 GiB (32-bit buffer offsets); set E3_WGRAD_NO_WINO=1");
    E3_REQUIRE(!a.dy_chunk || (a.dy_chunk == (size_t)a.N * a.D * a.H * a.W * 8 && chunked_layout_ok((size_t)a.N * a.D * a.H * a.W, a.Cout)), E3_ERR_INVALID,
               "Winograd wgrad: bad channel-chunked dy");
    constexpr int lds_bytes = G_LDS_FLOATS * 4;
    static bool set = false;
    if (!set) { E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); set = true; }
    const dim3 grid((unsigned)((size_t)splits * co_tiles * ci_tiles));
    hipLaunchKernelGGL(wgrad_wino_kernel, grid, dim3(256), lds_bytes, s, a, tD, tH, tW, tps, co_tiles, ci_tiles);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
