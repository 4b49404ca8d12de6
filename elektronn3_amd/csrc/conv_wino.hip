// 3x3x3 stride-1 convolution as Winograd F(2x2x2, 3x3x3) on the fp32 matrix cores (forward AND dgrad: the packer
// hands dgrad over as a forward conv with flipped taps and swapped channel roles).
//
// Replaces torch.nn.Conv3d(k=3, padding=1) inside elektronn3's conv3 blocks (unet.py:131-149) wherever the grid fills
// the chip.  64 multiplies per 2x2x2 output tile and (ci, co) pair instead of 216: 3.375x fewer matrix FLOPs than the
// direct implicit GEMM of conv_v3.hip, with the same fp32 arithmetic (F(2,3) only has 0, +-1, +-1/2 coefficients; the
// measured error against an fp64 convolution is the same 2-3e-7 rel-L2 as the direct fp32 kernel).
//
//   Y = A^T [ sum_ci (G g G^T)(ci,co)  (.)  (B^T d B)(ci) ] A          in each of the three dimensions
//
// Work decomposition (one workgroup = 4 waves, one wave per SIMD, 512 registers per lane):
//   * a brick of 2x2x8 tiles (4x4x16 output voxels, 6x6x18 input halo) x 32 output channels;
//   * the 64 Winograd positions (pd, ph, pw) are 64 independent GEMMs  M = 32 tiles, N = 32 channels, K = Cin.
//     Wave w owns the 16 positions with pd = w: 16 accumulator tiles of v_mfma_f32_32x32x2_f32 = 256 registers.
//   * K is walked in chunks of 8 input channels.  Per chunk the raw halo (648 voxels x 8 channels) is staged in LDS
//     (double buffered, one barrier per chunk).  Lane (tile i, half hf) reads the 2 d-planes its pd needs of ITS tile
//     and ITS 4 channels (32 ds_read_b128, parity-split + XOR-swizzled layout = conflict-free), does the B^T d B
//     transform in registers (the lane that computes a transformed value is the lane that feeds it to the MFMA: the
//     transformed tile never touches LDS) and issues 64 MFMAs.  The transformed weights U of the wave's 16 positions
//     come straight from L2 into registers one chunk ahead (they are private to the wave, LDS would buy nothing).
//   * epilogue: A^T m A over (ph, pw) in registers, over pd through LDS (each wave then owns one (oh, ow) of every
//     tile), then bias / folded eval-BN + ReLU / per-brick Welford statistics / store as in the direct kernels.
#include <type_traits>
#include "kernels.h"
#include "brick_order.h"

namespace {

constexpr int W_LD = 6, W_LH = 6, W_LW = 18;               // halo of a 4x4x16 brick
constexpr int W_NVOX = W_LD * W_LH * W_LW;                  // 648
constexpr int W_AI = W_LD;                                  // one thread stages one (zh, zw, 16-B half) column: 6 d-planes
constexpr int W_CLASS = 32;                                 // slots per (zh, zw) parity class of a plane (3 x 9 = 27 used)
constexpr int W_PLANE = 4 * W_CLASS * 8;                    // floats of one D-transformed plane (1024)
constexpr int W_BUF = 8 * W_PLANE;                          // 8 planes (td, pd) per buffer: 32 KB
constexpr int W_EX = 4 * 16 * 64 * 4;                       // epilogue exchange [pd][e4][lane][4] floats (64 KB)
constexpr int W_LDS_FLOATS = (2 * W_BUF > W_EX + 4 * 32 * 3 ? 2 * W_BUF : W_EX + 4 * 32 * 3);   // 64 KB + statistics scratch

// The LDS image holds the halo ALREADY TRANSFORMED ALONG D: plane (td, pd) = row pd of B^T applied to the 4 d-planes of
// tile depth td (the staging thread of a (zh, zw) column has all 6 d-values in registers, so this costs 8 packed ops per
// column instead of a D pass in every wave, and halves the LDS reads of the transform).  Float offset of (zh, zw), 16-B
// piece q inside a plane: parity classes keep the stride-2 tile origins contiguous, the piece is XOR-ed with bit 0 of
// zh/2 so that each ds_read_b128 lane group covers all 64 banks.
__device__ __forceinline__ int plane_slot(int zh, int zw, int q) {
    const int slot = ((zh & 1) * 2 + (zw & 1)) * W_CLASS + (zh >> 1) * 9 + (zw >> 1);
    return slot * 8 + 4 * (q ^ ((zh >> 1) & 1));
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

template <bool PRO>
__global__ __launch_bounds__(256, 1) void conv3_wino_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hf = lane >> 5;

    // block index -> (sample, brick, column tile).  The tile counts are powers of two for the usual crop sizes: shifts
    // instead of four scalar divisions (~200 SALU instructions of an otherwise latency-bound prologue).
    unsigned bid = blockIdx.x, nbricks = gridDim.x, split = 0;
    if (a.splitk > 1) {          // split-K: grid = splits x bricks, every split walks its own share of the input channels
        nbricks = gridDim.x / (unsigned)a.splitk;
        split = bid / nbricks;
        bid -= split * nbricks;
    }
    unsigned L = xcd_remap(bid, nbricks);
    auto divmod = [](unsigned& x, int d) {
        int r;
        if ((d & (d - 1)) == 0) { r = (int)(x & (unsigned)(d - 1)); x >>= __builtin_ctz((unsigned)d); }
        else { r = (int)(x % (unsigned)d); x /= (unsigned)d; }
        return r;
    };
    const int ntile = divmod(L, a.ntiles);
    const int tw_ = divmod(L, a.tilesW);
    const int th_ = divmod(L, a.tilesH);
    const int td_ = divmod(L, a.tilesD); const int nb = (int)L;
    const int d0 = (td_ + a.o_td) * 4 + a.org_d, h0 = (th_ + a.o_th) * 4 + a.org_h, w0 = (tw_ + a.o_tw) * 16 + a.org_w;      // (o_*, org_*: first brick / voxel origin of the needed region)
    const int n0 = ntile * 32;
    const int mtile = ((nb * a.tilesD + td_) * a.tilesH + th_) * a.tilesW + tw_;
    const int NCH = a.Cin >> 3;
    constexpr unsigned OOB = 0x80000000u;       // buffer offset beyond every descriptor below: loads return 0, stores are dropped

    // ---- buffer descriptors.  The activation descriptors start at the brick's first d-plane so that every offset is a
    // small non-negative number whatever the size of the tensor; an out-of-volume voxel gets the OOB offset and the
    // hardware supplies the zero padding (no select, no 64-bit address arithmetic in the loop).
    const int dlo = d0 > 0 ? d0 - 1 : 0;
    const size_t plane_x = (size_t)a.H * a.W * a.x_ldc, plane_y = (size_t)a.H * a.W * a.y_ldc;
    const size_t xrem = (size_t)(a.D - dlo) * plane_x * 4, yrem = (size_t)(a.D - d0) * plane_y * 4;
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x) + ((size_t)nb * a.D + dlo) * plane_x + split * (unsigned)a.sk_x, 0, (int)(xrem < 0x7fffffffu ? xrem : 0x7fffffffu), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(
        a.y + ((size_t)nb * a.D + d0) * plane_y + split * a.sk_y, 0, (int)(yrem < 0x7fffffffu ? yrem : 0x7fffffffu), 0x00020000);
    // transformed weights: U[ntile][chunk][pos 64][hf 2][co 32][4 ci]; wave = pd owns positions 16 pd .. 16 pd + 15
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.wt) + ((size_t)ntile * NCH * 64 + wave * 16) * 256 + (size_t)split * a.sk_w, 0, NCH * 64 * 1024, 0x00020000);
    const int b_voff = lane * 16;

    // ---- staging plan: thread -> column (zh, zw, 16-B half q) of the halo, all 6 d-planes (threads 216..255 idle)
    const bool col_on = tid < W_LH * W_LW * 2;
    const int cq = tid & 1, czw = (tid >> 1) % W_LW, czh = (tid >> 1) / W_LW;
    const int cgh = h0 + czh - 1, cgw = w0 + czw - 1;
    const bool col_ok = col_on && cgh >= 0 && cgh < a.H && cgw >= 0 && cgw < a.W;
    unsigned a_src[W_AI];
    unsigned a_ok = 0;
#pragma unroll
    for (int zd = 0; zd < W_AI; ++zd) {
        const int gd = d0 + zd - 1;
        const bool ok = col_ok && gd >= 0 && gd < a.D;
        a_src[zd] = ok ? (unsigned)(((((gd - dlo) * a.H + cgh) * a.W + cgw) * a.x_ldc + 4 * cq) * 4) : OOB;
        a_ok |= (ok ? 1u : 0u) << zd;
    }
    const int a_dst = col_on ? plane_slot(czh, czw, cq) : 0;

    // ---- read plan of lane (tile i = j, half hf): tile (td, th, tw) = (j >> 4, (j >> 3) & 1, j & 7), plane (td, pd = wave)
    float m1 = -1.f;
    asm volatile("" : "+s"(m1));     // opaque -1: a + m1*b becomes v_pk_fma_f32 (hipcc only packs fadd/ffma, never fsub)
    const int ttd = j >> 4, tth = (j >> 3) & 1, ttw = j & 7;
    const int lbase = (ttd * 4 + wave) * W_PLANE + (tth * 9 + ttw) * 8;
    // rows h = 0,1 of the tile have zh/2 = th, rows 2,3 have th + 1: the swizzle bit differs between the two
    int rdA[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) rdA[hh] = lbase + 4 * (hf ^ ((tth + hh) & 1));

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    const bool dbg = (E3_DBG_FLAGS(a.flags) & 1024) != 0;     // timing experiments: s_memtime stamps into the statistics buffer
    long long* dbgp = reinterpret_cast<long long*>(a.stats) + (size_t)blockIdx.x * 16;
    int dbgi = 0;
    auto stamp = [&]() { if (dbg && tid == 0 && dbgi < 14) dbgp[dbgi++] = (long long)__builtin_amdgcn_s_memtime(); };
    if (dbg && tid == 0) dbgp[14] = (long long)__builtin_amdgcn_s_memrealtime();
    stamp();
    f32x4 xr[W_AI], Bv[16];
    auto issue_raw = [&](int cb) {
#pragma unroll
        for (int it = 0; it < W_AI; ++it)
            xr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, a_src[it], cb * 4, 0));
    };
    auto write_raw = [&](float* buf, int cb) {
        if (PRO) {                             // BN + ReLU of the producer applied while staging; the padding stays 0
            const f32x4 psc = *reinterpret_cast<const f32x4*>(a.pro_scale + cb + 4 * cq);
            const f32x4 psh = *reinterpret_cast<const f32x4*>(a.pro_shift + cb + 4 * cq);
#pragma unroll
            for (int zd = 0; zd < W_AI; ++zd) {
                const bool ok = (a_ok >> zd) & 1u;
#pragma unroll
                for (int e = 0; e < 4; ++e) xr[zd][e] = ok ? fmaxf(__builtin_fmaf(xr[zd][e], psc[e], psh[e]), 0.f) : 0.f;
            }
        }
        if (col_on) {
            // D pass of B^T for the two tile depths: rows  x0 - x2,  x1 + x2,  x2 - x1,  x1 - x3  of planes (0..3) and (2..5)
#pragma unroll
            for (int td = 0; td < 2; ++td) {
                const f32x4 x0 = xr[2 * td], x1 = xr[2 * td + 1], x2 = xr[2 * td + 2], x3 = xr[2 * td + 3];
                *reinterpret_cast<f32x4*>(buf + (td * 4 + 0) * W_PLANE + a_dst) = x0 + m1 * x2;
                *reinterpret_cast<f32x4*>(buf + (td * 4 + 1) * W_PLANE + a_dst) = x1 + x2;
                *reinterpret_cast<f32x4*>(buf + (td * 4 + 2) * W_PLANE + a_dst) = x2 + m1 * x1;
                *reinterpret_cast<f32x4*>(buf + (td * 4 + 3) * W_PLANE + a_dst) = x1 + m1 * x3;
            }
        }
    };
    auto load_B = [&](int c, int g) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            Bv[g * 4 + p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_voff + p * 1024, (c * 64 + g * 4) * 1024, 0));
    };

    // one 8-channel chunk: raw halo in `cur`, next chunk's halo goes to `nxt`
    auto chunk = [&](int c, const float* cur, float* nxt) {
        const int cn = c + 1 < NCH ? c + 1 : c;     // (the last chunk harmlessly re-stages itself: no branch in the loop body)
        if (c == 1) stamp();
        issue_raw(cn * 8);
        // ---- H and W passes of B^T d B on this lane's tile of plane (td, pd), 4 channels at a time (f32x4 = the 4 k-steps)
        f32x4 t[4][4];
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int imm = (((h & 1) * 2 + (w & 1)) * W_CLASS + (h >> 1) * 9 + (w >> 1)) * 8;
                t[h][w] = *reinterpret_cast<const f32x4*>(cur + rdA[h >> 1] + imm);
            }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const f32x4 u0 = t[0][w] + m1 * t[2][w], u1 = t[1][w] + t[2][w], u2 = t[2][w] + m1 * t[1][w], u3 = t[1][w] + m1 * t[3][w];
            t[0][w] = u0; t[1][w] = u1; t[2][w] = u2; t[3][w] = u3;
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const f32x4 u0 = t[h][0] + m1 * t[h][2], u1 = t[h][1] + t[h][2], u2 = t[h][2] + m1 * t[h][1], u3 = t[h][1] + m1 * t[h][3];
            t[h][0] = u0; t[h][1] = u1; t[h][2] = u2; t[h][3] = u3;
        }
        // keep the transform packed (2 floats per VALU lane-op): without an opaque use hipcc scalarises every vector op
        // whose results are only ever extracted element-wise (the MFMA operands below)
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                f32x2 lo = {t[h][w][0], t[h][w][1]}, hi = {t[h][w][2], t[h][w][3]};
                asm("" : "+v"(lo)); asm("" : "+v"(hi));
                t[h][w][0] = lo[0]; t[h][w][1] = lo[1]; t[h][w][2] = hi[0]; t[h][w][3] = hi[1];
            }
        __builtin_amdgcn_sched_barrier(0);
        if (c == 1) stamp();
        // ---- 16 positions x 4 k-steps; groups of 4 positions keep 4 independent accumulators in flight
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    acc[g * 4 + p] = __builtin_amdgcn_mfma_f32_32x32x2f32(t[g][p][s], Bv[g * 4 + p][s], acc[g * 4 + p], 0, 0, 0);
            load_B(cn, g);                           // the group's registers are free again: fetch them for the next chunk
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c == 1) stamp();
        write_raw(nxt, cn * 8);
        __builtin_amdgcn_sched_barrier(0);
        if (c == 1) stamp();
        __syncthreads();
        if (c == 1) stamp();
    };

    float* buf0 = smem;
    float* buf1 = smem + W_BUF;
    issue_raw(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) load_B(0, g);
    write_raw(buf0, 0);
    __syncthreads();
    stamp();
    chunk(0, buf0, buf1);                   // peeled: the accumulators are known zeros here (MFMA with a literal 0 addend)
    for (int c = 1; c < NCH; c += 2) {
        chunk(c, buf1, buf0);
        if (c + 1 < NCH) chunk(c + 1, buf0, buf1);
    }
    stamp();

    // ---- epilogue.  acc[ph*4+pw][r]: position (pd = wave, ph, pw), tile row r -> tile t = (r&3) + 8 (r>>2) + 4 hf, channel j.
    f32x16 q[2][2];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {       // incremental: 4 accumulators -> 2 temporaries -> folded into q, low register pressure
        const f32x16 t0 = acc[ph * 4 + 0] + acc[ph * 4 + 1] + acc[ph * 4 + 2];
        const f32x16 t1 = acc[ph * 4 + 1] + m1 * acc[ph * 4 + 2] + m1 * acc[ph * 4 + 3];
        if (ph == 0) { q[0][0] = t0; q[0][1] = t1; }
        else if (ph == 1) { q[0][0] += t0; q[0][1] += t1; q[1][0] = t0; q[1][1] = t1; }
        else if (ph == 2) { q[0][0] += t0; q[0][1] += t1; q[1][0] += m1 * t0; q[1][1] += m1 * t1; }
        else { q[1][0] += m1 * t0; q[1][1] += m1 * t1; }
    }
    // (the barrier that ended the last chunk already separates the raw buffers from their reuse below)
    float* ex = smem;
#pragma unroll
    for (int oh = 0; oh < 2; ++oh)
#pragma unroll
        for (int ow = 0; ow < 2; ++ow)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = q[oh][ow][4 * k + e];
                *reinterpret_cast<f32x4*>(ex + ((wave * 16 + (oh * 2 + ow) * 4 + k) * 64 + lane) * 4) = v;
            }
    __syncthreads();
    stamp();
    // wave w now owns output offset (oh, ow) = (w >> 1, w & 1) of every tile and sums the pd axis: od = 0, 1
    const int oh = wave >> 1, ow = wave & 1;
    const int n = n0 + j;
    const bool nvalid = n < a.Ncols;
    const bool aff = a.epi_scale != nullptr;
    const float bias = (a.bias && nvalid) ? a.bias[n] : 0.f;
    float es = 1.f, eh = 0.f;
    if (aff && nvalid) { es = a.epi_scale[n]; eh = a.epi_shift[n]; }
    f32x4 y[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x4 m[4];
#pragma unroll
        for (int pd = 0; pd < 4; ++pd) m[pd] = *reinterpret_cast<const f32x4*>(ex + ((pd * 16 + wave * 4 + k) * 64 + lane) * 4);
        y[0][k] = m[0] + m[1] + m[2] + bias;
        y[1][k] = m[1] + m1 * m[2] + m1 * m[3] + bias;
    }
    if (aff) {
#pragma unroll
        for (int od = 0; od < 2; ++od)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) y[od][k][e] = fmaxf(__builtin_fmaf(y[od][k][e], es, eh), 0.f);
    }
    // value (od, r = 4k + e): tile t = (r&3) + 8 (r>>2) + 4 hf -> (td, th, tw) = (r >> 3, (r >> 2) & 1, (r & 3) + 4 hf),
    // voxel (d0 + 2 td + od, h0 + 2 th + oh, w0 + 2 tw + ow).  Lane part of the address in a VGPR, the rest is scalar.
    const int gw_l = w0 + 8 * hf + ow, gh_l = h0 + oh;
    const unsigned y_voff = (unsigned)(((gh_l * a.W + gw_l) * a.y_ldc + n) * 4);
    const bool full = d0 + 4 <= a.D && h0 + 4 <= a.H && w0 + 16 <= a.W && n0 + 32 <= a.Ncols;
    const bool do_stats = a.stats != nullptr && !dbg;
    float cnt = 0.f, sum = 0.f;
    unsigned okmask = 0xffffffffu;
    if (full) {
#pragma unroll
        for (int od = 0; od < 2; ++od)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int soff = ((((2 * (r >> 3) + od) * a.H + 2 * ((r >> 2) & 1)) * a.W + 2 * (r & 3)) * a.y_ldc) * 4;
                const float v = y[od][r >> 2][r & 3];   // (bit_cast of a vector-element lvalue reads element 0 with this hipcc)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rs, y_voff, soff, 0);
                sum += v;
            }
        cnt = 32.f;
    } else {
        okmask = 0u;
#pragma unroll
        for (int od = 0; od < 2; ++od)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gd = d0 + 2 * (r >> 3) + od, gh = gh_l + 2 * ((r >> 2) & 1), gw = gw_l + 2 * (r & 3);
                const bool ok = nvalid && gd < a.D && gh < a.H && gw < a.W;
                const int soff = ((((2 * (r >> 3) + od) * a.H + 2 * ((r >> 2) & 1)) * a.W + 2 * (r & 3)) * a.y_ldc) * 4;
                const float v = y[od][r >> 2][r & 3];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rs, ok ? y_voff : OOB, soff, 0);
                cnt += ok ? 1.f : 0.f;
                sum += ok ? v : 0.f;
                okmask |= (ok ? 1u : 0u) << (od * 16 + r);
            }
    }
    if (dbg) { asm volatile("" :: "v"(sum)); stamp(); if (tid == 0) dbgp[15] = (long long)__builtin_amdgcn_s_memrealtime(); }
    if (do_stats) {
        float mean = cnt > 0.f ? sum / cnt : 0.f, m2 = 0.f;
#pragma unroll
        for (int od = 0; od < 2; ++od)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = y[od][r >> 2][r & 3] - mean;
                m2 += ((okmask >> (od * 16 + r)) & 1u) ? d * d : 0.f;
            }
        const float cnt2 = __shfl_xor(cnt, 32), mean2 = __shfl_xor(mean, 32), m22 = __shfl_xor(m2, 32);
        welford_merge(cnt, mean, m2, cnt2, mean2, m22);
        float* scr = smem + W_EX;
        if (hf == 0) {
            float* sc = scr + (wave * 32 + j) * 3;
            sc[0] = cnt; sc[1] = mean; sc[2] = m2;
        }
        __syncthreads();
        if (tid < 32 && n < a.Ncols) {
            float c0 = 0.f, me = 0.f, mm = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float* sc = scr + (w * 32 + tid) * 3;
                welford_merge(c0, me, mm, sc[0], sc[1], sc[2]);
            }
            float* o = a.stats + ((size_t)mtile * a.Cout + n) * 3;
            o[0] = c0; o[1] = me; o[2] = mm;
        }
    }
}

// ---- persistent variant: one workgroup per CU walks the bricks L0, L0 + step, ... (the XCD-aware order of the plain kernel: XCD x owns a
// contiguous eighth of the brick range).  What differs from conv3_wino_kernel:
//   * the staging is software-pipelined over the whole (brick, chunk) sequence, two units ahead: during the MFMAs of unit u the wave
//     stores the D-transformed halo of unit u + 1 to LDS (raw data loaded during unit u - 1, transformed in the VALU phase of unit u)
//     and requests the raw halo of unit u + 2 -- of the next brick when the current one is done, so a brick starts with its data in LDS;
//   * fp32 VALU work shares the FMA lanes with the fp32 MFMA and is kept in one phase per unit (LDS reads, D transform of the staged
//     halo, H / W passes); EVERYTHING else -- LDS stores, global loads, the scalar bookkeeping of the staging cursor (mixed-radix brick
//     counters advanced without divisions or branches, descriptor and validity masks of the brick being staged) -- is issued between
//     the MFMAs (sched_group_barrier pipeline), where an instruction that is not VALU costs nothing;
//   * BatchNorm statistics: every lane keeps a running (count, mean, M2) of its channel over the workgroup's bricks (merged with
//     v_rcp instead of a division) and the workgroup writes ONE record at the end (`wgstats`; 256 / ntiles records per layer instead of
//     one per brick: no pre-merge launch, a tenth of the epilogue's statistic code per brick).  Where a workgroup's bricks do not all
//     belong to one column tile (unusual grids) the records stay per brick.
constexpr int W_PLDS_FLOATS = 2 * W_BUF + W_EX + 4 * 32 * 3 + 6 * 256;     // + the threads' running statistics and parked lane constants
constexpr int W_POOLX = 4 * 64 * 16;                                         // fused max-pool: [wave (oh, ow)][lane][16 channels] (16 KB)

struct WinoPArgs { BrickStep b; int wgstats; int org_d, org_h, org_w; };   // b: the logical brick order and the digits of the step gridDim / 8 between a workgroup's bricks (brick_order.h); org_*: voxel origin of the brick range's first brick

// AFF: the folded scale / shift + ReLU epilogue (inference; no statistics) -- a compile-time split: as a run-time branch its merge cost ~50 register
// moves per brick in both forms
// TR (launches without statistics: data gradients, the folded-epilogue inference form): TRANSPOSED accumulators -- the weights are the A operand, so a lane
// holds one tile (column) and, per register quad, 4 consecutive output channels (rows (r & 3) + 8 (r >> 2) + 4 hf): the brick leaves as 8 16-byte stores
// per lane instead of 32 dword stores.  Same operands, same arithmetic, same exchange; only the result orientation differs.
// POOL (AFF && TR, inference): the 2x2x2 ceil-mode max-pool of the output in the epilogue (ConvArgs::pool_out)
// HEAD (AFF && TR, inference, 32 output channels): the 1x1x1 head (+ softmax) on the activations in registers instead of storing them (ConvArgs::head_*)
template <bool AFF, bool TR = false, bool POOL = false, bool HEAD = false>
__global__ __launch_bounds__(256, 1) void conv3_wino_pkernel(const ConvArgs a, const unsigned nblk, const WinoPArgs pa) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hf = lane >> 5;
    const int NCH = a.Cin >> 3;
    constexpr unsigned OOB = 0x80000000u;
    // arguments that only the epilogue needs are re-read from the kernarg segment there (scalar loads) instead of occupying ~25 SGPRs across
    // the whole persistent loop (which made hipcc spill ~200 scalars into vector lanes, ~1k cycles of v_readlane / v_writelane per brick)
    typedef const __attribute__((address_space(4))) ConvArgs* KArgs;
    auto KA = []() -> KArgs { KArgs q = (KArgs)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(q)); return q; };
    const int D = a.D, H = a.H, W = a.W, xl = a.x_ldc;
    const int tilesD = a.tilesD, tilesH = a.tilesH, tilesW = a.tilesW, ntiles = a.ntiles;
    const unsigned plane_xb = (unsigned)((size_t)H * W * xl * 4);

    // ---- lane constants
    const bool col_on = tid < W_LH * W_LW * 2;
    const int cq = tid & 1, czw = (tid >> 1) % W_LW, czh = (tid >> 1) / W_LW;
    // (the 40 threads without a column store zeros into the 5 unused slots of the 4 parity classes: no divergent branch around the stores)
    int a_dst = col_on ? plane_slot(czh, czw, cq) : (((tid - W_LH * W_LW * 2) / 10) * W_CLASS + 27 + ((tid - W_LH * W_LW * 2) % 10) / 2) * 8 + 4 * (tid & 1);
    // staging: byte offset of the thread's halo column inside a d-plane relative to the brick's halo origin, and the two bits of the
    // brick's validity mask (6 d bits | 6 h bits | 18 w bits) it needs (all-ones never matches: threads without a column load zeros)
#ifdef E3_WINO_ABL_CONTIG      // developer builds, timing only (wrong data): a halo row's 32-byte pieces as one contiguous run (what channel-chunked planes would give)
    const unsigned col_rel = (unsigned)(((czh * W + 1) * xl + czw * 8 + 4 * cq) * 4);
#else
    const unsigned col_rel = (unsigned)(((czh * W + czw) * xl + 4 * cq) * 4);
#endif
    const unsigned col_bits = col_on ? (1u << (6 + czh)) | (1u << (12 + czw)) : 0xffffffffu;
    float m1 = -1.f;
    asm volatile("" : "+s"(m1));
    const int ttd = j >> 4, tth = (j >> 3) & 1, ttw = j & 7;
    const int lbase = (ttd * 4 + wave) * W_PLANE + (tth * 9 + ttw) * 8;
    int rdA[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) rdA[hh] = lbase + 4 * (hf ^ ((tth + hh) & 1));
    const int b_voff = lane * 16;

    // ---- brick cursors: mixed-radix digits (column tile, block-local w / h / d, block w / h / d, sample: brick_order.h) of the logical brick index L
    unsigned gdim = gridDim.x;
    asm volatile("" : "+s"(gdim));          // (kept in a register: a conditional use of gridDim.x becomes a branch around its load)
    struct Cur { int nt, tw, th, td, nb; unsigned bid; };      // bid: physical index blockIdx + k * gridDim (the brick exists while bid < nblk)
    auto advance = [&](Cur& c, bool go) {        // c += step if go (no branch, no division; tw, th, td count from the brick range's first brick)
        brick_advance(pa.b, go, ntiles, tilesW, tilesH, tilesD, c.nt, c.tw, c.th, c.td, c.nb);
        c.bid += go ? gdim : 0u;
    };
    auto range_mask = [](int lo, int n, int size) {      // bit z set: lo + z in [0, size), z in [0, n)
        const int first = lo < 0 ? -lo : 0, last = size - lo < n ? size - lo : n;
        return last > first ? ((1u << last) - 1u) & ~((1u << first) - 1u) : 0u;
    };
    // descriptor of the halo of brick c (origin = voxel (d0 - 1, h0 - 1, w0 - 1), possibly in front of the tensor: only valid lanes
    // form addresses from it) and its validity mask; a brick beyond the end of the range stages zeros
    __amdgpu_buffer_rsrc_t s_rs;
    unsigned s_mask_ = 0;
    auto make_stage = [&](const Cur& c) {
        const int d0 = c.td * 4 + pa.org_d, h0 = c.th * 4 + pa.org_h, w0 = c.tw * 16 + pa.org_w;
        const long long org = ((long long)c.nb * D + (d0 - 1)) * ((long long)H * W * xl) + ((long long)(h0 - 1) * W + (w0 - 1)) * xl;
        s_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + org, 0, 0x7fffffff, 0x00020000);
        const unsigned m = range_mask(d0 - 1, 6, D) | (range_mask(h0 - 1, 6, H) << 6) | (range_mask(w0 - 1, 18, W) << 12);
        s_mask_ = c.bid < nblk ? m : 0u;
    };

    f32x16 acc[16];
    f32x4 xr[W_AI], Bv[16], pD[8];
    // raw halo column of the staging cursor's brick, channels [cb, cb + 8): 6 d-planes (plane offset in the scalar offset, which the
    // range check ignores; an invalid plane or column is pushed out of range through the vector offset)
    auto issue_raw = [&](int cb, bool off = false) {      // off: nothing is read (mask 0: every lane out of range), the registers get zeros
        const unsigned s_mask = off ? 0u : s_mask_;
        const bool ok = (s_mask & col_bits) == col_bits;
        const unsigned voff = ok ? col_rel : OOB;
#pragma unroll
        for (int it = 0; it < W_AI; ++it) {
            const unsigned dsel = ((s_mask >> it) & 1u) ? 0u : OOB;
            xr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s_rs, voff | dsel, (int)(it * plane_xb) + cb * 4, 0));
        }
    };
    auto dtransform = [&]() {       // D pass of B^T for the two tile depths: rows x0 - x2, x1 + x2, x2 - x1, x1 - x3 of planes (0..3) and (2..5)
#pragma unroll
        for (int td = 0; td < 2; ++td) {
            const f32x4 x0 = xr[2 * td], x1 = xr[2 * td + 1], x2 = xr[2 * td + 2], x3 = xr[2 * td + 3];
            pD[td * 4 + 0] = x0 + m1 * x2; pD[td * 4 + 1] = x1 + x2; pD[td * 4 + 2] = x2 + m1 * x1; pD[td * 4 + 3] = x1 + m1 * x3;
        }
    };
    auto write_staged = [&](float* buf) {
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<f32x4*>(buf + q * W_PLANE + a_dst) = pD[q];
    };
    const float* b_base = nullptr;
    auto make_brs = [&](int ntile) { b_base = KA()->wt + ((size_t)ntile * NCH * 64 + wave * 16) * 256; };
    auto load_B = [&](int c, int g, bool off = false) {   // off: a descriptor of size 0 -- nothing is read, the registers get zeros
        const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b_base), 0, off ? 0 : NCH * 64 * 1024, 0x00020000);
#pragma unroll
        for (int p = 0; p < 4; ++p)
            Bv[g * 4 + p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_voff + p * 1024, (c * 64 + g * 4) * 1024, 0));
    };

    // ---- cursors: P = brick being computed, S = brick being staged (chunk sc of it is the next one to be requested)
    Cur P;
    {
        P.bid = blockIdx.x;
        brick_decode(xcd_remap(blockIdx.x, nblk), pa.b, ntiles, tilesW, tilesH, tilesD, P.nt, P.tw, P.th, P.td, P.nb);
    }
    Cur S = P;
    int sc = 0;
    auto stage_wrap = [&]() { sc = 0; advance(S, true); make_stage(S); };      // the cursor moves on to the workgroup's next brick
    auto stage_step = [&]() {                   // request the raw halo of staging unit (S, sc) and move the cursor to the next unit
        issue_raw(sc * 8);
        if (sc + 1 == NCH) stage_wrap(); else ++sc;
    };

    float* cur = smem;
    float* nxt = smem + W_BUF;
    float* ex = smem + 2 * W_BUF;
    float* scr = ex + W_EX;
    // running statistics of this lane's channel over the workgroup's bricks: [3][256] floats in LDS, touched by their own thread only (in
    // registers they were spilled around the main loop, and a scratch reload waits for every global load and store in flight)
    { float* const run = scr + 4 * 32 * 3 + tid; run[0] = 0.f; run[256] = 0.f; run[512] = 0.f; }
    // lane constants of the main loop that do not fit in registers across the output transform: parked in LDS and read back behind it (the
    // register allocator's own choice -- scratch -- makes the main loop wait for the output stores in front of the reload)
    int* const park = reinterpret_cast<int*>(scr + 4 * 32 * 3 + 3 * 256) + tid;
    park[0] = a_dst; park[256] = rdA[0]; park[512] = rdA[1];
    if (HEAD) {          // the head's weights [class][32] and biases [4] behind the parked constants (published by the prologue's barrier)
        float* const hw = scr + 4 * 32 * 3 + 6 * 256;
        const KArgs hk = KA();
        if (tid < 128) hw[tid] = tid < hk->head_cout * 32 ? hk->head_w[tid] : 0.f;
        else if (tid < 132) hw[tid] = (hk->head_b && tid - 128 < hk->head_cout) ? hk->head_b[tid - 128] : 0.f;
    }
#ifndef E3_WINO_ST_AUX
#define E3_WINO_ST_AUX 0      // cache-policy bits of the output stores (developer builds: 2 = non-temporal)
#endif
#ifndef E3_WINO_ABL
#define E3_WINO_ABL 0       // developer builds: bit mask of pieces left out of the MFMA phase (timing experiments, wrong results)
#endif
#ifdef E3_WINO_TIMING      // developer build (tools/phase_timing_pwino.py): s_memtime stamps of the workgroup's second brick instead of statistics
    long long* const tstamp = reinterpret_cast<long long*>(KA()->stats) + (size_t)blockIdx.x * 32;
    int tbrick = 0;
#define TSTAMP(i) do { if (tid == 0 && tbrick == 1) tstamp[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define TSTAMPC(i) do { if (tid == 0 && tbrick == 1) { const long long tm_ = (long long)__builtin_amdgcn_s_memtime(); if (tchunk == 1) tstamp[i] = tm_; if (tchunk == NCH - 1) tstamp[16 + i] = tm_; if (tchunk == NCH - 2) tstamp[24 + i] = tm_; } } while (0)
    if (tid == 0) tstamp[14] = (long long)__builtin_amdgcn_s_memrealtime();
#else
#define TSTAMP(i)
#define TSTAMPC(i)
#endif

    // ---- prologue: unit 0 staged, unit 1 requested, weights of chunk 0 requested
    make_stage(S);
    stage_step();
    dtransform();
    write_staged(cur);
    stage_step();
    make_brs(P.nt);
#pragma unroll
    for (int g = 0; g < 4; ++g) load_B(0, g);
    __syncthreads();

    // One 8-channel chunk (unit u) of brick P: transformed halo in `cur`.  ZERO (compile time) = first chunk of a brick: the accumulators
    // start from a literal zero.  `last` (run time, no branch) = last chunk of the brick: its requests read nothing (zero-sized descriptor
    // / empty mask) and the staging cursor stays -- the raw halo of unit u + 2 and the first weights of the next brick are requested in the
    // epilogue instead, after its register-hungry part (88 registers less to keep alive across the output transform).  `cB`: chunk whose
    // weights are requested.
    f32x4 tn[4][4];            // the next chunk's tile, read from LDS while the current chunk's last MFMAs run
    auto rd_imm = [](int h, int w) { return (((h & 1) * 2 + (w & 1)) * W_CLASS + (h >> 1) * 9 + (w >> 1)) * 8; };
    auto chunk = [&](auto zero_tag, bool last, int cB, int tchunk = 0) {
        constexpr bool ZERO = decltype(zero_tag)::value;
        TSTAMPC(1);
        // ---- VALU phase: H and W passes of B^T d B on this lane's tile of plane (td, pd), 4 channels at a time; D pass of the staged halo.
        // The tile's 16 LDS reads were issued during the last MFMAs of the previous chunk (`tn`, below); only a brick's first chunk reads here
        // (its predecessor's reads would have to stay alive across the epilogue).
        f32x4 t[4][4];
        if (ZERO) {
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int w = 0; w < 4; ++w) t[h][w] = *reinterpret_cast<const f32x4*>(cur + rdA[h >> 1] + rd_imm(h, w));
            __builtin_amdgcn_sched_barrier(0);      // (the LDS reads are issued before anything waits for the staged raw halo)
        } else {
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int w = 0; w < 4; ++w) t[h][w] = tn[h][w];
        }
        dtransform();
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const f32x4 u0 = t[0][w] + m1 * t[2][w], u1 = t[1][w] + t[2][w], u2 = t[2][w] + m1 * t[1][w], u3 = t[1][w] + m1 * t[3][w];
            t[0][w] = u0; t[1][w] = u1; t[2][w] = u2; t[3][w] = u3;
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const f32x4 u0 = t[h][0] + m1 * t[h][2], u1 = t[h][1] + t[h][2], u2 = t[h][2] + m1 * t[h][1], u3 = t[h][1] + m1 * t[h][3];
            t[h][0] = u0; t[h][1] = u1; t[h][2] = u2; t[h][3] = u3;
        }
        // pin the packed results in front of the MFMA block (and keep hipcc from scalarising vector ops whose results are only used element-wise)
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                f32x2 lo = {t[h][w][0], t[h][w][1]}, hi = {t[h][w][2], t[h][w][3]};
                asm volatile("" : "+v"(lo), "+v"(hi));
                t[h][w][0] = lo[0]; t[h][w][1] = lo[1]; t[h][w][2] = hi[0]; t[h][w][3] = hi[1];
            }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            f32x2 lo = {pD[q][0], pD[q][1]}, hi = {pD[q][2], pD[q][3]};
            asm volatile("" : "+v"(lo), "+v"(hi));
            pD[q][0] = lo[0]; pD[q][1] = lo[1]; pD[q][2] = hi[0]; pD[q][3] = hi[1];
        }
        __builtin_amdgcn_sched_barrier(0);
        TSTAMPC(2);
        // ---- MFMA phase: 16 quads of 4 MFMAs (4 positions of a row ph = g at one k-step: 4 independent accumulators in flight; a scheduling
        // fence after each quad keeps that order -- left alone, the scheduler issues the 4 k-steps of one accumulator back to back).  Each
        // quad carries its share of the rest: one LDS store of unit u + 1, one raw load of unit u + 2, one weight load of unit u + 1 (into a
        // register whose MFMAs are done), a piece of the staging cursor's bookkeeping (no branches: advanced by 0 or 1 unit / brick).
        asm volatile("" : "+s"(sc), "+s"(S.nt), "+s"(S.tw), "+s"(S.th), "+s"(S.td), "+s"(S.nb), "+s"(S.bid));    // (the scalar chain starts HERE, not in front of the VALU phase)
        const unsigned rmask = last ? 0u : s_mask_;      // (last chunk: empty mask, nothing is read)
        const bool rok = (rmask & col_bits) == col_bits;
        const unsigned rvoff = rok ? col_rel : OOB;
        const int rcb = sc * 8;
        const bool wrap = !last && sc + 1 == NCH;
        const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b_base), 0, last ? 0 : NCH * 64 * 1024, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rs = s_rs;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int g = q >> 2, ks = q & 3;
            if (q == 12) {
                // The chunk's barrier sits in front of the last 16 MFMAs: unit u + 1 is complete in `nxt` (its stores were issued with the
                // first MFMAs) and every wave is done with `cur`.  The last quads then carry the next chunk's LDS reads -- rows 0..2 of the
                // tile go into registers whose MFMAs are issued, row 3 follows the last quad -- so the next VALU phase starts with its operands.
                __syncthreads();
                TSTAMPC(4);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (q >= 12 && q < 15) {
#pragma unroll
                for (int w = 0; w < 4; ++w) tn[q - 12][w] = *reinterpret_cast<const f32x4*>(nxt + rdA[(q - 12) >> 1] + rd_imm(q - 12, w));
            }
            if (q < 8 && !(E3_WINO_ABL & 1)) *reinterpret_cast<f32x4*>(nxt + q * W_PLANE + a_dst) = pD[q];
            if (q >= 1 && q < 7 && !(E3_WINO_ABL & 2)) {
                const int it = q - 1;
                const unsigned dsel = ((rmask >> it) & 1u) ? 0u : OOB;
                xr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rs, rvoff | dsel, (int)(it * plane_xb) + rcb * 4, 0));
            }
            if (q >= 4 && !(E3_WINO_ABL & 4)) {      // weights of group (q - 4) >> 2, whose 16 MFMAs are issued
                const int gb = (q - 4) >> 2, pb = (q - 4) & 3;
                Bv[gb * 4 + pb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_voff + pb * 1024, (cB * 64 + gb * 4) * 1024, 0));
            }
            if (q == 8 && !(E3_WINO_ABL & 8)) { sc = last ? sc : (wrap ? 0 : sc + 1); advance(S, wrap); }
            if (q == 10 && !(E3_WINO_ABL & 8)) make_stage(S);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (ZERO && ks == 0) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    acc[g * 4 + p] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32(Bv[g * 4 + p][ks], t[g][p][ks], z, 0, 0, 0)
                                        : __builtin_amdgcn_mfma_f32_32x32x2f32(t[g][p][ks], Bv[g * 4 + p][ks], z, 0, 0, 0);
                } else {
                    acc[g * 4 + p] = TR ? __builtin_amdgcn_mfma_f32_32x32x2f32(Bv[g * 4 + p][ks], t[g][p][ks], acc[g * 4 + p], 0, 0, 0)
                                        : __builtin_amdgcn_mfma_f32_32x32x2f32(t[g][p][ks], Bv[g * 4 + p][ks], acc[g * 4 + p], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x004, 12, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x004, 12, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x004, 12, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x004, 12, 0);
            if (!(E3_WINO_ABL & 16)) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
            if (!(E3_WINO_ABL & 4)) Bv[12 + pb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_voff + pb * 1024, (cB * 64 + 12) * 1024, 0));
#pragma unroll
        for (int w = 0; w < 4; ++w) tn[3][w] = *reinterpret_cast<const f32x4*>(nxt + rdA[1] + rd_imm(3, w));
        TSTAMPC(3);
        { float* tsw = cur; cur = nxt; nxt = tsw; }
    };

    for (;;) {
        // (the staging cursor's chunk index at chunk c is (c + 2) mod NCH: it wraps in chunk NCH - 3, or -- NCH <= 2 -- in the epilogue)
        TSTAMP(0);
        chunk(std::true_type{}, NCH == 1, NCH == 1 ? 0 : 1);
        for (int c = 1; c < NCH; ++c) chunk(std::false_type{}, c + 1 == NCH, c + 1 == NCH ? 0 : c + 1, c);
        TSTAMP(5);

        // ---- epilogue.  acc[ph*4+pw][r]: position (pd = wave, ph, pw), tile row r -> tile t = (r&3) + 8 (r>>2) + 4 hf, channel j.
        // (the lane index is taken afresh: lane constants of the epilogue kept across the main loop were spilled, and a scratch reload in here
        // waits for every global request in flight -- the next brick's halo and weights, the output stores)
        int elane;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(elane));
        const int ej = elane & 31, ehf = elane >> 5;
        const int d0 = P.td * 4 + pa.org_d, h0 = P.th * 4 + pa.org_h, w0 = P.tw * 16 + pa.org_w, n0 = P.nt * 32;
        const KArgs e = KA();
        const int yl = e->y_ldc, eN = e->Ncols;
        const size_t plane_y = (size_t)H * W * yl;
        const size_t yrem = (size_t)(D - d0) * plane_y * 4;
        const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(
            e->y + ((size_t)P.nb * D + d0) * plane_y, 0, (int)(yrem < 0x7fffffffu ? yrem : 0x7fffffffu), 0x00020000);
        // per-channel constants of the epilogue, requested first: they arrive during the output transform, and waiting for them does not wait
        // for the requests of the next brick behind them (channels beyond Ncols: out of the descriptor's range, 0)
        const int n = n0 + ej;
        const bool nvalid = n < eN;
        constexpr bool aff = AFF;
        float bias, es, eh;
        f32x4 tbias[4], tes[4], teh[4];      // TR: the lane's 16 channels n0 + 8 k + 4 hf .. + 3
        if (TR) {
            const __amdgpu_buffer_rsrc_t c_rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e->bias), 0, e->bias ? eN * 4 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t c_rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e->epi_scale), 0, aff ? eN * 4 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t c_rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e->epi_shift), 0, aff ? eN * 4 : 0, 0x00020000);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int co = (n0 + 8 * k + 4 * ehf) * 4;
                tbias[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c_rs0, co, 0, 0));
                if (aff) {
                    tes[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c_rs1, co, 0, 0));
                    teh[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c_rs2, co, 0, 0));
                }
            }
            bias = es = eh = 0.f;
        } else
        {
            const __amdgpu_buffer_rsrc_t c_rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e->bias), 0, e->bias ? eN * 4 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t c_rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e->epi_scale), 0, aff ? eN * 4 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t c_rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e->epi_shift), 0, aff ? eN * 4 : 0, 0x00020000);
            bias = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c_rs0, n * 4, 0, 0));
            es = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c_rs1, n * 4, 0, 0));
            eh = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c_rs2, n * 4, 0, 0));
        }
        // request the raw halo of unit u + 2 and the first weights of the next brick
        Cur Pn = P;
        advance(Pn, true);
        const bool has_next = Pn.bid < nblk;
        if (!(E3_WINO_ABL & 32)) stage_step();
        make_brs(Pn.nt);
#pragma unroll
        for (int g = 0; g < 4; ++g) if (!(E3_WINO_ABL & 128)) load_B(0, g);
        // A^T m A over (ph, pw) in registers, in two halves of 8 accumulator rows (the next brick's weights and raw halo are live in
        // registers across the epilogue: the full-width form needed 80 more than there are)
#pragma unroll
        for (int hv = 0; hv < 2; ++hv) {
            f32x8 q[2][2];
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                f32x8 c[4];
#pragma unroll
                for (int pw = 0; pw < 4; ++pw)
#pragma unroll
                    for (int r = 0; r < 8; ++r) c[pw][r] = acc[ph * 4 + pw][hv * 8 + r];
                const f32x8 t0 = c[0] + c[1] + c[2];
                const f32x8 t1 = c[1] + m1 * c[2] + m1 * c[3];
                if (ph == 0) { q[0][0] = t0; q[0][1] = t1; }
                else if (ph == 1) { q[0][0] += t0; q[0][1] += t1; q[1][0] = t0; q[1][1] = t1; }
                else if (ph == 2) { q[0][0] += t0; q[0][1] += t1; q[1][0] += m1 * t0; q[1][1] += m1 * t1; }
                else { q[1][0] += m1 * t0; q[1][1] += m1 * t1; }
            }
#pragma unroll
            for (int oh = 0; oh < 2; ++oh)
#pragma unroll
                for (int ow = 0; ow < 2; ++ow)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = q[oh][ow][4 * k + e];
                        *reinterpret_cast<f32x4*>(ex + ((wave * 16 + (oh * 2 + ow) * 4 + hv * 2 + k) * 64 + elane) * 4) = v;
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        TSTAMP(6);
        // (per-channel constants of the epilogue: requested in front of the barrier, their latency is covered by it)
        __syncthreads();
        TSTAMP(7);
        const int oh = wave >> 1, ow = wave & 1;
        f32x4 y[2][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f32x4 m[4];
#pragma unroll
            for (int pd = 0; pd < 4; ++pd) m[pd] = *reinterpret_cast<const f32x4*>(ex + ((pd * 16 + wave * 4 + k) * 64 + elane) * 4);
            if (TR) {
                y[0][k] = m[0] + m[1] + m[2] + tbias[k];
                y[1][k] = m[1] + m1 * m[2] + m1 * m[3] + tbias[k];
            } else {
                y[0][k] = m[0] + m[1] + m[2] + bias;
                y[1][k] = m[1] + m1 * m[2] + m1 * m[3] + bias;
            }
        }
        if (aff) {
#pragma unroll
            for (int od = 0; od < 2; ++od)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[od][k][e] = TR ? fmaxf(__builtin_fmaf(y[od][k][e], tes[k][e], teh[k][e]), 0.f) : fmaxf(__builtin_fmaf(y[od][k][e], es, eh), 0.f);
        }
        if (TR) {
            // value (od, k, e): tile = lane column ej -> (td, th, tw) = (ej >> 4, (ej >> 3) & 1, ej & 7), channel n0 + 8 k + 4 hf + e:
            // voxel (d0 + 2 td + od, h0 + 2 th + oh, w0 + 2 tw + ow), 16 bytes per (od, k)
            const int ttd_ = ej >> 4, tth_ = (ej >> 3) & 1, ttw_ = ej & 7;
            const int gd = d0 + 2 * ttd_, gh = h0 + 2 * tth_ + oh, gw = w0 + 2 * ttw_ + ow;
            const unsigned t_voff = (unsigned)((((2 * ttd_ * H + gh) * W + gw) * yl + n0 + 4 * ehf) * 4);
            const bool vok = gh < H && gw < W;
            const bool ok0 = vok && gd < D, ok1 = vok && gd + 1 < D;
            const int od_off = (int)(plane_y * 4);
            if (!HEAD) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool cok = n0 + 8 * k + 4 * ehf < eN;
                if (!(E3_WINO_ABL & 64)) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y[0][k]), y_rs, (ok0 && cok) ? t_voff + 32 * k : OOB, 0, E3_WINO_ST_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y[1][k]), y_rs, (ok1 && cok) ? t_voff + 32 * k : OOB, od_off, E3_WINO_ST_AUX);
                }
            }
            }
            if (HEAD) {
                // conv_final_fwd_kernel's arithmetic on the registers: channel quad q = 2 k + hf of the voxel is summed as an fmaf chain from 0, the eight
                // quad sums meet as ((q0 + q1) + (q2 + q3)) + ((q4 + q5) + (q6 + q7)) -- this lane holds the quads of its half, the partner lane (^ 32) the others
                const float* const hw = scr + 4 * 32 * 3 + 6 * 256;
                const int hc = e->head_cout;
                float lg[2][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float p[2][4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(hw + c * 32 + 8 * k + 4 * ehf);
#pragma unroll
                        for (int od = 0; od < 2; ++od) {
                            float sacc = 0.f;
#pragma unroll
                            for (int e4 = 0; e4 < 4; ++e4) sacc = __builtin_fmaf(y[od][k][e4], wv[e4], sacc);
                            p[od][k] = sacc + __shfl_xor(sacc, 32);
                        }
                    }
#pragma unroll
                    for (int od = 0; od < 2; ++od) lg[od][c] = ((p[od][0] + p[od][1]) + (p[od][2] + p[od][3])) + hw[128 + c];
                }
                // this lane finishes the voxel od = hf
                float l4[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) l4[c] = ehf ? lg[1][c] : lg[0][c];
                if (e->head_softmax) {
                    float m = l4[0];
#pragma unroll
                    for (int c = 1; c < 4; ++c) m = c < hc ? fmaxf(m, l4[c]) : m;
                    float sm = 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) { l4[c] = c < hc ? __expf(l4[c] - m) : 0.f; sm += l4[c]; }
                    const float inv = 1.f / sm;
#pragma unroll
                    for (int c = 0; c < 4; ++c) l4[c] *= inv;
                }
                const int vd = gd + ehf;
                const bool inb = vok && vd < D && vd >= e->head_lo[0] && vd < e->head_hi[0] && gh >= e->head_lo[1] && gh < e->head_hi[1] && gw >= e->head_lo[2] && gw < e->head_hi[2];
                if (inb) {
                    float* const yo = e->head_y + (long long)P.nb * e->head_ys[0] + (long long)(vd - e->head_lo[0]) * e->head_ys[2] + (long long)(gh - e->head_lo[1]) * e->head_ys[3] + (gw - e->head_lo[2]);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < hc) yo[(long long)c * e->head_ys[1]] = l4[c];
                }
            }
            if (POOL) {
                // a Winograd tile = one pooling window: max over od in the lane, over (oh, ow) = the four waves through LDS (voxels outside the tensor
                // do not take part: ceil_mode; NaN propagates as in nn.MaxPool3d); wave w then stores channel quad k = w of every window
                float* const px = scr + 4 * 32 * 3 + 6 * 256;
                auto nmax = [](float a_, float b_) { return (b_ > a_ || b_ != b_) ? b_ : a_; };
                constexpr float NEG = -3.4028235e38f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f32x4 pm;
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) pm[e4] = nmax(ok0 ? y[0][k][e4] : NEG, ok1 ? y[1][k][e4] : NEG);
                    *reinterpret_cast<f32x4*>(px + ((wave * 64 + elane) * 16) + 4 * k) = pm;
                }
                __syncthreads();
                f32x4 best = *reinterpret_cast<const f32x4*>(px + ((0 * 64 + elane) * 16) + 4 * wave);
#pragma unroll
                for (int w2 = 1; w2 < 4; ++w2) {
                    const f32x4 o2 = *reinterpret_cast<const f32x4*>(px + ((w2 * 64 + elane) * 16) + 4 * wave);
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) best[e4] = nmax(best[e4], o2[e4]);
                }
                const int Dp = (D + 1) >> 1, Hp = (H + 1) >> 1, Wp = (W + 1) >> 1;
                const int pd_ = (d0 >> 1) + ttd_, ph_ = (h0 >> 1) + tth_, pw_ = (w0 >> 1) + ttw_;
                const bool pok = pd_ < Dp && ph_ < Hp && pw_ < Wp && n0 + 8 * wave + 4 * ehf < eN;
                if (pok) *reinterpret_cast<f32x4*>(e->pool_out + ((((size_t)P.nb * Dp + pd_) * Hp + ph_) * Wp + pw_) * eN + n0 + 8 * wave + 4 * ehf) = best;
            }
            TSTAMP(8);
            TSTAMP(9);
#ifdef E3_WINO_TIMING
            if (tid == 0 && tbrick == 1) tstamp[15] = (long long)__builtin_amdgcn_s_memrealtime();
            ++tbrick;
#endif
            if (!has_next) break;
            {
                const int* const pk = reinterpret_cast<const int*>(scr + 4 * 32 * 3 + 3 * 256) + wave * 64 + elane;
                a_dst = pk[0]; rdA[0] = pk[256]; rdA[1] = pk[512];
            }
            P = Pn;
            continue;
        }
        const int gw_l = w0 + 8 * ehf + ow, gh_l = h0 + oh;
        const unsigned y_voff = (unsigned)(((gh_l * W + gw_l) * yl + n) * 4);
        const bool full = d0 + 4 <= D && h0 + 4 <= H && w0 + 16 <= W && n0 + 32 <= eN;
#ifdef E3_WINO_TIMING
        const bool do_stats = false;
#else
        const bool do_stats = !AFF && e->stats != nullptr;
#endif
        float cnt, mean, m2;
        if (full) {                 // (uniform) no masks: packed sums
#pragma unroll
            for (int od = 0; od < 2; ++od)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int soff = ((((2 * (r >> 3) + od) * H + 2 * ((r >> 2) & 1)) * W + 2 * (r & 3)) * yl) * 4;
                    const float v = y[od][r >> 2][r & 3];
                    if (!(E3_WINO_ABL & 64)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rs, y_voff, soff, E3_WINO_ST_AUX);
                }
            cnt = 32.f; mean = 0.f; m2 = 0.f;
            if (do_stats) {
                const f32x4 s4 = (y[0][0] + y[0][1]) + (y[0][2] + y[0][3]) + ((y[1][0] + y[1][1]) + (y[1][2] + y[1][3]));
                mean = ((s4[0] + s4[1]) + (s4[2] + s4[3])) * (1.f / 32.f);
                f32x4 q4 = {0.f, 0.f, 0.f, 0.f};
                const float nm = m1 * mean;
#pragma unroll
                for (int od = 0; od < 2; ++od)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const f32x4 dv = y[od][k] + nm; q4 += dv * dv; }
                m2 = (q4[0] + q4[1]) + (q4[2] + q4[3]);
            }
        } else {
            unsigned okmask = 0u;
            float sum = 0.f;
            cnt = 0.f;
#pragma unroll
            for (int od = 0; od < 2; ++od)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gd = d0 + 2 * (r >> 3) + od, gh = gh_l + 2 * ((r >> 2) & 1), gw = gw_l + 2 * (r & 3);
                    const bool ok = nvalid && gd < D && gh < H && gw < W;
                    const int soff = ((((2 * (r >> 3) + od) * H + 2 * ((r >> 2) & 1)) * W + 2 * (r & 3)) * yl) * 4;
                    const float v = y[od][r >> 2][r & 3];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rs, ok ? y_voff : OOB, soff, E3_WINO_ST_AUX);
                    cnt += ok ? 1.f : 0.f;
                    sum += ok ? v : 0.f;
                    okmask |= (ok ? 1u : 0u) << (od * 16 + r);
                }
            mean = cnt > 0.f ? sum / cnt : 0.f;
            m2 = 0.f;
            if (do_stats) {
#pragma unroll
                for (int od = 0; od < 2; ++od)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float d = y[od][r >> 2][r & 3] - mean;
                        m2 += ((okmask >> (od * 16 + r)) & 1u) ? d * d : 0.f;
                    }
            }
        }
        TSTAMP(8);
        // flush: merge the lane records of a channel (two lane halves, four waves) in a fixed order and write one record
        auto flush = [&](float fc, float fm, float fs, size_t row) {
            const float c2 = __shfl_xor(fc, 32), mm2 = __shfl_xor(fm, 32), s2 = __shfl_xor(fs, 32);
            welford_merge(fc, fm, fs, c2, mm2, s2);
            if (ehf == 0) {
                float* sc_ = scr + (wave * 32 + ej) * 3;
                sc_[0] = fc; sc_[1] = fm; sc_[2] = fs;
            }
            __syncthreads();
            if (wave == 0 && elane < 32 && n0 + elane < eN) {
                float c0 = 0.f, me = 0.f, mm = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float* sc_ = scr + (w * 32 + elane) * 3;
                    welford_merge(c0, me, mm, sc_[0], sc_[1], sc_[2]);
                }
                float* o = KA()->stats + (row * KA()->Cout + n0 + elane) * 3;
                o[0] = c0; o[1] = me; o[2] = mm;
            }
        };
        if (do_stats) {
            if (pa.wgstats) {        // running record of this lane: Chan's merge with an approximate reciprocal (its error is far below the rounding of the sums)
                float* const run = scr + 4 * 32 * 3 + wave * 64 + elane;
                const float rn = run[0], rmean = run[256], rm2 = run[512];
                const float nn = rn + cnt;
                const float rf = cnt * __builtin_amdgcn_rcpf(fmaxf(nn, 1.f));
                const float dl = mean - rmean;
                const float nmean = rmean + dl * rf, nm2 = rm2 + m2 + dl * dl * rn * rf;
                run[0] = nn; run[256] = nmean; run[512] = nm2;
                if (!has_next) flush(nn, nmean, nm2, (size_t)((blockIdx.x & 7u) * (32u / (unsigned)ntiles) + (blockIdx.x >> 3) / (unsigned)ntiles));
            } else {
                const size_t mtile = (size_t)(((P.nb * tilesD + P.td) * tilesH + P.th) * tilesW + P.tw);
                flush(cnt, mean, m2, mtile);
            }
        }
        TSTAMP(9);
#ifdef E3_WINO_TIMING
        if (tid == 0 && tbrick == 1) tstamp[15] = (long long)__builtin_amdgcn_s_memrealtime();
        ++tbrick;
#endif
        if (!has_next) break;
        {
            const int* const pk = reinterpret_cast<const int*>(scr + 4 * 32 * 3 + 3 * 256) + wave * 64 + elane;
            a_dst = pk[0]; rdA[0] = pk[256]; rdA[1] = pk[512];
        }
        P = Pn;
    }
}

// torch weights -> U[ntile][chunk][pos][hf][co32][4]:  U = (G (x) G (x) G) g, evaluated in double.
//   dgrad == 0:  g[tap][n = co][k = ci] = w[co][ci][tap]           (w is (Cout, Cin, 27))
//   dgrad == 1:  g[tap][n = ci][k = co] = w[co][ci][26 - tap]      (rows/cols swapped, taps flipped)
// K = number of GEMM-K channels (multiple of 8), Ncols = real columns, NPad = padded to 32.
// layout 2 (conv_wino4.hip; 96 positions per chunk): the 256 floats of a (column tile, chunk, position) are [lane 64][ks 2][half 2] of v_mfma_f32_16x16x4_f32's A operand:
//   lane = kk * 16 + m  ->  k = chunk * 8 + 2 kk + ks,  n = ntile * 32 + 8 (m >> 2) + 4 half + (m & 3)
__device__ __forceinline__ void wino_pack_item(const float* __restrict__ w, float* __restrict__ out, int Cin, int dgrad, int K, int Ncols, size_t i, int layout = 0) {
    const int NCH = K >> 3;
    {
        const int e = i & 3, co = (i >> 2) & 31, hf = (i >> 7) & 1;
        const size_t r = i >> 8;
        const int ch = r % NCH, nt = r / NCH;
        int n = nt * 32 + co, k = ch * 8 + hf * 4 + e;
        if (layout >= 1) {
            const int ln = (int)((i >> 2) & 63), m = ln & 15, kq = ln >> 4, ks = e >> 1, half = e & 1;
            n = nt * 32 + 8 * (m >> 2) + 4 * half + (m & 3);
            k = ch * 8 + 2 * kq + ks;
        }
        double g[27];
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            float v = 0.f;
            if (n < Ncols) v = dgrad ? w[((size_t)k * Cin + n) * 27 + (26 - t)] : w[((size_t)n * Cin + k) * 27 + t];
            g[t] = v;
        }
        // separable G: rows [1,0,0], [.5,.5,.5], [.5,-.5,.5], [0,0,1]
        double u1[4][3][3], u2[4][4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double g0 = g[0 * 9 + b * 3 + c], g1 = g[1 * 9 + b * 3 + c], g2 = g[2 * 9 + b * 3 + c];
                u1[0][b][c] = g0; u1[1][b][c] = 0.5 * (g0 + g1 + g2); u1[2][b][c] = 0.5 * (g0 - g1 + g2); u1[3][b][c] = g2;
            }
#pragma unroll
        for (int pd = 0; pd < 4; ++pd)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double g0 = u1[pd][0][c], g1 = u1[pd][1][c], g2 = u1[pd][2][c];
                u2[pd][0][c] = g0; u2[pd][1][c] = 0.5 * (g0 + g1 + g2); u2[pd][2][c] = 0.5 * (g0 - g1 + g2); u2[pd][3][c] = g2;
            }
        if (layout == 2) {      // F(2x2x4) tiles (conv_wino4.hip): 96 positions (pd, ph, pw 6), G of F(4,3) along the w taps; lane layout as layout 1
            float* o4 = out + ((size_t)(nt * NCH + ch) * 96) * 256 + (i & 255);
#pragma unroll
            for (int pd = 0; pd < 4; ++pd)
#pragma unroll
                for (int ph = 0; ph < 4; ++ph) {
                    const double g0 = u2[pd][ph][0], g1 = u2[pd][ph][1], g2 = u2[pd][ph][2];
                    const int pos = (pd * 4 + ph) * 6;
                    o4[(size_t)(pos + 0) * 256] = (float)(0.25 * g0);
                    o4[(size_t)(pos + 1) * 256] = (float)(-(g0 + g1 + g2) / 6.0);
                    o4[(size_t)(pos + 2) * 256] = (float)(-(g0 - g1 + g2) / 6.0);
                    o4[(size_t)(pos + 3) * 256] = (float)(g0 / 24.0 + g1 / 12.0 + g2 / 6.0);
                    o4[(size_t)(pos + 4) * 256] = (float)(g0 / 24.0 - g1 / 12.0 + g2 / 6.0);
                    o4[(size_t)(pos + 5) * 256] = (float)g2;
                }
            return;
        }
        float* o = out + ((size_t)(nt * NCH + ch) * 64) * 256 + (hf * 32 + co) * 4 + e;
#pragma unroll
        for (int pd = 0; pd < 4; ++pd)
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const double g0 = u2[pd][ph][0], g1 = u2[pd][ph][1], g2 = u2[pd][ph][2];
                const int pos = (pd * 4 + ph) * 4;
                o[(size_t)(pos + 0) * 256] = (float)g0;
                o[(size_t)(pos + 1) * 256] = (float)(0.5 * (g0 + g1 + g2));
                o[(size_t)(pos + 2) * 256] = (float)(0.5 * (g0 - g1 + g2));
                o[(size_t)(pos + 3) * 256] = (float)g2;
            }
    }
}

__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int dgrad, int K, int Ncols, int NPad, int layout) {
    const size_t total = (size_t)(NPad >> 5) * (K >> 3) * 256;     // one thread per (ntile, chunk, hf, co, e): all 64 positions
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        wino_pack_item(w, out, Cin, dgrad, K, Ncols, i, layout);
}

// all layers of a network in ONE launch (each layer alone is a 4..256-workgroup, latency-bound kernel: 25 of them cost
// 0.19 ms per training step); workgroup -> job by binary search over the block prefix
struct WinoPackMultiArgs {
    const float* w[WINO_PACK_MAX_JOBS]; float* out[WINO_PACK_MAX_JOBS];
    int Cin[WINO_PACK_MAX_JOBS], K[WINO_PACK_MAX_JOBS], Ncols[WINO_PACK_MAX_JOBS], dgrad[WINO_PACK_MAX_JOBS], layout[WINO_PACK_MAX_JOBS];
    int bstart[WINO_PACK_MAX_JOBS + 1];
    int n;
};
__global__ __launch_bounds__(256) void wino_pack_multi_kernel(const WinoPackMultiArgs a) {
    const int b = blockIdx.x;
    int lo = 0, hi = a.n;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.bstart[mid] <= b) lo = mid; else hi = mid; }
    const int j = lo;
    const int K = a.K[j], Ncols = a.Ncols[j], NPad = (Ncols + 31) / 32 * 32;
    const size_t total = (size_t)(NPad >> 5) * (K >> 3) * 256;
    const size_t i = (size_t)(b - a.bstart[j]) * 256 + threadIdx.x;
    if (i < total) wino_pack_item(a.w[j], a.out[j], a.Cin[j], a.dgrad[j], K, Ncols, i, a.layout[j]);
}

}  // namespace

// ---- host side
size_t wino_packed_floats(int K, int ncols) { return (size_t)96 * K * (size_t)((ncols + 31) / 32 * 32); }      // (96 positions: the F(2x2x4) layout; F(2x2x2) needs 64)

int wino_bricks(int N, int D, int H, int W) { return N * cdiv(D, 4) * cdiv(H, 4) * cdiv(W, 16); }

// The bottom level of a U-Net has few bricks (cfg 2: 2 x 8 x 16 x 16 voxels = 16 bricks x Cout/32 column tiles = 64..128 workgroups for 256 CUs)
// but many input channels, walked serially by each workgroup: split the channels over 2 or 4 workgroups per brick (partial sums to a
// scratch tensor, fixed-order reduction by splitk_reduce_kernel).  Per SAMPLE, like conv_use_wino: independent of the batch size.
static int splitk_factor(size_t nblk1, int K) {
    static const bool enabled = getenv("E3_NO_SPLITK") == nullptr;
    int S = 1;
    while (enabled && S < 4 && nblk1 * S < 256 && K / (2 * S) >= 64 && K % (16 * S) == 0) S *= 2;
    return S;
}

bool conv_use_wino(ConvKind kind, int flags, int N, int D, int H, int W, int Cin, int ncols) {
    static const bool enabled = getenv("E3_CONV_NO_WINO") == nullptr;
    if (!enabled || kind != CONV_K3 || (flags & (CF_SCATTER_UP | CF_GATHER_UP | CF_NO_WINO)) != 0 || Cin < 8 || (Cin & 7)) return false;
    // decided per SAMPLE (not per batch) so that the algorithm, and with it every rounding, is independent of the batch size:
    // eval-mode outputs of a batch are bit-identical to those of its samples run one by one (tests/test_unet_gpu.py)
    (void)N;
    size_t grid = (size_t)wino_bricks(1, D, H, W) * ((ncols + 31) / 32);
    if (flags & CF_SPLITK_OK) grid *= splitk_factor(grid, Cin);
    return grid >= 64u;        // a Winograd workgroup does 3.4x less matrix work than a direct one: worth it from 1/4 of the CUs
}

int conv_wino_splitk(int D, int H, int W, int K, int ncols) {
    if (!conv_use_wino(CONV_K3, CF_SPLITK_OK, 1, D, H, W, K, ncols)) return 0;
    return splitk_factor((size_t)wino_bricks(1, D, H, W) * ((ncols + 31) / 32), K);
}

int launch_wino_pack_multi(const WinoPackJob* jobs, int njobs, hipStream_t s) {
    for (int j0 = 0; j0 < njobs; j0 += WINO_PACK_MAX_JOBS) {
        WinoPackMultiArgs a;
        a.n = njobs - j0 < WINO_PACK_MAX_JOBS ? njobs - j0 : WINO_PACK_MAX_JOBS;
        int b = 0;
        for (int j = 0; j < a.n; ++j) {
            const WinoPackJob& q = jobs[j0 + j];
            int K = q.dgrad ? q.Cout : q.Cin;
            const int ncols = q.dgrad ? q.Cin : q.Cout, NPad = (ncols + 31) / 32 * 32;
            const float* w = q.w;
            if (q.kn > 0) {      // a share of the GEMM-K channels: w is [Cout][Cin][27], K runs over Cin (forward) or Cout (dgrad)
                w += (size_t)q.k0 * 27 * (q.dgrad ? q.Cin : 1);
                K = q.kn;
            }
            a.w[j] = w; a.out[j] = q.out; a.Cin[j] = q.Cin; a.K[j] = K; a.Ncols[j] = ncols; a.dgrad[j] = q.dgrad; a.layout[j] = q.layout;
            a.bstart[j] = b;
            b += (int)(((size_t)(NPad >> 5) * (K >> 3) * 256 + 255) / 256);
        }
        a.bstart[a.n] = b;
        if (b > 0) hipLaunchKernelGGL(wino_pack_multi_kernel, dim3(b), dim3(256), 0, s, a);
        E3_CHECK_HIP(hipGetLastError());
    }
    return E3_OK;
}

int launch_wino_pack(const float* w, float* out, int Cout, int Cin, int dgrad, hipStream_t s, int layout = 0) {
    const int K = dgrad ? Cout : Cin, ncols = dgrad ? Cin : Cout;
    const int NPad = (ncols + 31) / 32 * 32;
    const size_t total = (size_t)(NPad >> 5) * (K >> 3) * 256;
    const int grid = (int)((total + 255) / 256);
    hipLaunchKernelGGL(wino_pack_kernel, dim3(grid), dim3(256), 0, s, w, out, Cout, Cin, dgrad, K, ncols, NPad, layout);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

size_t conv_packed_floats(ConvKind kind, int K, int ncols) {
    const int T = kind == CONV_K3 ? 27 : (kind == CONV_K3_PLANAR ? 9 : 1);
    const int ct = conv_col_tile(ncols);
    const size_t direct = (size_t)T * (cdiv(ncols, ct) * ct) * K;
    const size_t wino = kind == CONV_K3 ? wino_packed_floats(K, ncols) : (kind == CONV_K3_PLANAR ? wino2d_packed_floats(K, ncols) : 0);
    return direct > wino ? direct : wino;
}

int launch_pack_conv_auto(ConvKind kind, int dgrad, const float* w, float* out, int Cout, int Cin, int N, int D, int H, int W, int flags, hipStream_t s) {
    const int K = dgrad ? Cout : Cin, ncols = dgrad ? Cin : Cout;
    if (conv_use_wino(kind, flags, N, D, H, W, K, ncols)) return launch_wino_pack(w, out, Cout, Cin, dgrad, s, conv_wino_layout(flags, D, H, W, K, ncols, 1));
    if (conv_use_wino2d(kind, flags, N, D, H, W, K, ncols)) return launch_wino2d_pack(w, out, Cout, Cin, dgrad, s);
    const int T = kind == CONV_K3 ? 27 : 9, ct = conv_col_tile(ncols);
    return launch_pack_weights(dgrad ? PACK_CONV_DGRAD : PACK_CONV_FWD, w, out, Cout, Cin, T, cdiv(ncols, ct) * ct, s);
}

// does a Winograd launch of `nblk` workgroup-bricks take the persistent kernel?  (splits == 1 and no BN prologue are the caller's business)
static bool wino_persistent(size_t nblk, int flags) {
    static const bool persist = getenv("E3_WINO_NO_PERSIST") == nullptr;
    static const size_t pmin = getenv("E3_WINO_PERSIST_MIN") ? (size_t)atol(getenv("E3_WINO_PERSIST_MIN")) : 512;    // two bricks per workgroup are enough (measured: 512 = 1024 - 0.7 % of the step; tests force 1: every shape)
    return persist && nblk >= pmin && !(flags & (1024 | CF_NO_PERSIST));
}
// one statistics record per WORKGROUP (256 / ntiles rows of [Cout][3]) instead of one per brick: possible when every workgroup of the
// 256-workgroup persistent grid stays inside one column tile and the workgroups of a row cover all column tiles -- XCD ranges that start
// at multiples of ntiles (nblk / 8 a multiple of ntiles) and a step of 32 logical bricks that is one too
static bool wino_wgstats(size_t nblk, int ntiles, unsigned pgrid = 256u) {
    return pgrid == 256u && nblk >= 256 && nblk % 8 == 0 && (nblk / 8) % (size_t)ntiles == 0 && 32 % ntiles == 0;
}
int wino_stats_parts(int N, int D, int H, int W, int Cin, int ncols, int flags) {
    const int lay = conv_wino_layout(flags, D, H, W, Cin, ncols, 1);
    if (lay == 2) return wino4_stats_parts(N, D, H, W, ncols);
    const int bricks = wino_bricks(N, D, H, W), ntiles = (ncols + 31) / 32;
    const size_t nblk = (size_t)bricks * ntiles;
    return (wino_persistent(nblk, flags) && wino_wgstats(nblk, ntiles)) ? 256 / ntiles : bricks;
}

int launch_conv3_wino(ConvArgs a, hipStream_t s) {
    const int lay = conv_wino_layout(a.flags, a.D, a.H, a.W, a.Cin, a.Ncols, a.splitk);
    if (lay == 2) return launch_conv3_wino4(a, s);
    E3_REQUIRE(!a.x_chunk && !a.y_chunk, E3_ERR_UNSUPPORTED, "channel-chunked input / output: F(2x2x4) Winograd kernel only");
    a.tilesD = cdiv(a.D, 4); a.tilesH = cdiv(a.H, 4); a.tilesW = cdiv(a.W, 16);
    a.o_td = a.o_th = a.o_tw = 0;
    if (a.box_hi[0] > 0) {      // needed region: the bricks that meet the box
        E3_REQUIRE(!a.stats && a.splitk <= 1, E3_ERR_INVALID, "conv with a needed region: no statistics, no split-K");
        const int dims[3] = {a.D, a.H, a.W}, edge[3] = {4, 4, 16};
        int o[3], n[3];
        for (int i = 0; i < 3; ++i) {
            const int lo = a.box_lo[i] < 0 ? 0 : a.box_lo[i], hi = a.box_hi[i] > dims[i] ? dims[i] : a.box_hi[i];
            E3_REQUIRE(hi > lo, E3_ERR_INVALID, "conv with a needed region: empty box");
            // bricks start at the box's (even) low corner, not at a multiple of the brick edge: the 1-voxel margins that the box of a conv in front of
            // another conv carries would otherwise cost a whole extra brick per axis (cfg 5's level-0 concat conv: 26 x 50 x 14 -> 25 x 49 x 13 bricks).
            // EVEN origins keep every voxel in the 2x2x2 Winograd tile it has in the whole-tensor launch: same arithmetic, bit-identical values.
            static const bool brick_aligned = getenv("E3_WINO_BOX_ALIGNED") != nullptr;      // A/B switch: origins at multiples of the brick edge
            o[i] = brick_aligned ? lo / edge[i] * edge[i] : (lo & ~1);
            n[i] = cdiv(hi - o[i], edge[i]);
        }
        a.org_d = o[0]; a.org_h = o[1]; a.org_w = o[2];
        a.tilesD = n[0]; a.tilesH = n[1]; a.tilesW = n[2];
    }
    a.NPad = (a.Ncols + 31) / 32 * 32;
    a.ntiles = a.NPad / 32;
    if (a.stats) a.cu_reserve = 0;      // (the statistic records are sized for the full grid; only data gradients run beside a collective)
    const size_t nblk = (size_t)a.N * a.tilesD * a.tilesH * a.tilesW * a.ntiles;
    E3_REQUIRE(nblk > 0 && nblk < (1u << 31), E3_ERR_INVALID, "conv grid out of range");
    E3_REQUIRE((size_t)6 * a.H * a.W * (size_t)(a.x_ldc > a.y_ldc ? a.x_ldc : a.y_ldc) * 4 < 0x7fffffffu, E3_ERR_UNSUPPORTED,
               "six d-planes of the conv input/output view exceed 2^31 bytes (32-bit buffer offsets)");
    constexpr int lds_bytes = W_LDS_FLOATS * 4;
    static bool attr_set = false;
    if (!attr_set) {
        E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wino_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wino_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        attr_set = true;
    }
    // persistent variant (staging pipelined across bricks) where a CU gets several bricks; with one or two bricks per CU the plain kernel is as
    // fast.  E3_WINO_NO_PERSIST=1: A/B switch.
    const unsigned splits = a.splitk > 1 ? (unsigned)a.splitk : 1u;
    if (splits > 1) E3_REQUIRE(!a.bias && !a.stats && !a.epi_scale && !a.pro_scale && a.Cin == a.sk_x, E3_ERR_INVALID, "split-K conv: bias / statistics / fused prologue or epilogue are not available");
    if (splits == 1 && !a.pro_scale && wino_persistent(nblk, a.flags)) {
        constexpr int plds = W_PLDS_FLOATS * 4;
        static bool pattr = false;
        if (!pattr) {
            E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wino_pkernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, plds));
            E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wino_pkernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, plds));
            pattr = true;
        }
        E3_REQUIRE(!(a.epi_scale && a.stats), E3_ERR_INVALID, "Winograd conv: statistics and the folded epilogue exclude each other");
        // one workgroup per CU (256 is a multiple of the 8 XCDs; smaller grids run one brick each).  cu_reserve > 0 (a multiple of 8): that many
        // CUs are left to the resident workgroups of a collective on a side stream -- a persistent workgroup that finds no free CU would
        // only start when another one has finished ALL its bricks, i.e. the launch would take twice as long
        const unsigned pfull = 256u - (unsigned)((a.cu_reserve < 0 ? 0 : (a.cu_reserve > 128 ? 128 : a.cu_reserve)) & ~7);
        const unsigned pgrid = nblk >= pfull ? pfull : (unsigned)nblk;
        // a workgroup's bricks are L0, L0 + pgrid / 8, ... in the logical (XCD-blocked) order: digits of that step in the mixed radix
        // of brick_order.h for the division-free brick counters
        WinoPArgs pa{};
        pa.b = brick_step_make(pgrid == pfull ? pfull / 8u : 0u, pgrid / 8u, a.ntiles, a.tilesW, a.tilesH, a.tilesD, 64);
        pa.wgstats = (a.stats && wino_wgstats(nblk, a.ntiles, pgrid)) ? 1 : 0;
        pa.org_d = a.org_d + 4 * a.o_td; pa.org_h = a.org_h + 4 * a.o_th; pa.org_w = a.org_w + 16 * a.o_tw;      // (o_t*: first brick of a needed region)
        // launches without statistics whose output view allows 16-byte stores: transposed accumulators (E3_WINO_NO_TR=1: A/B switch)
        static const bool no_tr = getenv("E3_WINO_NO_TR") != nullptr;
        const bool tr = !no_tr && !a.stats && (a.Ncols & 3) == 0 && (a.y_ldc & 3) == 0 && ((uintptr_t)a.y & 15) == 0;
        if (tr) {
            static bool tattr = false;
            if (!tattr) {
                E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wino_pkernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, plds));
                E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wino_pkernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, plds));
                tattr = true;
            }
            static const bool no_head = getenv("E3_WINO_NO_HEAD") != nullptr;      // A/B switch
            if (a.epi_scale && a.head_w && a.head_done && !no_head && a.Ncols == 32 && a.head_cout >= 1 && a.head_cout <= 4 && !a.pool_out) {      // + the 1x1x1 head behind it
                constexpr int plds_head = (W_PLDS_FLOATS + W_POOLX) * 4;
                static bool hattr = false;
                if (!hattr) { E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wino_pkernel<true, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, plds_head)); hattr = true; }
                hipLaunchKernelGGL((conv3_wino_pkernel<true, true, false, true>), dim3(pgrid), dim3(256), plds_head, s, a, (unsigned)nblk, pa);
                *a.head_done = 1;
                E3_CHECK_HIP(hipGetLastError());
                return E3_OK;
            }
            static const bool no_pool = getenv("E3_WINO_NO_POOL") != nullptr;      // A/B switch
            if (a.epi_scale && a.pool_out && a.pool_done && !no_pool && a.box_hi[0] <= 0 && (a.Ncols & 31) == 0) {      // + the max-pool behind it
                constexpr int plds_pool = (W_PLDS_FLOATS + W_POOLX) * 4;
                static bool pattr2 = false;
                if (!pattr2) { E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wino_pkernel<true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, plds_pool)); pattr2 = true; }
                hipLaunchKernelGGL((conv3_wino_pkernel<true, true, true>), dim3(pgrid), dim3(256), plds_pool, s, a, (unsigned)nblk, pa);
                *a.pool_done = 1;
            } else if (a.epi_scale) hipLaunchKernelGGL((conv3_wino_pkernel<true, true>), dim3(pgrid), dim3(256), plds, s, a, (unsigned)nblk, pa);
            else hipLaunchKernelGGL((conv3_wino_pkernel<false, true>), dim3(pgrid), dim3(256), plds, s, a, (unsigned)nblk, pa);
        } else if (a.epi_scale) hipLaunchKernelGGL(conv3_wino_pkernel<true>, dim3(pgrid), dim3(256), plds, s, a, (unsigned)nblk, pa);
        else hipLaunchKernelGGL(conv3_wino_pkernel<false>, dim3(pgrid), dim3(256), plds, s, a, (unsigned)nblk, pa);
        E3_CHECK_HIP(hipGetLastError());
        return E3_OK;
    }
    if (a.pro_scale) hipLaunchKernelGGL(conv3_wino_kernel<true>, dim3((unsigned)nblk), dim3(256), lds_bytes, s, a);
    else hipLaunchKernelGGL(conv3_wino_kernel<false>, dim3((unsigned)nblk * splits), dim3(256), lds_bytes, s, a);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
