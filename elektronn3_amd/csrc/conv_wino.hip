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
    const int d0 = td_ * 4, h0 = th_ * 4, w0 = tw_ * 16;
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

    const bool dbg = (a.flags & 1024) != 0;     // timing experiments: s_memtime stamps into the statistics buffer
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

// ---- persistent variant: one workgroup per CU walks the bricks bid, bid + grid, ...; the input staging is software-pipelined ACROSS
// bricks: the last chunk of a brick, whose "next chunk" slot is idle, fetches / D-transforms / stages the first chunk of the next
// brick into the other LDS buffer, so a brick starts with its first chunk already in LDS (no exposed global-load latency, no
// wasted re-staging).  Nothing but a few scalars lives across the epilogue, which has its own LDS region (130 KB in total).
constexpr int W_PLDS_FLOATS = 2 * W_BUF + W_EX + 4 * 32 * 3;

__global__ __launch_bounds__(256, 1) void conv3_wino_pkernel(const ConvArgs a, const unsigned nblk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hf = lane >> 5;
    // The arguments are re-read from the kernarg segment (scalar loads) wherever a brick needs them instead of living in ~40 SGPRs
    // across the whole persistent loop (which made hipcc spill 200+ scalars into vector lanes).
    typedef const __attribute__((address_space(4))) ConvArgs* KArgs;
    auto KA = []() -> KArgs { KArgs q = (KArgs)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(q)); return q; };
    const int NCH = a.Cin >> 3;
    constexpr unsigned OOB = 0x80000000u;

    // ---- lane constants that do not depend on the brick
    const bool col_on = tid < W_LH * W_LW * 2;
    const int cq = tid & 1, czw = (tid >> 1) % W_LW, czh = (tid >> 1) / W_LW;
    const int a_dst = col_on ? plane_slot(czh, czw, cq) : 0;
    float m1 = -1.f;
    asm volatile("" : "+s"(m1));
    const int ttd = j >> 4, tth = (j >> 3) & 1, ttw = j & 7;
    const int lbase = (ttd * 4 + wave) * W_PLANE + (tth * 9 + ttw) * 8;
    int rdA[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) rdA[hh] = lbase + 4 * (hf ^ ((tth + hh) & 1));
    const int b_voff = lane * 16;

    // ---- per-brick state.  Out: where the brick's results go (scalars).  Stage: where the staging slot reads from -- the current
    // brick, or already the NEXT one during a brick's last chunk: a descriptor, per-d-plane scalar offsets / validity flags and ONE
    // vector register (byte offset of the thread's halo column inside a d-plane, OOB outside H x W).
    struct Out { int d0, h0, w0, nb, n0, ntile, mtile; };
    struct Stage { __amdgpu_buffer_rsrc_t x_rs; unsigned col_off; unsigned dflag[W_AI]; unsigned dsoff[W_AI]; };
    auto divmod = [](unsigned& x, int d) {
        int r;
        if ((d & (d - 1)) == 0) { r = (int)(x & (unsigned)(d - 1)); x >>= __builtin_ctz((unsigned)d); }
        else { r = (int)(x % (unsigned)d); x /= (unsigned)d; }
        return r;
    };
    auto make_out = [&](unsigned bid, Out& o) {
        const KArgs k = KA();
        unsigned L = xcd_remap(bid, nblk);
        o.ntile = divmod(L, k->ntiles);
        // (a 4 x 4 x 2 block of bricks per XCD and pass instead of this w-h-d order was measured: 11 % fewer bytes fetched over the step's 15
        // launches -- 4.90 -> 4.38 GB -- and 0.8 % MORE time, 13.13 -> 13.24 ms per step: the kernel is matrix/VALU-bound and the blocks
        // start their waves on colder L2 lines.  Not kept.)
        const int tw_ = divmod(L, k->tilesW);
        const int th_ = divmod(L, k->tilesH);
        const int td_ = divmod(L, k->tilesD);
        o.nb = (int)L;
        o.d0 = td_ * 4; o.h0 = th_ * 4; o.w0 = tw_ * 16; o.n0 = o.ntile * 32;
        o.mtile = ((o.nb * k->tilesD + td_) * k->tilesH + th_) * k->tilesW + tw_;
    };
    auto make_stage = [&](const Out& o, bool real, Stage& p) {
        const KArgs k = KA();
        const int D = k->D, H = k->H, W = k->W, xl = k->x_ldc;
        const size_t plane_x = (size_t)H * W * xl;
        const unsigned plane_xb = (unsigned)(plane_x * 4);
        const int dlo = o.d0 > 0 ? o.d0 - 1 : 0;
        const size_t xrem = (size_t)(D - dlo) * plane_x * 4;
        p.x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(k->x) + ((size_t)o.nb * D + dlo) * plane_x, 0,
                                                   (int)(xrem < 0x7fffffffu ? xrem : 0x7fffffffu), 0x00020000);
        const int cgh = o.h0 + czh - 1, cgw = o.w0 + czw - 1;
        const bool col_ok = real && col_on && cgh >= 0 && cgh < H && cgw >= 0 && cgw < W;
        p.col_off = col_ok ? (unsigned)(((cgh * W + cgw) * xl + 4 * cq) * 4) : OOB;
#pragma unroll
        for (int zd = 0; zd < W_AI; ++zd) {
            const int gd = o.d0 + zd - 1;
            const bool ok = gd >= 0 && gd < D;
            p.dflag[zd] = ok ? 0u : OOB;
            p.dsoff[zd] = ok ? (unsigned)(gd - dlo) * plane_xb : 0u;
        }
    };

    f32x16 acc[16];
    f32x4 xr[W_AI], Bv[16];
    Stage S;
    auto issue_raw = [&](int cb) {
#pragma unroll
        for (int it = 0; it < W_AI; ++it)
            xr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(S.x_rs, S.col_off | S.dflag[it], (int)S.dsoff[it] + cb * 4, 0));
    };
    auto write_raw = [&](float* buf) {
        if (col_on) {
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
    __amdgpu_buffer_rsrc_t b_rs;
    auto load_B = [&](int c, int g) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            Bv[g * 4 + p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_voff + p * 1024, (c * 64 + g * 4) * 1024, 0));
    };

    // one 8-channel chunk of the current brick: transformed halo in `cur`; the chunk staged meanwhile into `nxt` is channel
    // offset cb_next of stage S (the same brick, or the first chunk of the next brick when this is the brick's last chunk).
    // `cB` = chunk whose weights are fetched for the next iteration.
    auto chunk = [&](auto zero_tag, const float* cur, float* nxt, int cb_next, int cB) {
        constexpr bool ZERO = decltype(zero_tag)::value;
        issue_raw(cb_next);
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
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                f32x2 lo = {t[h][w][0], t[h][w][1]}, hi = {t[h][w][2], t[h][w][3]};
                asm("" : "+v"(lo)); asm("" : "+v"(hi));
                t[h][w][0] = lo[0]; t[h][w][1] = lo[1]; t[h][w][2] = hi[0]; t[h][w][3] = hi[1];
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if (ZERO && s == 0) {
                        f32x16 z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.f;
                        acc[g * 4 + p] = __builtin_amdgcn_mfma_f32_32x32x2f32(t[g][p][s], Bv[g * 4 + p][s], z, 0, 0, 0);
                    } else {
                        acc[g * 4 + p] = __builtin_amdgcn_mfma_f32_32x32x2f32(t[g][p][s], Bv[g * 4 + p][s], acc[g * 4 + p], 0, 0, 0);
                    }
                }
            load_B(cB, g);                            // (unconditional: a branch here makes hipcc drain every memory counter mid-MFMA)
        }
        __builtin_amdgcn_sched_barrier(0);
        write_raw(nxt);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };

    float* cur = smem;
    float* nxt = smem + W_BUF;
    float* ex = smem + 2 * W_BUF;
    float* scr = ex + W_EX;
    unsigned bid = blockIdx.x;
    Out P;
    make_out(bid, P);
    make_stage(P, true, S);
    issue_raw(0);
    write_raw(cur);
    __syncthreads();
    for (;;) {
        const unsigned nbid = bid + gridDim.x;
        const bool has_next = nbid < nblk;
        b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(KA()->wt) + ((size_t)P.ntile * NCH * 64 + wave * 16) * 256, 0, NCH * 64 * 1024, 0x00020000);
#pragma unroll
        for (int g = 0; g < 4; ++g) load_B(0, g);
        auto advance_stage = [&]() {                      // from here on the staging slot works for the NEXT brick (zeros if none)
            Out on;
            make_out(has_next ? nbid : bid, on);
            make_stage(on, has_next, S);
        };
        // chunk 0 starts the accumulators from a literal zero; the brick's last chunk stages the next brick's first chunk
        if (NCH == 1) advance_stage();
        chunk(std::true_type{}, cur, nxt, NCH == 1 ? 0 : 8, NCH == 1 ? 0 : 1);
        { float* tsw = cur; cur = nxt; nxt = tsw; }
        for (int c = 1; c < NCH; ++c) {
            const bool last = c + 1 == NCH;
            if (last) advance_stage();
            chunk(std::false_type{}, cur, nxt, last ? 0 : (c + 1) * 8, last ? 0 : c + 1);   // (last: a harmless re-fetch of chunk 0's weights)
            { float* tsw = cur; cur = nxt; nxt = tsw; }
        }

        // ---- epilogue (as in conv3_wino_kernel, with its own LDS region)
        const KArgs e = KA();
        const int d0 = P.d0, h0 = P.h0, w0 = P.w0, n0 = P.n0;
        const int eD = e->D, eH = e->H, eW = e->W, yl = e->y_ldc, eN = e->Ncols;
        const size_t plane_y = (size_t)eH * eW * yl;
        const size_t yrem = (size_t)(eD - d0) * plane_y * 4;
        const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(
            e->y + ((size_t)P.nb * eD + d0) * plane_y, 0, (int)(yrem < 0x7fffffffu ? yrem : 0x7fffffffu), 0x00020000);
        f32x16 q[2][2];
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const f32x16 t0 = acc[ph * 4 + 0] + acc[ph * 4 + 1] + acc[ph * 4 + 2];
            const f32x16 t1 = acc[ph * 4 + 1] + m1 * acc[ph * 4 + 2] + m1 * acc[ph * 4 + 3];
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
                for (int k = 0; k < 4; ++k) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = q[oh][ow][4 * k + e];
                    *reinterpret_cast<f32x4*>(ex + ((wave * 16 + (oh * 2 + ow) * 4 + k) * 64 + lane) * 4) = v;
                }
        __syncthreads();
        const int oh = wave >> 1, ow = wave & 1;
        const int n = n0 + j;
        const bool nvalid = n < eN;
        const bool aff = e->epi_scale != nullptr;
        const float bias = (e->bias && nvalid) ? e->bias[n] : 0.f;
        float es = 1.f, eh = 0.f;
        if (aff && nvalid) { es = e->epi_scale[n]; eh = e->epi_shift[n]; }
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
        const int gw_l = w0 + 8 * hf + ow, gh_l = h0 + oh;
        const unsigned y_voff = (unsigned)(((gh_l * eW + gw_l) * yl + n) * 4);
        const bool full = d0 + 4 <= eD && h0 + 4 <= eH && w0 + 16 <= eW && n0 + 32 <= eN;
        const bool do_stats = e->stats != nullptr;
        float cnt = 0.f, sum = 0.f;
        unsigned okmask = 0xffffffffu;
        if (full) {
#pragma unroll
            for (int od = 0; od < 2; ++od)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int soff = ((((2 * (r >> 3) + od) * eH + 2 * ((r >> 2) & 1)) * eW + 2 * (r & 3)) * yl) * 4;
                    const float v = y[od][r >> 2][r & 3];
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
                    const bool ok = nvalid && gd < eD && gh < eH && gw < eW;
                    const int soff = ((((2 * (r >> 3) + od) * eH + 2 * ((r >> 2) & 1)) * eW + 2 * (r & 3)) * yl) * 4;
                    const float v = y[od][r >> 2][r & 3];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rs, ok ? y_voff : OOB, soff, 0);
                    cnt += ok ? 1.f : 0.f;
                    sum += ok ? v : 0.f;
                    okmask |= (ok ? 1u : 0u) << (od * 16 + r);
                }
        }
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
            if (hf == 0) {
                float* sc = scr + (wave * 32 + j) * 3;
                sc[0] = cnt; sc[1] = mean; sc[2] = m2;
            }
            __syncthreads();
            if (tid < 32 && n < eN) {
                float c0 = 0.f, me = 0.f, mm = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float* sc = scr + (w * 32 + tid) * 3;
                    welford_merge(c0, me, mm, sc[0], sc[1], sc[2]);
                }
                float* o = e->stats + ((size_t)P.mtile * e->Cout + n) * 3;
                o[0] = c0; o[1] = me; o[2] = mm;
            }
        }
        if (!has_next) break;
        bid = nbid;
        make_out(bid, P);
    }
}

// torch weights -> U[ntile][chunk][pos][hf][co32][4]:  U = (G (x) G (x) G) g, evaluated in double.
//   dgrad == 0:  g[tap][n = co][k = ci] = w[co][ci][tap]           (w is (Cout, Cin, 27))
//   dgrad == 1:  g[tap][n = ci][k = co] = w[co][ci][26 - tap]      (rows/cols swapped, taps flipped)
// K = number of GEMM-K channels (multiple of 8), Ncols = real columns, NPad = padded to 32.
__device__ __forceinline__ void wino_pack_item(const float* __restrict__ w, float* __restrict__ out, int Cin, int dgrad, int K, int Ncols, size_t i) {
    const int NCH = K >> 3;
    {
        const int e = i & 3, co = (i >> 2) & 31, hf = (i >> 7) & 1;
        const size_t r = i >> 8;
        const int ch = r % NCH, nt = r / NCH;
        const int n = nt * 32 + co, k = ch * 8 + hf * 4 + e;
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

__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int dgrad, int K, int Ncols, int NPad) {
    const size_t total = (size_t)(NPad >> 5) * (K >> 3) * 256;     // one thread per (ntile, chunk, hf, co, e): all 64 positions
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        wino_pack_item(w, out, Cin, dgrad, K, Ncols, i);
}

// all layers of a network in ONE launch (each layer alone is a 4..256-workgroup, latency-bound kernel: 25 of them cost
// 0.19 ms per training step); workgroup -> job by binary search over the block prefix
struct WinoPackMultiArgs {
    const float* w[WINO_PACK_MAX_JOBS]; float* out[WINO_PACK_MAX_JOBS];
    int Cin[WINO_PACK_MAX_JOBS], K[WINO_PACK_MAX_JOBS], Ncols[WINO_PACK_MAX_JOBS], dgrad[WINO_PACK_MAX_JOBS];
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
    if (i < total) wino_pack_item(a.w[j], a.out[j], a.Cin[j], a.dgrad[j], K, Ncols, i);
}

}  // namespace

// ---- host side
size_t wino_packed_floats(int K, int ncols) { return (size_t)64 * K * (size_t)((ncols + 31) / 32 * 32); }

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
            a.w[j] = w; a.out[j] = q.out; a.Cin[j] = q.Cin; a.K[j] = K; a.Ncols[j] = ncols; a.dgrad[j] = q.dgrad;
            a.bstart[j] = b;
            b += (int)(((size_t)(NPad >> 5) * (K >> 3) * 256 + 255) / 256);
        }
        a.bstart[a.n] = b;
        if (b > 0) hipLaunchKernelGGL(wino_pack_multi_kernel, dim3(b), dim3(256), 0, s, a);
        E3_CHECK_HIP(hipGetLastError());
    }
    return E3_OK;
}

int launch_wino_pack(const float* w, float* out, int Cout, int Cin, int dgrad, hipStream_t s) {
    const int K = dgrad ? Cout : Cin, ncols = dgrad ? Cin : Cout;
    const int NPad = (ncols + 31) / 32 * 32;
    const size_t total = (size_t)(NPad >> 5) * (K >> 3) * 256;
    const int grid = (int)((total + 255) / 256);
    hipLaunchKernelGGL(wino_pack_kernel, dim3(grid), dim3(256), 0, s, w, out, Cout, Cin, dgrad, K, ncols, NPad);
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
    if (conv_use_wino(kind, flags, N, D, H, W, K, ncols)) return launch_wino_pack(w, out, Cout, Cin, dgrad, s);
    if (conv_use_wino2d(kind, flags, N, D, H, W, K, ncols)) return launch_wino2d_pack(w, out, Cout, Cin, dgrad, s);
    const int T = kind == CONV_K3 ? 27 : 9, ct = conv_col_tile(ncols);
    return launch_pack_weights(dgrad ? PACK_CONV_DGRAD : PACK_CONV_FWD, w, out, Cout, Cin, T, cdiv(ncols, ct) * ct, s);
}

int launch_conv3_wino(ConvArgs a, hipStream_t s) {
    a.tilesD = cdiv(a.D, 4); a.tilesH = cdiv(a.H, 4); a.tilesW = cdiv(a.W, 16);
    a.NPad = (a.Ncols + 31) / 32 * 32;
    a.ntiles = a.NPad / 32;
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
    // persistent variant (staging pipelined across bricks) where a CU gets several bricks: it hides the first-chunk load latency of
    // every brick but the first (-5..10 % on the 32-/64-channel layers at full resolution); with one or two bricks per CU the plain
    // kernel is as fast or 1-3 % faster.  E3_WINO_NO_PERSIST=1: A/B switch.
    static const bool persist = getenv("E3_WINO_NO_PERSIST") == nullptr;
    static const size_t pmin = getenv("E3_WINO_PERSIST_MIN") ? (size_t)atol(getenv("E3_WINO_PERSIST_MIN")) : 1024;   // (tests force 1: every shape)
    const unsigned splits = a.splitk > 1 ? (unsigned)a.splitk : 1u;
    if (splits > 1) E3_REQUIRE(!a.bias && !a.stats && !a.epi_scale && !a.pro_scale && a.Cin == a.sk_x, E3_ERR_INVALID, "split-K conv: bias / statistics / fused prologue or epilogue are not available");
    if (persist && splits == 1 && nblk >= pmin && !a.pro_scale && !(a.flags & (1024 | CF_NO_PERSIST))) {
        constexpr int plds = W_PLDS_FLOATS * 4;
        static bool pattr = false;
        if (!pattr) { E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_wino_pkernel), hipFuncAttributeMaxDynamicSharedMemorySize, plds)); pattr = true; }
        const unsigned pgrid = nblk >= 256 ? 256u : (unsigned)nblk;   // one workgroup per CU (256 is a multiple of the 8 XCDs; smaller grids run one brick each)
        hipLaunchKernelGGL(conv3_wino_pkernel, dim3(pgrid), dim3(256), plds, s, a, (unsigned)nblk);
        E3_CHECK_HIP(hipGetLastError());
        return E3_OK;
    }
    if (a.pro_scale) hipLaunchKernelGGL(conv3_wino_kernel<true>, dim3((unsigned)nblk), dim3(256), lds_bytes, s, a);
    else hipLaunchKernelGGL(conv3_wino_kernel<false>, dim3((unsigned)nblk * splits), dim3(256), lds_bytes, s, a);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
