// Weight gradient of the 3x3x3 / 1x3x3 convolution on the bf16 matrix cores (autograd's dW of nn.Conv3d as conv3() builds it,
// elektronn3/models/unet.py:131-149):
//
//   dW[co][ci][tap] = sum over voxels v:  dY[v][co] * X[v + tap][ci]
//
// A GEMM whose contraction index is the VOXEL, while both tensors are stored channel-contiguous (NDHWC).  gfx950's transposing LDS
// read (ds_read_b64_tr_b16) delivers exactly the k-major fragments v_mfma_f32_32x32x16_bf16 wants from [voxel][32 channels] images:
// within a 16-lane group, lane t supplies the address of 4 consecutive channels of voxel row t/4 and receives channel t of the
// four rows.  64-byte rows make those reads conflict-free at any tap shift, and let the LDS-DMA staging
// (buffer_load_dwordx4 ... lds) copy whole 64-byte row segments with no swizzle.
//
// Workgroup = (32 co x 32 ci) tile x ALL taps over a contiguous range of 2x8x16-voxel bricks; the 4 waves split the taps
// (7/7/7/6 accumulator tiles, resident over the whole range: no atomics, fixed summation order).  A k-step = the 16 voxels of one
// w-row; the dY fragment of a k-step is shared by the wave's taps.  Result: fp32 partial slabs part[split][tap][co][ci] in the
// layout of the fp32 path, finished by wgrad_reduce_kernel.
#include <type_traits>
#include "bf16.h"
#include "kernels.h"      // (WSkPart: the stream-K partition shared with the fp32 path)

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
// (a plain function: called straight from a kernel TEMPLATE, hipcc's host pass drops the kernel's stub)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, lds_ptr_t dst, int, unsigned voff, int, int, int) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, 0, 0, 0);
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr unsigned OOB = 0x80000000u;
constexpr int HH = 10, HW = 18;

__device__ __forceinline__ bf16x8 tr_frag(unsigned addr) {       // 8 k-values: rows 0..3 (addr) and 4..7 (addr + 4 rows)
    typedef s16x4 __attribute__((address_space(3))) * lp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)addr);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(addr + 256));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ unsigned range_mask(int lo, int n, int size) {      // bit z set: lo + z in [0, size), z in [0, n)
    const int first = lo < 0 ? -lo : 0, last = size - lo < n ? size - lo : n;
    return last > first ? (unsigned)(((1ull << last) - 1ull) & ~((1ull << first) - 1ull)) : 0u;
}

#ifdef E3_CONV_TIMING
__device__ unsigned long long* g_wtiming = nullptr;      // phase timestamps (tools/conv_phases.py --wgrad): [workgroup][brick < 8][4]
#define E3_WTICK(k) do { if (threadIdx.x == 0 && g_wtiming && brick - brick0 < 8 && blockIdx.x < 1024) g_wtiming[((size_t)blockIdx.x * 8 + (brick - brick0)) * 4 + (k)] = wall_clock64(); } while (0)
#else
#define E3_WTICK(k)
#endif

// One unit of work: the (32 co x 32 ci) tile (co0, ci0) of one layer over the bricks [brick0, brick1); the partial result goes to
// out[tap * tap_stride + row * row_stride + column].  The one-layer kernel runs one segment per workgroup, the cross-layer stream-K kernel (round 6) a list.
struct WSegB {
    const bf16_t* x; const bf16_t* x2; const bf16_t* dy;
    int x_split, x_ldc, dy_ldc, N, D, H, W, tilesD, tilesH, tilesW;
    int ci0, co0, brick0, brick1;
    float* out; int tap_stride, row_stride;
};

template <int KD>
__device__ __forceinline__ void wgrad_b16_segment(const WSegB& a, unsigned char* const smem) {
    constexpr int HD = 2 + (KD == 3 ? 2 : 0), PD = KD == 3 ? 1 : 0;
    constexpr int HV = HD * HH * HW;
    constexpr int XP = (HV * 4 + 63) / 64;        // 1 KB pieces of the X image (45 / 23)
    constexpr int XI = (XP + 3) / 4;
    constexpr int XIMG = XP * 1024;
    constexpr int TAPS = KD * 9, TPW = (TAPS + 3) / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesD = a.tilesD, tilesH = a.tilesH, tilesW = a.tilesW;
    const int ci0 = a.ci0, co0 = a.co0;
    const bool second = a.x2 && ci0 >= a.x_split;                       // this tile's input channels live in the second tensor
    const bf16_t* xsrc = second ? a.x2 : a.x;
    const int cisrc = second ? ci0 - a.x_split : ci0;
    const int brick0 = a.brick0, brick1 = a.brick1;

    // ---- staging plan.  Validity of a halo voxel is separable: one scalar mask per brick (4 d | 10 h | 18 w bits), one constant
    // 3-bit pattern per lane and piece
    unsigned xpm[XI], xrel[XI];
#pragma unroll
    for (int it = 0; it < XI; ++it) {
        const int idx = (it * 4 + wave) * 64 + lane;
        const int v = idx >> 2, q = idx & 3;
        const int zw = v % HW, zh = (v / HW) % HH, zd = v / (HW * HH);
        xpm[it] = v < HV ? (1u << zd) | (1u << (4 + zh)) | (1u << (14 + zw)) : 0xffffffffu;
        xrel[it] = (unsigned)((((zd * a.H + zh) * a.W + zw) * a.x_ldc + cisrc) * 2 + q * 16);
    }
    unsigned gpm[4], grel[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = (it * 4 + wave) * 64 + lane;
        const int v = idx >> 2, q = idx & 3;
        const int ww = v & 15, hh = (v >> 4) & 7, dd = v >> 7;
        gpm[it] = (1u << dd) | (1u << (4 + hh)) | (1u << (14 + ww));
        grel[it] = (unsigned)((((dd * a.H + hh) * a.W + ww) * a.dy_ldc + co0) * 2 + q * 16);
    }
    const size_t samp_x = (size_t)a.D * a.H * a.W * a.x_ldc, samp_g = (size_t)a.D * a.H * a.W * a.dy_ldc;

    // ---- fragment addresses
    const int G = lane >> 4, t = lane & 15;
    const int krow = 8 * (G >> 1) + (t >> 2), chb = (16 * (G & 1) + 4 * (t & 3)) * 2;
    const unsigned ybase = (unsigned)(XIMG + krow * 64 + chb);                  // + k-step * 1024
    unsigned xt[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        int tap = wave * TPW + i; tap = tap < TAPS ? tap : TAPS - 1;            // (the last wave repeats a tap in its spare slot)
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        xt[i] = (unsigned)((((kd * HH + kh) * HW + kw) + krow) * 64 + chb);     // + ((dd * HH + hh) * HW) * 64
    }

    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    for (int brick = brick0; brick < brick1; ++brick) {
        E3_WTICK(0);
        int Lt = brick;
        const int tw_ = Lt % tilesW; Lt /= tilesW; const int th_ = Lt % tilesH; Lt /= tilesH; const int td_ = Lt % tilesD; const int nb = Lt / tilesD;
        const int d0 = td_ * 2, h0 = th_ * 8, w0 = tw_ * 16;
        const unsigned xmask = range_mask(d0 - PD, HD, a.D) | (range_mask(h0 - 1, HH, a.H) << 4) | (range_mask(w0 - 1, HW, a.W) << 14);
        const unsigned gmask = range_mask(d0, 2, a.D) | (range_mask(h0, 8, a.H) << 4) | (range_mask(w0, 16, a.W) << 14);
        const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xsrc) + (size_t)nb * samp_x, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.dy) + (size_t)nb * samp_g, 0, 0x7fffffff, 0x00020000);
        const unsigned xbase = (unsigned)(((((d0 - PD) * a.H + h0 - 1) * a.W + w0 - 1) * a.x_ldc) * 2);   // wraps at the borders
        const unsigned gbase = (unsigned)((((d0 * a.H + h0) * a.W + w0) * a.dy_ldc) * 2);
#pragma unroll
        for (int it = 0; it < XI; ++it) {
            const int wi = it * 4 + wave;
            if (wi < XP) {
                const bool ok = (xmask & xpm[it]) == xpm[it];
                dma16(x_rs, (lds_ptr_t)(smem + wi * 1024), 16, ok ? xrel[it] + xbase : OOB, 0, 0, 0);
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const bool ok = (gmask & gpm[it]) == gpm[it];
            dma16(g_rs, (lds_ptr_t)(smem + XIMG + (it * 4 + wave) * 1024), 16, ok ? grel[it] + gbase : OOB, 0, 0, 0);
        }
        E3_WTICK(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        E3_WTICK(2);
        // the fragments of k-step s+1 (one dY fragment, one X fragment per tap) are requested before the MFMAs of step s: a transposing
        // read takes ~130 cycles, and left to the compiler every MFMA waits for its own read (lgkmcnt(0) in front of each of them)
        bf16x8 af[2], bfr[2][TPW];
        af[0] = tr_frag(ybase);
#pragma unroll
        for (int i = 0; i < TPW; ++i) bfr[0][i] = tr_frag(xt[i]);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (s + 1 < 16) {
                const int dd = (s + 1) >> 3, hh = (s + 1) & 7;
                af[(s + 1) & 1] = tr_frag(ybase + (s + 1) * 1024);
#pragma unroll
                for (int i = 0; i < TPW; ++i) bfr[(s + 1) & 1][i] = tr_frag(xt[i] + ((dd * HH + hh) * HW) * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TPW; ++i) acc[i] = E3_MFMA16(af[s & 1], bfr[s & 1][i], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        E3_WTICK(3);
        __syncthreads();
    }
    // ---- partial tile: lane holds column ci = lane & 31, rows (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tap = wave * TPW + i;
        if (tap >= TAPS) continue;
        float* dst = a.out + (size_t)tap * a.tap_stride + (lane & 31);
#pragma unroll
        for (int e = 0; e < 16; ++e) dst[(size_t)((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * a.row_stride] = acc[i][e];
    }
}

// ---- 3x3x3, sliding window along D (round 6).  The generic form stages the whole 4 x 10 x 18 halo of every 2 x 8 x 16 brick: 2.8 x the brick's voxels, and the kernel is
// bound by that L2 -> LDS traffic (61 KB per brick at the 25 - 40 GB/s a CU gets; DESIGN_LOG.md 3b).  Here a segment's bricks are walked D-FASTEST and the four halo
// d-planes live in a RING of four 12 KB slots (slot = global plane index & 3): the next brick of a column needs planes 2 td + 1, 2 td + 2 only -- the other two are the
// previous brick's -- so a brick stages 2 x 11.25 KB of X + 16 KB of dY = 39 KB instead of 61.  A plane's slot depends on td's parity: the k-step loop exists in two
// forms (uniform branch), each with compile-time plane slots; tap kd of wave w's i-th tap is a lane-invariant run-time value, so the four possible plane addresses of a
// tap are kept in registers (7 x 4).
constexpr int SW_PLANE = 12 * 1024;                 // 180 voxels x 64 B, padded to 12 DMA pieces
constexpr int SW_XIMG = 4 * SW_PLANE;
__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rs, lds_ptr_t dst, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, soff, 0, 0);
}

__device__ __forceinline__ void wgrad_b16_segment_sw(const WSegB& a, unsigned char* const smem) {
    constexpr int TAPS = 27, TPW = 7;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesD = a.tilesD, tilesH = a.tilesH, tilesW = a.tilesW;
    const int ci0 = a.ci0, co0 = a.co0;
    const bool second = a.x2 && ci0 >= a.x_split;
    const bf16_t* xsrc = second ? a.x2 : a.x;
    const int cisrc = second ? ci0 - a.x_split : ci0;
    const int brick0 = a.brick0, brick1 = a.brick1;

    // ---- staging plan of ONE plane: 12 pieces, piece 4 j + wave of this wave (j = 0..2); lane -> (voxel of the 10 x 18 plane, 16-byte quarter of its 64-byte row)
    unsigned xpm[3], xrel[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int idx = (j * 4 + wave) * 64 + lane;
        const int v = idx >> 2, q = idx & 3;
        const int zw = v % HW, zh = v / HW;
        xpm[j] = v < HH * HW ? (1u << (4 + zh)) | (1u << (14 + zw)) : 0xffffffffu;
        xrel[j] = (unsigned)(((zh * a.W + zw) * a.x_ldc + cisrc) * 2 + q * 16);
    }
    unsigned gpm[4], grel[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = (it * 4 + wave) * 64 + lane;
        const int v = idx >> 2, q = idx & 3;
        const int ww = v & 15, hh = (v >> 4) & 7, dd = v >> 7;
        gpm[it] = (1u << dd) | (1u << (4 + hh)) | (1u << (14 + ww));
        grel[it] = (unsigned)((((dd * a.H + hh) * a.W + ww) * a.dy_ldc + co0) * 2 + q * 16);
    }
    const size_t samp_x = (size_t)a.D * a.H * a.W * a.x_ldc, samp_g = (size_t)a.D * a.H * a.W * a.dy_ldc;
    const int plane_b = a.H * a.W * a.x_ldc * 2;       // bytes between d-planes of x (the launcher bounds a sample by 2^31 bytes)

    // ---- fragment addresses
    const int G = lane >> 4, t = lane & 15;
    const int krow = 8 * (G >> 1) + (t >> 2), chb = (16 * (G & 1) + 4 * (t & 3)) * 2;
    const unsigned ybase = (unsigned)(SW_XIMG + krow * 64 + chb);                  // + k-step * 1024
    // tap i: in-plane address (lane) and kd (wave-uniform): the plane (dd + kd) of a brick whose plane 0 sits in ring slot s0 is slot (s0 + dd + kd) & 3 --
    // a scalar offset added per read (one v_add beside the MFMAs; four precomputed addresses per tap would cost 21 registers this kernel does not have)
    unsigned xin[TPW]; int kdv[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        int tap = wave * TPW + i; tap = tap < TAPS ? tap : TAPS - 1;            // (the last wave repeats a tap in its spare slot)
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        xin[i] = (unsigned)(((kh * HW + kw) + krow) * 64 + chb);                // + (hh * HW) * 64
        kdv[i] = __builtin_amdgcn_readfirstlane(kd);
    }

    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    int prev_col = -1, prev_td = -2;
    for (int brick = brick0; brick < brick1; ++brick) {
        int Lt = brick;
        const int td_ = Lt % tilesD; Lt /= tilesD;      // D fastest: consecutive bricks of a segment share two of the four halo planes
        const int col = Lt;
        const int tw_ = Lt % tilesW; Lt /= tilesW; const int th_ = Lt % tilesH; const int nb = Lt / tilesH;
        const int d0 = td_ * 2, h0 = th_ * 8, w0 = tw_ * 16;
        const bool slide = col == prev_col && td_ == prev_td + 1;
        prev_col = col; prev_td = td_;
        const unsigned dmask = range_mask(d0 - 1, 4, a.D);
        const unsigned hwmask = (range_mask(h0 - 1, HH, a.H) << 4) | (range_mask(w0 - 1, HW, a.W) << 14);
        const unsigned gmask = range_mask(d0, 2, a.D) | (range_mask(h0, 8, a.H) << 4) | (range_mask(w0, 16, a.W) << 14);
        const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(xsrc) + (size_t)nb * samp_x, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.dy) + (size_t)nb * samp_g, 0, 0x7fffffff, 0x00020000);
        const unsigned xbase = (unsigned)((((h0 - 1) * a.W + w0 - 1) * a.x_ldc) * 2);   // (wraps at the borders; the plane offset rides in the scalar operand)
        const unsigned gbase = (unsigned)((((d0 * a.H + h0) * a.W + w0) * a.dy_ldc) * 2);
        const int s0 = (d0 - 1) & 3;                 // ring slot of the brick's plane 0 (global plane d0 - 1): 3 or 1
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < 2 && slide) continue;            // (uniform) planes 0, 1 are the previous brick's planes 2, 3
            const bool pok = (dmask >> k) & 1u;
            const int slot = (s0 + k) & 3;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const bool ok = pok && (hwmask & xpm[j]) == xpm[j];
                dma16s(x_rs, (lds_ptr_t)(smem + slot * SW_PLANE + (j * 4 + wave) * 1024), ok ? xrel[j] + xbase : OOB, (d0 - 1 + k) * plane_b);
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const bool ok = (gmask & gpm[it]) == gpm[it];
            dma16(g_rs, (lds_ptr_t)(smem + SW_XIMG + (it * 4 + wave) * 1024), 16, ok ? grel[it] + gbase : OOB, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        {
            // (s0 is a run-time scalar: ONE form of the k-step loop -- two forms with compile-time slots, selected by a uniform branch, spilled 160+ VGPRs)
            bf16x8 af[2], bfr[2][TPW];
            af[0] = tr_frag(ybase);
#pragma unroll
            for (int i = 0; i < TPW; ++i) bfr[0][i] = tr_frag(xin[i] + (unsigned)(((s0 + kdv[i]) & 3) * SW_PLANE));
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s + 1 < 16) {
                    const int dd = (s + 1) >> 3, hh = (s + 1) & 7;
                    af[(s + 1) & 1] = tr_frag(ybase + (s + 1) * 1024);
#pragma unroll
                    for (int i = 0; i < TPW; ++i) bfr[(s + 1) & 1][i] = tr_frag(xin[i] + (unsigned)(((s0 + dd + kdv[i]) & 3) * SW_PLANE) + (hh * HW) * 64);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TPW; ++i) acc[i] = E3_MFMA16(af[s & 1], bfr[s & 1][i], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    // ---- partial tile: lane holds column ci = lane & 31, rows (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tap = wave * TPW + i;
        if (tap >= TAPS) continue;
        float* dst = a.out + (size_t)tap * a.tap_stride + (lane & 31);
#pragma unroll
        for (int e = 0; e < 16; ++e) dst[(size_t)((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * a.row_stride] = acc[i][e];
    }
}

template <int KD, bool SLIDE = false>
__global__ __launch_bounds__(256, 2) void wgrad_b16_kernel(const WgradB16Args a, int tilesD, int tilesH, int tilesW, int bricks_per_split,
                                                           int co_tiles, int ci_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TAPS = KD * 9;
    unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int ci_t = L % ci_tiles; L /= ci_tiles;
    const int co_t = L % co_tiles; const int split = L / co_tiles;
    const int nbricks = a.N * tilesD * tilesH * tilesW;
    const int CoPad = co_tiles * 32, CiPad = ci_tiles * 32;
    WSegB g;
    g.x = a.x; g.x2 = a.x2; g.dy = a.dy; g.x_split = a.x_split; g.x_ldc = a.x_ldc; g.dy_ldc = a.dy_ldc;
    g.N = a.N; g.D = a.D; g.H = a.H; g.W = a.W; g.tilesD = tilesD; g.tilesH = tilesH; g.tilesW = tilesW;
    g.ci0 = ci_t * 32; g.co0 = co_t * 32;
    g.brick0 = split * bricks_per_split;
    g.brick1 = g.brick0 + bricks_per_split < nbricks ? g.brick0 + bricks_per_split : nbricks;
    g.out = a.part + ((size_t)split * TAPS * CoPad + g.co0) * CiPad + g.ci0; g.tap_stride = CoPad * CiPad; g.row_stride = CiPad;
    if constexpr (SLIDE) wgrad_b16_segment_sw(g, smem); else wgrad_b16_segment<KD>(g, smem);
}

// ---- cross-layer stream-K launch (round 6; the fp32 twin and the scheme: wgrad_wino.hip / kernels.h WSkPart): the 3x3x3 weight gradients of many layers in ONE
// launch of 512 workgroups (two per CU), a private [27][32][32] fp32 tile slab per (workgroup, tile pair) segment, reduced by launch_wgrad_sk_reduce
struct WSkLayerB { const bf16_t* x; const bf16_t* x2; const bf16_t* dy; int x_split, x_ldc, dy_ldc, N, D, H, W, tilesD, tilesH, tilesW; };
struct WSkArgsB { WSkPart p; WSkLayerB L[WSK_MAX_LAYERS]; };

template <bool SLIDE>
__global__ __launch_bounds__(256, 2) void wgrad_b16_sk_kernel(const WSkArgsB a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned wg = xcd_remap(blockIdx.x, gridDim.x);
    unsigned g = wg * a.p.q + (wg < a.p.r ? wg : a.p.r);
    const unsigned gend = (wg + 1) * a.p.q + (wg + 1 < a.p.r ? wg + 1 : a.p.r);
    int l = 0;
    while (g < gend) {
        while (l + 1 < a.p.n && a.p.L[l + 1].g0 <= g) ++l;
        const WSkPartLayer& Lp = a.p.L[l];
        const WSkLayerB& Ly = a.L[l];
        const WSkUnit u = wsk_unit(Lp, g - Lp.g0);
        const unsigned tp = u.tp, b0 = u.brick, want = gend - g, nb = want < u.left ? want : u.left;
        WSegB s;
        s.x = Ly.x; s.x2 = Ly.x2; s.dy = Ly.dy; s.x_split = Ly.x_split; s.x_ldc = Ly.x_ldc; s.dy_ldc = Ly.dy_ldc;
        s.N = Ly.N; s.D = Ly.D; s.H = Ly.H; s.W = Ly.W; s.tilesD = Ly.tilesD; s.tilesH = Ly.tilesH; s.tilesW = Ly.tilesW;
        s.ci0 = (int)(tp % (unsigned)Lp.ci_tiles) * 32; s.co0 = (int)(tp / (unsigned)Lp.ci_tiles) * 32;
        s.brick0 = (int)b0; s.brick1 = (int)(b0 + nb);
        s.out = a.p.slab + (size_t)(wg + Lp.c0 + u.block * (unsigned)Lp.tps + tp) * (27 * 1024); s.tap_stride = 1024; s.row_stride = 32;
        if constexpr (SLIDE) wgrad_b16_segment_sw(s, smem); else wgrad_b16_segment<3>(s, smem);      // (the brick loop ends with a barrier: the stage is free for the next segment)
        g += nb;
    }
}

}  // namespace

#ifdef E3_CONV_TIMING
extern "C" int e3_debug_wgrad_timing(void* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_wtiming), &buf, sizeof(buf)) == hipSuccess ? 0 : 1; }
#endif

int wgrad_b16_splits(int N, int D, int H, int W, int Cin, int Cout, int planar) {
    (void)planar;
    const int nbricks = N * cdiv(D, 2) * cdiv(H, 8) * cdiv(W, 16);
    const int tiles = (Cin / 32) * (Cout / 32);
    int splits = cdiv(512, tiles);
    if (splits > nbricks) splits = nbricks;
    if (splits < 1) splits = 1;
    const int per = cdiv(nbricks, splits);
    return cdiv(nbricks, per);                 // no empty split
}

static bool wgrad_b16_slide() { static const bool on = getenv("E3_B16_WGRAD_NO_SLIDE") == nullptr; return on; }      // A/B switch: the generic staging for the 3x3x3 form too

size_t wgrad_b16_sk_slab_floats(int tile_pairs) { return wgrad_sk_slab_floats(tile_pairs, 512); }

int launch_wgrad_b16_sk(const WgradSkB16Layer* layers, int n, float* slab, size_t slab_floats, hipStream_t s) {
    constexpr int lds = ((4 * HH * HW * 4 + 63) / 64) * 1024 + 16384;
    for (int l0 = 0; l0 < n; l0 += WSK_MAX_LAYERS) {
        WSkArgsB a{};
        const int m = n - l0 < WSK_MAX_LAYERS ? n - l0 : WSK_MAX_LAYERS;
        int Cin[WSK_MAX_LAYERS], Cout[WSK_MAX_LAYERS], nbr[WSK_MAX_LAYERS]; float* dw[WSK_MAX_LAYERS];
        for (int i = 0; i < m; ++i) {
            const WgradSkB16Layer& q = layers[l0 + i];
            E3_REQUIRE(q.Cin % 32 == 0 && q.Cout % 32 == 0, E3_ERR_UNSUPPORTED, "bf16 wgrad: channel counts must be multiples of 32");
            E3_REQUIRE(q.x_ldc % 8 == 0 && q.dy_ldc % 8 == 0, E3_ERR_INVALID, "bf16 wgrad: misaligned view");
            WSkLayerB& L = a.L[i];
            L.x = q.x; L.x2 = q.x2; L.dy = q.dy; L.x_split = q.x_split; L.x_ldc = q.x_ldc; L.dy_ldc = q.dy_ldc;
            L.N = q.N; L.D = q.D; L.H = q.H; L.W = q.W; L.tilesD = cdiv(q.D, 2); L.tilesH = cdiv(q.H, 8); L.tilesW = cdiv(q.W, 16);
            Cin[i] = q.Cin; Cout[i] = q.Cout; nbr[i] = q.N * L.tilesD * L.tilesH * L.tilesW; dw[i] = q.dw;
        }
        const int rc = wgrad_sk_partition(a.p, m, Cin, Cout, nbr, dw, 512, slab, slab_floats);
        if (rc) return rc;
        if (wgrad_b16_slide()) hipLaunchKernelGGL(wgrad_b16_sk_kernel<true>, dim3(512), dim3(256), SW_XIMG + 16384, s, a);
        else hipLaunchKernelGGL(wgrad_b16_sk_kernel<false>, dim3(512), dim3(256), lds, s, a);
        E3_CHECK_HIP(hipGetLastError());
        const int rr = launch_wgrad_sk_reduce(a.p, s);
        if (rr) return rr;
    }
    return E3_OK;
}

int launch_wgrad_b16(WgradB16Args a, hipStream_t s) {
    E3_REQUIRE(a.Cin % 32 == 0 && a.Cout % 32 == 0, E3_ERR_UNSUPPORTED, "bf16 wgrad: channel counts must be multiples of 32");
    E3_REQUIRE(a.x_ldc % 8 == 0 && a.dy_ldc % 8 == 0, E3_ERR_INVALID, "bf16 wgrad: misaligned view");
    const int tD = cdiv(a.D, 2), tH = cdiv(a.H, 8), tW = cdiv(a.W, 16);
    const int nbricks = a.N * tD * tH * tW;
    const int per = cdiv(nbricks, a.splits);
    const int co_tiles = a.Cout / 32, ci_tiles = a.Cin / 32;
    const unsigned grid = (unsigned)(a.splits * co_tiles * ci_tiles);
    if (a.planar) {
        constexpr int lds = ((2 * HH * HW * 4 + 63) / 64) * 1024 + 16384;
        hipLaunchKernelGGL(wgrad_b16_kernel<1>, dim3(grid), dim3(256), lds, s, a, tD, tH, tW, per, co_tiles, ci_tiles);
    } else {
        constexpr int lds = ((4 * HH * HW * 4 + 63) / 64) * 1024 + 16384;
        if (wgrad_b16_slide()) hipLaunchKernelGGL((wgrad_b16_kernel<3, true>), dim3(grid), dim3(256), SW_XIMG + 16384, s, a, tD, tH, tW, per, co_tiles, ci_tiles);
        else hipLaunchKernelGGL(wgrad_b16_kernel<3>, dim3(grid), dim3(256), lds, s, a, tD, tH, tW, per, co_tiles, ci_tiles);
    }
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
