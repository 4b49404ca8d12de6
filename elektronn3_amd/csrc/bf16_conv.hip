// 3x3x3 / 1x3x3 convolution (stride 1, zero padding 1) on the bf16 matrix cores: forward and data gradient of nn.Conv3d as the
// reference's conv3() builds it (elektronn3/models/unet.py:131-149, planar form :114-128), for bf16 NDHWC tensors.
//
// Implicit GEMM, no im2col buffer:  Y^T[co][v] = sum_tap sum_ci W_tap[co][ci] * X[v + tap][ci]
//   A operand (rows = output channels)  = the packed weights, straight from L2 into a register ring several taps deep
//   B operand (columns = voxels)        = the input brick's halo, staged ONCE per 32-channel chunk into LDS by LDS-DMA
//                                         (buffer_load_dwordx4 ... lds, zero padding = the descriptor's range check)
//   v_mfma_f32_32x32x16_bf16, fp32 accumulators.  The result tile has a VOXEL per lane and 16 output channels in registers
//   (4 consecutive ones per register quad), which is exactly the NDHWC store pattern: bias, bf16 rounding, 8-byte stores, no
//   transposition.  BatchNorm statistics (of the rounded values, shifted by the bias) are summed per lane and reduced through LDS.
//
// Workgroup = brick of BD x 8 x 16 (or BD x 4 x 32) voxels x (32 * CO_T) output channels, 4 waves; a wave owns BD tiles of 32 voxels.
// LDS image: [halo voxel][32 channels] = 64-byte rows, the four 16-byte pieces of a row XOR-swizzled by bits 2..3 of the voxel's
// w coordinate (applied to the SOURCE address of the DMA: the LDS side of a DMA is lane-linear), so that a wave's ds_read_b128 of
// 16 consecutive voxels covers all 64 banks; every (tile, kd, kh) is an immediate offset on one of 6 lane addresses (3 kw x 2 k-steps).
// One LDS buffer per workgroup, 2-3 workgroups per CU: one workgroup's staging overlaps the others' MFMA phases.
// Channels of the input may come from two tensors (x2 / x_split) and go to two (y2 / y_split): the halves of a U-Net concatenation.
#include "bf16.h"

namespace {

// GEMM row r of a 32-row tile holds output channel swap23(r) (bits 2 and 3 exchanged): the accumulator layout of v_mfma_f32_32x32x16 then gives
// lane (voxel, g) the channels 16 k + 8 g + 0..7 in registers 8 k .. 8 k + 7 -- 16-byte stores, two lanes = 32 contiguous bytes of a voxel's row
__host__ __device__ __forceinline__ int swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }
typedef __attribute__((address_space(3))) void* lds_ptr_t;
// (a plain function: called straight from a kernel TEMPLATE, hipcc's host pass drops the kernel's stub)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, lds_ptr_t dst, int, unsigned voff, int, int, int) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, 0, 0, 0);
}
constexpr unsigned OOB = 0x80000000u;

// Brick = BD x BH x BW voxels = BD x 4 tiles of 32 voxels per d-slice.  TW = 16: tile = 2 h-rows x 16 w (brick BD x 8 x 16; works for any
// W); TW = 32: tile = 1 h-row x 32 w (brick BD x 4 x 32): the 16 lanes of a ds_read_b128 group then read 16 CONSECUTIVE voxels of one
// row at any tap shift, which the swizzle maps to 16 distinct bank slots -- no conflicts (the 2 x 16 tile straddles two rows 18 voxels
// apart: a quarter more LDS cycles, measured 22 % of the LDS-active cycles of the 64->32 layer).
template <int BD, int KD, int TW, int CH = 32>
struct Geo {
    static constexpr int RPT = 32 / TW;                   // h-rows per tile
    // 3x3x3: a brick is BD slices x 4 row groups (wave = slice).  1x3x3 (planar block, dim = 2 network): the BD x 4 tiles all lie in ONE slice
    // (wave = BD consecutive row groups) -- no depth halo to pay for, so the brick grows in h instead: 18 x 34 halo voxels for 16 x 32 outputs (1.2x)
    // where BD slices of 6 x 34 cost 1.6x, and a depth-1 volume fills its tiles
    static constexpr bool FLAT = KD == 1;
    static constexpr int DZ = FLAT ? 1 : BD;              // brick depth
    static constexpr int BH = 4 * RPT * (FLAT ? BD : 1), BW = TW;
    static constexpr int HH = BH + 2, HW = BW + 2;
    static constexpr int HD = FLAT ? 1 : BD + 2;
    static constexpr int HV = HD * HH * HW;              // halo voxels
    static constexpr int RB = CH * 2;                    // bytes of a voxel's row in the LDS image (CH channels per chunk: 32, or 16 = one k-step)
    static constexpr int PPV = RB / 16;                  // 16-byte pieces per voxel
    static constexpr int KS = CH / 16;                   // MFMA k-steps per chunk
    static constexpr int NI = (HV * PPV + 63) / 64;      // 1 KB wave-pieces per chunk
    static constexpr int NIW = (NI + 3) / 4;             // per wave
    static constexpr int IMG = NI * 1024;                // bytes
    static constexpr int NV = BD;                        // 2x16-voxel tiles per wave
    static constexpr int TAPS = KD * 9;
};

#ifdef E3_CONV_TIMING
__device__ unsigned long long* g_timing = nullptr;       // phase timestamps (tools/conv_phases.py): [workgroup][16]
#define E3_TICK(k) do { if (threadIdx.x == 0 && g_timing && blockIdx.x < 8192) g_timing[(size_t)blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define E3_TICK(k)
#endif

// CH = 16: the image holds ONE k-step of channels (half the LDS), so that three workgroups instead of two share a CU (a fourth does not fit the
// register file: 64 accumulators + weight ring + fragments need > 128 registers) and a workgroup's staging meets two others' MFMA / store
// phases (same bricks, same halo traffic, twice the barriers)
template <int BD, int CO_T, int KD, int TW, int CH = 32>
__global__ __launch_bounds__(256, (CH == 16 || BD <= 2) ? 3 : 2) void conv_b16_kernel(const ConvB16Args a, int tilesD, int tilesH, int tilesW, int cgroups, int ksplit, int o_td, int o_th, int o_tw) {
    using G = Geo<BD, KD, TW, CH>;
    constexpr int RB = G::RB, PPV = G::PPV, KS = G::KS;
    constexpr int HH = G::HH, HW = G::HW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, g = lane >> 5;
    E3_TICK(0);
    unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = L % cgroups; L /= cgroups;
    const int ksp = L % ksplit; L /= ksplit;            // split-K: this workgroup sums the channel chunks [ch0, ch1) into partial[ksp]
    const int brick = (int)L;
    // brick order: the bricks an XCD works on at the same time (64 of them: 32 CUs x 2) should be neighbours in all three dimensions, so
    // that their shared halo planes are L2 hits -- all tilesD bricks of a 2 x 2 (h, w) column before the next column, instead of whole
    // d-slabs (whose 128 bricks x 140 KB outlive the 4 MB L2 before the next slab asks for the shared planes again)
    int tw, th, td, n;
    if (((tilesH | tilesW) & 1) == 0) {
        const unsigned per = (unsigned)tilesD * 4;
        const unsigned col = L / per, r = L % per;
        const unsigned ncol = (unsigned)(tilesH >> 1) * (tilesW >> 1);
        n = col / ncol; const unsigned c2 = col % ncol;
        td = r >> 2; th = 2 * (c2 / (tilesW >> 1)) + ((r >> 1) & 1); tw = 2 * (c2 % (tilesW >> 1)) + (r & 1);
    } else {
        tw = L % tilesW; L /= tilesW; th = L % tilesH; L /= tilesH; td = L % tilesD; n = L / tilesD;
    }
    const int d0 = td * G::DZ + o_td, h0 = th * G::BH + o_th, w0 = tw * G::BW + o_tw;      // (o_*: voxel origin of the needed region's first brick)
    const int co0 = cg * 32 * CO_T;
    constexpr int PD = KD == 3 ? 1 : 0;
    const int nch = a.Cin / CH;
    const int ch0 = ksp * (nch / ksplit), ch1 = ch0 + nch / ksplit;

    // ---- staging plan: wave-piece wi = it * 4 + wave, lane -> (halo voxel, LDS piece); source piece = LDS piece ^ swizzle
    const size_t samp = (size_t)a.D * a.H * a.W * a.x_ldc;
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x) + (size_t)n * samp, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t x2_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x2 ? a.x2 : a.x) + (size_t)n * samp, 0, 0x7fffffff, 0x00020000);
    const int xsplit_ch = a.x2 ? a.x_split / CH : nch;                  // chunks [0, xsplit_ch) from x, the rest from x2
    unsigned rel[G::NIW]; unsigned okmask = 0;
#pragma unroll
    for (int it = 0; it < G::NIW; ++it) {
        const int idx = (it * 4 + wave) * 64 + lane;
        const int v = idx / PPV, qp = idx % PPV;
        const int zw = v % HW, zh = (v / HW) % HH, zd = v / (HW * HH);
        const int q = PPV == 4 ? qp ^ ((zw >> 2) & 3) : qp ^ ((zw >> 3) & 1);      // (32-byte rows: 16 consecutive voxels x 2 pieces = all 64 banks)
        const int gd = d0 - PD + zd, gh = h0 - 1 + zh, gw = w0 - 1 + zw;
        const bool ok = v < G::HV && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
        rel[it] = (unsigned)((((gd * a.H + gh) * a.W + gw) * a.x_ldc) * 2 + q * 16);
        okmask |= ok ? (1u << it) : 0u;
    }

    // ---- lane read addresses (tap kd = kh = 0, tile 0 of the wave): 3 kw x 2 k-steps
    const int r = TW == 16 ? j >> 4 : 0, c = TW == 16 ? j & 15 : j;
    const int T0 = wave * G::NV, dT0 = G::FLAT ? 0 : T0 >> 2, hp0 = G::FLAT ? T0 : T0 & 3;
    unsigned rd[3][KS];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int hw = c + kw, sw = PPV == 4 ? (hw >> 2) & 3 : (hw >> 3) & 1;
        const int row = (dT0 * HH + G::RPT * hp0 + r) * HW + hw;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) rd[kw][ks] = (unsigned)(row * RB + (((2 * ks + g) ^ sw) << 4));
    }
    // ---- weights: packed [tap][chunk][k-step][CoPad][2][8]; lane (j, g) reads 16 B of row co0 + ct*32 + j
    const size_t wstep = (size_t)a.Cout * 16;                          // elements per (tap, chunk, k-step)
    const bf16_t* wlane = a.wt + (size_t)(co0 + j) * 16 + g * 8;

    f32x16 acc[CO_T][G::NV];
#pragma unroll
    for (int ct = 0; ct < CO_T; ++ct)
#pragma unroll
        for (int t = 0; t < G::NV; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ct][t][e] = 0.f;

    for (int ch = ch0; ch < ch1; ++ch) {
        constexpr int RING = CO_T == 1 ? 6 : 4;      // taps of weights in flight (L2 latency under load is several MFMA taps)
        bf16x8 wf[RING][CO_T][KS];
#define E3_LOAD_W(TAP, SLOT)                                                                                                        \
    _Pragma("unroll") for (int ct_ = 0; ct_ < CO_T; ++ct_)                                                                          \
        _Pragma("unroll") for (int ks_ = 0; ks_ < KS; ++ks_)                                                                        \
            wf[SLOT][ct_][ks_] = *reinterpret_cast<const bf16x8*>(wlane + ((size_t)((TAP) * nch + ch) * KS + ks_) * wstep + ct_ * 512)
#pragma unroll
        for (int t0 = 0; t0 < RING - 1; ++t0) { E3_LOAD_W(t0, t0); }
#pragma unroll
        for (int it = 0; it < G::NIW; ++it) {
            const int wi = it * 4 + wave;
            if (wi < G::NI)
                dma16(ch < xsplit_ch ? x_rs : x2_rs, (lds_ptr_t)(smem + wi * 1024), 16,
                                                         ((okmask >> it) & 1u) ? rel[it] + (unsigned)(ch < xsplit_ch ? ch : ch - xsplit_ch) * (unsigned)RB : OOB, 0, 0, 0);
        }
        E3_TICK(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        E3_TICK(2);
        __syncthreads();
        E3_TICK(3);
        {
            // (tap, k-step) steps; the wave's NV fragments of step s+1 are requested before the MFMAs of step s (a ds_read_b128 takes ~130
            // cycles, the compiler's own order puts two MFMAs = 64 cycles between a read and its use: taps 10.0 -> 5.5 us per brick of the
            // 32 -> 32 layer, tools/conv_phases.py), MFMAs tile-innermost (consecutive MFMAs never share an accumulator)
            constexpr int NSTEP = G::TAPS * KS;
            bf16x8 b[2][G::NV];
#pragma unroll
            for (int t = 0; t < G::NV; ++t) b[0][t] = *reinterpret_cast<const bf16x8*>(smem + rd[0][0] + (G::RPT * t * HW) * RB);
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                const int tap = st / KS, ks = st % KS;
                if (ks == 0 && tap + RING - 1 < G::TAPS) { E3_LOAD_W(tap + RING - 1, (tap + RING - 1) % RING); }
                if (st + 1 < NSTEP) {
                    const int tn = (st + 1) / KS, kn = (st + 1) % KS;
                    const int kd = tn / 9, kh = (tn / 3) % 3, kw = tn % 3;
#pragma unroll
                    for (int t = 0; t < G::NV; ++t)
                        b[(st + 1) & 1][t] = *reinterpret_cast<const bf16x8*>(smem + rd[kw][kn] + ((kd * HH + G::RPT * t + kh) * HW) * RB);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = 0; ct < CO_T; ++ct)
#pragma unroll
                    for (int t = 0; t < G::NV; ++t)
                        acc[ct][t] = E3_MFMA16(wf[tap % RING][ct][ks], b[st & 1][t], acc[ct][t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        E3_TICK(4);
        __syncthreads();
        E3_TICK(5);
#undef E3_LOAD_W
    }

    // ---- epilogue: lane (j, g) holds, for tile t and register e, the output of voxel j and channel (e&3) + 8*(e>>2) + 4*g
    if (ksplit > 1) {           // fp32 partial sums [split][voxel][Cout]; bias, rounding and statistics happen in splitk_reduce_b16_kernel
        const size_t vox = (size_t)a.N * a.D * a.H * a.W;
#pragma unroll
        for (int ct = 0; ct < CO_T; ++ct)
#pragma unroll
            for (int t = 0; t < G::NV; ++t) {
                const int d = d0 + dT0, h = h0 + G::RPT * (hp0 + t) + r, w = w0 + c;
                if (d < a.D && h < a.H && w < a.W) {
                    float* prow = a.partial + ((size_t)ksp * vox + (((size_t)n * a.D + d) * a.H + h) * a.W + w) * a.Cout + co0 + ct * 32 + 8 * g;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<f32x4*>(prow + 16 * (q >> 1) + 4 * (q & 1)) = f32x4{acc[ct][t][4 * q], acc[ct][t][4 * q + 1], acc[ct][t][4 * q + 2], acc[ct][t][4 * q + 3]};
                }
            }
        E3_TICK(6);
        return;
    }
    // Everything outside the MFMA phase costs a wave ~13-24 cycles per vector instruction while the CU's other workgroups run their MFMAs
    // (profiles/r03_bf16_conv_experiments.md), so the epilogue is written for few instructions: the two run-time choices -- folded scale / shift
    // + ReLU or bias, interior brick or a brick that sticks out of the volume -- select one of four straight-line bodies (no per-element
    // branches or selects), four channels at a time as vectors (packed fp32 ops, packed conversions).
    float ssum[CO_T][16], ssq[CO_T][16];
    const bool want_stats = a.stats != nullptr;
    const bool interior = d0 + G::DZ <= a.D && h0 + G::BH <= a.H && w0 + G::BW <= a.W;
    auto body = [&](auto aff_tag, auto full_tag) __attribute__((always_inline)) {
        constexpr bool AFF = decltype(aff_tag)::value, FULL = decltype(full_tag)::value;
#pragma unroll
        for (int ct = 0; ct < CO_T; ++ct) {
            f32x4 bq[4], sq[4], hq[4], s1[4], s2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cb = co0 + ct * 32 + 16 * (q >> 1) + 8 * g + 4 * (q & 1);
                bq[q] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + cb) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (AFF) { sq[q] = *reinterpret_cast<const f32x4*>(a.epi_scale + cb); hq[q] = *reinterpret_cast<const f32x4*>(a.epi_shift + cb); }
                s1[q] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int t = 0; t < G::NV; ++t) {
                const int d = d0 + dT0, h = h0 + G::RPT * (hp0 + t) + r, w = w0 + c;
                const bool valid = FULL || (d < a.D && h < a.H && w < a.W);
                const int cot = co0 + ct * 32;
                bf16_t* yrow = (a.y2 && cot >= a.y_split ? a.y2 + (cot - a.y_split) : a.y + cot) + ((((size_t)n * a.D + d) * a.H + h) * a.W + w) * a.y_ldc + 8 * g;
                bf16x4 rprev;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[ct][t][4 * q], acc[ct][t][4 * q + 1], acc[ct][t][4 * q + 2], acc[ct][t][4 * q + 3]};
                    if (AFF) {
                        v = v * sq[q] + hq[q];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    } else v = v + bq[q];
                    const bf16x4 rb = __builtin_convertvector(v, bf16x4);            // round to nearest even
                    if (q & 1) { if (valid) *reinterpret_cast<bf16x8*>(yrow + 16 * (q >> 1)) = __builtin_shufflevector(rprev, rb, 0, 1, 2, 3, 4, 5, 6, 7); }
                    else rprev = rb;
                    if (!AFF) {               // (statistics only exist without the folded epilogue)
                        f32x4 dv = __builtin_convertvector(rb, f32x4) - bq[q];
                        if (!FULL) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) dv[e] = valid ? dv[e] : 0.f;
                        }
                        s1[q] += dv; s2[q] += dv * dv;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) { ssum[ct][e] = s1[e >> 2][e & 3]; ssq[ct][e] = s2[e >> 2][e & 3]; }
        }
    };
    if (a.epi_scale) { if (interior) body(std::true_type{}, std::true_type{}); else body(std::true_type{}, std::false_type{}); }
    else { if (interior) body(std::false_type{}, std::true_type{}); else body(std::false_type{}, std::false_type{}); }
    E3_TICK(6);
    if (!want_stats) { E3_TICK(7); return; }
    // ---- statistics: S[wave][quantity][channel][33] floats in the (now free) image, column sums, (n, mean, M2) record per brick
    float* S = reinterpret_cast<float*>(smem);
    float* R = S + 4 * 2 * 32 * 33;                    // [2][4][32]
    const int nd = a.D - d0 < G::DZ ? a.D - d0 : G::DZ, nh = a.H - h0 < G::BH ? a.H - h0 : G::BH, nw = a.W - w0 < G::BW ? a.W - w0 : G::BW;
    const float cnt = (float)(nd * nh * nw);
#pragma unroll
    for (int ct = 0; ct < CO_T; ++ct) {
        if (ct) __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int col = 16 * (e >> 3) + 8 * g + (e & 7);
            S[((wave * 2 + 0) * 32 + col) * 33 + j] = ssum[ct][e];
            S[((wave * 2 + 1) * 32 + col) * 33 + j] = ssq[ct][e];
        }
        __syncthreads();
        {
            const int col = tid & 31, qn = (tid >> 5) & 1, wv = tid >> 6;
            const float* row = S + ((wv * 2 + qn) * 32 + col) * 33;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) s += row[k];
            R[(qn * 4 + wv) * 32 + col] = s;
        }
        __syncthreads();
        if (tid < 32) {
            const float s = (R[0 * 32 + tid] + R[1 * 32 + tid]) + (R[2 * 32 + tid] + R[3 * 32 + tid]);
            const float q2 = (R[4 * 32 + tid] + R[5 * 32 + tid]) + (R[6 * 32 + tid] + R[7 * 32 + tid]);
            const int co = co0 + ct * 32 + tid;
            const float b = a.bias ? a.bias[co] : 0.f;
            const float m = s / cnt;
            float* rec = a.stats + ((size_t)brick * a.Cout + co) * 3;
            rec[0] = cnt; rec[1] = b + m; rec[2] = fmaxf(q2 - s * m, 0.f);
        }
    }
    E3_TICK(7);
}
#undef E3_TICK

// ---------------------------------------------------------------- persistent form of the 3x3x3 kernel (level-0 shapes: thousands of bricks)
// Same arithmetic, same operands, same LDS image as conv_b16_kernel<4, 1, 3, 32, 16> (bricks of 4 x 4 x 32 voxels x 32 output channels, 16-channel
// images), but 512 workgroups (two per CU) WALK the work items and the latency chain of a brick -- kernel arguments, index decode, staging plan, first
// weights, halo DMA, barrier, taps, bias loads, stores, statistics, workgroup turnaround: 28 us per brick of which 3 us are matrix instructions
// (profiles/r03_bf16_conv_experiments.md) -- is taken apart:
//   * the image is DOUBLE-BUFFERED: the halo DMA of unit u + 1 (unit = 16-channel chunk of an item; the next item's first chunk after an item's last)
//     is issued right after the barrier that opens unit u, one barrier per unit;
//   * the weight ring (9 taps; 27 = 3 x 9, so slots are static) runs across units: the first taps of unit u + 1 are requested during the last taps of u;
//   * per-lane staging constants (offset relative to the brick origin, halo coordinates) are computed once per workgroup; per item the validity of a
//     piece is three unsigned compares, the brick coordinates come from multiply-high divisions of wave-uniform values;
//   * BatchNorm statistics are running per-lane sums over the workgroup's items (the workgroup's output-channel group is fixed: the per-XCD item
//     ranges and the stride are multiples of the group count), reduced through LDS ONCE: one (n, mean, M2) record per workgroup.
// XCD x owns a contiguous range of items (neighbouring bricks share halo planes in that XCD's L2), its 64 workgroups stride through it.
#ifndef E3_PABL
#define E3_PABL 0      // developer builds (tools/build_b16p_variants.sh), timing only, wrong results: 1 one LDS fragment per step, 2 no weight loads in the loop, 4 no DMA, 8 no MFMAs, 16 no stores
#endif
struct ConvB16PArgs {
    unsigned items, per_xcd;
    unsigned cgroups, m_cg, per, m_per, ncol, m_ncol, tw2, m_tw2;      // divisors and their multiply-high constants (0: divisor 1)
    unsigned ca, cb;                                                   // log2 of a column's cross-section in bricks (H, W)
};
__device__ __forceinline__ unsigned fdiv(unsigned x, unsigned m) { return m ? __umulhi(x, m) : x; }

template <int CO_T>
__global__ __launch_bounds__(256, 2) void conv_b16_pkernel(const ConvB16Args a, const ConvB16PArgs p) {
    using G = Geo<4, 3, 32, 16>;
    constexpr int RB = G::RB, HH = G::HH, HW = G::HW, NIW = G::NIW, NI = G::NI, IMG = G::IMG, NV = G::NV;
    constexpr int RING = 9, TAPS = 27;
    static_assert(G::PPV == 2 && G::KS == 1 && TAPS % RING == 0, "geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, g = lane >> 5;
    const unsigned xcd = blockIdx.x & 7u, wi = blockIdx.x >> 3, wstride = gridDim.x >> 3;
    const unsigned L_hi = (xcd + 1) * p.per_xcd, L_end = L_hi < p.items ? L_hi : p.items;
    unsigned L = xcd * p.per_xcd + wi;
    const int cg = (int)(wi - fdiv(wi, p.m_cg) * p.cgroups);            // the same for every item of this workgroup
    const int co0 = cg * 32 * CO_T;
    const int nch = a.Cin / 16;
    const int xsplit_ch = a.x2 ? a.x_split / 16 : nch;
    const size_t samp = (size_t)a.D * a.H * a.W * a.x_ldc;

    // ---- lane constants of the staging plan: wave-piece wi = it * 4 + wave, lane -> (halo voxel, LDS piece); source piece = LDS piece ^ swizzle
    unsigned relc[NIW], pz[NIW];
#pragma unroll
    for (int it = 0; it < NIW; ++it) {
        const int idx = (it * 4 + wave) * 64 + lane;
        const int v = idx >> 1, qp = idx & 1;
        const int zw = v % HW, zh = (v / HW) % HH, zd = v / (HW * HH);
        const int q = qp ^ ((zw >> 3) & 1);
        relc[it] = (unsigned)((((zd * a.H + zh) * a.W + zw) * a.x_ldc) * 2 + q * 16);
        pz[it] = v < G::HV ? (unsigned)(zd | (zh << 8) | (zw << 16)) : 0x00ffffffu;       // (a piece behind the halo: never valid)
    }
    // ---- lane read addresses (tap kd = kh = 0, tile 0 of the wave): 3 kw
    unsigned rd[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int hw = j + kw, sw = (hw >> 3) & 1;
        rd[kw] = (unsigned)(((wave * HH) * HW + hw) * RB + ((g ^ sw) << 4));
    }
    const size_t wstep = (size_t)a.Cout * 16;                          // elements per (tap, 16-channel chunk)
    const bf16_t* wlane = a.wt + (size_t)(co0 + j) * 16 + g * 8;

    struct Item { int d0, h0, w0, n; };
    auto decode = [&](unsigned Li) {
        unsigned r0 = fdiv(Li, p.m_cg);                                  // brick index (output-channel group innermost)
        const unsigned col = fdiv(r0, p.m_per), r = r0 - col * p.per;
        const unsigned n = fdiv(col, p.m_ncol), c2 = col - n * p.ncol;
        const unsigned c2h = fdiv(c2, p.m_tw2), c2w = c2 - c2h * p.tw2;
        Item it;
        it.n = (int)n; it.d0 = (int)(r >> (p.ca + p.cb)) * 4; it.h0 = (int)((c2h << p.ca) + ((r >> p.cb) & ((1u << p.ca) - 1u))) * 4; it.w0 = (int)((c2w << p.cb) + (r & ((1u << p.cb) - 1u))) * 32;
        return it;
    };
    unsigned voff[NIW];
    auto plan = [&](const Item& it) {
        const unsigned base = (unsigned)((((it.d0 - 1) * a.H + (it.h0 - 1)) * a.W + (it.w0 - 1)) * a.x_ldc * 2);
#pragma unroll
        for (int i = 0; i < NIW; ++i) {
            const unsigned gd = (unsigned)(it.d0 - 1) + (pz[i] & 0xffu), gh = (unsigned)(it.h0 - 1) + ((pz[i] >> 8) & 0xffu), gw = (unsigned)(it.w0 - 1) + (pz[i] >> 16);
            const bool ok = gd < (unsigned)a.D && gh < (unsigned)a.H && gw < (unsigned)a.W;
            voff[i] = ok ? relc[i] + base : OOB;
        }
    };
    auto issue = [&](const Item& it, int ch, int buf) {
        const bool first = ch < xsplit_ch;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(first ? a.x : a.x2) + (size_t)it.n * samp, 0, 0x7fffffff, 0x00020000);
        const unsigned coff = (unsigned)(first ? ch : ch - xsplit_ch) * (unsigned)RB;
#pragma unroll
        for (int i = 0; i < NIW; ++i) {
            const int w_i = i * 4 + wave;
            if (w_i < NI) dma16(rs, (lds_ptr_t)(smem + buf * IMG + w_i * 1024), 16, voff[i] == OOB ? OOB : voff[i] + coff, 0, 0, 0);
        }
    };

    float* const S = reinterpret_cast<float*>(smem);                   // statistics scratch (after the last unit): [wave][2][32][33] + [2][4][32]
    const bool want_stats = a.stats != nullptr;
    f32x4 s1[CO_T][4], s2[CO_T][4];
#pragma unroll
    for (int ct = 0; ct < CO_T; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) { s1[ct][q] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[ct][q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float cnt = 0.f;

    if (L < L_end) {
        Item cur = decode(L), dm = cur;
        unsigned Ldm = L; int chdm = 0, ch = 0, buf = 0;
        plan(dm);
        issue(dm, 0, 0);
        bf16x8 wf[RING][CO_T];
#define E3_LOAD_WP(TAP, CHK, SLOT)                                                                                                  \
    _Pragma("unroll") for (int ct_ = 0; ct_ < CO_T; ++ct_)                                                                          \
        wf[SLOT][ct_] = *reinterpret_cast<const bf16x8*>(wlane + (size_t)((TAP) * nch + (CHK)) * wstep + ct_ * 512)
#pragma unroll
        for (int t0 = 0; t0 < RING - 1; ++t0) { E3_LOAD_WP(t0, 0, t0); }
        f32x16 acc[CO_T][NV];
#pragma unroll
        for (int ct = 0; ct < CO_T; ++ct)
#pragma unroll
            for (int t = 0; t < NV; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ct][t][e] = 0.f;

        bool ring_ahead = true;            // the ring already holds requests of this unit's first taps (issued behind this unit's DMA)
        while (true) {
            // this unit's DMA is older than the RING - 1 weight requests of its first taps: loads return in order, so "at most RING - 1 outstanding"
            // means the image has landed whatever the (unordered) stores of an epilogue in between are doing
            if (ring_ahead) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((RING - 1) * CO_T) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();               // unit u's image has landed (every wave waited for its own pieces); everybody is done reading the other buffer
            // the staging pointer runs one unit ahead
            if (++chdm == nch) { chdm = 0; Ldm += wstride; if (Ldm < L_end) { dm = decode(Ldm); plan(dm); } }
            const bool more = Ldm < L_end;
            if (more && !(E3_PABL & 4)) issue(dm, chdm, buf ^ 1);
            const int chn = ch + 1 == nch ? 0 : ch + 1;        // chunk of the next unit (its first weights enter the ring during this unit's last taps)
            {
                const unsigned char* img = smem + buf * IMG;
                bf16x8 b[2][NV];
#pragma unroll
                for (int t = 0; t < NV; ++t) b[0][t] = *reinterpret_cast<const bf16x8*>(img + rd[0] + (t * HW) * RB);
#pragma unroll
                for (int st = 0; st < TAPS; ++st) {
                    if (!(E3_PABL & 2)) {
                    if (st + RING - 1 < TAPS) { E3_LOAD_WP(st + RING - 1, ch, (st + RING - 1) % RING); }
                    else if (more) { E3_LOAD_WP(st + RING - 1 - TAPS, chn, (st + RING - 1) % RING); }
                    }
                    if (st + 1 < TAPS) {
                        const int tn = st + 1;
                        const int kd = tn / 9, kh = (tn / 3) % 3, kw = tn % 3;
#pragma unroll
                        for (int t = 0; t < ((E3_PABL & 1) ? 1 : NV); ++t)
                            b[(st + 1) & 1][t] = *reinterpret_cast<const bf16x8*>(img + rd[kw] + ((kd * HH + t + kh) * HW) * RB);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ct = 0; ct < CO_T; ++ct)
#pragma unroll
                        for (int t = 0; t < NV; ++t) {
                            if (E3_PABL & 8) { if (t == 0) asm volatile("" :: "v"(b[st & 1][0]), "v"(b[st & 1][(E3_PABL & 1) ? 0 : 1]), "v"(b[st & 1][(E3_PABL & 1) ? 0 : 2]), "v"(b[st & 1][(E3_PABL & 1) ? 0 : 3]), "v"(wf[st % RING][ct])); }
                            else acc[ct][t] = E3_MFMA16(wf[st % RING][ct], b[st & 1][(E3_PABL & 1) ? 0 : t], acc[ct][t], 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            buf ^= 1;
            ring_ahead = more;
            if (++ch < nch) continue;
            // ---- the item is complete: lane (j, g) holds, for tile t and register e, the output of voxel j and channel (e&3) + 8*(e>>2) + 4*g
            {
                const int d0 = cur.d0, h0 = cur.h0, w0 = cur.w0, n = cur.n;
                const bool interior = d0 + 4 <= a.D && h0 + 4 <= a.H && w0 + 32 <= a.W;
                auto body = [&](auto aff_tag, auto full_tag) __attribute__((always_inline)) {
                    constexpr bool AFF = decltype(aff_tag)::value, FULL = decltype(full_tag)::value;
#pragma unroll
                    for (int ct = 0; ct < CO_T; ++ct) {
                        f32x4 bq[4], sq[4], hq[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int cb = co0 + ct * 32 + 16 * (q >> 1) + 8 * g + 4 * (q & 1);
                            bq[q] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + cb) : f32x4{0.f, 0.f, 0.f, 0.f};
                            if (AFF) { sq[q] = *reinterpret_cast<const f32x4*>(a.epi_scale + cb); hq[q] = *reinterpret_cast<const f32x4*>(a.epi_shift + cb); }
                        }
#pragma unroll
                        for (int t = 0; t < NV; ++t) {
                            const int d = d0 + wave, h = h0 + t, w = w0 + j;
                            const bool valid = FULL || (d < a.D && h < a.H && w < a.W);
                            const int cot = co0 + ct * 32;
                            bf16_t* yrow = (a.y2 && cot >= a.y_split ? a.y2 + (cot - a.y_split) : a.y + cot) + ((((size_t)n * a.D + d) * a.H + h) * a.W + w) * a.y_ldc + 8 * g;
                            bf16x4 rprev;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                f32x4 v = {acc[ct][t][4 * q], acc[ct][t][4 * q + 1], acc[ct][t][4 * q + 2], acc[ct][t][4 * q + 3]};
                                if (AFF) {
                                    v = v * sq[q] + hq[q];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                                } else v = v + bq[q];
                                const bf16x4 rb = __builtin_convertvector(v, bf16x4);            // round to nearest even
                                if (q & 1) { if (valid && !((E3_PABL & 16) && a.D > 0)) *reinterpret_cast<bf16x8*>(yrow + 16 * (q >> 1)) = __builtin_shufflevector(rprev, rb, 0, 1, 2, 3, 4, 5, 6, 7); }
                                else rprev = rb;
                                if (!AFF) {               // (statistics only exist without the folded epilogue)
                                    f32x4 dv = __builtin_convertvector(rb, f32x4) - bq[q];
                                    if (!FULL) {
#pragma unroll
                                        for (int e = 0; e < 4; ++e) dv[e] = valid ? dv[e] : 0.f;
                                    }
                                    s1[ct][q] += dv; s2[ct][q] += dv * dv;
                                }
                            }
#pragma unroll
                            for (int e = 0; e < 16; ++e) acc[ct][t][e] = 0.f;
                        }
                    }
                };
                if (a.epi_scale) { if (interior) body(std::true_type{}, std::true_type{}); else body(std::true_type{}, std::false_type{}); }
                else { if (interior) body(std::false_type{}, std::true_type{}); else body(std::false_type{}, std::false_type{}); }
                const int nd = a.D - d0 < 4 ? a.D - d0 : 4, nh = a.H - h0 < 4 ? a.H - h0 : 4, nw = a.W - w0 < 32 ? a.W - w0 : 32;
                cnt += (float)(nd * nh * nw);
            }
            ch = 0; L += wstride;
            if (L >= L_end) break;
            cur = dm;
        }
#undef E3_LOAD_WP
    }
    if (!want_stats) return;
    // ---- statistics of the workgroup's items: S[wave][quantity][channel][33] floats, column sums, ONE (n, mean, M2) record
    float* R = S + 4 * 2 * 32 * 33;                    // [2][4][32]
    const int part = (int)(xcd * (wstride / p.cgroups) + fdiv(wi, p.m_cg));
#pragma unroll
    for (int ct = 0; ct < CO_T; ++ct) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int col = 16 * (e >> 3) + 8 * g + (e & 7);
            S[((wave * 2 + 0) * 32 + col) * 33 + j] = s1[ct][e >> 2][e & 3];
            S[((wave * 2 + 1) * 32 + col) * 33 + j] = s2[ct][e >> 2][e & 3];
        }
        __syncthreads();
        {
            const int col = tid & 31, qn = (tid >> 5) & 1, wv = tid >> 6;
            const float* row = S + ((wv * 2 + qn) * 32 + col) * 33;
            float sm = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) sm += row[k];
            R[(qn * 4 + wv) * 32 + col] = sm;
        }
        __syncthreads();
        if (tid < 32) {
            const float sm = (R[0 * 32 + tid] + R[1 * 32 + tid]) + (R[2 * 32 + tid] + R[3 * 32 + tid]);
            const float q2 = (R[4 * 32 + tid] + R[5 * 32 + tid]) + (R[6 * 32 + tid] + R[7 * 32 + tid]);
            const int co = co0 + ct * 32 + tid;
            const float b = a.bias ? a.bias[co] : 0.f;
            const float m = cnt > 0.f ? sm / cnt : 0.f;
            float* rec = a.stats + ((size_t)part * a.Cout + co) * 3;
            rec[0] = cnt; rec[1] = b + m; rec[2] = fmaxf(q2 - sm * m, 0.f);
        }
    }
}

// torch (Cout, Cin, T) fp32 -> packed bf16 [tap][chunk][k-step][Cg][2][8], Cg = GEMM rows (output channels of THIS launch)
__global__ void pack_conv_b16_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin, int T, int dgrad) {
    const int Kg = dgrad ? Cout : Cin, Cg = dgrad ? Cin : Cout;       // GEMM-K channels, GEMM rows
    const size_t total = (size_t)T * Kg * Cg;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int e = r & 7; r >>= 3; const int g = r & 1; r >>= 1; const int row = r % Cg; r /= Cg;
        const int ks = r & 1; r >>= 1; const int ch = r % (Kg >> 5); const int tap = (int)(r / (Kg >> 5));
        const int k = ch * 32 + ks * 16 + g * 8 + e;
        const int rc = swap23(row);
        const float v = dgrad ? w[((size_t)k * Cin + rc) * T + (T - 1 - tap)] : w[((size_t)rc * Cin + k) * T + tap];
        out[i] = f2bf(v);
    }
}

// split-K epilogue: y[v][c] = bf16(sum_s partial[s][v][c] + bias[c]) (or relu(sum * scale + shift)), statistics of the stored values
// (shifted by the bias) as one (n, mean, M2) record per workgroup and channel.  Thread = 8 channels; the threads of a channel octet
// walk the voxels with a fixed stride, fixed summation order.
__global__ __launch_bounds__(256) void splitk_reduce_b16_kernel(const float* __restrict__ partial, int S, size_t vox, int C, const float* __restrict__ bias,
                                                                const float* __restrict__ epi_scale, const float* __restrict__ epi_shift,
                                                                bf16_t* __restrict__ y, int y_ldc, float* __restrict__ stats, bf16_t* __restrict__ y2, int y_split) {
    __shared__ float red[2][256][8];
    const int Q = C >> 3;
    const int BT = (256 / Q) * Q;
    const int tid = threadIdx.x;
    const bool active = tid < BT;
    const size_t i00 = (size_t)blockIdx.x * BT + tid;
    const int q = (int)(i00 % Q);
    const size_t vstride = (size_t)gridDim.x * BT / Q;
    float bs[8], sc[8], sh[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bs[e] = bias ? bias[8 * q + e] : 0.f;
        sc[e] = epi_scale ? epi_scale[8 * q + e] : 1.f; sh[e] = epi_scale ? epi_shift[8 * q + e] : 0.f;
        s1[e] = 0.f; s2[e] = 0.f;
    }
    float cnt = 0.f;
    for (size_t v = i00 / Q; active && v < vox; v += vstride) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int sp = 0; sp < S; ++sp) {
            const float* p = partial + ((size_t)sp * vox + v) * C + 8 * q;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(p), a1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[e] += a0[e]; acc[4 + e] += a1[e]; }
        }
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float val = epi_scale ? fmaxf(__builtin_fmaf(acc[e], sc[e], sh[e]), 0.f) : acc[e] + bs[e];
            o[e] = f2bf(val);
            const float dv = bf2f(o[e]) - bs[e];
            s1[e] += dv; s2[e] = __builtin_fmaf(dv, dv, s2[e]);
        }
        *reinterpret_cast<u16x8*>((y2 && 8 * q >= y_split ? y2 + (8 * q - y_split) : y + 8 * q) + v * y_ldc) = o;
        cnt += 1.f;
    }
    if (!stats) return;
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][tid][e] = active ? s1[e] : 0.f; red[1][tid][e] = active ? s2[e] : 0.f; }
    __shared__ float cn[256];
    cn[tid] = active ? cnt : 0.f;
    __syncthreads();
    for (int t = tid; t < Q * 8; t += 256) {
        const int e = t & 7, qq = t >> 3;
        float a1 = 0.f, a2 = 0.f, n = 0.f;
        for (int k = qq; k < BT; k += Q) { a1 += red[0][k][e]; a2 += red[1][k][e]; n += cn[k]; }
        const float b = bias ? bias[8 * qq + e] : 0.f;
        const float m = n > 0.f ? a1 / n : 0.f;
        float* rec = stats + ((size_t)blockIdx.x * C + 8 * qq + e) * 3;
        rec[0] = n; rec[1] = b + m; rec[2] = fmaxf(a2 - a1 * m, 0.f);
    }
}

// all weight packings of a pass in ONE launch (a dozen 5-us launches otherwise): block -> job by a prefix table in the kernel arguments
struct PackMultiArgs { PackB16Job job[PACK_B16_MAX_JOBS]; unsigned first_block[PACK_B16_MAX_JOBS + 1]; int njobs; };
__global__ void pack_multi_b16_kernel(const PackMultiArgs a) {
    int jb = 0;
    while (jb + 1 < a.njobs && blockIdx.x >= a.first_block[jb + 1]) ++jb;
    const PackB16Job J = a.job[jb];
    const size_t total = (size_t)J.T * J.Cin * J.Cout;
    const unsigned nblk = a.first_block[jb + 1] - a.first_block[jb];
    for (size_t i = (size_t)(blockIdx.x - a.first_block[jb]) * blockDim.x + threadIdx.x; i < total; i += (size_t)nblk * blockDim.x) {
        size_t r = i;
        const int e = r & 7; r >>= 3; const int g = r & 1; r >>= 1;
        float v;
        if (J.mode < 2) {           // conv: [tap][chunk][k-step][Cg][2][8]
            const int dgrad = J.mode;
            const int Kg = dgrad ? J.Cout : J.Cin, Cg = dgrad ? J.Cin : J.Cout;
            const int row = r % Cg; r /= Cg;
            const int ks = r & 1; r >>= 1; const int ch = r % (Kg >> 5); const int tap = (int)(r / (Kg >> 5));
            const int k = ch * 32 + ks * 16 + g * 8 + e;
            const int rc = swap23(row);
            v = dgrad ? J.w[((size_t)k * J.Cin + rc) * J.T + (J.T - 1 - tap)] : J.w[((size_t)rc * J.Cin + k) * J.T + tap];
        } else {                    // transposed conv: [row tile][k-step][32][2][8]; torch (Cin, Cout, T)
            const int dgrad = J.mode - 2;
            const int K = dgrad ? J.T * J.Cout : J.Cin;
            // (e and g were peeled off above as if g were the second index; the transposed-conv layout is [..][2][32][8])
            const size_t q5 = (r << 1) | g;
            const int rr = q5 & 31, gg = (q5 >> 5) & 1; r = q5 >> 6;
            const int ks = r % (K >> 4); const int rt = (int)(r / (K >> 4));
            const int row = rt * 32 + rr, k = ks * 16 + gg * 8 + e;
            int ci, co, tap;
            if (dgrad) { ci = row; tap = k / J.Cout; co = k % J.Cout; } else { tap = rt % J.T; co = (rt / J.T) * 32 + rr; ci = k; }
            v = J.w[((size_t)ci * J.Cout + co) * J.T + tap];
        }
        J.out[i] = f2bf(v);
    }
}

template <int BD, int CO_T, int KD, int TW, int CH = 32>
int launch_t(const ConvB16Args& a, int ksplit, hipStream_t s) {
    using G = Geo<BD, KD, TW, CH>;
    int tD = cdiv(a.D, G::DZ), tH = cdiv(a.H, G::BH), tW = cdiv(a.W, G::BW);
    int o[3] = {0, 0, 0};
    if (a.box_hi[0] > 0 && !G::FLAT) {      // needed region: the bricks that meet the box
        E3_REQUIRE(!a.stats, E3_ERR_INVALID, "bf16 conv with a needed region: no statistics");
        const int dims[3] = {a.D, a.H, a.W}, edge[3] = {G::DZ, G::BH, G::BW};
        int n[3];
        for (int i = 0; i < 3; ++i) {
            const int lo = a.box_lo[i] < 0 ? 0 : a.box_lo[i], hi = a.box_hi[i] > dims[i] ? dims[i] : a.box_hi[i];
            E3_REQUIRE(hi > lo, E3_ERR_INVALID, "bf16 conv with a needed region: empty box");
            // bricks start at the box's low corner (a direct conv has no tile alignment to keep: any origin gives the same values): the 1-voxel margins of
            // a box in front of another conv do not cost an extra brick per axis
            o[i] = lo; n[i] = cdiv(hi - lo, edge[i]);
        }
        tD = n[0]; tH = n[1]; tW = n[2];
    }
    const int cgroups = a.Cout / (32 * CO_T);
    const size_t grid = (size_t)a.N * tD * tH * tW * cgroups * ksplit;
    const int lds = G::IMG > 4 * 2 * 32 * 33 * 4 + 1024 ? G::IMG : 4 * 2 * 32 * 33 * 4 + 1024;
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)conv_b16_kernel<BD, CO_T, KD, TW, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_done = true; }
    hipLaunchKernelGGL((conv_b16_kernel<BD, CO_T, KD, TW, CH>), dim3((unsigned)grid), dim3(256), lds, s, a, tD, tH, tW, cgroups, ksplit, o[0], o[1], o[2]);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

// Work decomposition of one launch: brick depth (4x8x16 bricks -- each weight fragment feeds 4 tiles per wave, halo overhead 2.1x
// instead of 2.8x -- where they still fill the chip), output-channel tiles per workgroup, and for the low-resolution levels (a few
// dozen bricks for 256 CUs) a split of the input channels over several workgroups (fp32 partial sums, splitk_reduce_b16_kernel).
struct Decomp { int bd, co_t, ksplit, tw; long bricks; int persist; int ca, cb; };      // persist: 1 = conv_b16_pkernel (ca, cb: log2 of its columns' cross-section in bricks)
// cross-section of the persistent kernel's brick columns (bricks of 4 rows x 32 voxels; a column runs through D; an XCD's 64 workgroups work on 64 consecutive
// bricks of a column): 8 x 2 bricks = 32 rows x 64 voxels, clamped to what divides the brick counts.  Measured on cfg 3's shard (tools/b16_col_sweep.sh,
// profiles/r06_brick_order.md): against round 5's 2 x 2 columns the six launches of a step read 1 389 instead of 1 550 MB, the step is 0.3 - 0.5 % shorter;
// columns 4 bricks wide (128 voxels) read 13 % MORE.  E3_B16_COL=a,b (log2, H and W): A/B switch, 1,1 = round 5's columns.
static void conv_b16_column(unsigned tH, unsigned tW, int& ca, int& cb) {
    static const char* const env = getenv("E3_B16_COL");
    int wa = 3, wb = 1;
    if (env) sscanf(env, "%d,%d", &wa, &wb);
    ca = cb = 0;
    while (ca < wa && ca < 4 && tH % (2u << ca) == 0) ++ca;
    while (cb < wb && cb < 4 && tW % (2u << cb) == 0) ++cb;
}
constexpr int PGRID = 512;      // workgroups of the persistent form (two per CU)
Decomp conv_b16_decomp(int N, int D, int H, int W, int Cin, int Cout, int planar) {
    static const int forced = getenv("E3_B16_BD") ? atoi(getenv("E3_B16_BD")) : 0;
    static const bool no_split = getenv("E3_B16_NO_SPLITK") != nullptr;
    static const int forced_tw = getenv("E3_B16_TW") ? atoi(getenv("E3_B16_TW")) : 0;
    Decomp d;
    d.ca = d.cb = 1;
    // 1 x 32-voxel tiles (conflict-free LDS reads) where the rows are long enough to fill them, 2 x 16 otherwise
    d.tw = (forced_tw == 16 || forced_tw == 32) ? forced_tw : (W % 32 == 0 || W >= 96 ? 32 : 16);
    const int bh = d.tw == 32 ? 4 : 8;
    // (planar: the brick is one slice deep and bd row groups high, Geo::FLAT)
    auto bricks = [&](int bd) { return planar ? (long)N * D * cdiv(H, bh * bd) * cdiv(W, d.tw) : (long)N * cdiv(D, bd) * cdiv(H, bh) * cdiv(W, d.tw); };
    d.bd = (forced == 2 || forced == 4) ? forced : (bricks(4) >= 512 ? 4 : 2);
    d.bricks = bricks(d.bd);
    // two output tiles per workgroup halve the staging per FLOP but cost a workgroup per CU (256 registers) and, on small grids, force a
    // split-K pass: measured on cfg 2, they pay at level 1 (512 bricks: 69 vs 71, 118 vs 120 us) but neither at level 2 (128 bricks: one tile
    // fills the chip without split-K, 54 -> 39, 77 -> 67 us) nor at level 0 (4096 bricks: the three-workgroup 16-channel form wins, 264 -> 249 us)
    static const int cot = getenv("E3_B16_COT") ? atoi(getenv("E3_B16_COT")) : 0;      // A/B switch: 1 / 2 = always that many tiles where legal
    d.co_t = (Cout % 64 == 0 && (cot == 2 ? d.bricks * (Cout / 64) >= 256 : (cot != 1 && d.bricks * (Cout / 64) >= 512 && d.bricks <= 1024))) ? 2 : 1;
    const long wgs = d.bricks * (Cout / (32 * d.co_t));
    const int nch = Cin / 32;
    d.ksplit = 1;
    while (!no_split && wgs * d.ksplit < 512 && nch % (2 * d.ksplit) == 0 && d.ksplit < 8) d.ksplit *= 2;
    // persistent form (conv_b16_pkernel): 3x3x3, 4 x 4 x 32 bricks in the column order, one output tile per workgroup, several items per workgroup
    static const bool no_persist = getenv("E3_B16_NO_PERSIST") != nullptr;      // A/B switch
    static const long persist_min = getenv("E3_B16_PERSIST_MIN") ? atol(getenv("E3_B16_PERSIST_MIN")) : 4 * PGRID;      // (tests: 1 = every shape the kernel can take)
    const int cgroups = Cout / 32;
    d.persist = (!no_persist && !planar && d.bd == 4 && d.tw == 32 && d.co_t == 1 && d.ksplit == 1 && wgs >= persist_min && 64 % cgroups == 0 &&
                 (cdiv(H, 4) & 1) == 0 && (cdiv(W, 32) & 1) == 0 && wgs * 256 < (1l << 32)) ? 1 : 0;
    if (d.persist) {
        // multiply-high division x / d = umulhi(x, ceil(2^32 / d)) is exact only while x * (m d - 2^32) < 2^32: every dividend of the kernel is below the item
        // count (+ one XCD share for the padded ranges); a shape beyond that bound takes the one-brick kernels -- decided HERE, so that the statistics sizing
        // (conv_b16_stats_parts) and the launcher ask one predicate
        const unsigned tD = (unsigned)cdiv(D, 4), tH = (unsigned)cdiv(H, 4), tW = (unsigned)cdiv(W, 32);
        conv_b16_column(tH, tW, d.ca, d.cb);
        const unsigned long long items = (unsigned long long)N * tD * tH * tW * cgroups;
        unsigned long long per_xcd = (items + 7) / 8; per_xcd = (per_xcd + cgroups - 1) / cgroups * cgroups;
        for (unsigned dv : {(unsigned)cgroups, tD << (d.ca + d.cb), (tH >> d.ca) * (tW >> d.cb), tW >> d.cb}) {
            if (dv <= 1) continue;
            const unsigned long long m = ((1ull << 32) + dv - 1) / dv, e = m * dv - (1ull << 32), xmax = items + per_xcd + 256;
            if (e * xmax >= (1ull << 32)) d.persist = 0;
        }
    }
    return d;
}
int reduce_blocks(size_t vox, int C) {
    size_t g = (vox * (size_t)(C / 8) + 255) / 256;
    if (g > 512) g = 512;
    if (g == 0) g = 1;
    return (int)g;
}

}  // namespace

#ifdef E3_CONV_TIMING
extern "C" int e3_debug_conv_timing(void* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_timing), &buf, sizeof(buf)) == hipSuccess ? 0 : 1; }
#endif

int conv_b16_stats_parts(int N, int D, int H, int W, int Cin, int Cout, int planar) {
    const Decomp d = conv_b16_decomp(N, D, H, W, Cin, Cout, planar);
    return d.ksplit > 1 ? reduce_blocks((size_t)N * D * H * W, Cout) : (d.persist ? PGRID / (Cout / 32) : (int)d.bricks);
}

size_t conv_b16_partial_floats(int N, int D, int H, int W, int Cin, int Cout, int planar) {
    const Decomp d = conv_b16_decomp(N, D, H, W, Cin, Cout, planar);
    return d.ksplit > 1 ? (size_t)d.ksplit * N * D * H * W * Cout : 0;
}

size_t conv_b16_packed_elems(int Cin, int Cout, int planar) { return (size_t)(planar ? 9 : 27) * Cin * Cout; }

int launch_pack_conv_b16(const float* w, bf16_t* out, int Cout, int Cin, int planar, int dgrad, hipStream_t s) {
    const int T = planar ? 9 : 27;
    const size_t total = (size_t)T * Cin * Cout;
    const unsigned grid = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_conv_b16_kernel, dim3(grid), dim3(256), 0, s, w, out, Cout, Cin, T, dgrad);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_pack_multi_b16(const PackB16Job* jobs, int njobs, hipStream_t s) {
    E3_REQUIRE(njobs >= 0 && njobs <= PACK_B16_MAX_JOBS, E3_ERR_INVALID, "too many weight-packing jobs");
    if (njobs == 0) return E3_OK;
    PackMultiArgs a{};
    unsigned nb = 0;
    for (int i = 0; i < njobs; ++i) {
        a.job[i] = jobs[i];
        a.first_block[i] = nb;
        const size_t total = (size_t)jobs[i].T * jobs[i].Cin * jobs[i].Cout;
        size_t b = (total + 2047) / 2048; if (b > 256) b = 256; if (b == 0) b = 1;
        nb += (unsigned)b;
    }
    a.first_block[njobs] = nb; a.njobs = njobs;
    hipLaunchKernelGGL(pack_multi_b16_kernel, dim3(nb), dim3(256), 0, s, a);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_conv_b16(ConvB16Args a, hipStream_t s) {
    E3_REQUIRE(a.Cin % 32 == 0 && a.Cout % 32 == 0, E3_ERR_UNSUPPORTED, "bf16 conv: channel counts must be multiples of 32");
    E3_REQUIRE(a.x_ldc % 8 == 0 && a.y_ldc % 8 == 0 && ((uintptr_t)a.y & 15) == 0 && (!a.y2 || ((uintptr_t)a.y2 & 15) == 0), E3_ERR_INVALID, "bf16 conv: misaligned view");
    E3_REQUIRE((!a.x2 || (a.x_split % 32 == 0 && a.x_split > 0 && a.x_split < a.Cin)) && (!a.y2 || (a.y_split % 32 == 0 && a.y_split > 0 && a.y_split < a.Cout)),
               E3_ERR_INVALID, "bf16 conv: a two-tensor operand splits at a multiple of 32 channels");
    E3_REQUIRE((size_t)a.D * a.H * a.W * a.x_ldc < (1ull << 30), E3_ERR_UNSUPPORTED, "bf16 conv: sample larger than 2 GB");
    const Decomp d = conv_b16_decomp(a.N, a.D, a.H, a.W, a.Cin, a.Cout, a.planar);
    E3_REQUIRE(d.ksplit == 1 || a.partial, E3_ERR_INVALID, "bf16 conv: this shape needs the split-K scratch (conv_b16_partial_floats)");
    const bool two = d.co_t == 2;
    int rc;
#define E3_B16_LAUNCH(KD_, TW_)                                                                                              \
    (d.bd == 4 ? (two ? launch_t<4, 2, KD_, TW_>(a, d.ksplit, s) : launch_t<4, 1, KD_, TW_>(a, d.ksplit, s))                 \
               : (two ? launch_t<2, 2, KD_, TW_>(a, d.ksplit, s) : launch_t<2, 1, KD_, TW_>(a, d.ksplit, s)))
    if (d.persist && !(a.box_hi[0] > 0)) {
        const int tD = cdiv(a.D, 4), tH = cdiv(a.H, 4), tW = cdiv(a.W, 32);
        ConvB16PArgs pa{};
        auto magic = [](unsigned dv) { return dv == 1 ? 0u : (unsigned)(((1ull << 32) + dv - 1) / dv); };
        pa.cgroups = (unsigned)(a.Cout / 32); pa.m_cg = magic(pa.cgroups);
        pa.ca = (unsigned)d.ca; pa.cb = (unsigned)d.cb;
        pa.per = (unsigned)tD << (d.ca + d.cb); pa.m_per = magic(pa.per);
        pa.ncol = (unsigned)(tH >> d.ca) * (unsigned)(tW >> d.cb); pa.m_ncol = magic(pa.ncol);
        pa.tw2 = (unsigned)(tW >> d.cb); pa.m_tw2 = magic(pa.tw2);
        pa.items = (unsigned)((size_t)a.N * tD * tH * tW * pa.cgroups);
        pa.per_xcd = (pa.items + 7) / 8; pa.per_xcd = (pa.per_xcd + pa.cgroups - 1) / pa.cgroups * pa.cgroups;
        using GP = Geo<4, 3, 32, 16>;
        constexpr int lds = 2 * GP::IMG;
        static_assert(GP::IMG >= 4 * 2 * 32 * 33 * 4 + 1024, "statistics scratch fits one image");
        static bool pattr = false;
        if (!pattr) { E3_CHECK_HIP(hipFuncSetAttribute((const void*)conv_b16_pkernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); pattr = true; }
        hipLaunchKernelGGL((conv_b16_pkernel<1>), dim3(PGRID), dim3(256), lds, s, a, pa);      // (conv_b16_decomp has checked that the magic divisions are exact)
        E3_CHECK_HIP(hipGetLastError());
        return E3_OK;
    }
    // 16-channel LDS images, three workgroups per CU, where the shape allows (BD = 4, 3x3x3, 32-voxel rows, one 32-channel output tile, no split-K):
    // 150 -> 133, 247 -> 227, 154 -> 138 us on the level-0 forward convs of cfg 2, 5.89 -> 5.76 ms per step (the 32-channel-image form stays for the other decompositions)
    if (d.bd == 4 && d.tw == 32 && d.ksplit == 1 && !two) rc = a.planar ? launch_t<4, 1, 1, 32, 16>(a, d.ksplit, s) : launch_t<4, 1, 3, 32, 16>(a, d.ksplit, s);
    else if (a.planar) rc = d.tw == 32 ? E3_B16_LAUNCH(1, 32) : E3_B16_LAUNCH(1, 16);
    else rc = d.tw == 32 ? E3_B16_LAUNCH(3, 32) : E3_B16_LAUNCH(3, 16);
#undef E3_B16_LAUNCH
    if (rc || d.ksplit == 1) return rc;
    const size_t vox = (size_t)a.N * a.D * a.H * a.W;
    hipLaunchKernelGGL(splitk_reduce_b16_kernel, dim3(reduce_blocks(vox, a.Cout)), dim3(256), 0, s, a.partial, d.ksplit, vox, a.Cout, a.bias,
                       a.epi_scale, a.epi_shift, a.y, a.y_ldc, a.stats, a.y2, a.y_split);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
