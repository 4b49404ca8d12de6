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
    const int d0 = (td + o_td) * G::DZ, h0 = (th + o_th) * G::BH, w0 = (tw + o_tw) * G::BW;      // (o_*: first brick of the needed region)
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
                    float* prow = a.partial + ((size_t)ksp * vox + (((size_t)n * a.D + d) * a.H + h) * a.W + w) * a.Cout + co0 + ct * 32 + 4 * g;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<f32x4*>(prow + 8 * q) = f32x4{acc[ct][t][4 * q], acc[ct][t][4 * q + 1], acc[ct][t][4 * q + 2], acc[ct][t][4 * q + 3]};
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
                const int cb = co0 + ct * 32 + 8 * q + 4 * g;
                bq[q] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + cb) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (AFF) { sq[q] = *reinterpret_cast<const f32x4*>(a.epi_scale + cb); hq[q] = *reinterpret_cast<const f32x4*>(a.epi_shift + cb); }
                s1[q] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int t = 0; t < G::NV; ++t) {
                const int d = d0 + dT0, h = h0 + G::RPT * (hp0 + t) + r, w = w0 + c;
                const bool valid = FULL || (d < a.D && h < a.H && w < a.W);
                const int cot = co0 + ct * 32;
                bf16_t* yrow = (a.y2 && cot >= a.y_split ? a.y2 + (cot - a.y_split) : a.y + cot) + ((((size_t)n * a.D + d) * a.H + h) * a.W + w) * a.y_ldc + 4 * g;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[ct][t][4 * q], acc[ct][t][4 * q + 1], acc[ct][t][4 * q + 2], acc[ct][t][4 * q + 3]};
                    if (AFF) {
                        v = v * sq[q] + hq[q];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    } else v = v + bq[q];
                    const bf16x4 rb = __builtin_convertvector(v, bf16x4);            // round to nearest even
                    if (valid) *reinterpret_cast<bf16x4*>(yrow + 8 * q) = rb;
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
            const int col = (e & 3) + 8 * (e >> 2) + 4 * g;
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

// torch (Cout, Cin, T) fp32 -> packed bf16 [tap][chunk][k-step][Cg][2][8], Cg = GEMM rows (output channels of THIS launch)
__global__ void pack_conv_b16_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin, int T, int dgrad) {
    const int Kg = dgrad ? Cout : Cin, Cg = dgrad ? Cin : Cout;       // GEMM-K channels, GEMM rows
    const size_t total = (size_t)T * Kg * Cg;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int e = r & 7; r >>= 3; const int g = r & 1; r >>= 1; const int row = r % Cg; r /= Cg;
        const int ks = r & 1; r >>= 1; const int ch = r % (Kg >> 5); const int tap = (int)(r / (Kg >> 5));
        const int k = ch * 32 + ks * 16 + g * 8 + e;
        const float v = dgrad ? w[((size_t)k * Cin + row) * T + (T - 1 - tap)] : w[((size_t)row * Cin + k) * T + tap];
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
            v = dgrad ? J.w[((size_t)k * J.Cin + row) * J.T + (J.T - 1 - tap)] : J.w[((size_t)row * J.Cin + k) * J.T + tap];
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
            o[i] = lo / edge[i]; n[i] = cdiv(hi, edge[i]) - o[i];
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
struct Decomp { int bd, co_t, ksplit, tw; long bricks; };
Decomp conv_b16_decomp(int N, int D, int H, int W, int Cin, int Cout, int planar) {
    static const int forced = getenv("E3_B16_BD") ? atoi(getenv("E3_B16_BD")) : 0;
    static const bool no_split = getenv("E3_B16_NO_SPLITK") != nullptr;
    static const int forced_tw = getenv("E3_B16_TW") ? atoi(getenv("E3_B16_TW")) : 0;
    Decomp d;
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
    return d.ksplit > 1 ? reduce_blocks((size_t)N * D * H * W, Cout) : (int)d.bricks;
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
    E3_REQUIRE(a.x_ldc % 8 == 0 && a.y_ldc % 4 == 0, E3_ERR_INVALID, "bf16 conv: misaligned view");
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
