// nn.ConvTranspose3d(k = s = 2; planar: (1,2,2)) of the decoder (upconv2('transpose'), elektronn3/models/unet.py:152-165) for bf16
// NDHWC tensors: forward (scatter), data gradient (gather) and weight gradient.  The taps do not overlap, so
//   forward   Y^T[(tap, co)][v] = sum_ci  W[ci][co][tap] * X[v][ci]                 written to voxel up(v, tap) = (sd d + kd, 2h + kh, 2w + kw)
//   dgrad     dX^T[ci][v]       = sum_(tap, co) W[ci][co][tap] * dY[up(v, tap)][co]
// are plain GEMMs with a voxel per MFMA column: a wave owns 32 consecutive low-resolution voxels, reads its B fragments (16-byte
// pieces of voxel rows) straight from global memory and the packed weights (A fragments) from L2; the result tile has a voxel per
// lane and 4 consecutive channels per register quad = 8-byte NDHWC stores.  Both are HBM-bound (26 FLOP/B at L0): no LDS tiling.
//   wgrad     dW[ci][co][tap]   = sum_v X[v][ci] * dY[up(v, tap)][co]   contracts over voxels: [voxel][32 ch] LDS images read with
//             ds_read_b64_tr_b16 (see bf16_wgrad.hip); the dY image is staged de-interleaved (even / odd w in separate images: the
//             DMA's source address is free) so that the 16 voxels of a k-step are contiguous 64-byte rows.
#include "bf16.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
// (a plain function: called straight from a kernel TEMPLATE, hipcc's host pass drops the kernel's stub)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, lds_ptr_t dst, int, unsigned voff, int, int, int) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, 0, 0, 0);
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr unsigned OOB = 0x80000000u;

// packed weights: [row tile][k-step][2][32 rows][8] bf16 (k-group major: a wave's fragment is 1 KB, read lane-linearly from LDS or L2)
//   forward: rows = (co tile, tap, 32 channels), K = ci;   dgrad: rows = ci, K = (tap, co)
__global__ void pack_upconv_b16_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cin, int Cout, int T, int dgrad) {
    const int rows = dgrad ? Cin : T * Cout, K = dgrad ? T * Cout : Cin;
    const size_t total = (size_t)rows * K;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int e = r & 7; r >>= 3;
        const int rr = r & 31; r >>= 5; const int g = r & 1; r >>= 1;
        const int ks = r % (K >> 4); const int rt = (int)(r / (K >> 4));
        const int row = rt * 32 + rr, k = ks * 16 + g * 8 + e;
        int ci, co, tap;
        if (dgrad) { ci = row; tap = k / Cout; co = k % Cout; } else { tap = rt % T; co = (rt / T) * 32 + rr; ci = k; }
        out[i] = f2bf(w[((size_t)ci * Cout + co) * T + tap]);
    }
}

// GATHER = false: forward; true: data gradient.  One wave = NVT 32-voxel tiles of the LOW-resolution grid (each weight fragment, read
// from L2, feeds NVT MFMAs), RT row tiles per pass.  Results leave through a per-wave LDS tile so that 4 consecutive lanes write the
// 64 contiguous bytes of a voxel's 32 channels (16-byte stores) instead of every lane scattering 8 bytes into its own row.
template <bool GATHER, int RT, int NVT>
__global__ __launch_bounds__(256) void upconv_b16_kernel(const UpconvB16Args a, size_t nvox) {
    __shared__ float S[4][2][33];                                   // statistics exchange (forward only)
    __shared__ __attribute__((aligned(16))) unsigned char xp[4][32 * 80];   // per-wave transposition tile: [32 voxels][64 B + 16 pad]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, g = lane >> 5;
    const int T = a.sd * 4;
    const int K = GATHER ? T * a.Cout : a.Cin;            // GEMM-K
    const int rows = GATHER ? a.Cin : T * a.Cout;
    const int nks = K >> 4, nrt = rows >> 5;
    // per tile: this lane's voxel, the element offset of its output voxel for tap 0 and the taps whose output lies inside (Do, Ho, Wo)
    bool vin[NVT]; unsigned xoff[NVT], obase[NVT], okm[NVT];
#pragma unroll
    for (int t = 0; t < NVT; ++t) {
        const size_t v = (((size_t)blockIdx.x * 4 + wave) * NVT + t) * 32 + j;
        vin[t] = v < nvox;
        size_t r = vin[t] ? v : 0;
        const int w = r % a.W; r /= a.W; const int h = r % a.H; r /= a.H; const int d = r % a.D; const int n = (int)(r / a.D);
        xoff[t] = (unsigned)((vin[t] ? v : 0) * a.x_ldc);
        obase[t] = (unsigned)(((((size_t)n * a.Do + a.sd * d) * a.Ho + 2 * h) * a.Wo + 2 * w) * a.y_ldc);
        unsigned m = 0;
        for (int tap = 0; tap < T; ++tap) {
            const int od = a.sd * d + (a.sd == 2 ? (tap >> 2) : 0), oh = 2 * h + ((tap >> 1) & 1), ow = 2 * w + (tap & 1);
            if (vin[t] && od < a.Do && oh < a.Ho && ow < a.Wo) m |= 1u << tap;
        }
        okm[t] = m;
    }
    auto tapoff = [&](int tap) -> unsigned {              // element offset of tap's output voxel relative to tap 0's
        return (unsigned)((((a.sd == 2 ? (tap >> 2) : 0) * a.Ho + ((tap >> 1) & 1)) * a.Wo + (tap & 1)) * a.y_ldc);
    };
    float ssum[16], ssq[16], cntl = 0.f;                  // forward statistics of the current co tile, summed over its taps and tiles

    // blockIdx.y splits the row tiles (forward: one co tile = T row tiles per y; dgrad: RT ci tiles per y): more workgroups for the
    // low-resolution layers, whose few voxel tiles cannot fill the chip
    const int rt_lo = GATHER ? blockIdx.y * RT : blockIdx.y * T, rt_hi = GATHER ? rt_lo + RT : rt_lo + T;
    for (int rt0 = rt_lo; rt0 < rt_hi && rt0 < nrt; rt0 += RT) {
        f32x16 acc[RT][NVT];
#pragma unroll
        for (int q = 0; q < RT; ++q)
#pragma unroll
            for (int t = 0; t < NVT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[q][t][e] = 0.f;
        // K loop, two k-steps per pass, the operands of the next pass requested before the MFMAs of this one (both operands come
        // straight from global memory / L2: unpipelined, every k-step waited a full round trip -- 108 us for a 2 GFLOP layer)
        constexpr int U = 2;
        bf16x8 b[2][U][NVT], af[2][U][RT];
        auto fetch = [&](int ks0, bf16x8 (&bb)[U][NVT], bf16x8 (&aa)[U][RT]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ks = ks0 + u < nks ? ks0 + u : nks - 1;
#pragma unroll
                for (int t = 0; t < NVT; ++t) {
                    if constexpr (!GATHER) {
                        bb[u][t] = vin[t] ? *reinterpret_cast<const bf16x8*>(a.x + xoff[t] + ks * 16 + g * 8) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    } else {
                        const int kk = ks * 16, tap = kk / a.Cout, co = kk % a.Cout;
                        bb[u][t] = ((okm[t] >> tap) & 1u) ? *reinterpret_cast<const bf16x8*>(a.y + obase[t] + tapoff(tap) + co + g * 8) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    }
                }
#pragma unroll
                for (int q = 0; q < RT; ++q) {
                    const int rt = rt0 + q < nrt ? rt0 + q : nrt - 1;
                    aa[u][q] = *reinterpret_cast<const bf16x8*>(a.wt + ((size_t)rt * nks + ks) * 512 + g * 256 + j * 8);
                }
            }
        };
        fetch(0, b[0], af[0]);
        for (int ks0 = 0, par = 0; ks0 < nks; ks0 += U, par ^= 1) {
            if (par == 0) {
                if (ks0 + U < nks) fetch(ks0 + U, b[1], af[1]);
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (ks0 + u < nks)
#pragma unroll
                        for (int q = 0; q < RT; ++q)
#pragma unroll
                            for (int t = 0; t < NVT; ++t) acc[q][t] = E3_MFMA16(af[0][u][q], b[0][u][t], acc[q][t], 0, 0, 0);
            } else {
                if (ks0 + U < nks) fetch(ks0 + U, b[0], af[0]);
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (ks0 + u < nks)
#pragma unroll
                        for (int q = 0; q < RT; ++q)
#pragma unroll
                            for (int t = 0; t < NVT; ++t) acc[q][t] = E3_MFMA16(af[1][u][q], b[1][u][t], acc[q][t], 0, 0, 0);
            }
        }
        // ---- epilogue of this group of row tiles
#pragma unroll
        for (int q = 0; q < RT; ++q) {
            const int rt = rt0 + q;
            if (rt >= nrt) continue;
            const int tap = GATHER ? 0 : rt % T, cb = GATHER ? rt * 32 : (rt / T) * 32;       // forward: tap `rt % T` of the channels [cb, cb + 32)
            if (!GATHER && tap == 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
                cntl = 0.f;
            }
#pragma unroll
            for (int t = 0; t < NVT; ++t) {
                const bool ok = GATHER ? vin[t] : ((okm[t] >> tap) & 1u) != 0;
                if (!GATHER) cntl += ok ? 1.f : 0.f;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int c0 = cb + 8 * qq + 4 * g;
                    u16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float val = acc[q][t][4 * qq + e];
                        float bias = 0.f;
                        if constexpr (!GATHER) {
                            bias = a.bias ? a.bias[c0 + e] : 0.f;
                            if (a.epi_scale) val = fmaxf(__builtin_fmaf(val, a.epi_scale[c0 + e], a.epi_shift[c0 + e]), 0.f);
                            else val += bias;
                        }
                        const bf16_t rb = f2bf(val);
                        o[e] = rb;
                        if constexpr (!GATHER) {
                            const float dv = ok ? bf2f(rb) - bias : 0.f;
                            ssum[4 * qq + e] += dv; ssq[4 * qq + e] = __builtin_fmaf(dv, dv, ssq[4 * qq + e]);
                        }
                    }
                    *reinterpret_cast<u16x4*>(xp[wave] + j * 80 + (8 * qq + 4 * g) * 2) = o;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int jj = p * 16 + (lane >> 2), piece = lane & 3;
                    const u16x8 row = *reinterpret_cast<const u16x8*>(xp[wave] + jj * 80 + piece * 16);
                    if constexpr (GATHER) {
                        const unsigned xo = __shfl(xoff[t], jj);
                        const int vj = __shfl((int)vin[t], jj);
                        if (vj) *reinterpret_cast<u16x8*>(const_cast<bf16_t*>(a.x) + xo + cb + piece * 8) = row;
                    } else {
                        const unsigned ob = __shfl(obase[t], jj), om = __shfl(okm[t], jj);
                        if ((om >> tap) & 1u) *reinterpret_cast<u16x8*>(a.y + ob + tapoff(tap) + cb + piece * 8) = row;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            if constexpr (!GATHER) {
                if (a.stats && tap == T - 1) {
                    // one record per (workgroup, channel): count, mean, M2 over the workgroup's (voxel, tap) outputs
                    float cw = cntl;
#pragma unroll
                    for (int off = 16; off >= 1; off >>= 1) cw += __shfl_xor(cw, off);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
#pragma unroll
                        for (int off = 16; off >= 1; off >>= 1) { ssum[e] += __shfl_xor(ssum[e], off); ssq[e] += __shfl_xor(ssq[e], off); }
                    }
                    __syncthreads();
                    if (j == 0) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int col = (e & 3) + 8 * (e >> 2) + 4 * g;
                            S[wave][0][col] = ssum[e]; S[wave][1][col] = ssq[e];
                        }
                        if (g == 0) S[wave][0][32] = cw;
                    }
                    __syncthreads();
                    if (tid < 32) {
                        const float s = (S[0][0][tid] + S[1][0][tid]) + (S[2][0][tid] + S[3][0][tid]);
                        const float q2 = (S[0][1][tid] + S[1][1][tid]) + (S[2][1][tid] + S[3][1][tid]);
                        const float cnt = (S[0][0][32] + S[1][0][32]) + (S[2][0][32] + S[3][0][32]);
                        const int co = cb + tid;
                        const float bias = a.bias ? a.bias[co] : 0.f;
                        const float m = cnt > 0.f ? s / cnt : 0.f;
                        float* rec = a.stats + ((size_t)blockIdx.x * a.Cout + co) * 3;
                        rec[0] = cnt; rec[1] = bias + m; rec[2] = fmaxf(q2 - s * m, 0.f);
                    }
                }
            }
        }
    }
}

// Forward, the HBM-bound case (Cin = 64 or 128): persistent workgroups.  The weights of one 32-channel output tile (all taps, all of
// K: T x Cin x 32 bf16 = 32 / 64 KB) are staged into LDS ONCE per workgroup; the workgroup then walks 128-voxel tiles of the
// low-resolution grid, each staged by LDS-DMA (whole 128-byte row segments, double-buffered: the next tile's loads are in flight
// during the current tile's MFMAs and stores) and read back as conflict-free ds_read_b128 fragments (16-byte pieces of a row
// XOR-swizzled by bits 1..3 of the row).  8 accumulator tiles per wave (one per tap); stores leave through the transposition tile.
template <int CH, int SD>      // 64-channel chunks of K; depth stride (1: planar block / 2D network, taps (kh, kw) only)
__global__ __launch_bounds__(256, 2) void upconv_fwd_b16_kernel(const UpconvB16Args a, size_t nvox, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = 4 * SD;
    constexpr int WB = T * CH * 4 * 1024;                 // weights [tap][chunk][k-step][1 KB]
    constexpr int XB = CH * 128 * 128;                    // one X buffer [chunk][128 rows][128 B]
    unsigned char* xs = smem + WB;
    unsigned char* xp = smem + WB + 2 * XB;               // [4 waves][32 * 80]
    float* S = reinterpret_cast<float*>(xp + 4 * 32 * 80);   // [4][2][33]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, g = lane >> 5;
    const int cot = blockIdx.y, cb = cot * 32;
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.wt), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, 0x7fffffff, 0x00020000);
    // weights: packed [rt = cot*T + tap][ks over Cin/16][1 KB]; here [tap][chunk][ks] = the same order (ks = chunk*4 + ks4)
    for (int p = wave; p < T * CH * 4; p += 4)
        dma16(w_rs, (lds_ptr_t)(smem + p * 1024), 16, (unsigned)(((size_t)cot * T * CH * 4 + p) * 1024 + lane * 16), 0, 0, 0);
    auto stage = [&](int tile, int buf) {
        // piece index within the tile: [chunk][row][8 pieces]; 64 lanes = 8 rows
        for (int p = wave; p < CH * 16; p += 4) {
            const int c = p >> 4, row = ((p & 15) << 3) + (lane >> 3), pp = lane & 7;
            const int sp = pp ^ ((row >> 1) & 7);
            const size_t v = (size_t)tile * 128 + row;
            dma16(x_rs, (lds_ptr_t)(xs + buf * XB + p * 1024), 16, v < nvox ? (unsigned)(v * a.x_ldc * 2 + c * 128 + sp * 16) : OOB, 0, 0, 0);
        }
    };
    float ssum[16], ssq[16], cntl = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
    f32x4 bq[4], sq[4], hq[4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        const int c0 = cb + 8 * qq + 4 * g;
        bq[qq] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.epi_scale) { sq[qq] = *reinterpret_cast<const f32x4*>(a.epi_scale + c0); hq[qq] = *reinterpret_cast<const f32x4*>(a.epi_shift + c0); }
    }
    const int row = wave * 32 + j;
    const unsigned xrd = (unsigned)(row * 128), xsw = (unsigned)((row >> 1) & 7);
    const unsigned wrd = (unsigned)(g * 512 + j * 16);

    int it = 0;
    if ((int)blockIdx.x < ntiles) stage(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) stage(tile + gridDim.x, buf ^ 1);
        // this lane's voxel
        const size_t v = (size_t)tile * 128 + row;
        const bool vin = v < nvox;
        size_t r = vin ? v : 0;
        const int w = r % a.W; r /= a.W; const int h = r % a.H; r /= a.H; const int d = r % a.D; const int n = (int)(r / a.D);
        const unsigned obase = (unsigned)(((((size_t)n * a.Do + SD * d) * a.Ho + 2 * h) * a.Wo + 2 * w) * a.y_ldc);
        unsigned okm = 0;
#pragma unroll
        for (int tap = 0; tap < T; ++tap)
            if (vin && SD * d + (tap >> 2) < a.Do && 2 * h + ((tap >> 1) & 1) < a.Ho && 2 * w + (tap & 1) < a.Wo) okm |= 1u << tap;
        f32x16 acc[T];
#pragma unroll
        for (int tap = 0; tap < T; ++tap)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[tap][e] = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(xs + buf * XB + c * 16384 + xrd + (((2 * ks + g) ^ xsw) << 4));
#pragma unroll
                for (int tap = 0; tap < T; ++tap) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(smem + ((tap * CH + c) * 4 + ks) * 1024 + wrd);
                    acc[tap] = E3_MFMA16(af, b, acc[tap], 0, 0, 0);
                }
            }
        // epilogue: bias / folded BN, rounding, statistics, transposed 16-byte stores
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
            const bool ok = ((okm >> tap) & 1u) != 0;
            cntl += ok ? 1.f : 0.f;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                u16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float val = acc[tap][4 * qq + e];
                    if (a.epi_scale) val = fmaxf(__builtin_fmaf(val, sq[qq][e], hq[qq][e]), 0.f);
                    else val += bq[qq][e];
                    const bf16_t rb = f2bf(val);
                    o[e] = rb;
                    const float dv = ok ? bf2f(rb) - bq[qq][e] : 0.f;
                    ssum[4 * qq + e] += dv; ssq[4 * qq + e] = __builtin_fmaf(dv, dv, ssq[4 * qq + e]);
                }
                *reinterpret_cast<u16x4*>(xp + wave * (32 * 80) + j * 80 + (8 * qq + 4 * g) * 2) = o;
            }
            __builtin_amdgcn_wave_barrier();
            const unsigned toff = (unsigned)((((tap >> 2) * a.Ho + ((tap >> 1) & 1)) * a.Wo + (tap & 1)) * a.y_ldc);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int jj = p * 16 + (lane >> 2), piece = lane & 3;
                const u16x8 rowv = *reinterpret_cast<const u16x8*>(xp + wave * (32 * 80) + jj * 80 + piece * 16);
                const unsigned ob = __shfl(obase, jj), om = __shfl(okm, jj);
                if ((om >> tap) & 1u) *reinterpret_cast<u16x8*>(a.y + ob + toff + cb + piece * 8) = rowv;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (!a.stats) return;
    // one record per (workgroup, channel): count, mean, M2 over all (voxel, tap) outputs this workgroup produced
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) cntl += __shfl_xor(cntl, off);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) { ssum[e] += __shfl_xor(ssum[e], off); ssq[e] += __shfl_xor(ssq[e], off); }
    }
    if (j == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int col = (e & 3) + 8 * (e >> 2) + 4 * g;
            S[(wave * 2 + 0) * 33 + col] = ssum[e]; S[(wave * 2 + 1) * 33 + col] = ssq[e];
        }
        if (g == 0) S[(wave * 2 + 0) * 33 + 32] = cntl;
    }
    __syncthreads();
    if (tid < 32) {
        const float s = (S[0 * 33 + tid] + S[2 * 33 + tid]) + (S[4 * 33 + tid] + S[6 * 33 + tid]);
        const float q2 = (S[1 * 33 + tid] + S[3 * 33 + tid]) + (S[5 * 33 + tid] + S[7 * 33 + tid]);
        const float cnt = (S[0 * 33 + 32] + S[2 * 33 + 32]) + (S[4 * 33 + 32] + S[6 * 33 + 32]);
        const int co = cb + tid;
        const float bias = a.bias ? a.bias[co] : 0.f;
        const float m = cnt > 0.f ? s / cnt : 0.f;
        float* rec = a.stats + ((size_t)blockIdx.x * a.Cout + co) * 3;
        rec[0] = cnt; rec[1] = bias + m; rec[2] = fmaxf(q2 - s * m, 0.f);
    }
}

// Data gradient, the HBM-bound case (Cin = 64, Cout = 32: the full-resolution transposed conv of a start_filts = 32 network):
// persistent workgroups, the whole weight matrix (64 x 256 bf16 = 32 KB) in LDS, 64-voxel tiles of the LOW-resolution grid whose
// 8 x 64 gathered dY rows (64 B each) are staged by LDS-DMA (a lane always fetches the same voxel, one tap per instruction),
// double-buffered.  Wave (vt, rt) owns voxel tile vt and ci tile rt: 16 MFMAs per tile; 16-byte stores through the transposition tile.
template <int SD>
__global__ __launch_bounds__(256, 1) void upconv_dgrad_b16_kernel(const UpconvB16Args a, size_t nvox, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int T = 4 * SD, WB = 2 * 2 * T * 1024, YB = T * 64 * 64;        // weights [rt][ks][1 KB]; one dY buffer [tap][64 voxels][64 B]
    unsigned char* ys = smem + WB;
    unsigned char* xp = smem + WB + 2 * YB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, g = lane >> 5;
    const int vt = wave & 1, rt = wave >> 1;
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.wt), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, 0x7fffffff, 0x00020000);
    for (int p = wave; p < 4 * T; p += 4) dma16(w_rs, (lds_ptr_t)(smem + p * 1024), 16, (unsigned)(p * 1024 + lane * 16), 0, 0, 0);
    // staging: instruction k of a wave = tap k of the 16 voxels [16 wave, 16 wave + 16), lane = (voxel, 16-byte piece)
    const int sv = 16 * wave + (lane >> 2), sp = (lane & 3) ^ ((sv >> 2) & 3);
    auto stage = [&](int tile, int buf) {
        const size_t v = (size_t)tile * 64 + sv;
        const bool vin = v < nvox;
        size_t r = vin ? v : 0;
        const int w = r % a.W; r /= a.W; const int h = r % a.H; r /= a.H; const int d = r % a.D; const int n = (int)(r / a.D);
        const unsigned ob = (unsigned)((((((size_t)n * a.Do + SD * d) * a.Ho + 2 * h) * a.Wo + 2 * w) * a.y_ldc) * 2 + sp * 16);
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
            const bool ok = vin && SD * d + (tap >> 2) < a.Do && 2 * h + ((tap >> 1) & 1) < a.Ho && 2 * w + (tap & 1) < a.Wo;
            const unsigned toff = (unsigned)(((((tap >> 2) * a.Ho + ((tap >> 1) & 1)) * a.Wo + (tap & 1)) * a.y_ldc) * 2);
            dma16(y_rs, (lds_ptr_t)(ys + buf * YB + (tap * 64 + 16 * wave) * 64), 16, ok ? ob + toff : OOB, 0, 0, 0);
        }
    };
    const int row = vt * 32 + j;
    const unsigned yrd = (unsigned)(row * 64), ysw = (unsigned)((row >> 2) & 3);
    const unsigned wrd = (unsigned)(rt * 2 * T * 1024 + g * 512 + j * 16);
    int it = 0;
    if ((int)blockIdx.x < ntiles) stage(blockIdx.x, 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) stage(tile + gridDim.x, buf ^ 1);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2 * T; ++ks) {        // k = (tap = ks / 2, 16 channels of dY)
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(ys + buf * YB + (ks >> 1) * 4096 + yrd + (((2 * (ks & 1) + g) ^ ysw) << 4));
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(smem + ks * 1024 + wrd);
            acc = E3_MFMA16(af, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            u16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[4 * qq + e]);
            *reinterpret_cast<u16x4*>(xp + wave * (32 * 80) + j * 80 + (8 * qq + 4 * g) * 2) = o;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int jj = p * 16 + (lane >> 2), piece = lane & 3;
            const u16x8 rowv = *reinterpret_cast<const u16x8*>(xp + wave * (32 * 80) + jj * 80 + piece * 16);
            const size_t v = (size_t)tile * 64 + vt * 32 + jj;
            if (v < nvox) *reinterpret_cast<u16x8*>(const_cast<bf16_t*>(a.x) + v * a.x_ldc + rt * 32 + piece * 8) = rowv;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__device__ __forceinline__ bf16x8 tr_frag(unsigned addr, unsigned row_pitch = 64) {      // k-rows 0..3 at addr, 4..7 four rows further
    typedef s16x4 __attribute__((address_space(3))) * lp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)addr);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(addr + 4 * row_pitch));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// weight gradient.  Workgroup = (CIT x 32 ci) x 32 co, all taps (wave w owns taps {w, w + 4}; planar: tap w), over a contiguous
// range of low-resolution bricks of 4 (h) x 16 (w) voxels of one d-slice.  LDS: X image [64 voxels][CIT][64 B], dY images
// [tap][64 voxels][64 B] (tap = (kd, kh, kw): the row (sd d + kd, 2h + kh), its even / odd w).
template <int CIT>
__global__ __launch_bounds__(256, 3) void upconv_wgrad_b16_kernel(const bf16_t* __restrict__ x, int x_ldc, int Cin, const bf16_t* __restrict__ dy,
                                                                  int dy_ldc, int Cout, float* __restrict__ part, int N, int D, int H, int W,
                                                                  int Do, int Ho, int Wo, int sd, int bricks_per_split, int co_tiles, int ci_groups) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XIMG = 64 * CIT * 64;
    const int T = sd * 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int cig = L % ci_groups; L /= ci_groups;
    const int co_t = L % co_tiles; const int split = L / co_tiles;
    const int ci0 = cig * 32 * CIT, co0 = co_t * 32;
    const int tH = (H + 3) / 4, tW = (W + 15) / 16;
    const int nbricks = N * D * tH * tW;
    const int brick0 = split * bricks_per_split;
    const int brick1 = brick0 + bricks_per_split < nbricks ? brick0 + bricks_per_split : nbricks;
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dy), 0, 0x7fffffff, 0x00020000);

    const int G = lane >> 4, t = lane & 15;
    const int krow = 8 * (G >> 1) + (t >> 2), chb = (16 * (G & 1) + 4 * (t & 3)) * 2;
    f32x16 acc[2][CIT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < CIT; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][c][e] = 0.f;

    for (int brick = brick0; brick < brick1; ++brick) {
        int Lt = brick;
        const int tw_ = Lt % tW; Lt /= tW; const int th_ = Lt % tH; Lt /= tH; const int d = Lt % D; const int n = Lt / D;
        const int h0 = th_ * 4, w0 = tw_ * 16;
        // X: 64 voxels x CIT x 4 pieces
        for (int p = tid; p < 64 * CIT * 4; p += 256) {
            const int q = p & 3, c = (p >> 2) % CIT, vv = p / (4 * CIT);
            const int hh = vv >> 4, ww = vv & 15;
            const bool ok = h0 + hh < H && w0 + ww < W;
            const size_t off = ((((size_t)n * D + d) * H + h0 + hh) * W + w0 + ww) * x_ldc + ci0 + c * 32;
            dma16(x_rs, (lds_ptr_t)(smem + (p - lane) * 16), 16, ok ? (unsigned)(off * 2 + q * 16) : OOB, 0, 0, 0);
        }
        // dY: T images of 64 voxels x 4 pieces
        for (int p = tid; p < T * 64 * 4; p += 256) {
            const int q = p & 3, vv = (p >> 2) & 63, tap = p >> 8;
            const int hh = vv >> 4, ww = vv & 15;
            const int od = sd * d + (sd == 2 ? (tap >> 2) : 0), oh = 2 * (h0 + hh) + ((tap >> 1) & 1), ow = 2 * (w0 + ww) + (tap & 1);
            const bool ok = h0 + hh < H && w0 + ww < W && od < Do && oh < Ho && ow < Wo;
            const size_t off = ((((size_t)n * Do + od) * Ho + oh) * Wo + ow) * dy_ldc + co0;
            dma16(g_rs, (lds_ptr_t)(smem + XIMG + (p - lane) * 16), 16, ok ? (unsigned)(off * 2 + q * 16) : OOB, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; ++s) {            // k-step = one w-row of 16 voxels
            bf16x8 af[CIT];
#pragma unroll
            for (int c = 0; c < CIT; ++c) af[c] = tr_frag((unsigned)(((s * 16 + krow) * CIT + c) * 64 + chb), CIT * 64);
            // NOTE: with CIT = 2 the X rows are 128 B apart (4 k-rows: 2-way bank conflict on this fragment only)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int tap = wave + 4 * i;
                if (tap < T) {
                    const bf16x8 bfr = tr_frag((unsigned)(XIMG + ((tap * 64 + s * 16 + krow) * 64) + chb));
#pragma unroll
                    for (int c = 0; c < CIT; ++c) acc[i][c] = E3_MFMA16(af[c], bfr, acc[i][c], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // slab part[split][tap][CiPad][CoPad]: rows = ci, columns = co
    const int CiPad = ci_groups * 32 * CIT, CoPad = co_tiles * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int tap = wave + 4 * i;
        if (tap >= T) continue;
#pragma unroll
        for (int c = 0; c < CIT; ++c) {
            float* dst = part + (((size_t)split * T + tap) * CiPad + ci0 + c * 32) * CoPad + co0 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) dst[(size_t)((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * CoPad] = acc[i][c][e];
        }
    }
}

}  // namespace

constexpr int UP_NVT = 1;     // 32-voxel tiles per wave (generic kernel)
static bool upconv_fwd_persistent(int Cin, int sd) {
    static const bool off = getenv("E3_B16_UP_GENERIC") != nullptr;      // A/B switch
    return !off && (Cin == 64 || Cin == 128);
}
int upconv_b16_stats_parts(int N, int D, int H, int W, int sd, int Cin) {      // one record per workgroup
    const int tiles = (int)(((size_t)N * D * H * W + 128 * UP_NVT - 1) / (128 * UP_NVT));
    return upconv_fwd_persistent(Cin, sd) ? (tiles < 512 ? tiles : 512) : tiles;
}

size_t upconv_b16_packed_elems(int Cin, int Cout, int sd) { return (size_t)sd * 4 * Cin * Cout; }

int launch_pack_upconv_b16(const float* w, bf16_t* out, int Cin, int Cout, int sd, int dgrad, hipStream_t s) {
    const int T = sd * 4;
    const size_t total = (size_t)T * Cin * Cout;
    const unsigned grid = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_upconv_b16_kernel, dim3(grid), dim3(256), 0, s, w, out, Cin, Cout, T, dgrad);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_upconv_b16_fwd(UpconvB16Args a, hipStream_t s) {
    E3_REQUIRE(a.Cin % 32 == 0 && a.Cout % 32 == 0, E3_ERR_UNSUPPORTED, "bf16 transposed conv: channel counts must be multiples of 32");
    E3_REQUIRE(a.x_ldc % 8 == 0 && a.y_ldc % 4 == 0, E3_ERR_INVALID, "bf16 transposed conv: misaligned view");
    const size_t nvox = (size_t)a.N * a.D * a.H * a.W;
    E3_REQUIRE((size_t)a.N * a.Do * a.Ho * a.Wo * a.y_ldc < (1ull << 32) && nvox * a.x_ldc < (1ull << 31), E3_ERR_UNSUPPORTED, "bf16 transposed conv: tensor too large for 32-bit element offsets");
    if (upconv_fwd_persistent(a.Cin, a.sd)) {
        const int tiles = (int)((nvox + 127) / 128);
        const int gx = tiles < 512 ? tiles : 512;
        const int CH = a.Cin / 64, T = 4 * a.sd;
        const int lds = T * CH * 4 * 1024 + 2 * CH * 16384 + 4 * 32 * 80 + 4 * 2 * 33 * 4;
        static bool done = false;
        if (!done) {
            const auto cap = hipFuncAttributeMaxDynamicSharedMemorySize;
            (void)hipFuncSetAttribute((const void*)upconv_fwd_b16_kernel<1, 1>, cap, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)upconv_fwd_b16_kernel<1, 2>, cap, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)upconv_fwd_b16_kernel<2, 1>, cap, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)upconv_fwd_b16_kernel<2, 2>, cap, 160 * 1024);
            done = true;
        }
        auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(gx, a.Cout / 32), dim3(256), lds, s, a, nvox, tiles); };
        if (CH == 1) { if (a.sd == 2) go(upconv_fwd_b16_kernel<1, 2>); else go(upconv_fwd_b16_kernel<1, 1>); }
        else         { if (a.sd == 2) go(upconv_fwd_b16_kernel<2, 2>); else go(upconv_fwd_b16_kernel<2, 1>); }
        E3_CHECK_HIP(hipGetLastError());
        return E3_OK;
    }
    hipLaunchKernelGGL((upconv_b16_kernel<false, 4, UP_NVT>), dim3((unsigned)((nvox + 128 * UP_NVT - 1) / (128 * UP_NVT)), a.Cout / 32), dim3(256), 0, s, a, nvox);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_upconv_b16_dgrad(UpconvB16Args a, hipStream_t s) {
    E3_REQUIRE(a.Cin % 32 == 0 && a.Cout % 32 == 0, E3_ERR_UNSUPPORTED, "bf16 transposed conv: channel counts must be multiples of 32");
    E3_REQUIRE(a.y_ldc % 8 == 0 && a.x_ldc % 4 == 0, E3_ERR_INVALID, "bf16 transposed conv: misaligned view");
    const size_t nvox = (size_t)a.N * a.D * a.H * a.W;
    E3_REQUIRE((size_t)a.N * a.Do * a.Ho * a.Wo * a.y_ldc < (1ull << 32) && nvox * a.x_ldc < (1ull << 32), E3_ERR_UNSUPPORTED, "bf16 transposed conv: tensor too large for 32-bit element offsets");
    static const bool generic = getenv("E3_B16_UP_GENERIC") != nullptr;
    if (!generic && a.Cin == 64 && a.Cout == 32 && (size_t)a.N * a.Do * a.Ho * a.Wo * a.y_ldc < (1ull << 30)) {
        const int tiles = (int)((nvox + 63) / 64);
        const int gx = tiles < 256 ? tiles : 256;
        const int T = 4 * a.sd;
        const int lds = 4 * T * 1024 + 2 * T * 64 * 64 + 4 * 32 * 80;
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute((const void*)upconv_dgrad_b16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)upconv_dgrad_b16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            done = true;
        }
        auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(gx), dim3(256), lds, s, a, nvox, tiles); };
        if (a.sd == 2) go(upconv_dgrad_b16_kernel<2>); else go(upconv_dgrad_b16_kernel<1>);
        E3_CHECK_HIP(hipGetLastError());
        return E3_OK;
    }
    const unsigned gx = (unsigned)((nvox + 128 * UP_NVT - 1) / (128 * UP_NVT));
    // one ci tile per workgroup where four would leave most of the chip idle (lowest level of cfg 2: 32 voxel tiles x 2 -> x 8 workgroups)
    if ((size_t)gx * cdiv(a.Cin / 32, 4) < 192) hipLaunchKernelGGL((upconv_b16_kernel<true, 1, UP_NVT>), dim3(gx, a.Cin / 32), dim3(256), 0, s, a, nvox);
    else hipLaunchKernelGGL((upconv_b16_kernel<true, 4, UP_NVT>), dim3(gx, cdiv(a.Cin / 32, 4)), dim3(256), 0, s, a, nvox);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int upconv_b16_wgrad_splits(int N, int D, int H, int W) {
    const int nbricks = N * D * cdiv(H, 4) * cdiv(W, 16);
    int splits = nbricks < 1024 ? nbricks : 1024;
    const int per = cdiv(nbricks, splits);
    return cdiv(nbricks, per);
}

int launch_upconv_b16_wgrad(const bf16_t* x, int x_ldc, int Cin, const bf16_t* dy, int dy_ldc, int Cout, float* part,
                            int N, int D, int H, int W, int Do, int Ho, int Wo, int sd, int splits, hipStream_t s) {
    E3_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, E3_ERR_UNSUPPORTED, "bf16 transposed conv: channel counts must be multiples of 32");
    E3_REQUIRE((size_t)N * D * H * W * x_ldc < (1ull << 30) && (size_t)N * Do * Ho * Wo * dy_ldc < (1ull << 30), E3_ERR_UNSUPPORTED,
               "bf16 transposed-conv wgrad: tensor larger than 2 GB");
    const int nbricks = N * D * cdiv(H, 4) * cdiv(W, 16);
    const int per = cdiv(nbricks, splits);
    const int co_tiles = Cout / 32;
    const int T = sd * 4;
    if (Cin % 64 == 0) {
        const int cig = Cin / 64;
        const int lds = 64 * 2 * 64 + T * 64 * 64;
        static bool done = false;
        if (!done) { (void)hipFuncSetAttribute((const void*)upconv_wgrad_b16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 2 * 64 + 8 * 64 * 64); done = true; }
        hipLaunchKernelGGL(upconv_wgrad_b16_kernel<2>, dim3((unsigned)(splits * co_tiles * cig)), dim3(256), lds, s, x, x_ldc, Cin, dy, dy_ldc, Cout, part,
                           N, D, H, W, Do, Ho, Wo, sd, per, co_tiles, cig);
    } else {
        const int cig = Cin / 32;
        const int lds = 64 * 64 + T * 64 * 64;
        static bool done = false;
        if (!done) { (void)hipFuncSetAttribute((const void*)upconv_wgrad_b16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 64 + 8 * 64 * 64); done = true; }
        hipLaunchKernelGGL(upconv_wgrad_b16_kernel<1>, dim3((unsigned)(splits * co_tiles * cig)), dim3(256), lds, s, x, x_ldc, Cin, dy, dy_ldc, Cout, part,
                           N, D, H, W, Do, Ho, Wo, sd, per, co_tiles, cig);
    }
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
