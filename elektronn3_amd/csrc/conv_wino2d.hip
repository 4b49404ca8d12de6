// Planar 1x3x3 stride-1 convolution as Winograd F(2x2, 3x3) on the fp32 matrix cores (forward and dgrad).
//
// Replaces torch.nn.Conv3d(kernel_size=(1,3,3), padding=(0,1,1)) of the planar blocks (unet.py:114-128, 138-141) wherever
// the grid fills the chip: 16 multiplies per 2x2 output tile and (ci, co) pair instead of 36 (2.25x fewer matrix FLOPs),
// same fp32 arithmetic and the same transform matrices as the 3D kernel (conv_wino.hip), one dimension less.
//
// Work decomposition: a brick of 4x8 tiles (8x16 voxels of one d-slice, 10x18 halo) x 64 output channels.  The 16 positions
// (ph, pw) are 16 independent GEMMs  M = 32 tiles, N = 64, K = Cin;  wave w owns the 4 positions with ph = w:
// 4 x 2 accumulator tiles = 128 registers, so TWO workgroups share a CU and cover each other's staging, barriers and
// epilogue (the 3D kernel needs all 512 registers of a SIMD lane for one wave).  Per 8-channel chunk the raw halo is staged
// in LDS (parity-split + XOR-swizzled: conflict-free ds_read_b128), lane (tile, half) reads the two h-rows its ph needs,
// does the H and W passes of B^T d B in registers (8 + 8 packed ops) and feeds 32 MFMAs; the transformed weights of the
// wave's positions come from L2 into registers, re-fetched position by position as soon as their MFMAs are issued.
// Epilogue: A^T over pw in registers, over ph through LDS, then bias / folded eval-BN + ReLU / per-brick statistics / store.
#include "kernels.h"

namespace {

constexpr int P_LW = 18, P_NVOX = 10 * 18;                  // halo of an 8x16 brick
constexpr int P_AI = 2;                                     // 16-B pieces per thread and chunk (360 of 512 used)
constexpr int P_CL = 64, P_S = 12;                          // slots per (zh, zw) parity class / per zh/2 row (conflict-free with the swizzle)
constexpr int P_RAW = 4 * P_CL * 8;                         // floats of the raw image (8 KB)
constexpr int P_BUF = P_RAW + (P_AI * 256 - P_NVOX * 2) * 4;   // + landing zone of the pieces beyond the halo
constexpr int P_EX = 4 * 4 * 4 * 64 * 4;                    // epilogue exchange [ph][nt*2+ow][r/4][lane][4] floats (64 KB)
constexpr int P_LDS_FLOATS = (2 * P_BUF > P_EX + 4 * 64 * 3 ? 2 * P_BUF : P_EX + 4 * 64 * 3);
typedef float f32x2p __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int slot2(int zh, int zw, int q) {
    const int slot = ((zh & 1) * 2 + (zw & 1)) * P_CL + (zh >> 1) * P_S + (zw >> 1);
    return slot * 8 + 4 * (q ^ ((zh >> 1) & 1));
}

__global__ __launch_bounds__(256, 2) void conv2_wino_kernel(const ConvArgs a) {
    constexpr int NT = 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hf = lane >> 5;
    constexpr unsigned OOB = 0x80000000u;

    unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    auto divmod = [](unsigned& x, int d) {
        int r;
        if ((d & (d - 1)) == 0) { r = (int)(x & (unsigned)(d - 1)); x >>= __builtin_ctz((unsigned)d); }
        else { r = (int)(x % (unsigned)d); x /= (unsigned)d; }
        return r;
    };
    const int ntile = divmod(L, a.ntiles);
    const int tw_ = divmod(L, a.tilesW);
    const int th_ = divmod(L, a.tilesH);
    const int dz = divmod(L, a.D); const int nb = (int)L;
    const int h0 = th_ * 8, w0 = tw_ * 16, n0 = ntile * 32 * NT;
    const int mtile = ((nb * a.D + dz) * a.tilesH + th_) * a.tilesW + tw_;
    const int NCH = a.Cin >> 3;

    // one d-slice per descriptor: every offset is a small non-negative number, out-of-volume voxels get the OOB offset
    const size_t slice_x = (size_t)a.H * a.W * a.x_ldc, slice_y = (size_t)a.H * a.W * a.y_ldc;
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x) + ((size_t)nb * a.D + dz) * slice_x, 0, (int)(slice_x * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(a.y + ((size_t)nb * a.D + dz) * slice_y, 0, (int)(slice_y * 4), 0x00020000);
    // transformed weights: U[ntile][chunk][pos 16][nt 2][hf 2][co 32][4 ci]; wave = ph owns positions 4 ph .. 4 ph + 3
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.wt) + ((size_t)ntile * NCH * 16 + wave * 4) * NT * 256, 0, NCH * 16 * NT * 1024, 0x00020000);
    const int b_voff = lane * 16;

    // ---- staging plan: piece idx = tid + 256 it -> (halo voxel, 16-B half)
    unsigned a_src[P_AI]; int a_dst[P_AI];
#pragma unroll
    for (int it = 0; it < P_AI; ++it) {
        const int idx = tid + it * 256;
        const int v = idx >> 1, q = idx & 1;
        const int zw = v % P_LW, zh = v / P_LW;
        const int gh = h0 + zh - 1, gw = w0 + zw - 1;
        const bool inb = v < P_NVOX;
        const bool ok = inb && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
        a_src[it] = ok ? (unsigned)(((gh * a.W + gw) * a.x_ldc + 4 * q) * 4) : OOB;
        a_dst[it] = inb ? slot2(zh, zw, q) : P_RAW + (idx - P_NVOX * 2) * 4;
    }
    // ---- read plan of lane (tile i = j, half hf): tile (th, tw) = (j >> 3, j & 7); H pass of Winograd row ph = wave:
    //      0: r0 - r2   1: r1 + r2   2: r2 - r1   3: r1 - r3   (rows of the tile's 4x4 window)
    const int ha = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int hb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sgn = wave == 1 ? 1.f : -1.f;
    float m1 = -1.f;
    asm volatile("" : "+s"(m1));     // opaque -1: a + m1*b becomes v_pk_fma_f32 (hipcc only packs fadd/ffma, never fsub)
    const int tth = j >> 3, ttw = j & 7;
    auto row_base = [&](int h) {     // LDS float offset of window row h, column 0 (w adds an immediate), this lane's 16-B half
        const int b = tth + (h >> 1);
        return ((h & 1) * 2 * P_CL + b * P_S + ttw) * 8 + 4 * (hf ^ (b & 1));
    };
    const int rdA = row_base(ha), rdB = row_base(hb);

    f32x16 acc[4][NT];
    f32x4 xr[P_AI], Bv[4][NT];
    auto issue_raw = [&](int cb) {
#pragma unroll
        for (int it = 0; it < P_AI; ++it)
            xr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, a_src[it], cb * 4, 0));
    };
    auto write_raw = [&](float* buf) {
#pragma unroll
        for (int it = 0; it < P_AI; ++it) *reinterpret_cast<f32x4*>(buf + a_dst[it]) = xr[it];
    };
    auto load_B = [&](int c, int p) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            Bv[p][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_voff + nt * 1024, (c * 16 * NT + p * NT) * 1024, 0));
    };

    auto chunk = [&](int c, const float* cur, float* nxt, bool first) {
        const int cn = c + 1 < NCH ? c + 1 : c;     // (the last chunk harmlessly re-stages itself: no branch in the loop body)
        issue_raw(cn * 8);
        f32x4 t[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int imm = ((w & 1) * P_CL + (w >> 1)) * 8;
            const f32x4 xa = *reinterpret_cast<const f32x4*>(cur + rdA + imm);
            const f32x4 xb = *reinterpret_cast<const f32x4*>(cur + rdB + imm);
            t[w] = xa + sgn * xb;
        }
        const f32x4 u0 = t[0] + m1 * t[2], u1 = t[1] + t[2], u2 = t[2] + m1 * t[1], u3 = t[1] + m1 * t[3];
        t[0] = u0; t[1] = u1; t[2] = u2; t[3] = u3;
#pragma unroll
        for (int w = 0; w < 4; ++w) {          // keep the transform packed (see conv_wino.hip)
            f32x2p lo = {t[w][0], t[w][1]}, hi = {t[w][2], t[w][3]};
            asm("" : "+v"(lo)); asm("" : "+v"(hi));
            t[w][0] = lo[0]; t[w][1] = lo[1]; t[w][2] = hi[0]; t[w][3] = hi[1];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (first && s == 0) {
                        f32x16 z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.f;
                        acc[p][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(t[p][s], Bv[p][nt][s], z, 0, 0, 0);
                    } else
                        acc[p][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(t[p][s], Bv[p][nt][s], acc[p][nt], 0, 0, 0);
                }
            load_B(cn, p);                            // the position's registers are free again: fetch them for the next chunk
        }
        __builtin_amdgcn_sched_barrier(0);
        write_raw(nxt);
        __syncthreads();
    };

    float* buf0 = smem;
    float* buf1 = smem + P_BUF;
    issue_raw(0);
#pragma unroll
    for (int p = 0; p < 4; ++p) load_B(0, p);
    write_raw(buf0);
    __syncthreads();
    chunk(0, buf0, buf1, true);
    for (int c = 1; c < NCH; c += 2) {
        chunk(c, buf1, buf0, false);
        if (c + 1 < NCH) chunk(c + 1, buf0, buf1, false);
    }

    // ---- epilogue.  acc[pw][nt][r]: position (ph = wave, pw), tile row r -> tile t = (r&3) + 8 (r>>2) + 4 hf, channel 32 nt + j.
    float* ex = smem;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const f32x16 q0 = acc[0][nt] + acc[1][nt] + acc[2][nt];
        const f32x16 q1 = acc[1][nt] + m1 * acc[2][nt] + m1 * acc[3][nt];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f32x4 v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = q0[4 * k + e]; v1[e] = q1[4 * k + e]; }
            *reinterpret_cast<f32x4*>(ex + (((wave * 4 + nt * 2 + 0) * 4 + k) * 64 + lane) * 4) = v0;
            *reinterpret_cast<f32x4*>(ex + (((wave * 4 + nt * 2 + 1) * 4 + k) * 64 + lane) * 4) = v1;
        }
    }
    __syncthreads();
    // wave w now owns column tile nt = w >> 1 and output column offset ow = w & 1 of every tile and sums the ph axis: oh = 0, 1
    const int nt = wave >> 1, ow = wave & 1;
    const int n = n0 + 32 * nt + j;
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
        for (int ph = 0; ph < 4; ++ph) m[ph] = *reinterpret_cast<const f32x4*>(ex + (((ph * 4 + wave) * 4 + k) * 64 + lane) * 4);
        y[0][k] = m[0] + m[1] + m[2] + bias;
        y[1][k] = m[1] + m1 * m[2] + m1 * m[3] + bias;
    }
    if (aff) {
#pragma unroll
        for (int oh = 0; oh < 2; ++oh)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) y[oh][k][e] = fmaxf(__builtin_fmaf(y[oh][k][e], es, eh), 0.f);
    }
    // value (oh, r = 4k + e): tile t = (r&3) + 8 (r>>2) + 4 hf -> (th, tw) = (r >> 2, (r & 3) + 4 hf),
    // voxel (h0 + 2 th + oh, w0 + 2 tw + ow).  Lane part of the address in a VGPR, the rest is scalar.
    const int gw_l = w0 + 8 * hf + ow;
    const unsigned y_voff = (unsigned)(((h0 * a.W + gw_l) * a.y_ldc + n) * 4);
    const bool full = h0 + 8 <= a.H && w0 + 16 <= a.W && n0 + 32 * NT <= a.Ncols;
    const bool do_stats = a.stats != nullptr;
    float cnt = 0.f, sum = 0.f;
    unsigned okmask = 0xffffffffu;
    if (full) {
#pragma unroll
        for (int oh = 0; oh < 2; ++oh)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int soff = (((2 * (r >> 2) + oh) * a.W + 2 * (r & 3)) * a.y_ldc) * 4;
                const float v = y[oh][r >> 2][r & 3];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rs, y_voff, soff, 0);
                sum += v;
            }
        cnt = 32.f;
    } else {
        okmask = 0u;
#pragma unroll
        for (int oh = 0; oh < 2; ++oh)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gh = h0 + 2 * (r >> 2) + oh, gw = gw_l + 2 * (r & 3);
                const bool ok = nvalid && gh < a.H && gw < a.W;
                const int soff = (((2 * (r >> 2) + oh) * a.W + 2 * (r & 3)) * a.y_ldc) * 4;
                const float v = y[oh][r >> 2][r & 3];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y_rs, ok ? y_voff : OOB, soff, 0);
                cnt += ok ? 1.f : 0.f;
                sum += ok ? v : 0.f;
                okmask |= (ok ? 1u : 0u) << (oh * 16 + r);
            }
    }
    if (do_stats) {
        float mean = cnt > 0.f ? sum / cnt : 0.f, m2 = 0.f;
#pragma unroll
        for (int oh = 0; oh < 2; ++oh)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = y[oh][r >> 2][r & 3] - mean;
                m2 += ((okmask >> (oh * 16 + r)) & 1u) ? d * d : 0.f;
            }
        const float cnt2 = __shfl_xor(cnt, 32), mean2 = __shfl_xor(mean, 32), m22 = __shfl_xor(m2, 32);
        welford_merge(cnt, mean, m2, cnt2, mean2, m22);
        float* scr = smem + P_EX;                       // [wave][32][3]
        if (hf == 0) {
            float* sc = scr + (wave * 32 + j) * 3;
            sc[0] = cnt; sc[1] = mean; sc[2] = m2;
        }
        __syncthreads();
        if (tid < 32 * NT) {                            // channel 32 t + jj: waves 2t (ow = 0) and 2t + 1 (ow = 1)
            const int tt = tid >> 5, jj = tid & 31, nn = n0 + 32 * tt + jj;
            if (nn < a.Ncols) {
                float c0 = 0.f, me = 0.f, mm = 0.f;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const float* sc = scr + ((2 * tt + w) * 32 + jj) * 3;
                    welford_merge(c0, me, mm, sc[0], sc[1], sc[2]);
                }
                float* o = a.stats + ((size_t)mtile * a.Cout + nn) * 3;
                o[0] = c0; o[1] = me; o[2] = mm;
            }
        }
    }
}

// torch weights -> U[ntile64][chunk][pos 16][nt 2][hf][co32][4]:  U = (G (x) G) g, evaluated in double.
//   dgrad == 0:  g[tap][n = co][k = ci] = w[co][ci][tap]           (w is (Cout, Cin, 9))
//   dgrad == 1:  g[tap][n = ci][k = co] = w[co][ci][8 - tap]
__global__ void wino2d_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int dgrad, int K, int Ncols, int NPad) {
    const int NCH = K >> 3;
    const size_t total = (size_t)(NPad >> 6) * NCH * 512;     // one thread per (ntile, chunk, nt, hf, co, e): all 16 positions
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = i & 3, co = (i >> 2) & 31, hf = (i >> 7) & 1, nt = (i >> 8) & 1;
        const size_t r = i >> 9;
        const int ch = r % NCH, ntl = r / NCH;
        const int n = ntl * 64 + nt * 32 + co, k = ch * 8 + hf * 4 + e;
        double g[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float v = 0.f;
            if (n < Ncols) v = dgrad ? w[((size_t)k * Cin + n) * 9 + (8 - t)] : w[((size_t)n * Cin + k) * 9 + t];
            g[t] = v;
        }
        double u1[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double g0 = g[c], g1 = g[3 + c], g2 = g[6 + c];
            u1[0][c] = g0; u1[1][c] = 0.5 * (g0 + g1 + g2); u1[2][c] = 0.5 * (g0 - g1 + g2); u1[3][c] = g2;
        }
        float* o = out + ((size_t)(ntl * NCH + ch) * 16) * 512 + ((nt * 2 + hf) * 32 + co) * 4 + e;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const double g0 = u1[ph][0], g1 = u1[ph][1], g2 = u1[ph][2];
            o[(size_t)(ph * 4 + 0) * 512] = (float)g0;
            o[(size_t)(ph * 4 + 1) * 512] = (float)(0.5 * (g0 + g1 + g2));
            o[(size_t)(ph * 4 + 2) * 512] = (float)(0.5 * (g0 - g1 + g2));
            o[(size_t)(ph * 4 + 3) * 512] = (float)g2;
        }
    }
}

}  // namespace

size_t wino2d_packed_floats(int K, int ncols) { return (size_t)16 * K * (size_t)((ncols + 63) / 64 * 64); }

int wino2d_bricks(int N, int D, int H, int W) { return N * D * cdiv(H, 8) * cdiv(W, 16); }

bool conv_use_wino2d(ConvKind kind, int flags, int N, int D, int H, int W, int Cin, int ncols) {
    static const bool enabled = getenv("E3_CONV_NO_WINO") == nullptr && getenv("E3_CONV_NO_WINO2D") == nullptr;
    if (!enabled || kind != CONV_K3_PLANAR || (flags & (CF_SCATTER_UP | CF_GATHER_UP | CF_NO_WINO)) != 0 || Cin < 8 || (Cin & 7)) return false;
    (void)N;                                             // per-sample decision (batch-size independent results)
    const size_t grid = (size_t)wino2d_bricks(1, D, H, W) * ((ncols + 63) / 64);
    return grid >= 128u;                                 // two workgroups per CU: worth it from a quarter of the chip
}

int launch_wino2d_pack(const float* w, float* out, int Cout, int Cin, int dgrad, hipStream_t s) {
    const int K = dgrad ? Cout : Cin, ncols = dgrad ? Cin : Cout;
    const int NPad = (ncols + 63) / 64 * 64;
    const size_t total = (size_t)(NPad >> 6) * (K >> 3) * 512;
    hipLaunchKernelGGL(wino2d_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, out, Cout, Cin, dgrad, K, ncols, NPad);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int launch_conv2_wino(ConvArgs a, hipStream_t s) {
    E3_REQUIRE(!a.pro_scale, E3_ERR_UNSUPPORTED, "the planar Winograd kernel has no BN prologue (callers set CF_NO_WINO)");
    a.tilesD = a.D; a.tilesH = cdiv(a.H, 8); a.tilesW = cdiv(a.W, 16);
    a.NPad = (a.Ncols + 63) / 64 * 64;
    a.ntiles = a.NPad / 64;
    const size_t nblk = (size_t)a.N * a.D * a.tilesH * a.tilesW * a.ntiles;
    E3_REQUIRE(nblk > 0 && nblk < (1u << 31), E3_ERR_INVALID, "conv grid out of range");
    E3_REQUIRE((size_t)a.H * a.W * (size_t)(a.x_ldc > a.y_ldc ? a.x_ldc : a.y_ldc) * 4 < 0x7fffffffu, E3_ERR_UNSUPPORTED,
               "a d-slice of the conv input/output view exceeds 2^31 bytes (32-bit buffer offsets)");
    constexpr int lds_bytes = P_LDS_FLOATS * 4;
    static bool attr_set = false;
    if (!attr_set) {
        E3_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        attr_set = true;
    }
    hipLaunchKernelGGL(conv2_wino_kernel, dim3((unsigned)nblk), dim3(256), lds_bytes, s, a);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}
