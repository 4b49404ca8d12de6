// GridAttention of the decoder blocks (UNet(attention=True); reference: elektronn3/models/unet.py:452-541, called at unet.py:393).
//
//   theta_x = theta(x)                      k = s = 2 conv, no bias          x: encoder skip [N, D, H, W, C]
//   phi_g   = resize(phi(g))                1x1x1 conv + bias, tri-/bilinear  g: decoder input [N, D', H', W', 2C]
//   f       = relu(theta_x + phi_g)                                           [N, d, h, w, C/2]
//   s       = sigmoid(psi(f))               1x1x1 conv to one channel
//   att     = resize(s) to x's grid
//   out     = BatchNorm(W(att * x))         1x1x1 conv + bias, then nn.BatchNorm
//
// The block is off the headline configuration (attention=False everywhere in the reference's examples), so it is built from a few
// general fp32 building blocks rather than per-shape kernels: an LDS-tiled row GEMM on the fp32 matrix cores (rows = voxels; optional 2x2x2 gather on the
// K side or scatter on the N side, per-row scales, bias / affine epilogue, fused row dot product), its reduction-over-voxels
// counterpart for the weight gradients (fixed split order -> deterministic), a general linear resize and its adjoint in gather form,
// and the two elementwise gate kernels.  Everything is NDHWC fp32; channel counts are multiples of 4.
#include "kernels.h"

namespace {

struct Grid5 { int N, d, h, w, D, H, W, sd; };   // rows live on (N, d, h, w); taps address (N, D, H, W) at (sd*z + tz, 2y + ty, 2x + tx)

__device__ __forceinline__ size_t tap_voxel(const Grid5& g, size_t r, int t) {
    const int x = (int)(r % g.w); r /= g.w; const int y = (int)(r % g.h); r /= g.h; const int z = (int)(r % g.d); const size_t n = r / g.d;
    const int tx = t & 1, ty = (t >> 1) & 1, tz = t >> 2;            // torch kernel order (kd, kh, kw); sd == 1: four taps, tz = 0
    return ((n * g.D + (size_t)(g.sd * z + tz)) * g.H + (size_t)(2 * y + ty)) * g.W + (size_t)(2 * x + tx);
}

// the same split in two: tap 0's voxel of row r (32-bit divisions: rows < 2^32 is checked by the launchers) + the tap's offset
__device__ __forceinline__ size_t row_base(const Grid5& g, size_t r) {
    unsigned q = (unsigned)r;
    const unsigned x = q % (unsigned)g.w; q /= (unsigned)g.w; const unsigned y = q % (unsigned)g.h; q /= (unsigned)g.h;
    const unsigned z = q % (unsigned)g.d; const unsigned n = q / (unsigned)g.d;
    return (((size_t)n * g.D + (size_t)(g.sd * z)) * g.H + (size_t)(2 * y)) * g.W + (size_t)(2 * x);
}
__device__ __forceinline__ unsigned tap_off(const Grid5& g, int t) { return (unsigned)(((t >> 2) * g.H + ((t >> 1) & 1)) * g.W + (t & 1)); }

}  // namespace

struct RowGemmArgs {
    const float* A; int lda; int Ck; int Tk;          // K = Tk * Ck, k = (t, c) reads A[voxel(r, t) * lda + c]  (Tk == 1: voxel = r)
    const float* rs_in;                               // optional: row r of A is scaled by rs_in[r]
    const float* W; int wst, wsc, wsn;                // B(k, n) = W[wst * tap + wsc * c_k + wsn * c_n]  (tap of k or of n, whichever side has taps)
    int Cn; int Tn;                                   // N = Tn * Cn; Tn > 1: column (t, c) goes to voxel(r, t), channel c
    const float* bias; const float* epi_scale; const float* epi_shift;
    const float* rs_out;                              // optional: row r of the result is scaled by rs_out[r] (after the dot product below)
    float* out; int ldo; int accumulate;
    float* rowdot; const float* dotsrc; int ld_dot;   // optional: rowdot[r] = sum_n result[r][n] * dotsrc[r * ld_dot + n]
    size_t rows; Grid5 g;
};

__global__ __launch_bounds__(256) void att_rowgemm_kernel(RowGemmArgs a) {
    __shared__ __attribute__((aligned(16))) float As[16][68];     // [k][row]
    __shared__ __attribute__((aligned(16))) float Bs[16][68];     // [k][col]
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const size_t row0 = (size_t)blockIdx.x * 64;
    const int K = a.Tk * a.Ck, Nc = a.Tn * a.Cn;
    const int lr = tid >> 2, kq = (tid & 3) * 4;                  // staging role on the A side: one row, four consecutive k
    const size_t r = row0 + lr; const bool rok = r < a.rows;
    const float rsin = (rok && a.rs_in) ? a.rs_in[r] : 1.f;
    const int kb = tid >> 4, nq = (tid & 15) * 4;                 // staging role on the B side: one k, four consecutive n
    float dotacc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int n0 = 0; n0 < Nc; n0 += 64) {
        float acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        for (int k0 = 0; k0 < K; k0 += 16) {
            {
                const int k = k0 + kq;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (rok && k < K) {
                    const int t = k / a.Ck, c = k - t * a.Ck;
                    const size_t vi = a.Tk > 1 ? tap_voxel(a.g, r, t) : r;
                    v = *reinterpret_cast<const f32x4*>(a.A + vi * a.lda + c) * rsin;
                }
                As[kq + 0][lr] = v[0]; As[kq + 1][lr] = v[1]; As[kq + 2][lr] = v[2]; As[kq + 3][lr] = v[3];
                const int kk = k0 + kb;
                const int tk = kk / a.Ck, ck = kk - tk * a.Ck;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = n0 + nq + e;
                    float wv = 0.f;
                    if (kk < K && n < Nc) { const int tn = n / a.Cn, cn = n - tn * a.Cn; wv = a.W[(size_t)a.wst * (tk + tn) + (size_t)a.wsc * ck + (size_t)a.wsn * cn]; }
                    Bs[kb][nq + e] = wv;
                }
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(&As[kk][ty * 4]);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(&Bs[kk][tx * 4]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
            }
            __syncthreads();
        }
        const int n = n0 + tx * 4;
        if (n < Nc) {
            f32x4 bq = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = bq;
            if (a.bias) bq = *reinterpret_cast<const f32x4*>(a.bias + n);
            if (a.epi_scale) { sc = *reinterpret_cast<const f32x4*>(a.epi_scale + n); sh = *reinterpret_cast<const f32x4*>(a.epi_shift + n); }
            const int tn = n / a.Cn, cn = n - tn * a.Cn;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const size_t rr = row0 + ty * 4 + i;
                if (rr >= a.rows) continue;
                f32x4 v = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
                v = (v + bq) * sc + sh;
                if (a.rowdot) {
                    const f32x4 dv = *reinterpret_cast<const f32x4*>(a.dotsrc + rr * a.ld_dot + n);
                    dotacc[i] += v[0] * dv[0] + v[1] * dv[1] + v[2] * dv[2] + v[3] * dv[3];
                }
                if (a.rs_out) v *= a.rs_out[rr];
                const size_t vi = a.Tn > 1 ? tap_voxel(a.g, rr, tn) : rr;
                float* o = a.out + vi * a.ldo + cn;
                if (a.accumulate) v += *reinterpret_cast<const f32x4*>(o);
                *reinterpret_cast<f32x4*>(o) = v;
            }
        }
    }
    if (a.rowdot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = dotacc[i];
            v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
            const size_t rr = row0 + ty * 4 + i;
            if (tx == 0 && rr < a.rows) a.rowdot[rr] = v;
        }
    }
}

struct RedGemmArgs {
    const float* L; int ldl; int M;                   // left operand rows [rows][M]
    const float* A; int lda; int Ck; int Tk;          // right operand, as the A side of RowGemmArgs; K = Tk * Ck
    const float* rs;                                  // optional per-row scale of the right operand
    int ones;                                         // 1: one more column of ones (column sums of L = bias gradient)
    float* part;                                      // [S][M][K + ones]
    size_t rows; int S; Grid5 g;
};

__global__ __launch_bounds__(256) void att_redgemm_kernel(RedGemmArgs a) {
    __shared__ __attribute__((aligned(16))) float Ls[32][36];     // [voxel][m]
    __shared__ __attribute__((aligned(16))) float Rs[32][68];     // [voxel][k]
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int m0 = blockIdx.z * 32, k0 = blockIdx.y * 64, K = a.Tk * a.Ck, Kp = K + a.ones;
    size_t chunk = (a.rows + a.S - 1) / a.S; chunk = (chunk + 31) / 32 * 32;
    const size_t vbeg = (size_t)blockIdx.x * chunk;
    const size_t vend = vbeg + chunk < a.rows ? vbeg + chunk : a.rows;
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (size_t vb = vbeg; vb < vend; vb += 32) {
        {
            const int lv = tid >> 3, mq = (tid & 7) * 4;
            const size_t v = vb + lv;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int m = m0 + mq + e; Ls[lv][mq + e] = (v < vend && m < a.M) ? a.L[v * a.ldl + m] : 0.f; }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int lv = (tid >> 4) + 16 * p, kq = (tid & 15) * 4, k = k0 + kq;
            const size_t v = vb + lv;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (v < vend) {
                if (k < K) {
                    const int t = k / a.Ck, c = k - t * a.Ck;
                    const size_t vi = a.Tk > 1 ? tap_voxel(a.g, v, t) : v;
                    val = *reinterpret_cast<const f32x4*>(a.A + vi * a.lda + c);
                    if (a.rs) val *= a.rs[v];
                } else if (a.ones && k == K) val[0] = 1.f;
            }
            *reinterpret_cast<f32x4*>(&Rs[lv][kq]) = val;
        }
        __syncthreads();
#pragma unroll 8
        for (int vv = 0; vv < 32; ++vv) {
            const float l0 = Ls[vv][ty * 2], l1 = Ls[vv][ty * 2 + 1];
            const f32x4 rv = *reinterpret_cast<const f32x4*>(&Rs[vv][tx * 4]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[0][j] = fmaf(l0, rv[j], acc[0][j]); acc[1][j] = fmaf(l1, rv[j], acc[1][j]); }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + ty * 2 + i, k = k0 + tx * 4 + j;
            if (m < a.M && k < Kp) a.part[((size_t)blockIdx.x * a.M + m) * Kp + k] = acc[i][j];
        }
}

// ---- the same two GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: lane l holds A[row l & 31][k = l >> 5], B[k = l >> 5][col l & 31];
// the 16 accumulators of a lane are col = l & 31, rows (e & 3) + 8 (e >> 2) + 4 (l >> 5)).  The VALU kernels above stay as the A/B
// reference (E3_ATT_VALU=1).
// NT column tiles of 32 per pass: a 128-row block keeps NT x 16 accumulators per lane and reads its A rows once.  SC: scattered output rows
// (Tn > 1); DOT: the row dot product / output row scale of the gate's backward (compiled out elsewhere: they cost registers)
template <int NT, bool SC, bool DOT>
__global__ __launch_bounds__(256) void att_rowgemm_mfma_kernel(RowGemmArgs a) {
    __shared__ float As[16][129];              // [k][row]
    __shared__ float Bs[16][NT * 32 + 1];      // [k][col]
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const size_t row0 = (size_t)blockIdx.x * 128;
    const int K = a.Tk * a.Ck, Nc = a.Tn * a.Cn;
    const int lr = tid >> 1, ko = (tid & 1) * 8;                  // staging role on the A side: one row, eight consecutive k
    const size_t r = row0 + lr; const bool rok = r < a.rows;
    const float rsin = (rok && a.rs_in) ? a.rs_in[r] : 1.f;
    const size_t abase = (rok && a.Tk > 1) ? row_base(a.g, r) : r;      // voxel of the staged row (tap 0)
    unsigned obase[SC ? 16 : 1];                                         // voxels of this lane's 16 output rows (tap 0)
    if (SC) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const size_t rr = row0 + w * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            obase[SC ? e : 0] = rr < a.rows ? (unsigned)row_base(a.g, rr) : 0u;
        }
    }
    float dotacc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) dotacc[e] = 0.f;
    for (int n0 = 0; n0 < Nc; n0 += NT * 32) {
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = k0 + ko + 4 * h;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (rok && k < K) {
                    const int t = k / a.Ck, c = k - t * a.Ck;
                    const size_t vi = a.Tk > 1 ? abase + tap_off(a.g, t) : r;
                    v = *reinterpret_cast<const f32x4*>(a.A + vi * a.lda + c) * rsin;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) As[ko + 4 * h + e][lr] = v[e];
            }
#pragma unroll
            for (int i = 0; i < 2 * NT; ++i) {
                const int e = tid + 256 * i, kb = e / (NT * 32), nn = e - kb * (NT * 32);
                const int kk = k0 + kb, n = n0 + nn;
                float wv = 0.f;
                if (kk < K && n < Nc) {
                    const int tk = kk / a.Ck, ck = kk - tk * a.Ck, tn = n / a.Cn, cn = n - tn * a.Cn;
                    wv = a.W[(size_t)a.wst * (tk + tn) + (size_t)a.wsc * ck + (size_t)a.wsn * cn];
                }
                Bs[kb][nn] = wv;
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float av = As[2 * kk + hi][w * 32 + lo];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Bs[2 * kk + hi][t * 32 + lo], acc[t], 0, 0, 0);
            }
            __syncthreads();
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = n0 + t * 32 + lo;
            if (n >= Nc) continue;
            const float bq = a.bias ? a.bias[n] : 0.f, sc = a.epi_scale ? a.epi_scale[n] : 1.f, sh = a.epi_scale ? a.epi_shift[n] : 0.f;
            const int tn = n / a.Cn, cn = n - tn * a.Cn;
            const unsigned toff = a.Tn > 1 ? tap_off(a.g, tn) : 0u;
            // all loads of the tile first (old values, dot operands, row scales), then the stores: a store between two loads would serialise them
            float oldv[16], dv[DOT ? 16 : 1], rs[DOT ? 16 : 1];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const size_t rr = row0 + w * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                const bool ok = rr < a.rows;
                const size_t vi = SC ? (size_t)obase[SC ? e : 0] + toff : rr;
                oldv[e] = (ok && a.accumulate) ? a.out[vi * a.ldo + cn] : 0.f;
                if (DOT) {
                    dv[DOT ? e : 0] = (ok && a.rowdot) ? a.dotsrc[rr * a.ld_dot + n] : 0.f;
                    rs[DOT ? e : 0] = (ok && a.rs_out) ? a.rs_out[rr] : 1.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const size_t rr = row0 + w * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
                if (rr >= a.rows) continue;
                const size_t vi = SC ? (size_t)obase[SC ? e : 0] + toff : rr;
                const float v = (acc[t][e] + bq) * sc + sh;
                if (DOT) { dotacc[e] = fmaf(v, dv[DOT ? e : 0], dotacc[e]); a.out[vi * a.ldo + cn] = fmaf(v, rs[DOT ? e : 0], oldv[e]); }
                else a.out[vi * a.ldo + cn] = v + oldv[e];
            }
        }
    }
    if (DOT && a.rowdot) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v = dotacc[e];
            v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
            const size_t rr = row0 + w * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            if (lo == 0 && rr < a.rows) a.rowdot[rr] = v;
        }
    }
}

template <int NT>      // block tile: 32 rows of L^T (m) x NT * 32 columns (k); the four waves split each 64-voxel stage and are summed at the end
__global__ __launch_bounds__(256) void att_redgemm_mfma_kernel(RedGemmArgs a) {
    __shared__ float Ls[64][33];               // [voxel][m]
    __shared__ float Rs[64][NT * 32 + 1];      // [voxel][k]
    __shared__ float red[4][32][33];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.z * 32, k0 = blockIdx.y * (NT * 32), K = a.Tk * a.Ck, Kp = K + a.ones;
    size_t chunk = (a.rows + a.S - 1) / a.S; chunk = (chunk + 63) / 64 * 64;
    const size_t vbeg = (size_t)blockIdx.x * chunk;
    const size_t vend = vbeg + chunk < a.rows ? vbeg + chunk : a.rows;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const int lv = tid >> 2, q = tid & 3;          // staging role: one voxel, a quarter of the m / k range
    const bool lvec = (a.ldl % 4 == 0) && (a.M % 4 == 0);
    for (size_t vb = vbeg; vb < vend; vb += 64) {
        const size_t v = vb + lv;
        const bool vok = v < vend;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + q * 8 + 4 * h;
            f32x4 lvv = {0.f, 0.f, 0.f, 0.f};
            if (vok) {
                if (lvec) { if (m < a.M) lvv = *reinterpret_cast<const f32x4*>(a.L + v * a.ldl + m); }
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (m + e < a.M) lvv[e] = a.L[v * a.ldl + m + e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) Ls[lv][q * 8 + 4 * h + e] = lvv[e];
        }
        const float rsv = (vok && a.rs) ? a.rs[v] : 1.f;
        const size_t vbase = (vok && a.Tk > 1) ? row_base(a.g, v) : v;
#pragma unroll
        for (int i = 0; i < 2 * NT; ++i) {
            const int kq = q * (NT * 8) + 4 * i, k = k0 + kq;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (vok) {
                if (k < K) {
                    const int t = k / a.Ck, c = k - t * a.Ck;
                    const size_t vi = a.Tk > 1 ? vbase + tap_off(a.g, t) : v;
                    val = *reinterpret_cast<const f32x4*>(a.A + vi * a.lda + c) * rsv;
                } else if (a.ones && k == K) val[0] = 1.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) Rs[lv][kq + e] = val[e];
        }
        __syncthreads();
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const int vv = 16 * w + 2 * st + hi;
            const float av = Ls[vv][lo];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Rs[vv][t * 32 + lo], acc[t], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[w][(e & 3) + 8 * (e >> 2) + 4 * hi][lo] = acc[t][e];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = tid + 256 * i, mr = o >> 5, kc = o & 31;
            const float sum = (red[0][mr][kc] + red[1][mr][kc]) + (red[2][mr][kc] + red[3][mr][kc]);
            const int m = m0 + mr, k = k0 + t * 32 + kc;
            if (m < a.M && k < Kp) a.part[((size_t)blockIdx.x * a.M + m) * Kp + k] = sum;
        }
        __syncthreads();
    }
}

// out = sum over the S splits: 16 outputs per workgroup, 16 partial sums per output, then a fixed-order tree (deterministic; replaces the
// one-thread-per-output loop, which is latency-bound when S is in the hundreds)
__global__ __launch_bounds__(256) void att_red_reduce16_kernel(const float* __restrict__ part, int S, int M, int Ck, int Tk, int ones, float* __restrict__ out,
                                                               int osm, int osc, int ost, float* __restrict__ bias_out) {
    __shared__ float ps[16][17];
    const int K = Tk * Ck, Kp = K + ones;
    const int oi = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + oi;
    float sum = 0.f;
    if (i < M * Kp) for (int s = q; s < S; s += 16) sum += part[(size_t)s * M * Kp + i];
    ps[q][oi] = sum;
    __syncthreads();
    if (q == 0 && i < M * Kp) {
        float t8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t8[j] = ps[2 * j][oi] + ps[2 * j + 1][oi];
        const float tot = ((t8[0] + t8[1]) + (t8[2] + t8[3])) + ((t8[4] + t8[5]) + (t8[6] + t8[7]));
        const int m = i / Kp, k = i - m * Kp;
        if (k < K) { const int t = k / Ck, c = k - t * Ck; out[(size_t)m * osm + (size_t)c * osc + (size_t)t * ost] = tot; }
        else if (bias_out) bias_out[m] = tot;
    }
}

namespace {

// source index pair and weight of nn.functional.interpolate(mode='(tri|bi)linear', align_corners=False) along one axis
__device__ __forceinline__ void lin_src(int o, int in, int out, int& i0, int& i1, float& l1) {
    if (in == out) { i0 = i1 = o; l1 = 0.f; return; }
    const float scale = (float)in / (float)out;
    float src = scale * ((float)o + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src; if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0; l1 = fminf(fmaxf(l1, 0.f), 1.f);
}

template <int VEC>
__global__ void att_resize_fwd_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int N, int Ds, int Hs, int Ws, int Dd, int Hd, int Wd) {
    const int Q = C / VEC;
    const size_t total = (size_t)N * Dd * Hd * Wd * Q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % Q); size_t r = i / Q;
        const int x = (int)(r % Wd); r /= Wd; const int y = (int)(r % Hd); r /= Hd; const int z = (int)(r % Dd); const size_t n = r / Dd;
        int z0, z1, y0, y1, x0, x1; float lz, ly, lx;
        lin_src(z, Ds, Dd, z0, z1, lz); lin_src(y, Hs, Hd, y0, y1, ly); lin_src(x, Ws, Wd, x0, x1, lx);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int cz = 0; cz < 2; ++cz)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                for (int cx = 0; cx < 2; ++cx) {
                    const float wgt = (cz ? lz : 1.f - lz) * (cy ? ly : 1.f - ly) * (cx ? lx : 1.f - lx);
                    const float* p = src + (((n * Ds + (size_t)(cz ? z1 : z0)) * Hs + (size_t)(cy ? y1 : y0)) * Ws + (size_t)(cx ? x1 : x0)) * C + q * VEC;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] = fmaf(wgt, p[e], acc[e]);
                }
        float* o = dst + (i / Q) * C + q * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = acc[e];
    }
}

// weight with which destination index o reads source index i
__device__ __forceinline__ float lin_weight(int o, int i, int in, int out) {
    int i0, i1; float l1;
    lin_src(o, in, out, i0, i1, l1);
    return (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
}
// destination indices that can read source index i: src(o) in (i - 1, i + 1)
__device__ __forceinline__ void lin_range(int i, int in, int out, int& lo, int& hi) {
    if (in == out) { lo = hi = i; return; }
    const float inv = (float)out / (float)in;
    lo = (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1; hi = (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1;
    if (lo < 0) lo = 0;
    if (hi > out - 1) hi = out - 1;
}

// adjoint of att_resize_fwd_kernel in gather form: gsrc[i] = sum_o weight(o -> i) * gdst[o], summed in a fixed order
template <int VEC>
__global__ void att_resize_bwd_kernel(const float* __restrict__ gdst, float* __restrict__ gsrc, int C, int N, int Ds, int Hs, int Ws, int Dd, int Hd, int Wd) {
    const int Q = C / VEC;
    const size_t total = (size_t)N * Ds * Hs * Ws * Q;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % Q); size_t r = i / Q;
        const int x = (int)(r % Ws); r /= Ws; const int y = (int)(r % Hs); r /= Hs; const int z = (int)(r % Ds); const size_t n = r / Ds;
        int zl, zh, yl, yh, xl, xh;
        lin_range(z, Ds, Dd, zl, zh); lin_range(y, Hs, Hd, yl, yh); lin_range(x, Ws, Wd, xl, xh);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int oz = zl; oz <= zh; ++oz) {
            const float wz = lin_weight(oz, z, Ds, Dd);
            if (wz == 0.f) continue;
            for (int oy = yl; oy <= yh; ++oy) {
                const float wy = lin_weight(oy, y, Hs, Hd);
                if (wy == 0.f) continue;
                for (int ox = xl; ox <= xh; ++ox) {
                    const float wx = lin_weight(ox, x, Ws, Wd);
                    if (wx == 0.f) continue;
                    const float wgt = wz * wy * wx;
                    const float* p = gdst + (((n * Dd + (size_t)oz) * Hd + (size_t)oy) * Wd + (size_t)ox) * C + q * VEC;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] = fmaf(wgt, p[e], acc[e]);
                }
            }
        }
        float* o = gsrc + (i / Q) * C + q * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = acc[e];
    }
}

// f = relu(theta + phi) in place over theta; s = sigmoid(psi_w . f + psi_b)
__global__ void att_gate_fwd_kernel(float* __restrict__ f, const float* __restrict__ phi, const float* __restrict__ psi_w, const float* __restrict__ psi_b,
                                    float* __restrict__ sgm, size_t rows, int Ci) {
    for (size_t v = blockIdx.x * (size_t)blockDim.x + threadIdx.x; v < rows; v += (size_t)gridDim.x * blockDim.x) {
        float dot = psi_b[0];
        for (int i = 0; i < Ci; i += 4) {
            f32x4 a = *reinterpret_cast<const f32x4*>(f + v * Ci + i);
            const f32x4 p = *reinterpret_cast<const f32x4*>(phi + v * Ci + i);
            const f32x4 w = *reinterpret_cast<const f32x4*>(psi_w + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = fmaxf(a[e] + p[e], 0.f); dot = fmaf(w[e], a[e], dot); }
            *reinterpret_cast<f32x4*>(f + v * Ci + i) = a;
        }
        sgm[v] = 1.f / (1.f + expf(-dot));
    }
}

// dpsi = ds * s * (1 - s); df = dpsi * psi_w where f > 0
__global__ void att_gate_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ sgm, const float* __restrict__ f, const float* __restrict__ psi_w,
                                    float* __restrict__ dpsi, float* __restrict__ df, size_t rows, int Ci) {
    for (size_t v = blockIdx.x * (size_t)blockDim.x + threadIdx.x; v < rows; v += (size_t)gridDim.x * blockDim.x) {
        const float sv = sgm[v], dp = ds[v] * sv * (1.f - sv);
        dpsi[v] = dp;
        for (int i = 0; i < Ci; i += 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(f + v * Ci + i);
            const f32x4 w = *reinterpret_cast<const f32x4*>(psi_w + i);
            f32x4 g;
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = a[e] > 0.f ? dp * w[e] : 0.f;
            *reinterpret_cast<f32x4*>(df + v * Ci + i) = g;
        }
    }
}

inline int grid_for(size_t items) { size_t g = (items + 255) / 256; return (int)(g < 1 ? 1 : (g > 65535 * 16 ? 65535 * 16 : g)); }

Grid5 grid5(const AttDims& d) { return Grid5{d.N, d.d, d.h, d.w, d.D, d.H, d.W, d.sd}; }

const bool g_att_valu = getenv("E3_ATT_VALU") != nullptr;      // A/B switch: the VALU GEMMs instead of the matrix-core ones

int run_rowgemm(RowGemmArgs a, hipStream_t s) {
    E3_REQUIRE(a.Ck % 4 == 0 && a.Cn % 4 == 0 && a.lda % 4 == 0 && a.ldo % 4 == 0, E3_ERR_UNSUPPORTED, "attention: channel counts must be multiples of 4");
    E3_REQUIRE(a.Tk == 1 || a.Tn == 1, E3_ERR_INVALID, "attention GEMM: taps on one side only");
    E3_REQUIRE(a.rows < ((size_t)1 << 32), E3_ERR_UNSUPPORTED, "attention GEMM: more than 2^32 voxel rows");
    if (a.rows == 0) return E3_OK;
    if (g_att_valu) hipLaunchKernelGGL(att_rowgemm_kernel, dim3((unsigned)((a.rows + 63) / 64)), dim3(256), 0, s, a);
    else {
        const dim3 g((unsigned)((a.rows + 127) / 128)), b(256);
        const int tiles = cdiv(a.Tn * a.Cn, 32);
        const bool sc = a.Tn > 1, dot = a.rowdot != nullptr || a.rs_out != nullptr;
        E3_REQUIRE(!sc || (size_t)a.g.N * a.g.D * a.g.H * a.g.W < ((size_t)1 << 32), E3_ERR_UNSUPPORTED, "attention GEMM: more than 2^32 scattered voxels");
        E3_REQUIRE(!(sc && dot), E3_ERR_INVALID, "attention GEMM: scatter and row dot product are separate launches");
#define E3_ROWGEMM(NT)                                                                                         \
        do {                                                                                                   \
            if (sc) hipLaunchKernelGGL((att_rowgemm_mfma_kernel<NT, true, false>), g, b, 0, s, a);           \
            else if (dot) hipLaunchKernelGGL((att_rowgemm_mfma_kernel<NT, false, true>), g, b, 0, s, a);     \
            else hipLaunchKernelGGL((att_rowgemm_mfma_kernel<NT, false, false>), g, b, 0, s, a);             \
        } while (0)
        if (tiles <= 1) E3_ROWGEMM(1);
        else if (tiles <= 2) E3_ROWGEMM(2);
        else E3_ROWGEMM(4);      // (8 tiles per pass cost 350-400 registers: one wave per SIMD; two passes over L2-resident rows are faster)
#undef E3_ROWGEMM
    }
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int red_nt(int Kp) { return Kp <= 32 ? 1 : (Kp <= 64 ? 2 : 4); }      // column tiles of 32 per block of the matrix-core reduce GEMM

int red_splits(size_t rows, int M, int Kp) {
    const int tiles = g_att_valu ? cdiv(Kp, 64) * cdiv(M, 32) : cdiv(Kp, red_nt(Kp) * 32) * cdiv(M, 32);
    int S = 1024 / tiles; if (S < 1) S = 1;
    const size_t cap = (rows + 255) / 256;
    if ((size_t)S > cap) S = (int)(cap < 1 ? 1 : cap);
    return S;
}

// weight gradient: out = L^T . R (+ bias column), through `part`
int run_redgemm(RedGemmArgs a, float* out, int osm, int osc, int ost, float* bias_out, hipStream_t s) {
    E3_REQUIRE(a.Ck % 4 == 0 && a.lda % 4 == 0, E3_ERR_UNSUPPORTED, "attention: channel counts must be multiples of 4");
    const int K = a.Tk * a.Ck, Kp = K + a.ones;
    a.S = red_splits(a.rows, a.M, Kp);
    if (g_att_valu) hipLaunchKernelGGL(att_redgemm_kernel, dim3(a.S, cdiv(Kp, 64), cdiv(a.M, 32)), dim3(256), 0, s, a);
    else {
        const int nt = red_nt(Kp);
        const dim3 g(a.S, cdiv(Kp, nt * 32), cdiv(a.M, 32)), b(256);
        if (nt == 1) hipLaunchKernelGGL(att_redgemm_mfma_kernel<1>, g, b, 0, s, a);
        else if (nt == 2) hipLaunchKernelGGL(att_redgemm_mfma_kernel<2>, g, b, 0, s, a);
        else hipLaunchKernelGGL(att_redgemm_mfma_kernel<4>, g, b, 0, s, a);
    }
    E3_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(att_red_reduce16_kernel, dim3(cdiv(a.M * Kp, 16)), dim3(256), 0, s, a.part, a.S, a.M, a.Ck, a.Tk, a.ones, out, osm, osc, ost, bias_out);
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

int run_resize(bool bwd, const float* src, float* dst, int C, int N, int Ds, int Hs, int Ws, int Dd, int Hd, int Wd, hipStream_t s) {
    // forward: src (Ds..) -> dst (Dd..); backward: gdst = src argument on the (Dd..) grid -> gsrc = dst argument on the (Ds..) grid
    const size_t items = (size_t)N * (bwd ? (size_t)Ds * Hs * Ws : (size_t)Dd * Hd * Wd) * (C % 4 == 0 ? C / 4 : C);
    if (items == 0) return E3_OK;
    const dim3 g(grid_for(items)), b(256);
    if (C % 4 == 0) {
        if (bwd) hipLaunchKernelGGL(att_resize_bwd_kernel<4>, g, b, 0, s, src, dst, C, N, Ds, Hs, Ws, Dd, Hd, Wd);
        else hipLaunchKernelGGL(att_resize_fwd_kernel<4>, g, b, 0, s, src, dst, C, N, Ds, Hs, Ws, Dd, Hd, Wd);
    } else {
        if (bwd) hipLaunchKernelGGL(att_resize_bwd_kernel<1>, g, b, 0, s, src, dst, C, N, Ds, Hs, Ws, Dd, Hd, Wd);
        else hipLaunchKernelGGL(att_resize_fwd_kernel<1>, g, b, 0, s, src, dst, C, N, Ds, Hs, Ws, Dd, Hd, Wd);
    }
    E3_CHECK_HIP(hipGetLastError());
    return E3_OK;
}

}  // namespace

bool att_resized(const AttDims& d) { return d.gd != d.d || d.gh != d.h || d.gw != d.w; }

size_t att_part_floats(const AttDims& d) {
    const int C = d.C, Ci = d.C / 2, G = 2 * d.C, T = d.sd * 4;
    auto need = [&](size_t rows, int M, int Kp) { return (size_t)red_splits(rows, M, Kp) * M * Kp; };
    const size_t fine = (size_t)d.N * d.D * d.H * d.W, coarse = (size_t)d.N * d.d * d.h * d.w, dec = (size_t)d.N * d.gd * d.gh * d.gw;
    size_t m = need(fine, C, C);
    const size_t b = need(coarse, 1, Ci + 1), c = need(coarse, Ci, T * C), e = need(dec, Ci, G + 1);
    if (b > m) m = b;
    if (c > m) m = c;
    if (e > m) m = e;
    return m;
}

#define RUN(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

// gate: x [fine, C] (ldx), g [dec grid, 2C] (ldg) -> f [coarse, C/2], sgm [coarse], att [fine]
int launch_att_gate_fwd(const AttDims& d, const float* x, int ldx, const float* g, int ldg, const AttParams& p, float* f, float* sgm, float* att,
                        float* phi_tmp, float* phi_res, hipStream_t s) {
    const int C = d.C, Ci = C / 2, G = 2 * C, T = d.sd * 4;
    E3_REQUIRE(C % 8 == 0, E3_ERR_UNSUPPORTED, "attention: channels must be a multiple of 8");
    E3_REQUIRE(d.d >= 1 && d.h >= 1 && d.w >= 1, E3_ERR_INVALID, "attention: the skip tensor is smaller than the 2x2x2 kernel of theta");
    const size_t coarse = (size_t)d.N * d.d * d.h * d.w, dec = (size_t)d.N * d.gd * d.gh * d.gw;
    {   // theta_x: k = s = 2 conv without bias (unet.py:498-501)
        RowGemmArgs a{};
        a.A = x; a.lda = ldx; a.Ck = C; a.Tk = T; a.W = p.theta_w; a.wst = 1; a.wsc = T; a.wsn = C * T; a.Cn = Ci; a.Tn = 1;
        a.out = f; a.ldo = Ci; a.rows = coarse; a.g = grid5(d);
        RUN(run_rowgemm(a, s));
    }
    {   // phi(g) on the decoder grid (unet.py:502-505)
        RowGemmArgs a{};
        a.A = g; a.lda = ldg; a.Ck = G; a.Tk = 1; a.W = p.phi_w; a.wst = 0; a.wsc = 1; a.wsn = G; a.Cn = Ci; a.Tn = 1; a.bias = p.phi_b;
        a.out = phi_tmp; a.ldo = Ci; a.rows = dec; a.g = grid5(d);
        RUN(run_rowgemm(a, s));
    }
    const float* phi = phi_tmp;
    if (d.gd != d.d || d.gh != d.h || d.gw != d.w) {      // F.interpolate(..., size=theta_x.shape[2:]) (unet.py:517): the identity when the grids agree
        RUN(run_resize(false, phi_tmp, phi_res, Ci, d.N, d.gd, d.gh, d.gw, d.d, d.h, d.w, s));
        phi = phi_res;
    }
    hipLaunchKernelGGL(att_gate_fwd_kernel, dim3(grid_for(coarse)), dim3(256), 0, s, f, phi, p.psi_w, p.psi_b, sgm, coarse, Ci);
    E3_CHECK_HIP(hipGetLastError());
    RUN(run_resize(false, sgm, att, 1, d.N, d.d, d.h, d.w, d.D, d.H, d.W, s));      // unet.py:525
    return E3_OK;
}

// W(att * x) + bias (training: raw tensor for the BatchNorm) or folded with the eval-mode BatchNorm (scale, shift; bias inside shift)
int launch_att_out_fwd(const AttDims& d, const float* x, int ldx, const float* att, const AttParams& p, const float* epi_scale, const float* epi_shift,
                       float* out, int ldo, hipStream_t s) {
    RowGemmArgs a{};
    a.A = x; a.lda = ldx; a.Ck = d.C; a.Tk = 1; a.rs_in = att; a.W = p.w_w; a.wst = 0; a.wsc = 1; a.wsn = d.C; a.Cn = d.C; a.Tn = 1;
    a.bias = epi_scale ? nullptr : p.w_b; a.epi_scale = epi_scale; a.epi_shift = epi_shift;
    a.out = out; a.ldo = ldo; a.rows = (size_t)d.N * d.D * d.H * d.W; a.g = grid5(d);
    return run_rowgemm(a, s);
}

// everything behind the BatchNorm backward: dz = gradient of W's output [fine, C]
//   -> gradients of w.0.weight, theta, phi, psi; dx [fine, C] (overwritten); dphi [dec grid, C/2] for launch_att_bwd_gate_input
int launch_att_bwd(const AttDims& d, const float* dz, const float* x, int ldx, const float* g, int ldg, const float* f, const float* sgm, const float* att,
                   const AttParams& p, const AttParams& grad, float* dx, float* dphi, float* tmp_fine, float* tmp_coarse, float* df, float* part,
                   hipStream_t s) {
    const int C = d.C, Ci = C / 2, G = 2 * C, T = d.sd * 4;
    const size_t fine = (size_t)d.N * d.D * d.H * d.W, coarse = (size_t)d.N * d.d * d.h * d.w, dec = (size_t)d.N * d.gd * d.gh * d.gw;
    const bool resized = att_resized(d);
    float* datt = tmp_fine;                      // [fine]
    float* dsg = tmp_coarse;                     // [coarse]
    float* dpsi = tmp_coarse + coarse;           // [coarse]
    {   // dW = dz^T (att * x)
        RedGemmArgs a{};
        a.L = dz; a.ldl = C; a.M = C; a.A = x; a.lda = ldx; a.Ck = C; a.Tk = 1; a.rs = att; a.ones = 0; a.part = part; a.rows = fine; a.g = grid5(d);
        RUN(run_redgemm(a, grad.w_w, C, 1, 0, nullptr, s));
    }
    {   // dy = dz W; dx = att * dy; datt = sum_c dy * x
        RowGemmArgs a{};
        a.A = dz; a.lda = C; a.Ck = C; a.Tk = 1; a.W = p.w_w; a.wst = 0; a.wsc = C; a.wsn = 1; a.Cn = C; a.Tn = 1;
        a.rs_out = att; a.out = dx; a.ldo = C; a.rowdot = datt; a.dotsrc = x; a.ld_dot = ldx; a.rows = fine; a.g = grid5(d);
        RUN(run_rowgemm(a, s));
    }
    RUN(run_resize(true, datt, dsg, 1, d.N, d.d, d.h, d.w, d.D, d.H, d.W, s));
    hipLaunchKernelGGL(att_gate_bwd_kernel, dim3(grid_for(coarse)), dim3(256), 0, s, dsg, sgm, f, p.psi_w, dpsi, df, coarse, Ci);
    E3_CHECK_HIP(hipGetLastError());
    {   // psi: dW = dpsi^T f, db = sum dpsi
        RedGemmArgs a{};
        a.L = dpsi; a.ldl = 1; a.M = 1; a.A = f; a.lda = Ci; a.Ck = Ci; a.Tk = 1; a.ones = 1; a.part = part; a.rows = coarse; a.g = grid5(d);
        RUN(run_redgemm(a, grad.psi_w, Ci, 1, 0, grad.psi_b, s));
    }
    {   // theta: dW[i][c][t] = sum_v df[v][i] x[2v + t][c]
        RedGemmArgs a{};
        a.L = df; a.ldl = Ci; a.M = Ci; a.A = x; a.lda = ldx; a.Ck = C; a.Tk = T; a.ones = 0; a.part = part; a.rows = coarse; a.g = grid5(d);
        RUN(run_redgemm(a, grad.theta_w, C * T, T, 1, nullptr, s));
    }
    {   // dx[2v + t][c] += sum_i df[v][i] theta[i][c][t]  (every voxel of x belongs to at most one 2x2x2 patch)
        RowGemmArgs a{};
        a.A = df; a.lda = Ci; a.Ck = Ci; a.Tk = 1; a.W = p.theta_w; a.wst = 1; a.wsc = C * T; a.wsn = T; a.Cn = C; a.Tn = T;
        a.out = dx; a.ldo = C; a.accumulate = 1; a.rows = coarse; a.g = grid5(d);
        RUN(run_rowgemm(a, s));
    }
    const float* dp = df;      // (att_resized(d) == false: the caller hands `df` to launch_att_bwd_gate_input)
    if (resized) { RUN(run_resize(true, df, dphi, Ci, d.N, d.gd, d.gh, d.gw, d.d, d.h, d.w, s)); dp = dphi; }
    {   // phi: dW = dphi^T g, db = sum dphi
        RedGemmArgs a{};
        a.L = dp; a.ldl = Ci; a.M = Ci; a.A = g; a.lda = ldg; a.Ck = G; a.Tk = 1; a.ones = 1; a.part = part; a.rows = dec; a.g = grid5(d);
        RUN(run_redgemm(a, grad.phi_w, G, 1, 0, grad.phi_b, s));
    }
    return E3_OK;
}

// dg += dphi . phi_w   (the gating signal is the decoder block's input: its gradient also gets the transposed conv's share)
int launch_att_bwd_gate_input(const AttDims& d, const float* dphi, const AttParams& p, float* dg, int ldg, hipStream_t s) {
    const int C = d.C, Ci = C / 2, G = 2 * C;
    RowGemmArgs a{};
    a.A = dphi; a.lda = Ci; a.Ck = Ci; a.Tk = 1; a.W = p.phi_w; a.wst = 0; a.wsc = G; a.wsn = 1; a.Cn = G; a.Tn = 1;
    a.out = dg; a.ldo = ldg; a.accumulate = 1; a.rows = (size_t)d.N * d.gd * d.gh * d.gw; a.g = grid5(d);
    return run_rowgemm(a, s);
}

// ---- plain 1x1x1 convolutions over voxel rows (the ResUNet's shortcut projections, resunet.py:247-251); w is torch's (Cout, Cin)
size_t pw_part_floats(size_t rows, int Cout, int Cin) { return (size_t)red_splits(rows, Cout, Cin + 1) * Cout * (Cin + 1); }

int launch_pw_fwd(const float* x, int ldx, int Cin, const float* w, const float* b, float* out, int ldo, int Cout, size_t rows, hipStream_t s) {
    RowGemmArgs a{};
    a.A = x; a.lda = ldx; a.Ck = Cin; a.Tk = 1; a.W = w; a.wst = 0; a.wsc = 1; a.wsn = Cin; a.Cn = Cout; a.Tn = 1; a.bias = b;
    a.out = out; a.ldo = ldo; a.rows = rows;
    return run_rowgemm(a, s);
}
// dx += dy . w
int launch_pw_dgrad_acc(const float* dy, int ldy, int Cout, const float* w, float* dx, int ldx, int Cin, size_t rows, hipStream_t s) {
    RowGemmArgs a{};
    a.A = dy; a.lda = ldy; a.Ck = Cout; a.Tk = 1; a.W = w; a.wst = 0; a.wsc = Cin; a.wsn = 1; a.Cn = Cin; a.Tn = 1;
    a.out = dx; a.ldo = ldx; a.accumulate = 1; a.rows = rows;
    return run_rowgemm(a, s);
}
// dw = dy^T x, db = column sums of dy; part: pw_part_floats(rows, Cout, Cin) floats
int launch_pw_wgrad(const float* dy, int ldy, int Cout, const float* x, int ldx, int Cin, float* part, float* dw, float* db, size_t rows, hipStream_t s) {
    RedGemmArgs a{};
    a.L = dy; a.ldl = ldy; a.M = Cout; a.A = x; a.lda = ldx; a.Ck = Cin; a.Tk = 1; a.ones = 1; a.part = part; a.rows = rows;
    return run_redgemm(a, dw, Cin, 1, 0, db, s);
}
