// Common definitions for libe3unet (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define E3_OK 0
#define E3_ERR_INVALID 1
#define E3_ERR_HIP 2
#define E3_ERR_UNSUPPORTED 3
#define E3_ERR_WORKSPACE 4

void e3_set_error(const std::string& msg);

#define E3_CHECK_HIP(expr)                                                                     \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            e3_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                   \
            return E3_ERR_HIP;                                                                 \
        }                                                                                      \
    } while (0)

#define E3_REQUIRE(cond, code, msg)                                                            \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            e3_set_error(std::string(msg) + " [" #cond "]");                                   \
            return (code);                                                                     \
        }                                                                                      \
    } while (0)

// Activations live in HBM as NDHWC fp32.  A view addresses `C` channels starting at `ptr`
// inside rows of `ldc` floats per voxel, so producers can write straight into one half of a
// concat buffer (torch.cat disappears, unet.py:398-399).
struct View {
    float* ptr;
    int N, D, H, W, C;
    int ldc;
    __host__ __device__ size_t voxels() const { return (size_t)N * D * H * W; }
};

static inline View make_view(float* p, int N, int D, int H, int W, int C, int ldc = 0) {
    View v; v.ptr = p; v.N = N; v.D = D; v.H = H; v.W = W; v.C = C; v.ldc = ldc ? ldc : C; return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bijective XCD-aware block remap (cdna guide T1): physical block b runs on XCD b % 8; give each
// XCD a contiguous range of logical ids so that neighbouring tiles share that XCD's L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned xcd = bid & 7u, q = nblk >> 3, r = nblk & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// The network's activation: ReLU (slope 0), LeakyReLU(slope) [get_activation 'leaky' = 0.1, unet.py:183-199], identity (slope 1,
// 'lin') or SiLU (ACT_SILU).  Written so that slope == 0 reproduces fmaxf(z, 0) bit for bit (+0 + -0 = +0) and the mask form stays sign-of-zero clean.
// Activation argument of the kernels: a constant slope, or -- nn.PReLU(num_parameters=1), 'prelu' -- a learnable slope read from
// device memory (no host sync).  Implicitly constructible from a float, so constant-slope call sites stay as they are.
struct ActArg {
    float slope; const float* ptr;
    // nn.RReLU in TRAIN mode (get_activation 'rrelu', unet.py:183-199 -> nn.RReLU(1/8, 1/3)): the slope of a negative input is drawn per
    // element from U(lo, hi).  seed != 0 selects it; the draw is a counter-based hash of (seed, element index), so the backward pass
    // recomputes the forward's slopes instead of storing a noise tensor.
    unsigned seed; float lo, hi;
    __host__ __device__ ActArg(float s = 0.f) : slope(s), ptr(nullptr), seed(0u), lo(0.f), hi(0.f) {}
    __host__ __device__ ActArg(float s, const float* p) : slope(s), ptr(p), seed(0u), lo(0.f), hi(0.f) {}
    __host__ __device__ ActArg rrelu(unsigned sd, float l, float h) const { ActArg a = *this; a.seed = sd; a.lo = l; a.hi = h; return a; }
    // (a learned slope that is exactly 2.0f must not be mistaken for the ACT_SILU code below: moved by one ulp)
    __device__ __forceinline__ float get() const { if (!ptr) return slope; const float v = *ptr; return v == 2.f ? 2.0000002f : v; }
};
// slope == ACT_SILU selects nn.SiLU ('silu'): z * sigmoid(z), derivative sig * (1 + z * (1 - sig)).
constexpr float ACT_SILU = 2.f;
constexpr float ACT_PRELU = 3.f;     // (plan-level code only: the kernels see the learnable slope through ActArg::ptr)
// slope of element idx (= voxel * C + channel of the activation's tensor): the uniform one, or the train-mode RReLU draw
__device__ __forceinline__ float act_slope_at(const ActArg& a, float slope, unsigned idx) {
    if (a.seed == 0u) return slope;
    unsigned h = idx * 747796405u + a.seed;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return a.lo + (a.hi - a.lo) * ((float)(h >> 8) * 5.9604645e-8f);
}
__device__ __forceinline__ float act_fwd(float z, float slope) {
    if (slope == ACT_SILU) return z / (1.f + expf(-z));
    return fmaxf(z, 0.f) + slope * fminf(z, 0.f);
}
__device__ __forceinline__ float act_bwd(float z, float g, float slope) {     // z = PRE-activation value
    if (slope == ACT_SILU) { const float sg = 1.f / (1.f + expf(-z)); return g * (sg * (1.f + z * (1.f - sg))); }
    return z > 0.f ? g : slope * g + 0.f;
}

// Chan/Welford merge of (count, mean, M2) records; robust for na == 0 or nb == 0.
__device__ __forceinline__ void welford_merge(float& na, float& ma, float& sa, float nb, float mb, float sb) {
    const float n = na + nb;
    if (n > 0.f) {
        const float d = mb - ma;
        const float f = nb / n;
        ma += d * f;
        sa += sb + d * d * na * f;
        na = n;
    }
}

// Timing experiments (flag bits 256: skip the staging, 512: skip the stores, 1024: s_memtime stamps instead of statistics; wrong results) exist in developer
// builds only (-DE3_TIMING, e.g. E3_HIPCC_EXTRA=-DE3_TIMING python -m elektronn3_amd.build --force into a copy of the tree): in the library that ships the
// bits read as zero and the code behind them is compiled out.
#ifdef E3_TIMING
#define E3_DBG_FLAGS(f) (f)
#else
#define E3_DBG_FLAGS(f) 0
#endif
