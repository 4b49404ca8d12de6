// Launcher interface of the native bf16 path (BASELINE.json configs[2]: the same U-Net with bf16 storage).
//
// Activations and their gradients live in HBM as bf16 NDHWC (same (ptr, ldc) views as the fp32 path), weights are packed to bf16
// per call, every accumulation is fp32 (v_mfma_f32_32x32x16_bf16), BatchNorm statistics / coefficients / parameter gradients are
// fp32.  Like a `model.to(torch.bfloat16)` reference module, every op rounds its result to bf16 once (the conv output incl. bias,
// the normalised activation, every gradient tensor); the statistics are those of the ROUNDED conv output, as nn.BatchNorm3d sees it.
// Reference: the reduced-precision switches of elektronn3 are Trainer(mixed_precision) (training/trainer.py:367,519) and
// Predictor(float16=True) / model.half() (inference/inference.py:445-446, benchmark/pred_benchmark.py:55,71).
#pragma once
#include "common.h"

typedef unsigned short bf16_t;                                   // storage type (raw bits)
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
// The whole 16-bit path is compiled twice: as is for bfloat16, and with -DE3_F16 -include f16_names.h for IEEE half (the reference's own
// reduced-precision mode: torch.cuda.amp.autocast defaults to float16, trainer.py:367,519; Predictor(float16=True) = model.half(),
// inference.py:445-446).  Only the element type, its conversions and the matrix instruction differ; f16_names.h renames every external
// symbol of these translation units (launch_*_b16 -> launch_*_f16, e3_*_bf16 -> e3_*_f16).
#ifdef E3_F16
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }     // round to nearest even, overflow -> inf (as torch)
#define E3_MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {                 // round to nearest even (NaN stays NaN)
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}
#define E3_MFMA16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif
__device__ __forceinline__ float round_bf(float f) { return bf2f(f2bf(f)); }

// ---------------------------------------------------------------- 3x3x3 / 1x3x3 conv, stride 1, 'same' padding (bf16_conv.hip)
// y[v][co] = bf16( sum_{tap, ci} x[v + tap][ci] * w[co][ci][tap] (+ bias[co]) )      forward (unet.py:131-149) and, with the flipped /
// transposed packing, the data gradient.  Cin % 32 == 0, Cout % 32 == 0.
struct ConvB16Args {
    const bf16_t* x; int x_ldc; int Cin;
    const bf16_t* wt;                      // packed by launch_pack_conv_b16: [tap][Cin/32][2][CoPad][2][8]
    const float* bias;                     // [Cout] fp32 or null
    bf16_t* y; int y_ldc;
    int N, D, H, W, Cout;
    int planar;                            // 1: 1x3x3 taps (planar blocks, unet.py:114-128)
    const float* epi_scale; const float* epi_shift;   // non-null: y = bf16(relu(acc * scale[co] + shift[co]))   (eval mode: folded BN)
    float* stats;                          // non-null: records [conv_b16_stats_parts][Cout][3] = (n, mean, M2) of the stored values
    float* partial;                        // scratch of conv_b16_partial_floats() floats (split-K of the low-resolution levels; may be null if 0)
    // concat without a concat buffer, halves kept as two tensors (64-byte rows at 32 channels: a chunk of one half then reads whole
    // rows instead of half of every 128-byte row): input channels >= x_split come from x2 (same x_ldc), output channels >= y_split go
    // to y2 (same y_ldc).  x_split / y_split are multiples of 32; 0 = single tensor.
    const bf16_t* x2; int x_split;
    bf16_t* y2; int y_split;
    // needed region (inference, 3x3x3 only): when box_hi[0] > 0 only the bricks that meet the voxel box [box_lo, box_hi) are computed, the rest of y
    // is left untouched.  No statistics.
    int box_lo[3], box_hi[3];
};
int conv_b16_stats_parts(int N, int D, int H, int W, int Cin, int Cout, int planar);
size_t conv_b16_partial_floats(int N, int D, int H, int W, int Cin, int Cout, int planar);
size_t conv_b16_packed_elems(int Cin, int Cout, int planar);
// torch (Cout, Cin, T) fp32 weights -> packed bf16; dgrad = 1: the data-gradient form (taps flipped, roles of Cin / Cout swapped)
int launch_pack_conv_b16(const float* w, bf16_t* out, int Cout, int Cin, int planar, int dgrad, hipStream_t s);
int launch_conv_b16(ConvB16Args a, hipStream_t s);
// every weight packing of a pass in one launch.  mode: 0 conv forward, 1 conv data-gradient, 2 transposed-conv forward, 3 its data gradient;
// w: torch layout (conv (Cout, Cin, T), transposed conv (Cin, Cout, T)); out: the layout the respective kernel reads
constexpr int PACK_B16_MAX_JOBS = 40;
struct PackB16Job { const float* w; bf16_t* out; int Cout, Cin, T, mode; };
int launch_pack_multi_b16(const PackB16Job* jobs, int njobs, hipStream_t s);

// ---------------------------------------------------------------- weight gradient of the same convs (bf16_wgrad.hip)
// part[split][tap][CoPad][CiPad] fp32 (the slab layout of the fp32 path: launch_wgrad_reduce* finish it)
struct WgradB16Args {
    const bf16_t* x; int x_ldc; int Cin;
    const bf16_t* dy; int dy_ldc; int Cout;
    float* part;
    int N, D, H, W;
    int planar;
    int splits;
    const bf16_t* x2; int x_split;         // input channels >= x_split come from x2 (same x_ldc); 0 = single tensor
};
int wgrad_b16_splits(int N, int D, int H, int W, int Cin, int Cout, int planar);
int launch_wgrad_b16(WgradB16Args a, hipStream_t s);
// ... and the 3x3x3 weight gradients of many layers in one stream-K launch (bf16_wgrad.hip; scheme: kernels.h WSkPart): dw = fp32 torch layout (Cout, Cin, 27)
struct WgradSkB16Layer { const bf16_t* x; const bf16_t* x2; int x_split; int x_ldc; int Cin; const bf16_t* dy; int dy_ldc; int Cout; int N, D, H, W; float* dw; };
size_t wgrad_b16_sk_slab_floats(int tile_pairs);
int launch_wgrad_b16_sk(const WgradSkB16Layer* layers, int n, float* slab, size_t slab_floats, hipStream_t s);

// ---------------------------------------------------------------- transposed conv k = s = (sd, 2, 2) (bf16_upconv.hip)
// forward: y[(sd d + kd, 2h + kh, 2w + kw)][co] = bf16(bias[co] + sum_ci x[(d,h,w)][ci] * w[ci][co][tap]) for output voxels inside
// (Do, Ho, Wo) (autocrop of the up-convolved tensor, unet.py:289-299); data gradient: the gather form; weight gradient: slab partials.
struct UpconvB16Args {
    const bf16_t* x; int x_ldc; int Cin;          // low-resolution side (N, D, H, W)
    bf16_t* y; int y_ldc; int Cout;               // high-resolution side (N, Do, Ho, Wo)
    const bf16_t* wt;                              // packed by launch_pack_upconv_b16
    const float* bias;
    int N, D, H, W, Do, Ho, Wo, sd;
    const float* epi_scale; const float* epi_shift;
    float* stats;                                  // forward: (n, mean, M2) records of the stored values
};
int upconv_b16_stats_parts(int N, int D, int H, int W, int sd, int Cin);
size_t upconv_b16_packed_elems(int Cin, int Cout, int sd);
int launch_pack_upconv_b16(const float* w /*torch (Cin, Cout, T)*/, bf16_t* out, int Cin, int Cout, int sd, int dgrad, hipStream_t s);
int launch_upconv_b16_fwd(UpconvB16Args a, hipStream_t s);
// dx[(d,h,w)][ci] = bf16(sum_{tap, co} dy[(sd d + kd, ...)][co] * w[ci][co][tap]); here x/Cin describe dx, y/Cout describe dy
int launch_upconv_b16_dgrad(UpconvB16Args a, hipStream_t s);
int upconv_b16_wgrad_splits(int N, int D, int H, int W);
// part[split][tap][CiPad][CoPad] fp32 (rows = ci: the transposed = 1 form of launch_wgrad_reduce*)
int launch_upconv_b16_wgrad(const bf16_t* x, int x_ldc, int Cin, const bf16_t* dy, int dy_ldc, int Cout, float* part,
                            int N, int D, int H, int W, int Do, int Ho, int Wo, int sd, int splits, hipStream_t s);

// ---------------------------------------------------------------- HBM-bound passes on bf16 tensors (bf16_ew.hip)
// a = bf16(relu(x * scale + shift)) (+ pooled = maxpool_{kd,2,2}(a), ceil mode); x == a-typed raw conv output
int launch_bn_relu_apply_b16(const bf16_t* x, int x_ldc, const float* scale, const float* shift, bf16_t* a, int a_ldc,
                             bf16_t* pooled, int kd, int N, int D, int H, int W, int C, hipStream_t s);
int launch_maxpool_b16(const bf16_t* a, int a_ldc, bf16_t* pooled, int kd, int N, int D, int H, int W, int C, hipStream_t s);
struct BnBwdB16Args {
    const bf16_t* x; int x_ldc;             // raw conv output (BN input)
    const float *mean, *invstd, *gamma, *scale, *shift;
    const bf16_t* g1; int g1_ldc;           // gradient w.r.t. the activation (null: only gpool)
    const bf16_t* gpool;                    // gradient w.r.t. the pooled output (packed, ceil dims) or null
    const bf16_t* pooled;                   // pooled forward output
    int kd, N, D, H, W, C;
    float* part; int parts;                 // [parts][3][C]: sum dz, sum dz*xhat, (apply) sum dx
    const float* coef;                      // apply: [2][C] = (sum dz / n, sum dz*xhat / n)
    bf16_t* dx; int dx_ldc;
    // the last unit: dA is recomputed from the head's (fp32, NCDHW) logits gradient and weights instead of being read
    const float* head_dy; const float* head_w; int head_cout; size_t head_S;
    int pool_one_lane;                      // set by the launcher: one lane per pooling window instead of one per window column
    // head form, criterion variant (head_dy == nullptr, hl_logits set; 2..4 classes): the logits gradient is formed per voxel from the logits the
    // forward wrote, the target and the criterion's finalised coefficients (loss.hip) -- no dlogits tensor; the REDUCE pass then also takes the
    // head's own gradient sums head_part[parts][cout * C + cout] (what conv_final_b16_bwd_kernel computes from a second pass over x)
    const float* hl_logits; const long long* hl_target; const float* hl_cw; const float* hl_coef; const float* hl_gout; float* head_part;
};
int bn_bwd_b16_parts(size_t voxels, int C);
int launch_bn_bwd_b16_reduce(BnBwdB16Args a, hipStream_t s);
int launch_bn_bwd_b16_apply(BnBwdB16Args a, hipStream_t s);

// first conv (Cin = in_channels < 8): x is the module's bf16 input (NDHWC == NCDHW for one channel); direct VALU conv
// first conv with ONE input channel on the matrix cores (bf16_first.hip); same bricks, records and slab as the conv_small_* kernels
bool conv_first_b16_supported(int Cin, int Cout, int planar);
int launch_conv_first_b16_fwd(const bf16_t* x, const float* w, const float* bias, bf16_t* y, int y_ldc, int N, int D, int H, int W, int Cout,
                              const float* epi_scale, const float* epi_shift, float* stats, hipStream_t s);
int launch_conv_first_b16_wgrad(const bf16_t* x, const bf16_t* dy, int dy_ldc, float* part, int N, int D, int H, int W, int Cout, int tiles_per_split, int splits,
                                hipStream_t s);
int conv_small_b16_stats_parts(int N, int D, int H, int W, int planar = 0);
int conv_first_b16_stats_parts(int N, int D, int H, int W, int Cout);      // > 0: launch_conv_first_b16_fwd writes that many records (persistent kernel)
int conv_small_b16_stats_parts2(int N, int D, int H, int W, int planar, int Cin, int Cout);      // record count of launch_conv_small_b16_fwd for this shape
int launch_conv_small_b16_fwd(const bf16_t* x, int Cin, const float* w /*torch (Cout,Cin,T) fp32*/, const float* bias, bf16_t* y, int y_ldc,
                              int N, int D, int H, int W, int Cout, int planar, const float* epi_scale, const float* epi_shift, float* stats, hipStream_t s);
int conv_small_b16_wgrad_splits(int N, int D, int H, int W, int planar = 0);
int launch_conv_small_b16_wgrad(const bf16_t* x, int Cin, const bf16_t* dy, int dy_ldc, float* part /*[splits][T][Cout][Cin]*/,
                                int N, int D, int H, int W, int Cout, int planar, hipStream_t s);
// 1x1x1 head: a (bf16, optionally BN+ReLU applied while loading) -> fp32 NCDHW logits (+ softmax); backward: dW/db partials (+ da)
int launch_conv_final_b16_fwd(const bf16_t* a, int a_ldc, int C, const float* w, const float* bias, float* y_ncdhw, int Cout,
                              size_t voxels_per_sample, int N, int softmax, const float* pro_scale, const float* pro_shift, hipStream_t s, size_t ychan = 0)   /* ychan: voxels between the channel planes of y (0 = S; larger when y is a range of d-planes of a bigger tensor) */;
int conv_final_b16_bwd_parts(size_t total_voxels);
int launch_conv_final_b16_bwd(const bf16_t* a, int a_ldc, int C, const float* w, const float* dy_ncdhw, bf16_t* da, int da_ldc,
                              float* part /*[parts][Cout][C+1]*/, int Cout, size_t voxels_per_sample, int N,
                              const float* pro_scale, const float* pro_shift, hipStream_t s);
// module boundary
int launch_ncdhw_to_ndhwc_b16(const bf16_t* src, bf16_t* dst, int N, int C, size_t S, hipStream_t s);
int launch_ndhwc_to_ncdhw_b16(const bf16_t* src, int src_ldc, bf16_t* dst, int N, int C, size_t S, hipStream_t s);
int launch_f32_to_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t s);
