// Kernel-launcher interface of libe3unet (internal; the public C ABI is include/e3unet.h).
// All tensors are fp32 NDHWC in HBM unless stated otherwise.
#pragma once
#include "common.h"

// ---------------------------------------------------------------- implicit-GEMM conv on f32 MFMA
enum ConvKind {
    CONV_K3 = 0,      // 3x3x3, pad 1           (unet.py:131-149 conv3)
    CONV_K3_PLANAR,   // 1x3x3, pad (0,1,1)     (unet.py:114-128,138-141)
    CONV_POINT,       // 1x1x1 GEMM over voxels (building block of the transposed conv)
};
enum ConvFlags {
    CF_SCATTER_UP = 1,  // POINT only: columns are (tap, co); row p is written to voxel 2p+tap   (ConvTranspose3d fwd)
    CF_GATHER_UP = 2,   // POINT only: K runs over (tap, c); row p reads voxel 2p+tap            (ConvTranspose3d dgrad)
    CF_NO_KSPLIT = 4,   // keep the 256-voxel decomposition (required with a BN+ReLU prologue)
    CF_NO_WINO = 8,     // direct kernels only (set by callers that pass a BN+ReLU prologue)
    CF_NO_PERSIST = 16, // one brick per workgroup even on large grids: a collective may hold CUs while this kernel runs, and a static
                        // 256-workgroup kernel that does not get all 256 CUs at once needs a full second round
    CF_BNRED = 64,      // data-gradient launch that carries the REDUCE pass of the BatchNorm backward in front (ConvArgs::br_*): takes conv3_wino4_kernel
    CF_WINO4 = 128,     // the launch may take the F(2x2x4) Winograd tiles of conv_wino4.hip (eval-mode forwards with the folded epilogue, data gradients:
                        // the caller's statement that no ReLU / arg-max decision of a training step hangs on this launch's rounding)
    CF_SPLITK_OK = 32,  // the caller can run the conv split over its input channels (conv_wino_splitk): count the splits when deciding
                        // whether the Winograd grid is large enough
};

struct ConvArgs {
    const float* x; int x_ldc;   // input view (pointer already offset to its first channel)
    int Cin;                     // input channels per tap (multiple of 8)
    const float* wt;             // packed weights [G][T][NPad][Cin]  (see pack_conv_weights)
    const float* bias;           // [Cout] or null
    float* y; int y_ldc;         // output view
    int N, D, H, W;              // dims of the GEMM-row (voxel) space
    int Do, Ho, Wo;              // SCATTER_UP: dims of y (after autocrop); GATHER_UP: dims of x
    int sd;                      // up-sampling factor along D (2, or 1 for planar blocks); H and W are always 2
    int Cout;                    // real output channels
    int Ncols;                   // GEMM columns: Cout, or taps*Cout for SCATTER_UP
    int NPad;                    // Ncols rounded up to the block's column tile
    const float* pro_scale; const float* pro_shift;  // non-null: x := relu(x*scale[c]+shift[c]) while staging
    const float* epi_scale; const float* epi_shift;  // non-null: y := relu(acc*scale[co]+shift[co])  (bias folded by caller)
    float* stats;                // non-null: per-(tile,channel) (count, mean, M2) of the stored values
    int tilesD, tilesH, tilesW, ntiles;
    int G;                       // gather taps (1 unless GATHER_UP)
    int flags;
    // split-K (Winograd 3x3x3 only, conv_wino_splitk()): the grid is splitk x the bricks; split s reads the channels [s * sk_x, (s + 1) * sk_x)
    // (Cin = sk_x), its own packed weights (wt + s * sk_w) and writes its partial sums to y + s * sk_y.  bias / stats / epilogue must be off.
    int splitk; int sk_x; unsigned sk_w; size_t sk_y;
    // channel-chunked input (conv_wino4.hip only; 0 = the usual [voxel][x_ldc] rows): x is laid out [Cin / 8][N][D][H][W][8], x_chunk = floats between two
    // chunk planes (= N D H W 8).  A halo row of one 8-channel chunk is then ONE contiguous run instead of 32 bytes out of every voxel's row (the staging's
    // line efficiency, profiles/r05_w4_phases.md section 5).  Written by the APPLY pass of the BatchNorm backward (BnBwdArgs::dx_chunk).
    size_t x_chunk;
    // channel-chunked OUTPUT (conv_wino4.hip only, plain store epilogue -- no fused pool / head / statistics): y is written [Ncols / 8][N][D][H][W][8],
    // y_chunk = floats between two chunk planes (= N D H W 8); y_ldc is ignored.  For a tensor whose only consumer stages 8-channel chunks
    // (the conv1 -> conv2 chains of an inference forward).
    size_t y_chunk;
    size_t pool_chunk;           // != 0: the fused pool's output (pool_out) is channel-chunked as well, pool_chunk = N Dp Hp Wp 8
    // store box (conv_wino4.hip only, inference): when sbox_hi[0] > 0 only the voxels [sbox_lo, sbox_hi) (d, h, w) of y are WRITTEN -- everything is still computed
    // (a fused pool sees all of it).  The skip activation of an encoder block when the decoder conv that reads it has a needed region (ConvArgs::box_*): the rest of
    // the tensor is never read.  The other kernels ignore it and store everything.
    int sbox_lo[3], sbox_hi[3];
    // needed region (Winograd 3x3x3 kernels only, inference): when box_hi[0] > 0 only the bricks that meet the voxel box [box_lo, box_hi)
    // (d, h, w) are computed -- the rest of y is left untouched.  The other kernels ignore it and compute everything.  No statistics.
    int box_lo[3], box_hi[3];
    int o_td, o_th, o_tw;        // (set by the launcher: first brick of the box per axis)
    int org_d, org_h, org_w;     // (set by the launcher: voxel origin of brick (0, 0, 0) -- EVEN, so that a needed region keeps the Winograd tile alignment of the whole tensor)
    // compute units to leave alone (0 = use the whole chip; a multiple of 8 = one share per XCD): kernels that size their grid for ONE
    // residency round of one workgroup per CU (the persistent Winograd kernel) launch 256 - cu_reserve workgroups, so that they all fit
    // beside the resident workgroups of a collective running on a side stream (data-parallel backward)
    int cu_reserve;
    // REDUCE pass of a BatchNorm backward fused into the epilogue of a DATA-GRADIENT launch (conv_wino4.hip only, flag CF_BNRED): the conv's output y IS dA,
    // the gradient w.r.t. the activation of the unit in front; with that unit's raw tensor br_x and its constants the store phase -- which holds dA as
    // whole voxel rows -- also takes sum dz and sum dz * xhat per channel (dz = dA * act'(x*scale + shift), xhat = (x - mean) * invstd): one record per
    // workgroup, br_part[row][3][Ncols] rows 0 and 1 like bn_bwd_kernel's, conv_wino4_bnred_parts() rows -- so the separate pass over (dA, x)
    // disappears.  Constant slope activations only.
    const float* br_x; int br_ldc; const float *br_scale, *br_shift, *br_mean, *br_invstd; float br_slope; float* br_part;
    // br_cols != 0 (a multiple of 32): y holds dA of the unit in front only in its first br_cols channels (the data gradient of a decoder block's first conv writes
    // the gradient of the whole concat buffer; the up-convolution's BatchNorm owns the first half); br_part rows are then [3][br_cols]
    int br_cols;
    // inference: nn.MaxPool3d(2, ceil_mode=True) of the (folded-epilogue) output taken in the conv's epilogue -- a Winograd output tile IS a pooling
    // window -- into pool_out [N][ceil(D/2)][ceil(H/2)][ceil(W/2)][Ncols] (packed).  Honoured by the persistent Winograd kernel's transposed form and by conv_wino4.hip;
    // the launcher sets *pool_done = 1 when it took the pooling along (the caller runs the pooling pass otherwise).
    float* pool_out; int* pool_done;
    // inference: the network's 1x1x1 head (conv_final, unet.py:881,912; + Softmax(1) of the Predictor) taken in the epilogue of the LAST 3x3x3 conv -- in
    // the transposed-accumulator form a voxel's 32 activations sit in two lanes -- so the last activation tensor is neither written nor re-read.
    // head_w [head_cout][32], head_b [head_cout] (or null); voxel (d, h, w) inside [head_lo, head_hi) of sample n, class c goes to
    // head_y + n ys[0] + c ys[1] + (d - lo[0]) ys[2] + (h - lo[1]) ys[3] + (w - lo[2]).  Same arithmetic and summation order as conv_final_fwd_kernel
    // (bit-identical results).  Honoured by the persistent kernel's folded-epilogue transposed form and by conv_wino4.hip at Ncols == 32; the launcher sets *head_done = 1.
    const float* head_w; const float* head_b; int head_cout, head_softmax; float* head_y; long long head_ys[4]; int head_lo[3], head_hi[3]; int* head_done;
};
// number of stats records (rows of [Cout][3]) the conv will write
int conv_stats_parts(ConvKind kind, int flags, int N, int D, int H, int W, int sd, int Cin, int ncols);
// chosen work decomposition: ks = 1 (256-voxel bricks) or 4 (64-voxel bricks, waves split K); nt = 32-column tiles per workgroup
void conv_decomposition(ConvKind kind, int flags, int N, int D, int H, int W, int Cin, int ncols, int* ks, int* nt);
int launch_conv_mfma(ConvKind kind, ConvArgs a, hipStream_t s);
int launch_conv3_v3(ConvKind kind, ConvArgs a, int nt, hipStream_t s);   // conv_v3.hip
// Winograd F(2x2x2,3x3x3) path (conv_wino.hip).  conv_use_wino() is THE routing predicate: the weight packer, the
// statistics sizing and launch_conv_mfma() all ask it, with K = GEMM-K channels and ncols = GEMM columns.
bool conv_use_wino(ConvKind kind, int flags, int N, int D, int H, int W, int K, int ncols);
constexpr int WINO_PACK_MAX_JOBS = 40;
struct WinoPackJob { const float* w; float* out; int Cout, Cin, dgrad; int k0 = 0, kn = 0; int layout = 0; };   // kn > 0: only the GEMM-K channels [k0, k0 + kn); layout: conv_wino_layout() of the launch that will read them
// Winograd decomposition of a 3x3x3 launch: 0 = F(2x2x2) tiles in 32-tile bricks (conv_wino.hip), 2 = F(2x2x4) tiles in 16-tile bricks (conv_wino4.hip).
// THE predicate for the packed-weight layout, the statistics sizing and the launcher (K = GEMM-K channels, ncols = GEMM columns; per-sample grid).
int conv_wino_layout(int flags, int D, int H, int W, int K, int ncols, int splitk);
int wino4_stats_parts(int N, int D, int H, int W, int ncols);
// CF_BNRED launches: 0 when the launch cannot carry the reduction (grid does not tile into one column tile per workgroup), else the number of partial rows it writes
int conv_wino4_bnred_parts(int N, int D, int H, int W, int K, int ncols);
int launch_conv3_wino4(ConvArgs a, hipStream_t s);
// split-K factor (1, 2 or 4) of a Winograd 3x3x3 conv whose bricks cannot fill the chip (decided per sample, like conv_use_wino)
int conv_wino_splitk(int D, int H, int W, int K, int ncols);   // (0 if the conv does not use the Winograd kernel even with the splits)
// dst[u][c] (ldc) = sum_s src[s * src_stride + u * C + c] (+ bias[c]); stats (optional): crop_stats_parts(units, C) records per channel
int launch_splitk_reduce(const float* src, int nsrc, size_t src_stride, const float* bias, float* dst, int dst_ldc, int C, size_t units,
                         float* stats, hipStream_t s);
int launch_wino_pack_multi(const WinoPackJob* jobs, int njobs, hipStream_t s);   // the Winograd weight transforms of many layers in one launch
int wino_bricks(int N, int D, int H, int W);
int wino_stats_parts(int N, int D, int H, int W, int Cin, int ncols, int flags);   // statistic records a Winograd launch writes (per brick, or per workgroup of the persistent kernel)
int launch_conv3_wino(ConvArgs a, hipStream_t s);
// planar 1x3x3: Winograd F(2x2,3x3) (conv_wino2d.hip), same contract
bool conv_use_wino2d(ConvKind kind, int flags, int N, int D, int H, int W, int K, int ncols);
int wino2d_bricks(int N, int D, int H, int W);
size_t wino2d_packed_floats(int K, int ncols);
int launch_wino2d_pack(const float* w, float* out, int Cout, int Cin, int dgrad, hipStream_t s);
int launch_conv2_wino(ConvArgs a, hipStream_t s);
// floats of packed-weight workspace a stride-1 conv (fwd or dgrad) may need, whichever algorithm is chosen
size_t conv_packed_floats(ConvKind kind, int K, int ncols);
// packs torch (Cout,Cin,T) weights for the forward (dgrad = 0) or the input-gradient (dgrad = 1) launch of a conv over
// an (N,D,H,W) grid, in the layout of the algorithm conv_use_wino() selects
int launch_pack_conv_auto(ConvKind kind, int dgrad, const float* w, float* out, int Cout, int Cin, int N, int D, int H, int W, int flags, hipStream_t s);
// transposed conv (POINT + SCATTER_UP / GATHER_UP) as a plain LDS-tiled GEMM (upconv_gemm.hip); Cx = channels per voxel of x
bool upconv_gemm_ok(int flags, int Cx, int Cout, int ncols);
int upconv_stats_parts(int N, int D, int H, int W, int sd, int Cx, int Cout);
int launch_upconv_gemm(ConvArgs a, hipStream_t s);
struct WgradArgs;
bool upconv_wgrad_ok(int Cin, int Cout, int sd);                         // weight gradient of the transposed conv, all taps per workgroup
int upconv_wgrad_splits(int N, int D, int H, int W, int Cin, int Cout);
int launch_upconv_wgrad(WgradArgs a, hipStream_t s);
int conv_col_tile(int ncols);  // 32 or 64: column tile the launcher will use for `ncols` GEMM columns

// ---------------------------------------------------------------- weight packing
enum PackMode {
    PACK_CONV_FWD = 0,   // torch (Cout,Cin,T)   -> [T][NPad][Cin]         B[k=ci][n=co] of tap t
    PACK_CONV_DGRAD,     // torch (Cout,Cin,T)   -> [T][NPad(Cin)][Cout]   flipped taps, roles swapped
    PACK_UP_FWD,         // torch (Cin,Cout,T)   -> [1][NPad(T*Cout)][Cin] column n = t*Cout+co
    PACK_UP_DGRAD,       // torch (Cin,Cout,T)   -> [T][NPad(Cin)][Cout]   gather tap t, column n = ci
};
int launch_pack_weights(PackMode mode, const float* w, float* out, int Cout, int Cin, int T, int NPad, hipStream_t s);

// ---------------------------------------------------------------- small-channel direct convs (HBM-bound)
// first layer: Cin < 8 (in_channels of the network), K3 or planar; x is NDHWC (== NCDHW when Cin == 1)
struct ConvSmallArgs {
    const float* x; int Cin;
    const float* w;              // torch layout (Cout,Cin,T)
    const float* bias;
    float* y; int y_ldc;
    int N, D, H, W, Cout;        // Cout multiple of 4, <= 256
    int planar;
    const float* epi_scale; const float* epi_shift;
    float* stats;
    long long xs_n, xs_d, xs_h;  // xs_d != 0: x is a view inside a larger volume (element strides of sample, d-plane, h-row; w stride = Cin)
    size_t y_chunk;              // != 0 (persistent matrix-core kernel only, conv_first_chunk_ok()): y is written channel-chunked, [Cout / 8][N][D][H][W][8] (ConvArgs::y_chunk)
};
bool conv_first_chunk_ok(int N, int D, int H, int W, int planar, int Cin, int Cout);      // the first conv of this shape can write a channel-chunked output
int conv_small_stats_parts(int N, int D, int H, int W, int planar);
int conv_small_stats_parts2(int N, int D, int H, int W, int planar, int Cin, int Cout);      // record count of launch_conv_small_fwd for this shape
int launch_conv_small_fwd(ConvSmallArgs a, hipStream_t s);
// dW partials for the first layer: part[split][T][CoPad=Cout][CiPad=Cin] ; returns number of splits used
int conv_small_wgrad_splits(int N, int D, int H, int W, int planar);
// optional fusion of the BN + ReLU backward (APPLY pass) of the conv's own output into the staging of dy; biaspart [splits][Cout]
struct SmallWgradFuse { const float* x1; int x1_ldc; const float* g; int g_ldc; const float *scale, *shift, *mean, *invstd, *gamma, *coef; float* biaspart; ActArg act; };
int launch_conv_small_wgrad(const float* x, int Cin, const float* dy, int dy_ldc, float* part,
                            int N, int D, int H, int W, int Cout, int planar, hipStream_t s, const SmallWgradFuse* fuse = nullptr);

// final 1x1x1 conv: C (multiple of 4) -> Cout (<= 16); output and its gradient are NCDHW (the module boundary)
int launch_conv_final_fwd(const float* a, int a_ldc, int C, const float* w, const float* bias, float* y_ncdhw,
                          int Cout, size_t voxels_per_sample, int N, int softmax, hipStream_t s,
                          const float* pro_scale = nullptr, const float* pro_shift = nullptr, ActArg pro_act = ActArg(0.f));   // a := act(a*scale + shift) while loading
// box form (e3_unet_forward_tile): only the voxels [lo, lo + size) of the (D, H, W) grid, written to a view of a larger NCDHW volume (element strides n, c, d, h)
int launch_conv_final_fwd_box(const float* a, int a_ldc, int C, const float* w, const float* bias, float* y, int Cout, int N, int D, int H, int W,
                              const int lo[3], const int size[3], const long long ystride[4], int softmax, hipStream_t s,
                              const float* pro_scale = nullptr, const float* pro_shift = nullptr, ActArg pro_act = ActArg(0.f));
int launch_conv_final_fwd_loss(const float* a, int a_ldc, int C, const float* w, const float* bias, float* y_ncdhw, int Cout, size_t voxels_per_sample, int N,
                               hipStream_t s, const float* pro_scale, const float* pro_shift, ActArg pro_act, const long long* target, const float* class_w,
                               float* partial /*[rows][2 + 3 Cout]*/, int max_rows, int* rows);     // head + partial sums of the CE + Dice criterion (loss.hip)
int conv_final_bwd_parts(size_t total_voxels);
int launch_conv_final_bwd(const float* a, int a_ldc, int C, const float* w, const float* dy_ncdhw, float* da, int da_ldc,
                          float* part /*[parts][Cout][C+1]*/, int Cout, size_t voxels_per_sample, int N, hipStream_t s,
                          const float* pro_scale = nullptr, const float* pro_shift = nullptr, ActArg pro_act = ActArg(0.f));

// ---------------------------------------------------------------- weighted CE + Dice criterion (loss.hip)
size_t ce_dice_workspace_floats(int C);
constexpr int CE_DICE_MAX_ROWS = 4096;      // partial rows the workspace holds (ce_dice_fwd_kernel writes 1024, the fused head up to 4096)
// loss and backward coefficients from `rows` partial rows already in the workspace (the second half of launch_ce_dice_fwd)
int launch_ce_dice_finalize(const float* w, int C, int rows, float a, float b, float eps, float smooth, float* workspace, float* loss_out, hipStream_t s);
int launch_ce_dice_sums_rows(int C, int rows, const float* workspace, double* sums, hipStream_t s);
int launch_ce_dice_fwd(const float* logits, const long long* target, const float* w, int C, int N, size_t vps, float a, float b,
                       float eps, float smooth, float* workspace, float* loss_out, hipStream_t s);
int launch_ce_dice_sums(const float* logits, const long long* target, const float* w, int C, int N, size_t vps, float* workspace,
                        double* sums, hipStream_t s);
int launch_ce_dice_from_sums(const double* sums, const float* w, int C, float a, float b, float eps, float smooth, float* workspace,
                             float* loss_out, hipStream_t s);
int launch_ce_dice_bwd(const float* logits, const long long* target, const float* w, int C, int N, size_t vps,
                       const float* workspace, const float* gout, float* dlogits, hipStream_t s);

// up_mode='resizeconv_*' building blocks (elementwise.hip): nearest up-sampling by (sd,2,2) and its backward, the autocrop of the
// up-convolved tensor fused with the statistics of the cropped tensor (records [crop_stats_parts][C][3]) and its backward
int launch_upsample_nearest(const float* x, int x_ldc, float* out, int C, int N, int Di, int Hi, int Wi, int sd, hipStream_t s, int linear = 0);   // linear: tri-/bilinear, align_corners=False
int launch_downsample_sum(const float* g, float* dx, int dx_ldc, int C, int N, int Di, int Hi, int Wi, int sd, hipStream_t s, int linear = 0);
int launch_embed_center_tap(const float* w1, float* wT, size_t pairs, int T, hipStream_t s);     // (Cout*Cin) 1x1x1 weights -> centre tap of T-tap kernels
int launch_extract_center_tap(const float* gT, float* g1, size_t pairs, int T, hipStream_t s);
int crop_stats_parts(size_t voxels, int C);
int launch_crop_stats(const float* src, float* dst, int C, int N, int Ds, int Hs, int Ws, int Dd, int Hd, int Wd, float* stats, hipStream_t s,
                      int od = 0, int oh = 0, int ow = 0);            // box at (od, oh, ow) inside src
int launch_pad_box(const float* src, float* dst, int C, int N, int Ds, int Hs, int Ws, int Dd, int Hd, int Wd, hipStream_t s,
                   int od = 0, int oh = 0, int ow = 0, int src_ldc = 0);  // src at (od, oh, ow) inside dst, zeros elsewhere
int launch_crop_copy(const float* src, float* dst, int dst_ldc, int C, int N, int Ds, int Hs, int Ws, int Dd, int Hd, int Wd, int od, int oh, int ow, hipStream_t s);
constexpr int FOLD_MAX_JOBS = 40;
struct FoldJob { const float *gamma, *beta, *rm, *rv, *bias; float *scale, *shift; int C; };   // gamma == nullptr: no norm (scale 1, shift bias)
int launch_fold_multi(const FoldJob* jobs, int njobs, float eps, hipStream_t s);   // eval-mode BN folds / bias folds of all conv units at once
constexpr int COLSUM_MAX_JOBS = 40;
struct ColsumJob { const float* part; int parts, stride, offset, C; float* out; };
int launch_colsum_multi(const ColsumJob* jobs, int njobs, hipStream_t s);   // out[c] = sum_p part[p*stride + offset + c], many at once
// PReLU: dslope[0] = sum over channels and partial rows of row 2 of the REDUCE pass' `part` (fixed order); tmp: [C] floats
int launch_prelu_dslope(const float* part, int parts, int C, float* tmp, float* dslope, hipStream_t s);
int launch_gn_bwd_coef(const float* dgamma, const float* dbeta, const float* gamma, const float* invstd, int C, int group, float inv_n,
                       float* coef, hipStream_t s);   // GroupNorm: rewrites coef after launch_bn_bwd_finalize
int launch_fill(float* p, float v, size_t n, hipStream_t s);
int launch_add_views(const float* a, int a_ldc, const float* b, int b_ldc, float* out, int out_ldc, size_t vox, int C, hipStream_t s);   // out = a + b
int launch_bias_fold(const float* conv_bias, float* scale, float* shift, int C, hipStream_t s);   // scale = 1, shift = bias

// ---------------------------------------------------------------- optimizer (optim.hip)
int launch_swa(int n_tensors, void* const* params, void* const* bufs, const long long* numels, double decay, int swap, hipStream_t s);
size_t adamw_state_floats(int n_tensors, const long long* numels);
size_t adamw_state_offset(int n_tensors, const long long* numels, int tensor);
int launch_adamw(int n_tensors, void* const* params, void* const* grads, const long long* numels, float* exp_avg, float* exp_avg_sq,
                 float* step, float* coef, double lr, double beta1, double beta2, double eps, double weight_decay,
                 const float* grad_scale, const float* found_inf, hipStream_t s, int bf16 = 0);

// ---------------------------------------------------------------- wgrad on f32 MFMA
struct WgradArgs {
    const float* x; int x_ldc; int Cin;     // conv input activation view
    const float* dy; int dy_ldc; int Cout;  // gradient w.r.t. conv output
    size_t dy_chunk;                        // != 0 (wgrad_wino.hip only): dy is channel-chunked, [Cout / 8][N][D][H][W][8], dy_chunk = N D H W 8 (ConvArgs::x_chunk)
    float* part;                            // [splits][T][CoPad][CiPad]
    int N, D, H, W;
    int CoPad, CiPad, splits;
    // POINT (transposed conv): x has dims (N,D,H,W); dy has dims (N,Do,Ho,Wo), row p pairs with dy voxel 2p+tap
    int Do, Ho, Wo, sd;
    int cu_reserve;                         // as ConvArgs::cu_reserve: the one-round kernels split the voxels over 256 - cu_reserve workgroups (3x3x3 Winograd kernel only)
};
// channel-chunked tensors ([C / 8][N][D][H][W][8]: ConvArgs::x_chunk, WgradArgs::dy_chunk, BnBwdArgs::dx_chunk): the chunk planes are addressed through
// 32-bit buffer offsets, so the whole tensor has to stay below 2 GiB
inline bool chunked_layout_ok(size_t vox, int C) { return C % 8 == 0 && vox * (size_t)C * 4 < 0x7fffffffu; }
int wgrad_splits(ConvKind kind, int N, int D, int H, int W, int Cin, int Cout, int cu_reserve = 0);
int launch_wgrad_mfma(ConvKind kind, WgradArgs a, hipStream_t s);
// Winograd F(3x3x3, 2x2x2) variant for CONV_K3 (wgrad_wino.hip): same bricks, splits and partial-slab layout
bool wgrad_use_wino(ConvKind kind);
int launch_wgrad_wino(WgradArgs a, int tD, int tH, int tW, int tps, int co_tiles, int ci_tiles, int splits, hipStream_t s);
// ... and the weight gradients of MANY layers in one stream-K launch (wgrad_wino.hip, round 6): dw = torch layout (Cout, Cin, 27), written directly
struct WgradSkLayer { const float* x; int x_ldc; int Cin; const float* dy; int dy_ldc; int Cout; size_t dy_chunk; int N, D, H, W; float* dw; };
size_t wgrad_wino_sk_slab_floats(int tile_pairs);      // tile_pairs = sum over the layers of ceil(Cout / 32) * ceil(Cin / 32)
// the partition of such a launch (shared by the fp32 and the 16-bit kernels and their common reduction).  A layer's bricks are cut into `nblocks` blocks of B bricks
// (about one workgroup's share); its units of work are ordered (block, tile pair, brick of the block): the workgroups that run side by side work on the SAME bricks for
// different tile pairs, so the X / dY lines a brick needs are fetched from HBM once and shared through the XCD's L2 (with the order (tile pair, brick) every tile pair
// re-read its operands from HBM long after the others: 3.3 GB of reads for 0.8 GB of tensors, profiles/r06_pmc_f32.md of the first version).  A (block, tile pair)
// cell has the global index c = c0 + block * tps + tile pair; workgroup w of nwg owns the units [w q + min(w, r), ...); the partial tile of (workgroup w, cell c) is
// slab w + c -- both grow along the list, so no two segments share a slab.
constexpr int WSK_MAX_LAYERS = 16;
struct WSkPartLayer { float* dw; int Cin, Cout, ci_tiles, tps, nbricks, B, nblocks; unsigned g0, t0, c0; };
struct WSkPart { WSkPartLayer L[WSK_MAX_LAYERS]; int n; unsigned total, q, r, ntp, ncells, nwg; float* slab; };
// unit `rel` (relative to the layer's g0) -> block, tile pair, first brick, bricks left in the cell
struct WSkUnit { unsigned block, tp, brick, left; };
__host__ __device__ inline WSkUnit wsk_unit(const WSkPartLayer& L, unsigned rel) {
    const unsigned per = (unsigned)L.B * (unsigned)L.tps;
    unsigned b = rel / per;
    if (b >= (unsigned)L.nblocks) b = (unsigned)L.nblocks - 1;           // (the last block may be the short one: its cells are smaller)
    const unsigned bsz = b + 1 == (unsigned)L.nblocks ? (unsigned)L.nbricks - b * (unsigned)L.B : (unsigned)L.B;
    const unsigned r2 = rel - b * per, tp = r2 / bsz, off = r2 - tp * bsz;
    return WSkUnit{b, tp, b * (unsigned)L.B + off, bsz - off};
}
size_t wgrad_sk_slab_floats(int tile_pairs, int workgroups);      // slabs of ANY partition: ids w + c < 3 workgroups + tile pairs (cells < total / q + tile pairs < 2 workgroups + tile pairs)
int wgrad_sk_partition(WSkPart& p, int n, const int* Cin, const int* Cout, const int* nbricks, float* const* dw, int workgroups, float* slab, size_t slab_floats);
int launch_wgrad_sk_reduce(const WSkPart& p, hipStream_t s);
int launch_wgrad_wino_sk(const WgradSkLayer* layers, int n, float* slab, size_t slab_floats, hipStream_t s);
// planar: Winograd F(3x3, 2x2) (wgrad_wino2d.hip); 64-channel granularity on the co side, 32 on the ci side
bool wgrad_use_wino2d(ConvKind kind, int Cin, int Cout);
int wgrad_wino2d_splits(int N, int D, int H, int W, int Cin, int Cout);
int launch_wgrad_wino2d(WgradArgs a, hipStream_t s);
// out (torch layout): transposed==0: (Cout,Cin,T) from part rows=co, cols=ci ; transposed==1: (Cin,Cout,T), part rows=ci, cols=co
struct WgradReduceJob { const float* part; float* out; int splits, T, RPad, CPad, R, C; };
int launch_wgrad_reduce_multi(const WgradReduceJob* jobs, int njobs, hipStream_t s);   // many slab reductions in one launch (same results)
int launch_wgrad_reduce(const float* part, float* out, int splits, int T, int RPad, int CPad, int R, int C, hipStream_t s);

// ---------------------------------------------------------------- batch-norm / relu / pool (HBM-bound elementwise)
// merges `parts` records of (count, mean, M2) per channel -> mean, invstd, scale, shift; updates running stats
struct BnFinalizeArgs {
    const float* stats; int parts; int C;
    const float* gamma; const float* beta;
    float* running_mean; float* running_var;   // may be null
    float momentum; float eps;
    float* mean; float* invstd; float* scale; float* shift;   // each [C]
    float* scratch;                           // optional: BN_PRERED * C * 3 floats; many records are first merged into BN_PRERED coalesced partials
    int group;                                // 0/1: statistics per channel (BatchNorm); > 1: merged over groups of `group` consecutive channels (GroupNorm)
};
constexpr int BN_PRERED = 64;
int launch_bn_finalize(BnFinalizeArgs a, hipStream_t s);
// eval mode: scale = gamma/sqrt(rv+eps); shift = beta + (conv_bias - rm)*scale   (BN folded into the conv epilogue)
int launch_bn_fold(const float* gamma, const float* beta, const float* rm, const float* rv, const float* conv_bias,
                   float eps, float* scale, float* shift, int C, hipStream_t s);
int launch_bn_frozen(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                     float* mean, float* invstd, float* scale, float* shift, int C, hipStream_t s);   // running statistics as (mean, invstd, scale, shift)
// a = relu(x*scale+shift) written to `a` (any ldc); optionally also p = maxpool_{kd,2,2}(a), ceil mode
int launch_bn_relu_apply(const float* x, int x_ldc, const float* scale, const float* shift, float* a, int a_ldc,
                         float* pooled /*null or packed NDHWC (ceil dims)*/, int kd,
                         int N, int D, int H, int W, int C, hipStream_t s, ActArg slope = ActArg(0.f));   // slope: 0 ReLU, 0.1 LeakyReLU, 1 identity, ...
int launch_maxpool(const float* a, int a_ldc, float* pooled, int kd, int N, int D, int H, int W, int C, hipStream_t s);

// backward of x -> (scale,shift) -> relu, with dA = g1 (+ unpool(gpool) through the max-pool of `a`)
struct BnBwdArgs {
    const float* x; int x_ldc;            // raw conv output (BN input)
    const float* mean; const float* invstd; const float* gamma; const float* scale; const float* shift;
    const float* g1; int g1_ldc;          // gradient w.r.t. relu output (may be null if gpool given)
    const float* gpool;                   // gradient w.r.t. pooled output (packed, ceil dims) or null
    const float* a; int a_ldc;            // relu output (needed for the pool arg-max) or null
    const float* pooled;                  // pooled forward output (packed)
    int kd;
    int N, D, H, W, C;
    float* part;                          // [parts][3][C]: sum dz, sum dz*xhat, (apply pass) sum dx
    int parts;
    const float* coef;                    // apply pass: [4][C] = (c1 = sum dz / n, c2 = sum dz*xhat / n, k1, k2): dx = g*istd*(dz - c1 - xh*c2) - (k1 + xh*k2)
    float* dx; int dx_ldc;                // apply pass output
    size_t dx_chunk;                      // != 0: the apply pass writes dx channel-chunked, [C / 8][N][D][H][W][8] (dx_chunk = N D H W 8), for consumers that stage 8-channel chunks
    size_t nt_bytes;                      // set by the launcher: tensors above this size are read with non-temporal loads
    // non-pool path, g1 == nullptr: the incoming gradient is that of the 1x1x1 head, g[v][c] = sum_co head_dy[n][co][sp] * head_w[co][c],
    // recomputed from the (tiny) NCDHW logits gradient instead of being written by conv_final_bwd and re-read twice
    const float* head_dy; const float* head_w; int head_cout; size_t head_S;
    // ... or (head_dy == nullptr, hl_logits set; head_cout <= 4) formed from the criterion e3_unet_forward_loss evaluated in the head: the logits it
    // wrote, the target and the finalised coefficients (loss.hip: coef[0] = a / Ws, [1 + k], [1 + C + k]) -- no dlogits tensor exists.
    // hl_gout: device scalar d(final)/d(loss) or null (= 1).
    const float* hl_logits; const long long* hl_target; const float* hl_cw; const float* hl_coef; const float* hl_gout;
    // REDUCE pass of the head form, head_part != nullptr: also the head's own gradients (what conv_final_bwd_kernel computes from a second
    // pass over x): partial sums [parts][head_cout * C + head_cout] of dW[co][c] = sum g[co] * act(z)[c] and db[co] = sum g[co]
    float* head_part;
    ActArg act;                           // activation: slope 0 = ReLU, > 0 = LeakyReLU(slope), 1 = identity, ACT_SILU; or a PReLU pointer
    // PReLU: the REDUCE pass also writes sum dA*min(z,0) per channel into row 2 of `part` (the APPLY pass overwrites it later)
};
int bn_bwd_parts(size_t voxels, int C);
int launch_bn_bwd_reduce(BnBwdArgs a, hipStream_t s);
// sums `parts` rows of `rows` x C -> out rows; used for (dgamma,dbeta) + coefficients and for bias grads
int launch_bn_bwd_finalize(const float* part, int parts, int C, float inv_n, float* dgamma, float* dbeta, float* coef, hipStream_t s);
int launch_bn_bwd_apply(BnBwdArgs a, hipStream_t s);
int launch_colsum_finalize(const float* part, int parts, int part_stride, int offset, int C, float* out, hipStream_t s);

// ---------------------------------------------------------------- GridAttention of the decoder blocks (attention.hip; unet.py:452-541)
struct AttDims {
    int N, C;               // batch, channels of the skip tensor x (the gating signal g has 2C, the gate works on C/2)
    int D, H, W;            // grid of x (after autocrop)
    int d, h, w;            // grid of theta(x): floor(D / sd), H / 2, W / 2
    int gd, gh, gw;         // grid of g
    int sd;                 // depth stride of theta: 2 (dim=3) or 1 (dim=2: 2x2 kernel on a depth-1 volume)
};
struct AttParams { float *w_w, *w_b, *theta_w, *phi_w, *phi_b, *psi_w, *psi_b; };   // torch layouts; also used for the gradients
bool att_resized(const AttDims& d);           // phi(g) needs the linear resize (the grids of theta(x) and g differ)
size_t att_part_floats(const AttDims& d);     // scratch for the split partial sums of the weight gradients
int launch_att_gate_fwd(const AttDims& d, const float* x, int ldx, const float* g, int ldg, const AttParams& p, float* f, float* sgm, float* att,
                        float* phi_tmp, float* phi_res, hipStream_t s);
int launch_att_out_fwd(const AttDims& d, const float* x, int ldx, const float* att, const AttParams& p, const float* epi_scale, const float* epi_shift,
                       float* out, int ldo, hipStream_t s);
int launch_att_bwd(const AttDims& d, const float* dz, const float* x, int ldx, const float* g, int ldg, const float* f, const float* sgm, const float* att,
                   const AttParams& p, const AttParams& grad, float* dx, float* dphi, float* tmp_fine, float* tmp_coarse, float* df, float* part,
                   hipStream_t s);
int launch_att_bwd_gate_input(const AttDims& d, const float* dphi, const AttParams& p, float* dg, int ldg, hipStream_t s);
// plain 1x1x1 convolutions over voxel rows on the same GEMM kernels (ResUNet shortcut projections); w: torch (Cout, Cin)
size_t pw_part_floats(size_t rows, int Cout, int Cin);
int launch_pw_fwd(const float* x, int ldx, int Cin, const float* w, const float* b, float* out, int ldo, int Cout, size_t rows, hipStream_t s);
int launch_pw_dgrad_acc(const float* dy, int ldy, int Cout, const float* w, float* dx, int ldx, int Cin, size_t rows, hipStream_t s);   // dx += dy . w
int launch_pw_wgrad(const float* dy, int ldy, int Cout, const float* x, int ldx, int Cin, float* part, float* dw, float* db, size_t rows, hipStream_t s);

// ---------------------------------------------------------------- layout helpers
int launch_ncdhw_to_ndhwc(const float* src, float* dst, int N, int C, size_t S, hipStream_t s);
int launch_ndhwc_to_ncdhw(const float* src, int src_ldc, float* dst, int N, int C, size_t S, hipStream_t s);
