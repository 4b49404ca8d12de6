/*
 * e3unet.h -- C ABI of libe3unet.so: the MI355X (gfx950) implementation of elektronn3's 3D U-Net hot path.
 *
 * elektronn3 has no FFI or plugin registry: its boundary for this path is Python duck typing on
 * torch.nn.Module.  This header is what a binding for that boundary calls: plain pointers and
 * sizes, no torch types.  Every entry point cites the reference code it replaces (paths relative to the
 * reference repo root).  The reference-side binding (ctypes) is shown in INTEGRATION.md and implemented in
 * elektronn3_amd/_lib.py.
 *
 * Conventions
 *   - All pointers are DEVICE pointers owned by the caller (e.g. the torch caching allocator) unless a parameter
 *     is documented as host memory.  The library never allocates or frees device memory.
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream).  Calls only enqueue work and
 *     return; they never synchronise.  Results are ordered on that stream.
 *   - Return value: 0 (E3_OK) on success, otherwise an E3_ERR_* code; e3_last_error() returns a message for the
 *     calling thread.  No process-global mutable state besides that thread-local string, so replicas driven from
 *     different Python threads (nn.DataParallel, benchmark/train_benchmark.py:109-110) do not interfere.
 *   - Activations inside the library are fp32 NDHWC ("channels-last-3d").  Module-boundary tensors (network
 *     input, logits, their gradients) are contiguous NCDHW like the reference's.
 *   - "view": (ptr, ldc) addresses C channels at `ptr` inside voxel rows of `ldc` floats, so an op can read or
 *     write one half of a concat buffer (replaces torch.cat, elektronn3/models/unet.py:398-399).
 */
#ifndef E3UNET_H
#define E3UNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define E3_OK 0
#define E3_ERR_INVALID 1
#define E3_ERR_HIP 2
#define E3_ERR_UNSUPPORTED 3
#define E3_ERR_WORKSPACE 4

/* Message of the last failing call on this thread ("" if none). */
const char* e3_last_error(void);
/* Library version string, e.g. "e3unet 0.1 gfx950". */
const char* e3_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Whole-network API: UNet.forward / its autograd backward
 *   replaces elektronn3/models/unet.py:894-916 (UNet.forward), :244-253 (DownConv.forward), :384-408
 *   (UpConv.forward), :256-325 (autocrop) and the autograd graph torch builds for them (Trainer._train_step,
 *   elektronn3/training/trainer.py:520,539).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct e3_unet_cfg {
    int32_t in_channels;    /* UNet(in_channels=...)            unet.py:757 */
    int32_t out_channels;   /* UNet(out_channels=...), 1..16    unet.py:758 */
    int32_t n_blocks;       /* UNet(n_blocks=...), 1..8         unet.py:759 */
    int32_t start_filts;    /* UNet(start_filts=...), multiple of 8   unet.py:760 */
    uint32_t planar_mask;   /* bit i set <=> i in planar_blocks unet.py:763,827 */
    int32_t normalization;  /* 1 = 'batch', 0 = 'none' (nn.Identity, unet.py:77-80), 2 = 'group<G>' (nn.GroupNorm(num_groups, C): affine, no running
                             * statistics, per-SAMPLE statistics => call with N = 1; 'instance' = 'batch' with N = 1 and unit affine constants) */
    float bn_eps;           /* nn.BatchNorm3d eps (1e-5) */
    int32_t full_norm;      /* 1: a norm after every (transposed) conv; 0: only after the last conv of a block (unet.py:238-242,369-375) */
    int32_t merge_add;      /* 0: merge_mode='concat' (torch.cat((up, skip), 1)); 1: merge_mode='add' (up + skip), unet.py:398-401 */
    int32_t num_groups;     /* normalization = 2: number of groups (8 for 'group', G for 'group<G>', unet.py:81-90) */
    int32_t up_resize;      /* 0: up_mode='transpose' (nn.ConvTranspose3d k=s=2); 1: 'resizeconv_nearest', 2: 'resizeconv_linear', 3 / 4: the same with a 1x1x1 conv ('resizeconv_nearest1' / 'resizeconv_linear1') (ResizeConv: nn.Upsample(nearest | tri-/bilinear, align_corners=False) + conv3,
                             * unet.py:152-176,411-449; parameters 'up_convs.i.upconv.conv.weight/bias') */
    int32_t conv_valid;     /* 0: conv_mode='same' (padding 1); 1: conv_mode='valid' (padding 0 in the 3x3x3 convs of the blocks, unet.py:217,347: the
                             * output is smaller than the input, see e3_unet_out_dims) */
    float act_slope;        /* activation (get_activation, unet.py:183-199): 0 = 'relu', 0.1 = 'leaky' (LeakyReLU(0.1)), 1 = 'lin' (identity), 2 = 'silu',
                             * 3 = 'prelu' (nn.PReLU(1) per activation: parameters '<block>.act<k>.weight' join the table) */
    int32_t attention;      /* 0: attention=False (DummyAttention); 1: attention=True, dim=3 (GridAttention with a 2x2x2 theta, unet.py:376-379,452-541);
                             * 2: attention=True, dim=2 (2x2 theta on the depth-1 volume).  Parameters 'up_convs.i.attention.{w.0,w.1,theta,phi,psi}.*'
                             * join the table after the block's up-convolution; fp32 path only */
    int32_t resunet;        /* 0: elektronn3.models.unet.UNet; 1: elektronn3.models.resunet.UNet (ConvBlock / DownBlock / UpBlock, resunet.py:212-457:
                             * parameters '<block>.convs.<k>.{conv1,norm1,act1,conv2,norm2,act2,proj}.*', a norm after every conv, full_norm ignored) */
    int32_t enc_res_blocks; /* resunet: ConvBlocks per encoder block = max(1, enc_res_blocks); >= 1: with residual shortcuts (not from the input image) */
    int32_t dec_res_blocks; /* resunet: the same for the decoder blocks (resunet.py:705-707,809-810) */
} e3_unet_cfg;

typedef struct e3_unet_plan e3_unet_plan;

int e3_unet_plan_create(const e3_unet_cfg* cfg, e3_unet_plan** out);
void e3_unet_plan_destroy(e3_unet_plan* plan);

/* The parameter table: pointers are passed in THIS order.  Names are the reference's state_dict keys
 * (e.g. "down_convs.0.conv1.weight", "up_convs.1.norm0.running_var"); num_batches_tracked is not part of the
 * table (the binding increments it).  kind: 0 = trainable parameter, 1 = buffer (running statistic). */
int e3_unet_param_count(const e3_unet_plan* plan);
int e3_unet_param_info(const e3_unet_plan* plan, int index, char* name, int name_len, int64_t* numel, int* kind);
/* Number of BatchNorm layers, in table order (momenta[] below has one entry per BN). */
int e3_unet_bn_count(const e3_unet_plan* plan);

/* Bytes of caller-provided device memory needed for an (N, in_channels, D, H, W) batch.
 *   saved_bytes:   activations kept from a training forward for the backward (0 for inference)
 *   scratch_bytes: temporaries; may be shared by consecutive calls on one stream */
/* Spatial size of the logits for a (D, H, W) input: equal to the input with conv_mode='same'; smaller with 'valid'. */
int e3_unet_out_dims(const e3_unet_plan* plan, int D, int H, int W, int* Do, int* Ho, int* Wo);
int e3_unet_sizes(const e3_unet_plan* plan, int N, int D, int H, int W, int training,
                  size_t* saved_bytes, size_t* scratch_bytes);

#define E3_FWD_TRAINING 1u  /* batch statistics + running-stat update (module.training) */
#define E3_FWD_SOFTMAX 2u   /* y = softmax over channels (Predictor's nn.Sequential(model, nn.Softmax(1)), inference.py:443-444) */
#define E3_FWD_FROZEN_BN 4u /* with E3_FWD_TRAINING: save activations for a backward, but normalise with the RUNNING statistics and leave them
                             * untouched -- autograd through a module in eval mode (frozen-BN fine-tuning, training/recalibration.py:53-73) */
#define E3_FWD_REUSE_PACKED 8u /* inference only (ignored with E3_FWD_TRAINING): the caller states that the PREVIOUS call on this plan was an inference
                                * forward with the same scratch buffer, N, D, H, W and parameter VALUES, and that nothing has written the scratch since: the
                                * packed / Winograd-transformed weights and the folded BatchNorm constants are used as they lie (no packing launches).  The
                                * tile loop of inference.Predictor (inference.py:153-199) sets it from its second tile on. */
#define E3_BWD_FROZEN_BN 1u /* e3_unet_backward2: the matching backward (BatchNorm statistics are constants) */
/* e3_unet_backward2, with a bucket_event: compute units (a multiple of 8, at most 128: one share per XCD) that the kernels launched AFTER the
 * event leave alone -- the kernels that fill the chip with exactly one workgroup per CU (persistent Winograd data gradients, Winograd weight
 * gradients) launch 256 - n workgroups, so that all of them are resident beside the workgroups of the collective the caller starts on a
 * side stream at the event (one that found no free CU would wait for a whole round: measured +55 % on the step with ONE foreign wave).
 * Results do not depend on it bit for bit only for the data gradients; the weight-gradient partial sums are split differently. */
#define E3_BWD_CU_RESERVE(n) ((((uint32_t)(n) >> 3) & 0x1fu) << 8)

/* y[N,out,D,H,W] = UNet(x[N,in,D,H,W]).
 *   params : e3_unet_param_count() device pointers in table order
 *   momenta: HOST array, one exponential-average factor per BN (module.momentum at call time; SWA.bn_update
 *            mutates it, training/swa.py:317-342).  Ignored unless E3_FWD_TRAINING.
 *   saved  : written when training (needed by e3_unet_backward), may be NULL otherwise */
int e3_unet_forward(e3_unet_plan* plan, void* stream, const float* x, int N, int D, int H, int W,
                    void* const* params, const float* momenta, float* y,
                    void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, uint32_t flags);

/* e3_unet_forward with the training example's criterion evaluated INSIDE the 1x1x1 head (SURVEY.md 8f rank 1 "loss on device fused with
 * conv_final"; replaces criterion(model(x), target) of training/trainer.py:520-524 with CombinedLoss([CrossEntropyLoss(w),
 * DiceLoss(apply_softmax=True, w)], (ce_weight, dice_weight)), modules/loss.py:19-49,158-234): y = the logits as before; *loss_out
 * (device scalar) = the loss of (y, target); `workspace` (e3_ce_dice_workspace_bytes(out_channels) bytes) receives the coefficients
 * that e3_ce_dice_bwd(logits = y, target, workspace, gout) turns into dLoss/dy for e3_unet_backward.  target: [N, D', H', W'] int64
 * class indices on the grid of y.  No E3_FWD_SOFTMAX. */
typedef struct e3_ce_dice_args {
    const long long* target; const float* class_weight /* [out_channels] or NULL */;
    float ce_weight, dice_weight, eps, smooth;
    void* workspace; size_t workspace_bytes; float* loss_out;
    /* a minibatch sharded over ranks (one process per GPU; the reference's loss is ONE loss over the gathered batch, trainer.py:520-524):
     * non-NULL -> e3_unet_forward_loss stops at the criterion's 2 + 3 out_channels sums of THIS shard (device doubles, the layout of
     * e3_ce_dice_sums); the caller adds them over the ranks (one all-reduce) and e3_ce_dice_from_sums(sums, ..., workspace, loss_out) writes
     * the batch-wide loss and the coefficients e3_unet_backward_loss reads.  loss_out is not touched by the forward then. */
    double* sums_out;
} e3_ce_dice_args;
int e3_unet_forward_loss(e3_unet_plan* plan, void* stream, const float* x, int N, int D, int H, int W,
                         void* const* params, const float* momenta, float* y,
                         void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, uint32_t flags, const e3_ce_dice_args* loss);

/* Inference forward (no E3_FWD_TRAINING) of which the caller keeps only the output voxels [roi[0..2], roi[3..5]) (d, h, w on the grid of
 * y) -- the tile loop of inference.Predictor, which crops the overlap off every tile's output (inference.py:496-525, tiled_apply :134-199):
 * y holds the same values as after e3_unet_forward INSIDE that region and unspecified values outside it.  The decoder's 3x3x3 convs then
 * compute only the bricks that the region needs (it grows by one voxel per conv and halves per transposed conv on the way back; the
 * encoder is computed in full).  Configurations without the facility (valid convs, attention, ResizeConv, ResUNet, planar blocks) simply
 * compute everything.  The 16-bit executors have the same entry point (e3_unet_forward_roi_bf16 / _f16 below). */
int e3_unet_forward_roi(e3_unet_plan* plan, void* stream, const float* x, int N, int D, int H, int W,
                        void* const* params, float* y, void* scratch, size_t scratch_bytes, uint32_t flags, const int roi[6]);

/* Gradients of a scalar loss w.r.t. every trainable parameter (and optionally x), given dy = dLoss/dy.
 *   grads : table-ordered device pointers (entries of buffers are ignored, may be NULL); every trainable entry
 *           is OVERWRITTEN with the gradient (accumulation into .grad stays with autograd)
 *   dx    : NULL or [N,in,D,H,W]
 *   bucket_event / bucket_after_down_block: if bucket_event (a hipEvent_t) is non-NULL it is recorded on
 *           `stream` as soon as the gradients of every layer EXCEPT down_convs[0..bucket_after_down_block-1] are
 *           complete, so a data-parallel binding can start the RCCL all-reduce of that bucket on a side stream
 *           while the rest of the backward runs (replaces nn.DataParallel's reduce_add_coalesced,
 *           benchmark/train_benchmark.py:109-110). */
int e3_unet_backward(e3_unet_plan* plan, void* stream, const float* dy, const float* x, int N, int D, int H, int W,
                     void* const* params, void* const* grads, float* dx,
                     void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                     void* bucket_event, int bucket_after_down_block);

/* e3_unet_backward with flags (E3_BWD_FROZEN_BN). */
int e3_unet_backward2(e3_unet_plan* plan, void* stream, const float* dy, const float* x, int N, int D, int H, int W,
                      void* const* params, void* const* grads, float* dx,
                      void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                      void* bucket_event, int bucket_after_down_block, uint32_t flags);

/* e3_unet_backward2 behind e3_unet_forward_loss: the loss IS the criterion of that forward, so dLoss/dlogits never becomes a tensor -- the
 * backward kernels of the head form it in registers from y (the logits that forward wrote), loss->target, loss->class_weight and the
 * coefficients in loss->workspace, and the head's own weight / bias gradients come out of the pass over the network's last raw tensor that
 * the BatchNorm backward makes anyway (no e3_ce_dice_bwd pass, no separate head-backward pass; criterion(out, target).backward() of
 * training/trainer.py:520-524,539).  gout: device scalar d(final)/d(loss), or NULL = 1 (a GradScaler's scale).  2..4 classes and a
 * normalisation in front of the head; anything else returns E3_ERR_UNSUPPORTED and the caller takes e3_ce_dice_bwd + e3_unet_backward2. */
int e3_unet_backward_loss(e3_unet_plan* plan, void* stream, const float* y, const e3_ce_dice_args* loss, const float* gout, const float* x,
                          int N, int D, int H, int W, void* const* params, void* const* grads, float* dx,
                          void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                          void* bucket_event, int bucket_after_down_block, uint32_t flags);

/* Per-layer profiling hook used by bench.py for the roofline line: when `layer` >= 0, hipEvents are recorded
 * around that layer's dominant kernel in every subsequent forward (which=0), dgrad (1) or wgrad (2); read the
 * mean duration of all recorded launches (and reset) with e3_unet_profile_read.  `layer` indexes the conv list
 * returned by e3_unet_conv_info. */
int e3_unet_conv_count(const e3_unet_plan* plan);
int e3_unet_conv_info(const e3_unet_plan* plan, int layer, char* name, int name_len, int* cin, int* cout, int* taps, int* level);
/* nn.RReLU in TRAIN mode (activation='rrelu', unet.py:183-199 get_activation -> nn.RReLU(): lower 1/8, upper 1/3): the slope of every
 * negative pre-activation is drawn from U(lower, upper).  seed != 0 arms it for the following TRAINING forwards of this plan (fp32 path;
 * every unit derives its own stream, the draw is a hash of (seed, unit, element index)); the backward of such a forward must run with the SAME
 * seed set (the slopes are recomputed, no noise tensor is stored).  seed == 0: the fixed slope cfg.act_slope (eval mode: (lower+upper)/2). */
int e3_unet_set_rrelu(e3_unet_plan* plan, double lower, double upper, unsigned seed);
/* attention=True: the attention map of decoder block `block` (= index into UNet.up_convs) as the last e3_unet_forward left it in its
 * workspaces -- `sigm_psi_f` of GridAttention.forward (unet.py:521-525), what UpConvBlock stores as `self.att` (unet.py:394-395) and the
 * Trainer plots.  Same (N, D, H, W, training) and the same `saved` / `scratch` buffers as that forward; call it on the forward's stream before
 * anything else reuses `scratch`.  out: (N, 1, Do, Ho, Wo) floats, or NULL to query the size only. */
int e3_unet_attention_map(const e3_unet_plan* plan, void* stream, int N, int D, int H, int W, int training, void* saved, void* scratch,
                          int block, float* out, int* Do, int* Ho, int* Wo);
int e3_unet_profile_select(e3_unet_plan* plan, int layer, int which);
int e3_unet_profile_read(e3_unet_plan* plan, double* mean_ms, int* launches);

/* ------------------------------------------------------------------------------------------------------------
 * Per-op API (unit parity against the oracle; also usable on their own).  Tensors are NDHWC views.
 * ---------------------------------------------------------------------------------------------------------- */

/* nn.Conv3d(k=3, pad=1) or, planar != 0, nn.Conv3d(k=(1,3,3), pad=(0,1,1))   [unet.py:131-149]
 *   w: torch layout (Cout, Cin, kd, 3, 3); bias may be NULL.
 *   pro_scale/pro_shift (NULL or [Cin]):  x := relu(x*scale+shift) applied while loading (fused BN+ReLU of the
 *       producer layer).
 *   epi_scale/epi_shift (NULL or [Cout]): y := relu(y*scale+shift) (eval-mode BatchNorm folded; bias is NOT added
 *       separately in that case -- fold it into epi_shift).
 *   stats (NULL or [e3_conv3d_stats_parts][Cout][3]): per-tile (count, mean, M2) of y for train-mode BatchNorm.
 *   workspace: e3_conv3d_workspace_bytes() bytes (packed weights). */
size_t e3_conv3d_workspace_bytes(int Cin, int Cout, int planar);
int e3_conv3d_stats_parts(int Cin, int Cout, int N, int D, int H, int W, int planar);
int e3_conv3d_fwd(void* stream, const float* x, int x_ldc, int Cin, const float* w, const float* bias,
                  float* y, int y_ldc, int Cout, int N, int D, int H, int W, int planar,
                  const float* pro_scale, const float* pro_shift, const float* epi_scale, const float* epi_shift,
                  float* stats, void* workspace, size_t workspace_bytes);
/* dx = d/dx of the above (no prologue/epilogue), Cout >= 8. */
int e3_conv3d_dgrad(void* stream, const float* dy, int dy_ldc, int Cout, const float* w, float* dx, int dx_ldc, int Cin,
                    int N, int D, int H, int W, int planar, void* workspace, size_t workspace_bytes);
/* dw (torch layout) = d/dw.  workspace: e3_conv3d_wgrad_workspace_bytes(). */
size_t e3_conv3d_wgrad_workspace_bytes(int Cin, int Cout, int N, int D, int H, int W, int planar);
int e3_conv3d_wgrad(void* stream, const float* x, int x_ldc, int Cin, const float* dy, int dy_ldc, int Cout, float* dw,
                    int N, int D, int H, int W, int planar, void* workspace, size_t workspace_bytes);

/* nn.ConvTranspose3d(Cin, Cout, kernel=stride=(sd,2,2)), sd in {1,2}   [unet.py:152-165]; w: (Cin, Cout, sd, 2, 2).
 * (D,H,W) are the INPUT dims; the output view has dims (Do,Ho,Wo) <= (sd*D, 2H, 2W): positions beyond are dropped,
 * which implements autocrop()'s crop of the up-convolved tensor (unet.py:289-299). */
size_t e3_convT_workspace_bytes(int Cin, int Cout, int sd);
int e3_convT_stats_parts(int Cin, int Cout, int N, int D, int H, int W, int sd);
int e3_convT_fwd(void* stream, const float* x, int x_ldc, int Cin, const float* w, const float* bias, float* y, int y_ldc,
                 int Cout, int N, int D, int H, int W, int sd, int Do, int Ho, int Wo, float* stats,
                 void* workspace, size_t workspace_bytes);
int e3_convT_dgrad(void* stream, const float* dy, int dy_ldc, int Cout, const float* w, float* dx, int dx_ldc, int Cin,
                   int N, int D, int H, int W, int sd, int Do, int Ho, int Wo, void* workspace, size_t workspace_bytes);
size_t e3_convT_wgrad_workspace_bytes(int Cin, int Cout, int N, int D, int H, int W, int sd);
int e3_convT_wgrad(void* stream, const float* x, int x_ldc, int Cin, const float* dy, int dy_ldc, int Cout, float* dw,
                   int N, int D, int H, int W, int sd, int Do, int Ho, int Wo, void* workspace, size_t workspace_bytes);

/* nn.BatchNorm3d training statistics from the per-tile records a conv wrote   [unet.py:77-105]:
 * mean/invstd/scale/shift are [C] outputs (scale = gamma*invstd, shift = beta - mean*scale);
 * running_mean/var are updated in place with `momentum` (NULL to skip). */
int e3_bn_finalize(void* stream, const float* stats, int parts, int C, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps,
                   float* mean, float* invstd, float* scale, float* shift);
/* a = relu(x*scale + shift); if pooled != NULL also pooled = MaxPool3d(k=(kd,2,2), ceil_mode=True)(a)
 * [unet.py:183-186, 67-74, 225-230].  pooled is packed NDHWC with ceil dims. */
int e3_bn_relu_apply(void* stream, const float* x, int x_ldc, const float* scale, const float* shift, float* a, int a_ldc,
                     float* pooled, int kd, int N, int D, int H, int W, int C);
int e3_maxpool(void* stream, const float* a, int a_ldc, float* pooled, int kd, int N, int D, int H, int W, int C);
/* Backward of x -> BatchNorm(train) -> ReLU [-> MaxPool]:
 *   dA = g1 (may be NULL) + unpool(gpool) (may be NULL; needs a, pooled);  dx, dgamma, dbeta, dxsum (= sum_p dx,
 *   the gradient of the producing conv's bias) are outputs.  workspace: e3_bn_bwd_workspace_bytes(). */
size_t e3_bn_bwd_workspace_bytes(int N, int D, int H, int W, int C);
int e3_bn_relu_bwd(void* stream, const float* x, int x_ldc, const float* mean, const float* invstd, const float* gamma,
                   const float* scale, const float* shift, const float* g1, int g1_ldc, const float* gpool,
                   const float* a, int a_ldc, const float* pooled, int kd, int N, int D, int H, int W, int C,
                   float* dx, int dx_ldc, float* dgamma, float* dbeta, float* dxsum, void* workspace, size_t workspace_bytes);

/* conv_final: nn.Conv3d(C, Cout, 1)   [unet.py:178-180, 881, 912].  y and dy are NCDHW (module boundary). */
int e3_conv1_fwd(void* stream, const float* a, int a_ldc, int C, const float* w, const float* bias, float* y_ncdhw,
                 int Cout, int N, int D, int H, int W, int softmax);
size_t e3_conv1_bwd_workspace_bytes(int C, int Cout, int N, int D, int H, int W);
int e3_conv1_bwd(void* stream, const float* a, int a_ldc, int C, const float* w, const float* dy_ncdhw, float* da, int da_ldc,
                 float* dw, float* db, int Cout, int N, int D, int H, int W, void* workspace, size_t workspace_bytes);

/* Criterion of the reference's training example on device:
 *   loss = ce_weight * CrossEntropyLoss(weight=w)(logits, target) + dice_weight * DiceLoss(apply_softmax=True, weight=w, smooth)(logits, target)
 * [elektronn3/modules/loss.py:19-49 CombinedLoss, :158-189 dice_loss (eps = 1e-4), :192-234 DiceLoss; examples/train_unet_neurodata.py:294-296].
 * logits/dlogits: fp32 NCDHW (N, C, D, H, W), 2 <= C <= 16; target: int64 (N, D, H, W) class indices; w: [C] or NULL (all ones).
 * The forward writes the scalar loss to loss_out (device) and leaves the coefficients of the backward in `workspace`
 * (e3_ce_dice_workspace_bytes(C) bytes, to be passed unchanged to e3_ce_dice_bwd); gout: device scalar dL/dloss or NULL (= 1). */
size_t e3_ce_dice_workspace_bytes(int C);
int e3_ce_dice_fwd(void* stream, const float* logits, const long long* target, const float* w, int C, int N, int D, int H, int W,
                   float ce_weight, float dice_weight, float eps, float smooth, void* workspace, size_t workspace_bytes, float* loss_out);
/* The same criterion over a minibatch that is sharded over ranks (the reference computes ONE loss over the batch that
 * nn.DataParallel gathers on GPU 0, training/trainer.py:520-524; weighted CE and Dice are ratios of batch-wide sums, modules/loss.py:181-186,
 * so per-rank losses do not average to it).  e3_ce_dice_sums writes this rank's 2 + 3C sums (fp64, device):
 *   [0] sum_v w[t_v] (-log p[t_v]), [1] sum_v w[t_v], [2 + c] sum p_c [t = c], [2 + C + c] sum p_c, [2 + 2C + c] sum [t = c];
 * the caller adds them over ranks (one all-reduce) and e3_ce_dice_from_sums turns the totals into the loss and the backward's
 * coefficients in `workspace`; e3_ce_dice_bwd then gives d(global loss)/d(local logits). */
int e3_ce_dice_sums(void* stream, const float* logits, const long long* target, const float* w, int C, int N, int D, int H, int W,
                    void* workspace, size_t workspace_bytes, double* sums);
int e3_ce_dice_from_sums(void* stream, const double* sums, const float* w, int C, float ce_weight, float dice_weight, float eps, float smooth,
                         void* workspace, size_t workspace_bytes, float* loss_out);
int e3_ce_dice_bwd(void* stream, const float* logits, const long long* target, const float* w, int C, int N, int D, int H, int W,
                   const void* workspace, size_t workspace_bytes, const float* gout, float* dlogits);

/* torch.optim.AdamW(model.parameters(), lr, betas, eps, weight_decay) [examples/train_unet_neurodata.py:257-262; stepped by
 * training/trainer.py:539-542 through GradScaler.step] as ONE launch over all parameter tensors:
 *   p *= 1 - lr*wd;  m += (1-b1)(g - m);  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
 * params/grads/numels: HOST arrays of n_tensors device pointers / element counts (fp32 tensors, each with its own
 * allocation; grads[i] may be NULL = tensor skipped).  exp_avg/exp_avg_sq: flat device buffers of
 * e3_adamw_state_floats() floats, zero-initialised by the caller; tensor i's moments start at e3_adamw_state_offset(.., i).
 * step: device float (number of steps taken, incremented here); coef: device scratch of 8 floats.
 * grad_scale/found_inf: NULL, or the device scalars a torch GradScaler registers on an optimizer with
 * _step_supports_amp_scaling: gradients are divided by grad_scale[0]; found_inf[0] != 0 skips the whole step (no host sync). */
size_t e3_adamw_state_floats(int n_tensors, const long long* numels);
size_t e3_adamw_state_offset(int n_tensors, const long long* numels, int tensor);
int e3_adamw_step(void* stream, int n_tensors, void* const* params, void* const* grads, const long long* numels,
                  float* exp_avg, float* exp_avg_sq, float* step, float* coef,
                  double lr, double beta1, double beta2, double eps, double weight_decay,
                  const float* grad_scale, const float* found_inf);
/* The same step for a module whose parameters (and therefore gradients) are bfloat16 (model.to(torch.bfloat16), BASELINE configs[2]'s
 * storage): params / grads point to bf16 tensors, the moments stay fp32, every parameter is rounded to bf16 once per step. */
int e3_adamw_step_bf16(void* stream, int n_tensors, void* const* params, void* const* grads, const long long* numels,
                  float* exp_avg, float* exp_avg_sq, float* step, float* coef,
                  double lr, double beta1, double beta2, double eps, double weight_decay,
                  const float* grad_scale, const float* found_inf);

/* Stochastic weight averaging of the reference's SWA(optimizer) wrapper [elektronn3/training/swa.py:145-176 update_swa_group,
 * :184-202 swap_swa_sgd; driven by training/trainer.py:681-700] as ONE launch over all parameter tensors:
 *   e3_swa_update:  swa_buffer += (p - swa_buffer) * (1 / (n_avg + 1))   (fp32, the reference's two rounded operations: bit-identical)
 *   e3_swa_swap:    exchanges every parameter with its running average.
 * params / swa_buffers: host arrays of n_tensors device pointers (contiguous fp32, numels[i] elements each). */
int e3_swa_update(void* stream, int n_tensors, void* const* params, void* const* swa_buffers, const long long* numels, long long n_avg);
int e3_swa_swap(void* stream, int n_tensors, void* const* params, void* const* swa_buffers, const long long* numels);

/* ------------------------------------------------------------------------------------------------------------
 * Native bf16 path (BASELINE.json configs[2]: the same U-Net with bf16 storage).
 *   The reference's reduced-precision switches are Trainer(mixed_precision=True) (torch.cuda.amp.autocast,
 *   elektronn3/training/trainer.py:367,519), Predictor(float16=True) / model.half() (elektronn3/inference/inference.py:408,445-446)
 *   and pred_benchmark's model.to(device, dtype) (benchmark/pred_benchmark.py:55,71); their bf16 counterpart is
 *   model.to(torch.bfloat16) with bf16 inputs, or torch.autocast('cuda', dtype=torch.bfloat16) around a fp32 module.
 *   Activations and their gradients are bf16 NDHWC in HBM, convolutions run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation,
 *   BatchNorm statistics / coefficients / parameter gradients are fp32; like a bf16 torch module every op rounds its result once.
 *   x: [N,in,D,H,W] bf16 NCDHW; params / grads: the SAME fp32 tables as e3_unet_forward / e3_unet_backward (conv weights are
 *   rounded to bf16 while they are packed); y / dy: fp32 NCDHW logits (values representable in bf16) and their gradient.
 *   dx must be NULL (no input gradient on this path).  e3_unet_bf16_supported(): 1 when the plan's configuration is covered
 *   (any planar blocks incl. dim 2, 'batch' norm with full_norm, ReLU, 'transpose', 'concat', 'same', in_channels < 8,
 *   start_filts % 32 == 0, out_channels <= 8); other configurations compute in fp32 on up-cast copies.
 * ---------------------------------------------------------------------------------------------------------- */
int e3_unet_bf16_supported(const e3_unet_plan* plan);
int e3_unet_sizes_bf16(const e3_unet_plan* plan, int N, int D, int H, int W, int training, size_t* saved_bytes, size_t* scratch_bytes);
int e3_unet_forward_bf16(e3_unet_plan* plan, void* stream, const void* x, int N, int D, int H, int W,
                         void* const* params, const float* momenta, float* y,
                         void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, uint32_t flags);
/* e3_unet_forward_roi for ONE TILE OF A LARGER VOLUME, without the tile copy in front of the model and the crop copy behind it (the tile loop of
 * tiled_apply, inference.py:153-199: `inp_tile = inp_padded[tile].contiguous(); out[out_slice] = model(inp_tile)[crop]`): x is read through
 * strides as a (N, in, D, H, W) view of the padded volume (the tile's borders are zero-padded like a contiguous tile's), and the voxels of `roi`
 * go straight to their place in the output volume: output voxel (d, h, w) of the region, channel c, sample n is written to
 * y + n*y_stride[0] + c*y_stride[1] + (d - roi[0])*y_stride[2] + (h - roi[1])*y_stride[3] + (w - roi[2]).  Strides in elements, w stride 1.
 * fp32 path, in_channels == 1, 'same' convolutions (anything else: E3_ERR_UNSUPPORTED, callers take e3_unet_forward_roi on a copied tile). */
typedef struct e3_tile_view {
    const float* x; long long x_stride[3];      /* sample, d-plane, h-row */
    float* y; long long y_stride[4];            /* sample, channel, d-plane, h-row */
} e3_tile_view;
int e3_unet_forward_tile(e3_unet_plan* plan, void* stream, const e3_tile_view* view, int N, int D, int H, int W,
                         void* const* params, void* scratch, size_t scratch_bytes, uint32_t flags, const int roi[6]);

/* e3_unet_forward_roi on this path (inference; the needed region of the Predictor's central crop, see e3_unet_forward_roi) */
int e3_unet_forward_roi_bf16(e3_unet_plan* plan, void* stream, const void* x, int N, int D, int H, int W,
                             void* const* params, float* y, void* scratch, size_t scratch_bytes, uint32_t flags, const int roi[6]);
int e3_unet_backward_bf16(e3_unet_plan* plan, void* stream, const float* dy, const void* x, int N, int D, int H, int W,
                          void* const* params, void* const* grads, void* dx,
                          void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                          void* bucket_event, int bucket_after_down_block);
/* e3_unet_backward_loss on the native 16-bit paths (the criterion itself is evaluated on the fp32 logits y by e3_ce_dice_fwd / e3_ce_dice_sums + e3_ce_dice_from_sums,
 * whose workspace this call reads): no dlogits tensor, no e3_ce_dice_bwd / head-backward pass.  2..4 classes. */
int e3_unet_backward_loss_bf16(e3_unet_plan* plan, void* stream, const float* y, const e3_ce_dice_args* loss, const float* gout, const void* x,
                               int N, int D, int H, int W, void* const* params, void* const* grads, void* dx,
                               void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                               void* bucket_event, int bucket_after_down_block);
/* Per-op entry points of the bf16 path (unit parity tests): bf16 NDHWC views, fp32 torch-layout weights / weight gradients.
 *   e3_conv3d_*_bf16    nn.Conv3d k=3 (planar: (1,3,3)), stride 1, padding 1            [unet.py:131-149]; Cin, Cout multiples of 32, or -- the
 *                       network's first conv, fwd / wgrad only -- Cin < 8 with a dense [voxel][Cin] input (x_ldc = Cin), 3x3x3, Cout % 4 == 0
 *   e3_convT_*_bf16     nn.ConvTranspose3d k = s = 2 with the autocrop box (Do,Ho,Wo)   [unet.py:152-165,289-299]
 * stats: NULL or e3_*_stats_parts_bf16() records of (count, mean, M2) per channel of the stored (rounded) output. */
size_t e3_conv3d_workspace_bytes_bf16(int Cin, int Cout, int N, int D, int H, int W, int planar);
int e3_conv3d_stats_parts_bf16(int Cin, int Cout, int N, int D, int H, int W, int planar);
int e3_conv3d_fwd_bf16(void* stream, const void* x, int x_ldc, int Cin, const float* w, const float* bias, void* y, int y_ldc, int Cout,
                       int N, int D, int H, int W, int planar, const float* epi_scale, const float* epi_shift, float* stats,
                       void* workspace, size_t workspace_bytes);
int e3_conv3d_dgrad_bf16(void* stream, const void* dy, int dy_ldc, int Cout, const float* w, void* dx, int dx_ldc, int Cin,
                         int N, int D, int H, int W, int planar, void* workspace, size_t workspace_bytes);
int e3_conv3d_wgrad_bf16(void* stream, const void* x, int x_ldc, int Cin, const void* dy, int dy_ldc, int Cout, float* dw,
                         int N, int D, int H, int W, int planar, void* workspace, size_t workspace_bytes);
size_t e3_convT_workspace_bytes_bf16(int Cin, int Cout, int N, int D, int H, int W);
int e3_convT_stats_parts_bf16(int Cin, int N, int D, int H, int W);
int e3_convT_fwd_bf16(void* stream, const void* x, int x_ldc, int Cin, const float* w, const float* bias, void* y, int y_ldc, int Cout,
                      int N, int D, int H, int W, int Do, int Ho, int Wo, float* stats, void* workspace, size_t workspace_bytes);
int e3_convT_dgrad_bf16(void* stream, const void* dy, int dy_ldc, int Cout, const float* w, void* dx, int dx_ldc, int Cin,
                        int N, int D, int H, int W, int Do, int Ho, int Wo, void* workspace, size_t workspace_bytes);
int e3_convT_wgrad_bf16(void* stream, const void* x, int x_ldc, int Cin, const void* dy, int dy_ldc, int Cout, float* dw,
                        int N, int D, int H, int W, int Do, int Ho, int Wo, void* workspace, size_t workspace_bytes);

/* The same path compiled for IEEE half (float16): the reference's OWN reduced-precision mode -- torch.cuda.amp.autocast defaults to
 * float16 (Trainer(mixed_precision=True), elektronn3/training/trainer.py:367,519) and Predictor(float16=True) calls model.half()
 * (elektronn3/inference/inference.py:445-446).  Same signatures and semantics with float16 in place of bfloat16 (v_mfma_f32_32x32x16_f16,
 * overflow -> inf as in torch); gradients of a float16 net want a scaled loss (torch.cuda.amp.GradScaler, trainer.py:368,539-542). */
int e3_unet_f16_supported(const e3_unet_plan* plan);
int e3_unet_sizes_f16(const e3_unet_plan* plan, int N, int D, int H, int W, int training, size_t* saved_bytes, size_t* scratch_bytes);
int e3_unet_forward_f16(e3_unet_plan* plan, void* stream, const void* x, int N, int D, int H, int W,
                         void* const* params, const float* momenta, float* y,
                         void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, uint32_t flags);
int e3_unet_forward_roi_f16(e3_unet_plan* plan, void* stream, const void* x, int N, int D, int H, int W,
                            void* const* params, float* y, void* scratch, size_t scratch_bytes, uint32_t flags, const int roi[6]);
int e3_unet_backward_f16(e3_unet_plan* plan, void* stream, const float* dy, const void* x, int N, int D, int H, int W,
                          void* const* params, void* const* grads, void* dx,
                          void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                          void* bucket_event, int bucket_after_down_block);
int e3_unet_backward_loss_f16(e3_unet_plan* plan, void* stream, const float* y, const e3_ce_dice_args* loss, const float* gout, const void* x,
                              int N, int D, int H, int W, void* const* params, void* const* grads, void* dx,
                              void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                              void* bucket_event, int bucket_after_down_block);
size_t e3_conv3d_workspace_bytes_f16(int Cin, int Cout, int N, int D, int H, int W, int planar);
int e3_conv3d_stats_parts_f16(int Cin, int Cout, int N, int D, int H, int W, int planar);
int e3_conv3d_fwd_f16(void* stream, const void* x, int x_ldc, int Cin, const float* w, const float* bias, void* y, int y_ldc, int Cout,
                       int N, int D, int H, int W, int planar, const float* epi_scale, const float* epi_shift, float* stats,
                       void* workspace, size_t workspace_bytes);
int e3_conv3d_dgrad_f16(void* stream, const void* dy, int dy_ldc, int Cout, const float* w, void* dx, int dx_ldc, int Cin,
                         int N, int D, int H, int W, int planar, void* workspace, size_t workspace_bytes);
int e3_conv3d_wgrad_f16(void* stream, const void* x, int x_ldc, int Cin, const void* dy, int dy_ldc, int Cout, float* dw,
                         int N, int D, int H, int W, int planar, void* workspace, size_t workspace_bytes);
size_t e3_convT_workspace_bytes_f16(int Cin, int Cout, int N, int D, int H, int W);
int e3_convT_stats_parts_f16(int Cin, int N, int D, int H, int W);
int e3_convT_fwd_f16(void* stream, const void* x, int x_ldc, int Cin, const float* w, const float* bias, void* y, int y_ldc, int Cout,
                      int N, int D, int H, int W, int Do, int Ho, int Wo, float* stats, void* workspace, size_t workspace_bytes);
int e3_convT_dgrad_f16(void* stream, const void* dy, int dy_ldc, int Cout, const float* w, void* dx, int dx_ldc, int Cin,
                        int N, int D, int H, int W, int Do, int Ho, int Wo, void* workspace, size_t workspace_bytes);
int e3_convT_wgrad_f16(void* stream, const void* x, int x_ldc, int Cin, const void* dy, int dy_ldc, int Cout, float* dw,
                        int N, int D, int H, int W, int Do, int Ho, int Wo, void* workspace, size_t workspace_bytes);

/* Layout conversion at the module boundary. */
int e3_ncdhw_to_ndhwc(void* stream, const float* src, float* dst, int N, int C, int D, int H, int W);
int e3_ndhwc_to_ncdhw(void* stream, const float* src, int src_ldc, float* dst, int N, int C, int D, int H, int W);

#ifdef __cplusplus
}
#endif
#endif /* E3UNET_H */
