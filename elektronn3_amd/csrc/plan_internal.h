// Internal plan structures shared by the fp32 executor (unet_plan.cpp) and the bf16 executor (unet_bf16.cpp).
#pragma once
#include <mutex>
#include <string>
#include <vector>

#include "../../include/e3unet.h"
#include "kernels.h"

struct ParamSlot { std::string name; int64_t numel; int kind; };

struct ConvUnit {           // conv (3x3x3 | 1x3x3 | transposed 2x2x2) followed by BatchNorm + ReLU
    std::string name;       // e.g. "down_convs.0.conv1"
    std::string bn_name;    // e.g. "down_convs.0.norm0"
    int cin, cout, level;   // level = resolution level of the OUTPUT
    int planar;             // planar block (1x3x3 / (1,2,2))
    int is_up;              // 1: transposed conv, 2: ResizeConv = nearest up-sampling + 3x3x3 conv (input at level+1); 0: plain conv
    int p_w, p_b, p_g, p_be, p_rm, p_rv;   // indices into the param table
    int p_a;                // nn.PReLU weight of the activation after this conv ('prelu'), -1 otherwise
    int bn_index;           // -1: no normalisation after this conv (nn.Identity): conv -> bias -> ReLU
    bool has_norm() const { return bn_index >= 0; }
    // role in the network (set by the plan builder; the executors never parse names)
    bool is_down = false;   // encoder unit
    bool enc_last = false;  // last unit of an encoder block: its activation is the skip connection (and is pooled below the last level)
    bool to_cat = false;    // first conv of a decoder block: its input is the merged (concat / add) tensor
    // ResUNet (models/resunet.py:212-262): the second conv of a residual ConvBlock adds the block's input -- through a 1x1x1 projection when the
    // channel counts differ -- BEFORE its norm:  y = conv2(..); y += proj(inp); norm2; act2
    int res_in = -1;        // index of the ConvBlock's first conv unit (whose input is `inp`), -1: no residual
    int p_pw = -1, p_pb = -1;   // projection parameters, -1: identity shortcut
};

// GridAttention of a decoder block (UNet(attention=True), unet.py:376-379,452-541): indices into the param table
struct AttUnit { int p_ww, p_wb, p_g, p_be, p_rm, p_rv, p_theta, p_phi_w, p_phi_b, p_psi_w, p_psi_b, bn_index; };

struct Arena {              // bump allocator used twice: once with base == nullptr to size, once to place
    char* base; size_t off;
    explicit Arena(void* b) : base((char*)b), off(0) {}
    float* take(size_t floats) {
        float* p = base ? (float*)(base + off) : nullptr;
        off += align_up(floats * sizeof(float), 256);
        return p;
    }
};

struct LevelDims { int D, H, W; size_t vox; };

struct e3_unet_plan {
    e3_unet_cfg cfg;
    std::vector<ParamSlot> params;
    std::vector<ConvUnit> units;      // execution order of the forward
    int p_final_w, p_final_b;
    int n_bn;
    std::vector<AttUnit> att;         // per level < n_blocks - 1 (cfg.attention != 0)
    std::vector<int> enc_last_unit;   // per level: index of the encoder block's last unit
    std::vector<int> up_unit;         // per level < n_blocks - 1: index of the decoder block's up-convolution unit
    int enc_convs = 2, dec_convs = 2; // plain conv units per encoder / decoder block (UNet: 2; ResUNet: 2 per ConvBlock)
    // nn.RReLU in train mode: per-call seed (0 = off: fixed slope cfg.act_slope) and the slope interval.  e3_unet_set_rrelu stores them PER
    // CALLING THREAD (keyed by the plan's uid): a plan is shared by every module with its configuration, and two threads driving two such
    // modules -- one RReLU-train, one not; nn.DataParallel replicas -- must not see each other's seed between their set_rrelu and their call.
    unsigned uid = 0;
    struct RRelu { unsigned seed = 0; float lo = 0.125f, hi = 1.f / 3.f; };
    RRelu& rrelu_state() const;                            // this thread's state for this plan (unet_plan.cpp)
    ActArg rrelu_of(ActArg a, int unit) const {           // unit's activation with its own stream of slopes
        const RRelu& r = rrelu_state();
        if (!r.seed) return a;
        const unsigned sd = (r.seed * 0x9E3779B1u) ^ ((unsigned)(unit + 1) * 0x85EBCA77u);
        return a.rrelu(sd | 1u, r.lo, r.hi);
    }
    // profiling (e3_unet_profile_select / _read: a measurement aid of bench.py and tools/layer_table.py, plan-wide on purpose -- the backward of a
    // step runs on autograd's worker thread -- and therefore the one piece of plan state that is NOT per thread; the event list is mutex-guarded)
    std::mutex prof_mutex;
    int prof_layer = -1, prof_which = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    size_t prof_used = 0;

    bool planar(int level) const { return (cfg.planar_mask >> level) & 1u; }
    int chan(int level) const { return cfg.start_filts << level; }
};


// Dimensions of every tensor of the network for an (N, D, H, W) input.  conv_mode='same': one size per resolution level.
// conv_mode='valid' (padding 0, unet.py:217,347): every 3x3x3 conv shrinks its grid by 2 (planar: H and W only), so each unit has its own
// input / output size; the up-convolved tensor is cropped by one voxel where its size differs from the skip's by an odd amount and the
// skip is centre-cropped to it (autocrop, unet.py:256-325).
struct UnitDims { LevelDims in, out; int od, oh, ow; };   // in: the unit's input tensor = the grid its conv kernel runs on (transposed conv /
                                                          // ResizeConv: the LOW-resolution input); out: its output tensor; (od, oh, ow): where
                                                          // `out` sits inside the conv grid (valid convs: 1 voxel in, 0 along D for planar)
struct NetDims {
    std::vector<UnitDims> u;
    std::vector<LevelDims> E, X;       // per level: the encoder's skip activation (before the pool), the level's input
    std::vector<int> sd_, sh_, sw_;    // per level: offset of the centre crop of the skip inside E
    LevelDims Y;                       // output of the last unit = grid of the logits
    bool ok;
};

void net_dims(const e3_unet_plan* p, int N, int D, int H, int W, NetDims& nd);

// Needed regions of an inference forward whose caller keeps only the output voxels [roi[0..2], roi[3..5]) (e3_unet_forward_roi*): per unit the
// box of ITS output that the layers behind it read for those voxels (on = false: the whole tensor).  The box grows by the 3x3x3 reach per
// conv and halves per transposed conv on the way back through the decoder; the encoder is needed in full (the bottom level sees all of it).
struct NeedBox { int lo[3], hi[3]; bool on = false; };
std::vector<NeedBox> need_boxes(const e3_unet_plan* plan, const NetDims& ND, const int* roi);

// What the packed / folded weights lying in a scratch buffer belong to (E3_FWD_REUSE_PACKED is honoured only when the caller's claim can be checked):
// recorded by a successful fp32 inference forward, forgotten by every other call that is handed the same scratch pointer (unet_plan.cpp).
struct PackedSig {
    const void* plan = nullptr; int N = 0, D = 0, H = 0, W = 0; uint64_t params = 0; uint32_t mode = 0;
    bool operator==(const PackedSig& o) const { return plan == o.plan && N == o.N && D == o.D && H == o.H && W == o.W && params == o.params && mode == o.mode; }
};
void packed_sig_forget(const void* scratch);
