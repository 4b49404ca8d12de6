// Whole-network executor of the native bf16 path (BASELINE.json configs[2]): the same fixed launch sequence as unet_plan.cpp
// (UNet.forward elektronn3/models/unet.py:894-916, DownConv :244-253, UpConv :384-408, autocrop :256-325 and their backward), with
// bf16 NDHWC activations / gradients, bf16 MFMA convolutions (bf16_conv.hip, bf16_wgrad.hip, bf16_upconv.hip) and fp32 statistics,
// coefficients and parameter gradients.  The parameter table stays fp32 (the Python side hands over up-cast copies of a bf16
// module's parameters: 22 MB for cfg 2); conv weights are rounded to bf16 when they are packed, once per call.
//
// Covers the configuration BASELINE names (and its siblings): dim = 3 / 2 with any planar blocks, normalization='batch' with
// full_norm, ReLU, up_mode='transpose', merge_mode='concat', conv_mode='same', in_channels < 8, start_filts % 32 == 0.
// e3_unet_bf16_supported() says so; everything else keeps the fp32 kernels on up-cast copies.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "bf16.h"
#include "plan_internal.h"

namespace {

using std::max;

struct ArenaB {             // byte-granular bump allocator (sizing pass with base == nullptr)
    char* base; size_t off;
    explicit ArenaB(void* b) : base((char*)b), off(0) {}
    void* take_bytes(size_t bytes) {
        void* p = base ? (void*)(base + off) : nullptr;
        off += align_up(bytes, 256);
        return p;
    }
    bf16_t* take_h(size_t elems) { return (bf16_t*)take_bytes(elems * 2); }
    float* take_f(size_t elems) { return (float*)take_bytes(elems * 4); }
};

struct UnitB { bf16_t* raw; bf16_t* act; int act_ldc; float *mean, *invstd, *scale, *shift; float* bnpart; float* slab; bf16_t *wpk_f, *wpk_d; };

struct BufB {
    std::vector<UnitB> ub;
    std::vector<bf16_t*> catA, catB, pooled, g1, g2, dcatA, dcatB;     // concat halves as separate tensors: A = transposed-conv output, B = encoder skip
    bf16_t* xin; bf16_t* wpack; bf16_t* evalA;
    float *stats, *small, *slab, *bnred, *skws;
    // round 6: 3x3x3 weight gradients deferred to ONE stream-K launch at the end of the backward (launch_wgrad_b16_sk): the unit's dZ in a buffer of its own
    std::vector<bf16_t*> dz_u; float* wsk_slab = nullptr; size_t wsk_floats = 0;
    size_t saved_bytes, scratch_bytes;
};

int up_unit_of(int nb, int j) { return 2 * nb + 3 * (nb - 2 - j); }

void plan_b16(const e3_unet_plan* p, int N, int D, int H, int W, bool training, void* saved, void* scratch, BufB& B) {
    const int nb = p->cfg.n_blocks;
    NetDims ND; net_dims(p, N, D, H, W, ND);
    ArenaB S(saved), T(scratch);
    ArenaB& A = training ? S : T;
    const size_t nu = p->units.size();
    B.ub.assign(nu, UnitB{});
    B.catA.assign(nb, nullptr); B.catB.assign(nb, nullptr); B.pooled.assign(nb, nullptr); B.g1.assign(nb, nullptr); B.g2.assign(nb, nullptr);
    B.dcatA.assign(nb, nullptr); B.dcatB.assign(nb, nullptr);
    B.xin = p->cfg.in_channels > 1 ? A.take_h(ND.X[0].vox * p->cfg.in_channels) : nullptr;
    for (int j = 0; j + 1 < nb; ++j) {
        B.catA[j] = A.take_h(ND.u[up_unit_of(nb, j)].out.vox * p->chan(j));
        B.catB[j] = A.take_h(ND.u[up_unit_of(nb, j)].out.vox * p->chan(j));
        B.pooled[j] = A.take_h(ND.X[j + 1].vox * p->chan(j));
    }
    for (size_t k = 0; k < nu; ++k) {
        const ConvUnit& u = p->units[k];
        UnitB& b = B.ub[k];
        const size_t n = ND.u[k].out.vox * u.cout;
        b.raw = training ? A.take_h(n) : nullptr;
        const bool enc_skip = u.enc_last && u.level < nb - 1;
        if (enc_skip) { b.act = B.catB[u.level]; b.act_ldc = u.cout; }
        else if (u.is_up) { b.act = B.catA[u.level]; b.act_ldc = u.cout; }
        else if (training && k + 1 == nu) { b.act = nullptr; b.act_ldc = u.cout; }      // the head applies BN + ReLU while loading `raw`
        else { b.act = A.take_h(n); b.act_ldc = u.cout; }
        b.mean = A.take_f(u.cout); b.invstd = A.take_f(u.cout); b.scale = A.take_f(u.cout); b.shift = A.take_f(u.cout);
    }
    B.saved_bytes = S.off;
    size_t wmax = 0, statmax = 0, slabmax = 0, skmax = 0;
    B.dz_u.assign(nu, nullptr); B.wsk_slab = nullptr; B.wsk_floats = 0;
    int sk_tile_pairs = 0;
    const int Cmax = p->chan(nb - 1);
    for (size_t k = 0; k < nu; ++k) {
        const ConvUnit& u = p->units[k];
        const LevelDims& li = ND.u[k].in;
        UnitB& b = B.ub[k];
        size_t own = 0;
        const int pl = u.planar, sd = pl ? 1 : 2, taps = pl ? 9 : 27;      // planar block (unet.py:114-128): 1x3x3 convs, (1,2,2) pooling / up-convolution
        if (u.is_up) {
            wmax = max(wmax, upconv_b16_packed_elems(u.cin, u.cout, sd));
            statmax = max(statmax, (size_t)upconv_b16_stats_parts(N, li.D, li.H, li.W, sd, u.cin) * u.cout * 3);
            own = (size_t)upconv_b16_wgrad_splits(N, li.D, li.H, li.W) * sd * 4 * u.cin * u.cout;
        } else if (u.cin < 8) {
            statmax = max(statmax, (size_t)conv_small_b16_stats_parts(N, li.D, li.H, li.W, pl) * u.cout * 3);
            slabmax = max(slabmax, (size_t)conv_small_b16_wgrad_splits(N, li.D, li.H, li.W, pl) * taps * u.cout * u.cin);
        } else {
            wmax = max(wmax, conv_b16_packed_elems(u.cin, u.cout, pl));
            statmax = max(statmax, (size_t)conv_b16_stats_parts(N, li.D, li.H, li.W, u.cin, u.cout, pl) * u.cout * 3);
            skmax = max(skmax, conv_b16_partial_floats(N, li.D, li.H, li.W, u.cin, u.cout, pl));
            if (training) skmax = max(skmax, conv_b16_partial_floats(N, li.D, li.H, li.W, u.cout, u.cin, pl));
            own = (size_t)wgrad_b16_splits(N, li.D, li.H, li.W, u.cin, u.cout, pl) * taps * u.cin * u.cout;
        }
        if (u.is_up || u.cin >= 8) {          // packed bf16 weights: forward form, data-gradient form (all packed by one launch per pass)
            const size_t pe = u.is_up ? upconv_b16_packed_elems(u.cin, u.cout, sd) : conv_b16_packed_elems(u.cin, u.cout, pl);
            b.wpk_f = T.take_h(pe);
            b.wpk_d = training ? T.take_h(pe) : nullptr;
        }
        if (training) {
            b.bnpart = T.take_f((size_t)bn_bwd_b16_parts(ND.u[k].out.vox, u.cout) * 3 * u.cout);
            // (which layers: ALL of them on this path -- the 16-bit level-0 dZ is 134 MB and the per-layer launches' 512 slabs per layer cost more than its trip to HBM:
            // same box, cfg-3 step: per layer 5.273 ms, layers up to 80 MB 5.118, all 5.091; the fp32 executor stops at 80 MB, unet_plan.cpp)
            static const bool no_defer = getenv("E3_WGRAD_NO_DEFER") != nullptr;
            static const double defer_max_mb = getenv("E3_WGRAD_DEFER_MAX_MB") ? atof(getenv("E3_WGRAD_DEFER_MAX_MB")) : 1e9;
            const bool defer = !no_defer && !u.is_up && u.cin >= 8 && !pl && (double)ND.u[k].out.vox * u.cout * 2.0 <= defer_max_mb * 1048576.0;
            if (defer) { B.dz_u[k] = T.take_h(ND.u[k].out.vox * u.cout); sk_tile_pairs += (u.cin / 32) * (u.cout / 32); slabmax = max(slabmax, own); own = 0; }      // (the per-layer fallback of such a unit takes the shared slab)
            b.slab = own ? T.take_f(own) : nullptr;
        }
    }
    if (sk_tile_pairs) { B.wsk_floats = wgrad_b16_sk_slab_floats(sk_tile_pairs); B.wsk_slab = T.take_f(B.wsk_floats); }
    if (training) slabmax = max(slabmax, (size_t)conv_final_b16_bwd_parts(ND.Y.vox) * (p->cfg.out_channels * p->chan(0) + p->cfg.out_channels));
    (void)wmax; B.wpack = nullptr;
    B.stats = T.take_f(statmax);
    B.skws = skmax ? T.take_f(skmax) : nullptr;
    B.small = T.take_f((size_t)5 * Cmax + 64);
    B.bnred = T.take_f((size_t)BN_PRERED * Cmax * 3);
    B.slab = training ? T.take_f(slabmax) : nullptr;
    B.evalA = nullptr;
    if (training) {
        for (int j = 0; j < nb; ++j) {
            const size_t n = ND.X[j].vox * p->chan(j);
            B.g1[j] = T.take_h(n); B.g2[j] = T.take_h(n);
            if (j + 1 < nb) { B.dcatA[j] = T.take_h(n); B.dcatB[j] = T.take_h(n); }
        }
    }
    B.scratch_bytes = T.off;
}

struct ProfB {      // same per-layer HIP-event profiling hook as the fp32 executor (e3_unet_profile_select / _read)
    e3_unet_plan* p; hipStream_t s; bool on;
    ProfB(e3_unet_plan* plan, hipStream_t st, int layer, int which) : p(plan), s(st) {
        on = plan->prof_layer >= 0 && plan->prof_layer == layer && plan->prof_which == which;
        if (on) {
            std::lock_guard<std::mutex> lk(p->prof_mutex);
            if (p->prof_used == p->prof_events.size()) {
                hipEvent_t a, b;
                if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
                p->prof_events.push_back({a, b});
            }
            (void)hipEventRecord(p->prof_events[p->prof_used].first, s);
        }
    }
    ~ProfB() { if (on) { (void)hipEventRecord(p->prof_events[p->prof_used].second, s); p->prof_used++; } }
};

#define RUN(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

bool supported(const e3_unet_cfg& c) {
    return c.normalization == 1 && c.full_norm && c.act_slope == 0.f && c.up_resize == 0 && !c.merge_add && !c.conv_valid &&
           c.in_channels < 8 && c.start_filts % 32 == 0 && c.out_channels <= 8 && !c.attention && !c.resunet;
}

}  // namespace

extern "C" {

int e3_unet_bf16_supported(const e3_unet_plan* plan) { return plan && supported(plan->cfg) ? 1 : 0; }

int e3_unet_sizes_bf16(const e3_unet_plan* plan, int N, int D, int H, int W, int training, size_t* saved_bytes, size_t* scratch_bytes) {
    E3_REQUIRE(plan && N > 0 && D > 0 && H > 0 && W > 0, E3_ERR_INVALID, "bad shape");
    E3_REQUIRE(supported(plan->cfg), E3_ERR_UNSUPPORTED, "configuration not on the native bf16 path");
    BufB B;
    plan_b16(plan, N, D, H, W, training != 0, nullptr, nullptr, B);
    if (saved_bytes) *saved_bytes = B.saved_bytes;
    if (scratch_bytes) *scratch_bytes = B.scratch_bytes;
    return E3_OK;
}

static int forward_b16_impl(e3_unet_plan* plan, void* stream, const void* x, int N, int D, int H, int W,
                            void* const* params, const float* momenta, float* y,
                            void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, uint32_t flags, const int* roi);

int e3_unet_forward_bf16(e3_unet_plan* plan, void* stream, const void* x, int N, int D, int H, int W,
                         void* const* params, const float* momenta, float* y,
                         void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, uint32_t flags) {
    return forward_b16_impl(plan, stream, x, N, D, H, W, params, momenta, y, saved, saved_bytes, scratch, scratch_bytes, flags, nullptr);
}

int e3_unet_forward_roi_bf16(e3_unet_plan* plan, void* stream, const void* x, int N, int D, int H, int W,
                             void* const* params, float* y, void* scratch, size_t scratch_bytes, uint32_t flags, const int roi[6]) {
    E3_REQUIRE(roi, E3_ERR_INVALID, "forward_roi: null region");
    E3_REQUIRE(!(flags & (E3_FWD_TRAINING | E3_FWD_FROZEN_BN)), E3_ERR_INVALID, "forward_roi: inference only");
    for (int i = 0; i < 3; ++i) E3_REQUIRE(roi[i] >= 0 && roi[3 + i] > roi[i], E3_ERR_INVALID, "forward_roi: empty or negative region");
    return forward_b16_impl(plan, stream, x, N, D, H, W, params, nullptr, y, nullptr, 0, scratch, scratch_bytes, flags, roi);
}

static int forward_b16_impl(e3_unet_plan* plan, void* stream, const void* x, int N, int D, int H, int W,
                            void* const* params, const float* momenta, float* y,
                            void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, uint32_t flags, const int* roi) {
    E3_REQUIRE(plan && x && y && params && scratch, E3_ERR_INVALID, "null argument");
    packed_sig_forget(scratch);
    E3_REQUIRE(supported(plan->cfg), E3_ERR_UNSUPPORTED, "configuration not on the native bf16 path");
    hipStream_t s = (hipStream_t)stream;
    const bool training = (flags & E3_FWD_TRAINING) != 0;
    const e3_unet_cfg& cfg = plan->cfg;
    const int nb = cfg.n_blocks;
    E3_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, E3_ERR_INVALID, "bad shape");
    E3_REQUIRE(!training || (saved && momenta), E3_ERR_INVALID, "training forward needs `saved` and `momenta`");
    BufB B;
    plan_b16(plan, N, D, H, W, training, saved, scratch, B);
    E3_REQUIRE(!training || saved_bytes >= B.saved_bytes, E3_ERR_WORKSPACE, "`saved` buffer too small");
    E3_REQUIRE(scratch_bytes >= B.scratch_bytes, E3_ERR_WORKSPACE, "`scratch` buffer too small");
    NetDims ND; net_dims(plan, N, D, H, W, ND);
    auto P = [&](int i) { return (float*)params[i]; };
    const size_t nu = plan->units.size();
    const std::vector<NeedBox> need = need_boxes(plan, ND, training ? nullptr : roi);      // (e3_unet_forward_roi_bf16; plan_internal.h)

    if (!training) {     // eval mode: BN folded into every conv epilogue (running statistics), all folds in one launch
        std::vector<FoldJob> jobs;
        for (size_t k = 0; k < nu; ++k) {
            const ConvUnit& u = plan->units[k];
            jobs.push_back({P(u.p_g), P(u.p_be), P(u.p_rm), P(u.p_rv), P(u.p_b), B.ub[k].scale, B.ub[k].shift, u.cout});
        }
        RUN(launch_fold_multi(jobs.data(), (int)jobs.size(), cfg.bn_eps, s));
    }
    {   // conv weights -> bf16 in the kernels' layouts, every layer in one launch
        std::vector<PackB16Job> jobs;
        for (size_t k = 0; k < nu; ++k) {
            const ConvUnit& u = plan->units[k];
            if (B.ub[k].wpk_f) jobs.push_back({P(u.p_w), B.ub[k].wpk_f, u.cout, u.cin, u.is_up ? (u.planar ? 4 : 8) : (u.planar ? 9 : 27), u.is_up ? 2 : 0});
        }
        RUN(launch_pack_multi_b16(jobs.data(), (int)jobs.size(), s));
    }
    const bf16_t* cur = (const bf16_t*)x; int cur_ldc = cfg.in_channels;
    const bf16_t* cur2 = nullptr;                        // second half of a concatenated input (the encoder skip), or null
    if (cfg.in_channels > 1) { RUN(launch_ncdhw_to_ndhwc_b16((const bf16_t*)x, B.xin, N, cfg.in_channels, ND.X[0].vox / N, s)); cur = B.xin; }

    for (size_t k = 0; k < nu; ++k) {
        const ConvUnit& u = plan->units[k];
        UnitB& b = B.ub[k];
        const LevelDims& lo = ND.u[k].out;
        const LevelDims& li = ND.u[k].in;
        const bool is_enc_conv2 = u.enc_last;
        const bool pool_after = is_enc_conv2 && u.level < nb - 1;
        const int pl = u.planar, sd = pl ? 1 : 2;
        bf16_t* dst = training ? b.raw : b.act;
        const int dst_ldc = training ? u.cout : b.act_ldc;
        const float* es = training ? nullptr : b.scale; const float* eh = training ? nullptr : b.shift;
        int parts = 0;
        if (u.is_up) {
            UpconvB16Args a{};
            a.x = cur; a.x_ldc = cur_ldc; a.Cin = u.cin; a.y = dst; a.y_ldc = dst_ldc; a.Cout = u.cout; a.wt = b.wpk_f;
            a.bias = training ? P(u.p_b) : nullptr; a.N = N; a.D = li.D; a.H = li.H; a.W = li.W; a.Do = lo.D; a.Ho = lo.H; a.Wo = lo.W; a.sd = sd;
            a.epi_scale = es; a.epi_shift = eh; a.stats = training ? B.stats : nullptr;
            if (need[k].on && N == 1 && !training) {      // needed region along D (no halo: a range of input planes is a smaller tensor), as in the fp32 executor
                const int p0 = need[k].lo[0] / sd, p1 = (need[k].hi[0] + sd - 1) / sd < li.D ? (need[k].hi[0] + sd - 1) / sd : li.D;
                if (p1 > p0 && (p0 > 0 || p1 < li.D)) {
                    const int o0 = p0 * sd, o1 = p1 * sd < lo.D ? p1 * sd : lo.D;
                    a.x = cur + (size_t)p0 * li.H * li.W * cur_ldc; a.D = p1 - p0;
                    a.y = dst + (size_t)o0 * lo.H * lo.W * dst_ldc; a.Do = o1 - o0;
                }
            }
            parts = upconv_b16_stats_parts(N, li.D, li.H, li.W, sd, u.cin);
            { ProfB pr(plan, s, (int)k, 0); RUN(launch_upconv_b16_fwd(a, s)); }
        } else if (u.cin < 8) {
            parts = conv_small_b16_stats_parts2(N, li.D, li.H, li.W, pl, u.cin, u.cout);
            ProfB pr(plan, s, (int)k, 0);
            RUN(launch_conv_small_b16_fwd(cur, u.cin, P(u.p_w), training ? P(u.p_b) : nullptr, dst, dst_ldc, N, li.D, li.H, li.W, u.cout, pl, es, eh,
                                          training ? B.stats : nullptr, s));
        } else {
            ConvB16Args a{};
            a.x = cur; a.x_ldc = cur_ldc; a.Cin = u.cin; a.wt = b.wpk_f; a.bias = training ? P(u.p_b) : nullptr;
            a.x2 = cur2; a.x_split = cur2 ? u.cin / 2 : 0;
            a.y = dst; a.y_ldc = dst_ldc; a.N = N; a.D = li.D; a.H = li.H; a.W = li.W; a.Cout = u.cout; a.planar = pl;
            a.epi_scale = es; a.epi_shift = eh; a.stats = training ? B.stats : nullptr; a.partial = B.skws;
            parts = conv_b16_stats_parts(N, li.D, li.H, li.W, u.cin, u.cout, pl);
            if (need[k].on && !pl && !a.stats)
                for (int i = 0; i < 3; ++i) { a.box_lo[i] = need[k].lo[i]; a.box_hi[i] = need[k].hi[i]; }
            { ProfB pr(plan, s, (int)k, 0); RUN(launch_conv_b16(a, s)); }
        }
        if (training) {
            BnFinalizeArgs f{};
            f.stats = B.stats; f.parts = parts; f.C = u.cout; f.gamma = P(u.p_g); f.beta = P(u.p_be); f.group = 1;
            f.running_mean = P(u.p_rm); f.running_var = P(u.p_rv); f.momentum = momenta[u.bn_index]; f.eps = cfg.bn_eps;
            f.mean = b.mean; f.invstd = b.invstd; f.scale = b.scale; f.shift = b.shift; f.scratch = B.bnred;
            RUN(launch_bn_finalize(f, s));
            if (k + 1 < nu)
                RUN(launch_bn_relu_apply_b16(b.raw, u.cout, b.scale, b.shift, b.act, b.act_ldc, pool_after ? B.pooled[u.level] : nullptr, sd,
                                             N, lo.D, lo.H, lo.W, u.cout, s));
        } else if (pool_after) {
            RUN(launch_maxpool_b16(b.act, b.act_ldc, B.pooled[u.level], sd, N, lo.D, lo.H, lo.W, u.cout, s));
        }
        cur2 = nullptr;
        if (pool_after) { cur = B.pooled[u.level]; cur_ldc = u.cout; }
        else if (u.is_up) { cur = B.catA[u.level]; cur2 = B.catB[u.level]; cur_ldc = u.cout; }
        else { cur = b.act; cur_ldc = b.act_ldc; }
    }
    {
        const ConvUnit& lu = plan->units.back();
        const UnitB& lb = B.ub.back();
        ProfB pr(plan, s, (int)nu, 0);
        if (roi && !training && N == 1 && (roi[0] > 0 || roi[3] < ND.Y.D)) {
            // needed region along D: the head reads and writes the d-planes [roi[0], roi[3]) only (a contiguous range of voxels of one sample)
            const int p0 = roi[0], p1 = roi[3] < ND.Y.D ? roi[3] : ND.Y.D;
            const size_t hw = (size_t)ND.Y.H * ND.Y.W;
            RUN(launch_conv_final_b16_fwd(cur + (size_t)p0 * hw * cur_ldc, cur_ldc, plan->chan(0), P(plan->p_final_w), P(plan->p_final_b), y + (size_t)p0 * hw,
                                          cfg.out_channels, (size_t)(p1 - p0) * hw, 1, (flags & E3_FWD_SOFTMAX) ? 1 : 0, nullptr, nullptr, s, ND.Y.vox));
        } else
        RUN(launch_conv_final_b16_fwd(training ? lb.raw : cur, training ? lu.cout : cur_ldc, plan->chan(0), P(plan->p_final_w), P(plan->p_final_b), y,
                                      cfg.out_channels, ND.Y.vox / N, N, (flags & E3_FWD_SOFTMAX) ? 1 : 0, training ? lb.scale : nullptr,
                                      training ? lb.shift : nullptr, s));
    }
    return E3_OK;
}

// the criterion's request of e3_unet_backward_loss_bf16: dLoss/dlogits is formed inside the head's backward kernels
struct HeadLossReqB { const float* logits; const long long* target; const float* cw; const float* coef; const float* gout; };
static int backward_b16_impl(e3_unet_plan* plan, void* stream, const float* dy, const HeadLossReqB* hl, const void* x, int N, int D, int H, int W,
                             void* const* params, void* const* grads, void* dx,
                             void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                             void* bucket_event, int bucket_after_down_block);

int e3_unet_backward_bf16(e3_unet_plan* plan, void* stream, const float* dy, const void* x, int N, int D, int H, int W,
                          void* const* params, void* const* grads, void* dx,
                          void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                          void* bucket_event, int bucket_after_down_block) {
    E3_REQUIRE(dy, E3_ERR_INVALID, "null argument");
    return backward_b16_impl(plan, stream, dy, nullptr, x, N, D, H, W, params, grads, dx, saved, saved_bytes, scratch, scratch_bytes, bucket_event, bucket_after_down_block);
}

int e3_unet_backward_loss_bf16(e3_unet_plan* plan, void* stream, const float* y, const e3_ce_dice_args* loss, const float* gout, const void* x,
                               int N, int D, int H, int W, void* const* params, void* const* grads, void* dx,
                               void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                               void* bucket_event, int bucket_after_down_block) {
    E3_REQUIRE(plan && y && loss && loss->target && loss->workspace, E3_ERR_INVALID, "null argument");
    const int C = plan->cfg.out_channels;
    E3_REQUIRE(loss->workspace_bytes >= e3_ce_dice_workspace_bytes(C), E3_ERR_WORKSPACE, "ce_dice workspace too small");
    if (!(C >= 2 && C <= 4)) { e3_set_error("e3_unet_backward_loss: needs 2..4 classes"); return E3_ERR_UNSUPPORTED; }
    const HeadLossReqB hl{y, loss->target, loss->class_weight, (const float*)loss->workspace + (size_t)CE_DICE_MAX_ROWS * (2 + 3 * C), gout};
    return backward_b16_impl(plan, stream, nullptr, &hl, x, N, D, H, W, params, grads, dx, saved, saved_bytes, scratch, scratch_bytes, bucket_event, bucket_after_down_block);
}

static int backward_b16_impl(e3_unet_plan* plan, void* stream, const float* dy, const HeadLossReqB* hl, const void* x, int N, int D, int H, int W,
                             void* const* params, void* const* grads, void* dx,
                             void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                             void* bucket_event, int bucket_after_down_block) {
    E3_REQUIRE(plan && (dy || hl) && x && params && grads && saved && scratch, E3_ERR_INVALID, "null argument");
    packed_sig_forget(scratch);
    E3_REQUIRE(supported(plan->cfg), E3_ERR_UNSUPPORTED, "configuration not on the native bf16 path");
    E3_REQUIRE(dx == nullptr, E3_ERR_UNSUPPORTED, "the native bf16 path does not compute the gradient of the network input");
    hipStream_t s = (hipStream_t)stream;
    const e3_unet_cfg& cfg = plan->cfg;
    const int nb = cfg.n_blocks;
    BufB B;
    plan_b16(plan, N, D, H, W, true, saved, scratch, B);
    E3_REQUIRE(saved_bytes >= B.saved_bytes, E3_ERR_WORKSPACE, "`saved` buffer too small");
    E3_REQUIRE(scratch_bytes >= B.scratch_bytes, E3_ERR_WORKSPACE, "`scratch` buffer too small");
    NetDims ND; net_dims(plan, N, D, H, W, ND);
    auto P = [&](int i) { return (float*)params[i]; };
    auto G = [&](int i) { return (float*)grads[i]; };
    const int C0 = plan->chan(0);
    const int nunits = (int)plan->units.size();

    if (!hl) {   // 1x1x1 head: dW, db (its data gradient is recomputed by the last unit's BN backward); e3_unet_backward_loss: that pass takes them along
        const UnitB& last = B.ub[nunits - 1];
        const int parts = conv_final_b16_bwd_parts(ND.Y.vox);
        const int ps = cfg.out_channels * C0 + cfg.out_channels;
        { ProfB pr(plan, s, nunits, 1);
          RUN(launch_conv_final_b16_bwd(last.raw, C0, C0, P(plan->p_final_w), dy, nullptr, C0, B.slab, cfg.out_channels, ND.Y.vox / N, N,
                                        last.scale, last.shift, s)); }
        RUN(launch_colsum_finalize(B.slab, parts, ps, 0, cfg.out_channels * C0, G(plan->p_final_w), s));
        RUN(launch_colsum_finalize(B.slab, parts, ps, cfg.out_channels * C0, cfg.out_channels, G(plan->p_final_b), s));
    }
    {   // data-gradient forms of the conv weights, every layer in one launch
        std::vector<PackB16Job> jobs;
        for (int k = 1; k < nunits; ++k) {
            const ConvUnit& u = plan->units[k];
            if (B.ub[k].wpk_d) jobs.push_back({P(u.p_w), B.ub[k].wpk_d, u.cout, u.cin, u.is_up ? (u.planar ? 4 : 8) : (u.planar ? 9 : 27), u.is_up ? 3 : 1});
        }
        RUN(launch_pack_multi_b16(jobs.data(), (int)jobs.size(), s));
    }
    const bf16_t* g = B.g1[0]; int g_ldc = C0;
    bool event_done = bucket_event == nullptr;
    std::vector<WgradReduceJob> wred;
    std::vector<ColsumJob> bias_jobs;
    // (not with the overlapped all-reduce -- its early bucket wants final gradients -- and not while a per-layer weight-gradient profile is taken)
    const bool defer_wgrad = bucket_event == nullptr && !(plan->prof_layer >= 0 && plan->prof_which == 2);
    std::vector<WgradSkB16Layer> wsk_layers;
    for (int k = nunits - 1; k >= 0; --k) {
        const ConvUnit& u = plan->units[k];
        const UnitB& b = B.ub[k];
        const LevelDims& lo = ND.u[k].out;
        const LevelDims& li = ND.u[k].in;
        const int j = u.level;
        const bool is_down = u.is_down;
        const bool is_enc_conv2 = u.enc_last;
        const bool pooled_unit = is_enc_conv2 && j < nb - 1;
        const int pl = u.planar, sd = pl ? 1 : 2, taps = pl ? 9 : 27;
        if (!event_done && is_down && is_enc_conv2 && j == bucket_after_down_block - 1) {
            if (!bias_jobs.empty()) { RUN(launch_colsum_multi(bias_jobs.data(), (int)bias_jobs.size(), s)); bias_jobs.clear(); }
            if (!wred.empty()) { RUN(launch_wgrad_reduce_multi(wred.data(), (int)wred.size(), s)); wred.clear(); }
            E3_CHECK_HIP(hipEventRecord((hipEvent_t)bucket_event, s)); event_done = true;
        }
        // ---- BN + ReLU (+ pool, + skip) backward -> dxr
        bf16_t* dxr = (defer_wgrad && B.dz_u[k]) ? B.dz_u[k] : B.g2[j];
        {
            BnBwdB16Args a{};
            a.x = b.raw; a.x_ldc = u.cout; a.mean = b.mean; a.invstd = b.invstd; a.gamma = P(u.p_g); a.scale = b.scale; a.shift = b.shift;
            if (k == nunits - 1) {
                a.g1 = nullptr; a.head_dy = dy; a.head_w = P(plan->p_final_w); a.head_cout = cfg.out_channels; a.head_S = ND.Y.vox / N;
                if (hl) { a.hl_logits = hl->logits; a.hl_target = hl->target; a.hl_cw = hl->cw; a.hl_coef = hl->coef; a.hl_gout = hl->gout; a.head_part = B.slab; }
            }
            else if (pooled_unit) { a.g1 = B.dcatB[j]; a.g1_ldc = u.cout; a.gpool = g; a.pooled = B.pooled[j]; }
            else { a.g1 = g; a.g1_ldc = g_ldc; }
            a.kd = sd; a.N = N; a.D = lo.D; a.H = lo.H; a.W = lo.W; a.C = u.cout;
            a.parts = bn_bwd_b16_parts(lo.vox, u.cout); a.part = b.bnpart; a.coef = B.small; a.dx = dxr; a.dx_ldc = u.cout;
            RUN(launch_bn_bwd_b16_reduce(a, s));
            if (a.head_part) {      // the head's gradients from the partial sums of that pass
                const int ps = cfg.out_channels * C0 + cfg.out_channels;
                RUN(launch_colsum_finalize(B.slab, a.parts, ps, 0, cfg.out_channels * C0, G(plan->p_final_w), s));
                RUN(launch_colsum_finalize(B.slab, a.parts, ps, cfg.out_channels * C0, cfg.out_channels, G(plan->p_final_b), s));
                a.head_part = nullptr;
            }
            RUN(launch_bn_bwd_finalize(a.part, a.parts, u.cout, (float)(1.0 / (double)lo.vox), G(u.p_g), G(u.p_be), B.small, s));
            RUN(launch_bn_bwd_b16_apply(a, s));
            bias_jobs.push_back({a.part, a.parts, 3 * u.cout, 2 * u.cout, u.cout, G(u.p_b)});
        }
        // ---- input activation of this conv
        const bf16_t* xin; int xin_ldc; const bf16_t* xin2 = nullptr;
        if (k == 0) { xin = cfg.in_channels > 1 ? B.xin : (const bf16_t*)x; xin_ldc = cfg.in_channels; }
        else {
            const ConvUnit& pu = plan->units[k - 1];
            const bool prev_pooled = pu.enc_last && pu.level < nb - 1 && is_down;
            if (prev_pooled) { xin = B.pooled[pu.level]; xin_ldc = pu.cout; }
            else if (pu.is_up) { xin = B.catA[pu.level]; xin2 = B.catB[pu.level]; xin_ldc = pu.cout; }
            else { xin = B.ub[k - 1].act; xin_ldc = B.ub[k - 1].act_ldc; }
            if (u.is_up) { xin = B.ub[k - 1].act; xin_ldc = B.ub[k - 1].act_ldc; xin2 = nullptr; }
        }
        // ---- weight gradient
        if (u.is_up) {
            const int splits = upconv_b16_wgrad_splits(N, li.D, li.H, li.W);
            { ProfB pr(plan, s, k, 2);
              RUN(launch_upconv_b16_wgrad(xin, xin_ldc, u.cin, dxr, u.cout, u.cout, b.slab, N, li.D, li.H, li.W, lo.D, lo.H, lo.W, sd, splits, s)); }
            wred.push_back({b.slab, G(u.p_w), splits, sd * 4, u.cin, u.cout, u.cin, u.cout});
        } else if (u.cin < 8) {
            const int splits = conv_small_b16_wgrad_splits(N, li.D, li.H, li.W, pl);
            { ProfB pr(plan, s, k, 2); RUN(launch_conv_small_b16_wgrad(xin, u.cin, dxr, u.cout, B.slab, N, li.D, li.H, li.W, u.cout, pl, s)); }
            RUN(launch_wgrad_reduce(B.slab, G(u.p_w), splits, taps, u.cout, u.cin, u.cout, u.cin, s));
        } else if (defer_wgrad && B.dz_u[k]) {      // one stream-K launch for all of them behind the loop
            wsk_layers.push_back(WgradSkB16Layer{xin, xin2, xin2 ? u.cin / 2 : 0, xin_ldc, u.cin, dxr, u.cout, u.cout, N, li.D, li.H, li.W, G(u.p_w)});
        } else {
            WgradB16Args a{};
            a.x = xin; a.x_ldc = xin_ldc; a.Cin = u.cin; a.dy = dxr; a.dy_ldc = u.cout; a.Cout = u.cout; a.part = b.slab ? b.slab : B.slab;
            a.x2 = xin2; a.x_split = xin2 ? u.cin / 2 : 0;
            a.N = N; a.D = li.D; a.H = li.H; a.W = li.W; a.planar = pl;
            a.splits = wgrad_b16_splits(N, li.D, li.H, li.W, u.cin, u.cout, pl);
            { ProfB pr(plan, s, k, 2); RUN(launch_wgrad_b16(a, s)); }
            if (b.slab) wred.push_back({b.slab, G(u.p_w), a.splits, taps, u.cout, u.cin, u.cout, u.cin});
            else RUN(launch_wgrad_reduce(B.slab, G(u.p_w), a.splits, taps, u.cout, u.cin, u.cout, u.cin, s));      // (a deferrable unit on the per-layer path: the shared slab, reduced on the spot)
        }
        // ---- data gradient -> g of the previous unit
        if (k == 0) break;
        if (u.is_up) {
            UpconvB16Args a{};
            a.x = B.g1[j + 1]; a.x_ldc = u.cin; a.Cin = u.cin; a.y = dxr; a.y_ldc = u.cout; a.Cout = u.cout; a.wt = b.wpk_d;
            a.N = N; a.D = li.D; a.H = li.H; a.W = li.W; a.Do = lo.D; a.Ho = lo.H; a.Wo = lo.W; a.sd = sd;
            { ProfB pr(plan, s, k, 1); RUN(launch_upconv_b16_dgrad(a, s)); }
            g = B.g1[j + 1]; g_ldc = u.cin;
        } else {
            const bool to_cat = u.to_cat;     // UpConv.conv1: gradient of the concat buffer
            bf16_t* out = to_cat ? B.dcatA[j] : B.g1[j];
            ConvB16Args a{};
            a.x = dxr; a.x_ldc = u.cout; a.Cin = u.cout; a.wt = b.wpk_d; a.bias = nullptr; a.y = out; a.y_ldc = to_cat ? u.cin / 2 : u.cin;
            if (to_cat) { a.y2 = B.dcatB[j]; a.y_split = u.cin / 2; }
            a.N = N; a.D = li.D; a.H = li.H; a.W = li.W; a.Cout = u.cin; a.planar = pl; a.partial = B.skws;
            { ProfB pr(plan, s, k, 1); RUN(launch_conv_b16(a, s)); }
            g = out; g_ldc = to_cat ? u.cin / 2 : u.cin;      // concat: the next unit (upconv) reads the first half
        }
    }
    if (!bias_jobs.empty()) RUN(launch_colsum_multi(bias_jobs.data(), (int)bias_jobs.size(), s));
    if (!wsk_layers.empty()) RUN(launch_wgrad_b16_sk(wsk_layers.data(), (int)wsk_layers.size(), B.wsk_slab, B.wsk_floats, s));
    if (!wred.empty()) RUN(launch_wgrad_reduce_multi(wred.data(), (int)wred.size(), s));
    if (!event_done) E3_CHECK_HIP(hipEventRecord((hipEvent_t)bucket_event, s));
    return E3_OK;
}

}  // extern "C"
