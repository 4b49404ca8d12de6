// Whole-network executor: UNet.forward and its backward as a fixed sequence of stream-ordered launches.
//
// Replaces the nn.Module graph walk + autograd tape of elektronn3/models/unet.py:894-916 (UNet.forward),
// :244-253 (DownConv.forward), :384-408 (UpConv.forward), :256-325 (autocrop).  Host code only; every launch goes
// to the caller's HIP stream, nothing synchronises, no device memory is allocated here.
//
// Memory plan (fp32 NDHWC, sized for 288 GB HBM: nothing is recomputed, nothing is re-read that need not be):
//   saved   per conv unit: raw conv output x (BN input), activation a = relu(bn(x)); encoder skip activations are
//           written straight into the second half of the decoder's concat buffer, the up-convolved activation
//           into the first half (torch.cat, unet.py:398-399, never runs); pooled outputs; per-BN mean/invstd/
//           scale/shift vectors.
//   scratch packed weights, BN statistic records, gradient ping-pong buffers per level, split-K slabs.
#include <atomic>
#include <cstdlib>
#include <map>
#include <unordered_map>
#include <vector>

#include "plan_internal.h"

namespace {

int add_param(e3_unet_plan* p, const std::string& name, int64_t numel, int kind) {
    p->params.push_back({name, numel, kind});
    return (int)p->params.size() - 1;
}

void add_unit(e3_unet_plan* p, const std::string& conv, const std::string& bn, int cin, int cout, int level, int planar, int is_up, bool norm,
              const std::string& act = "") {
    const bool group = p->cfg.normalization == 2;    // nn.GroupNorm: weight and bias only, no running statistics
    ConvUnit u;
    u.name = conv; u.bn_name = bn; u.cin = cin; u.cout = cout; u.level = level; u.planar = planar; u.is_up = is_up;
    const int taps = is_up == 1 ? (planar ? 4 : 8) : (planar ? 9 : 27);
    const bool k1 = is_up == 2 && p->cfg.up_resize >= 3;      // ResizeConv(kernel_size=1): 'resizeconv_nearest1' / 'resizeconv_linear1'
    u.p_w = add_param(p, conv + ".weight", (int64_t)cin * cout * (k1 ? 1 : taps), 0);
    u.p_b = add_param(p, conv + ".bias", cout, 0);
    u.p_g = u.p_be = u.p_rm = u.p_rv = -1; u.bn_index = -1;
    if (norm) {
        u.p_g = add_param(p, bn + ".weight", cout, 0);
        u.p_be = add_param(p, bn + ".bias", cout, 0);
        if (!group) {
            u.p_rm = add_param(p, bn + ".running_mean", cout, 1);
            u.p_rv = add_param(p, bn + ".running_var", cout, 1);
        }
        u.bn_index = p->n_bn++;
    }
    u.p_a = (p->cfg.act_slope == ACT_PRELU && !act.empty()) ? add_param(p, act + ".weight", 1, 0) : -1;
    p->units.push_back(u);
}

LevelDims mkdims(int N, int D, int H, int W) { return {D, H, W, (size_t)N * (size_t)(D > 0 ? D : 0) * (size_t)(H > 0 ? H : 0) * (size_t)(W > 0 ? W : 0)}; }

}  // namespace

e3_unet_plan::RRelu& e3_unet_plan::rrelu_state() const {
    static thread_local std::unordered_map<unsigned, RRelu> tls;      // (uid, not the address: a destroyed plan's address may be reused)
    return tls[uid];
}

void net_dims(const e3_unet_plan* p, int N, int D, int H, int W, NetDims& nd) {
    const int nb = p->cfg.n_blocks;
    const bool valid = p->cfg.conv_valid != 0;
    nd.u.assign(p->units.size(), UnitDims{});
    nd.E.assign(nb, LevelDims{}); nd.X.assign(nb, LevelDims{});
    nd.sd_.assign(nb, 0); nd.sh_.assign(nb, 0); nd.sw_.assign(nb, 0);
    nd.ok = true;
    auto pymod2 = [](int a) { return ((a % 2) + 2) % 2; };
    LevelDims cur = mkdims(N, D, H, W);
    size_t k = 0;
    auto plain = [&](int planar) {
        UnitDims& ud = nd.u[k++];
        ud.in = cur;
        const int m = planar ? 0 : 1;
        ud.od = valid ? m : 0; ud.oh = ud.ow = valid ? 1 : 0;
        ud.out = valid ? mkdims(N, cur.D - 2 * m, cur.H - 2, cur.W - 2) : cur;
        cur = ud.out;
        if (cur.D < 1 || cur.H < 1 || cur.W < 1) nd.ok = false;
    };
    for (int i = 0; i < nb; ++i) {
        nd.X[i] = cur;
        for (int c = 0; c < p->enc_convs; ++c) plain(p->planar(i));
        nd.E[i] = cur;
        if (i + 1 < nb) cur = mkdims(N, cdiv(cur.D, p->planar(i) ? 1 : 2), cdiv(cur.H, 2), cdiv(cur.W, 2));   // MaxPool3d(ceil_mode=True), unet.py:229
    }
    for (int j = nb - 2; j >= 0; --j) {
        UnitDims& ud = nd.u[k++];
        ud.in = cur; ud.od = ud.oh = ud.ow = 0;
        const int fd = cur.D * (p->planar(j) ? 1 : 2), fh = cur.H * 2, fw = cur.W * 2;      // full size of the up-convolved tensor
        const LevelDims& e = nd.E[j];
        ud.out = mkdims(N, fd - pymod2(fd - e.D), fh - pymod2(fh - e.H), fw - pymod2(fw - e.W));
        if (ud.out.D > e.D || ud.out.H > e.H || ud.out.W > e.W || ud.out.D < 1 || ud.out.H < 1 || ud.out.W < 1) nd.ok = false;
        nd.sd_[j] = (e.D - ud.out.D) / 2; nd.sh_[j] = (e.H - ud.out.H) / 2; nd.sw_[j] = (e.W - ud.out.W) / 2;
        cur = ud.out;
        for (int c = 0; c < p->dec_convs; ++c) plain(p->planar(j));
    }
    nd.Y = cur;
}

namespace {

// ---- buffers -----------------------------------------------------------------------------------------------
struct UnitBufs { float* raw; float* act; int act_ldc; float* mean; float* invstd; float* scale; float* shift; };

struct Buffers {
    // saved
    std::vector<UnitBufs> ub;            // per unit
    std::vector<float*> cat;             // per level < nb-1: [vox][2C]
    std::vector<float*> pooled;          // per level < nb-1: [vox(level+1)][C(level)]
    std::vector<float*> sum;             // merge_mode='add': per level < nb-1 [vox][C] = up + skip (the two halves of `cat`)
    float* xin;                          // NDHWC copy of the input when in_channels > 1
    size_t saved_bytes;
    // scratch
    float* wpack; float* stats; float* bnpart; float* slab; float* small;   // small: coef / fold vectors
    float* bnred;                                                              // pre-merged BN statistic records
    float* ones; float* zeros;                                                 // [Cmax] / [2*Cmax] constants for units without a norm
    std::vector<float*> ups;             // ResizeConv units: the up-sampled input [N, sd*D', 2H', 2W', Cin] (kept for the weight gradient)
    float* wemb; float* gemb;            // ResizeConv(kernel_size=1): weights / weight gradient embedded as the centre tap of a 27-tap kernel
    std::vector<float*> slab_u;          // per unit: own wgrad slab where the slab reduction is deferred to ONE launch (nullptr: B.slab, reduced on the spot)
    // Round 6: the Winograd weight gradients of all plain 3x3x3 convs run as ONE stream-K launch at the end of the backward (launch_wgrad_wino_sk): such a unit keeps
    // the gradient of its raw output in a buffer of its own until then, and the launch has one pool of tile slabs (E3_WGRAD_NO_DEFER=1: one launch per layer)
    std::vector<float*> dz_u;
    float* wsk_slab = nullptr; size_t wsk_floats = 0;
    float* skws = nullptr;               // split-K partial sums of the bottom-level convs (training only)
    float* rtmp; float* rpad; float* rdu; // ResizeConv scratch: conv output / padded gradient at the up-sampled size, gradient of the up-sampled input
    float* biaspart0;                    // [splits][Cout] conv-bias gradient partials of the first conv when its BN backward is fused into its wgrad
    std::vector<float*> bnpart_u;        // per unit: block partials of the BN backward [parts][3][C]; row 2 (sum dx = conv-bias gradient) is summed for all units at once
    std::vector<float*> wpk_u;           // inference: per unit, its OWN buffer for what is otherwise packed on the spot into the shared `wpack` -- packed weights then survive the call (E3_FWD_REUSE_PACKED)
    std::vector<float*> wpk_f, wpk_d;    // per unit: Winograd-transformed weights (forward / dgrad form), all packed by ONE launch; nullptr = packed on the spot into wpack
    std::vector<float*> g1, g2, dcat;    // gradient buffers per level
    std::vector<float*> gskip;           // conv_mode='valid' / attention: gradient of the (un-cropped) skip activation per level
    // GridAttention (cfg.attention): per level the saved gate tensors; shared scratch sized for the largest level
    struct AttBufs { float *f, *sgm, *att, *raw, *mean, *invstd, *scale, *shift, *x, *bnpart, *gx; };   // x: centre-cropped skip ('valid'); gx: its gradient
    std::vector<AttBufs> att;
    // ResUNet residual units: [2][vox][C] = (conv accumulations, shortcut) summed by the statistics pass; per level the gradient of such a sum
    float* res2 = nullptr;
    std::vector<float*> gres;
    float *att_phi = nullptr, *att_phires = nullptr, *att_tmp = nullptr, *att_tf = nullptr, *att_tc = nullptr, *att_df = nullptr, *att_dphi = nullptr,
          *att_part = nullptr;
    float* evalA; float* evalB;          // inference ping-pong (level-0 sized)
    size_t scratch_bytes;
};

size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

int pad_cols(int n) { const int t = conv_col_tile(n); return cdiv(n, t) * t; }

// grids of the GridAttention of the decoder block whose output is level j (unit index `up`): x = the skip cropped to the block's grid
AttDims att_dims(const e3_unet_plan* p, const NetDims& ND, int N, int j, size_t up) {
    AttDims d{};
    const LevelDims& lo = ND.u[up].out; const LevelDims& li = ND.u[up].in;
    d.N = N; d.C = p->chan(j); d.D = lo.D; d.H = lo.H; d.W = lo.W;
    d.sd = p->cfg.attention == 2 ? 1 : 2;
    d.d = lo.D / d.sd; d.h = lo.H / 2; d.w = lo.W / 2;        // theta: kernel = stride = 2, no padding
    d.gd = li.D; d.gh = li.H; d.gw = li.W;
    return d;
}
AttParams att_params(void* const* t, const AttUnit& au) {
    auto P = [&](int i) { return (float*)t[i]; };
    return AttParams{P(au.p_ww), P(au.p_wb), P(au.p_theta), P(au.p_phi_w), P(au.p_phi_b), P(au.p_psi_w), P(au.p_psi_b)};
}

void plan_buffers(const e3_unet_plan* p, int N, int D, int H, int W, bool training, void* saved, void* scratch, Buffers& B) {
    const int nb = p->cfg.n_blocks;
    NetDims ND; net_dims(p, N, D, H, W, ND);
    const bool valid = p->cfg.conv_valid != 0;
    auto up_unit = [&](int j) { return (size_t)p->up_unit[j]; };      // index of the up-conv unit whose output is level j
    Arena S(saved), T(scratch);
    B.ub.assign(p->units.size(), UnitBufs{});
    B.cat.assign(nb, nullptr); B.pooled.assign(nb, nullptr); B.sum.assign(nb, nullptr);
    B.ups.assign(p->units.size(), nullptr); B.rtmp = B.rpad = B.rdu = nullptr;
    B.g1.assign(nb, nullptr); B.g2.assign(nb, nullptr); B.dcat.assign(nb, nullptr); B.gskip.assign(nb, nullptr); B.gres.assign(nb, nullptr);
    B.xin = nullptr; B.evalA = B.evalB = nullptr;
    Arena& A = training ? S : T;   // in inference everything is scratch
    if (p->cfg.in_channels > 1) B.xin = A.take(ND.X[0].vox * p->cfg.in_channels);
    for (int j = 0; j + 1 < nb; ++j) {
        const size_t cvox = ND.u[up_unit(j)].out.vox;            // the decoder works on the (cropped) up-convolved grid
        B.cat[j] = A.take(cvox * 2 * p->chan(j));
        B.pooled[j] = A.take(ND.X[j + 1].vox * p->chan(j));
        if (p->cfg.merge_add) B.sum[j] = A.take(cvox * p->chan(j));
    }
    for (size_t k = 0; k < p->units.size(); ++k) {
        const ConvUnit& u = p->units[k];
        UnitBufs& b = B.ub[k];
        const size_t n = ND.u[k].out.vox * u.cout;
        // without batch statistics to wait for, the conv writes relu(acc*scale + shift) directly; other activations are not in the conv
        // epilogues and take the two-pass route (raw tensor, then the apply pass)
        b.raw = ((training && u.has_norm()) || p->cfg.act_slope != 0.f || u.is_up == 2 || (valid && !u.is_up) || u.res_in >= 0) ? A.take(n) : nullptr;
        if (u.is_up == 2 && p->cfg.up_resize < 3) {       // (the 1x1x1 variants convolve at low resolution: no up-sampled input)
            const LevelDims& li = ND.u[k].in;
            B.ups[k] = A.take((size_t)N * li.D * (u.planar ? 1 : 2) * li.H * 2 * li.W * 2 * u.cin);
        }
        // where does the activation go?
        const bool enc_skip = u.enc_last && u.level < nb - 1;
        // ('valid': the skip is larger than the decoder's grid and gets centre-cropped into the concat buffer by a copy)
        if (enc_skip && !valid && !p->cfg.attention) { b.act = B.cat[u.level] ? B.cat[u.level] + u.cout : nullptr; b.act_ldc = 2 * u.cout; }
        else if (u.is_up) { b.act = B.cat[u.level]; b.act_ldc = 2 * u.cout; }
        else { b.act = A.take(n); b.act_ldc = u.cout; }
        if (!saved && !scratch) { b.act = nullptr; }
        b.mean = A.take(u.cout); b.invstd = A.take(u.cout); b.scale = A.take(u.cout); b.shift = A.take(u.cout);
    }
    const int att_on = p->cfg.attention;
    B.att.assign(att_on ? nb : 0, Buffers::AttBufs{});
    size_t att_fine = 0, att_coarse = 0, att_dec = 0, att_cf = 0, att_partmax = 0;      // maxima over the levels (voxels; att_cf: fine voxels x C)
    for (int j = 0; att_on && j + 1 < nb; ++j) {
        const AttDims ad = att_dims(p, ND, N, j, up_unit(j));
        const size_t fine = ND.u[up_unit(j)].out.vox, coarse = (size_t)N * (ad.d > 0 ? ad.d : 0) * ad.h * ad.w, dec = ND.u[up_unit(j)].in.vox;
        const int C = ad.C, Ci = C / 2;
        Buffers::AttBufs& ab = B.att[j];
        ab.f = A.take(coarse * Ci); ab.sgm = A.take(coarse); ab.att = A.take(fine);
        ab.raw = training ? A.take(fine * C) : nullptr;
        ab.mean = A.take(C); ab.invstd = A.take(C); ab.scale = A.take(C); ab.shift = A.take(C);
        ab.x = valid ? A.take(fine * C) : nullptr;
        att_fine = max_sz(att_fine, fine); att_coarse = max_sz(att_coarse, coarse); att_dec = max_sz(att_dec, dec * Ci); att_cf = max_sz(att_cf, fine * C);
        if (training && ad.d > 0 && ad.h > 0 && ad.w > 0) att_partmax = max_sz(att_partmax, att_part_floats(ad));
    }
    B.saved_bytes = S.off;
    // scratch
    size_t wmax = 0, statmax = 0, slabmax = 0, bnpartmax = 0, rtmpmax = 0, rdumax = 0;
    std::vector<size_t> slab_own(p->units.size(), 0);     // slab of a unit whose reduction can be deferred (transposed conv, plain conv with cin >= 8)
    for (size_t k = 0; k < p->units.size(); ++k) {
        const ConvUnit& u = p->units[k];
        const LevelDims& lo = ND.u[k].out;
        const LevelDims& ci = ND.u[k].in;          // plain convs: the grid the conv kernel runs on
        if (u.is_up == 2) {      // ResizeConv: a 3x3x3 / 1x3x3 conv on the up-sampled grid
            const LevelDims& li = ND.u[k].in;
            const int Ud = li.D * (u.planar ? 1 : 2), Uh = li.H * 2, Uw = li.W * 2, taps = u.planar ? 9 : 27;
            const ConvKind kind = u.planar ? CONV_K3_PLANAR : CONV_K3;
            const size_t uvox = (size_t)N * Ud * Uh * Uw;
            wmax = max_sz(wmax, max_sz(conv_packed_floats(kind, u.cin, u.cout), conv_packed_floats(kind, u.cout, u.cin)));
            statmax = max_sz(statmax, (size_t)conv_stats_parts(kind, 0, N, Ud, Uh, Uw, 2, u.cin, u.cout) * u.cout * 3);
            statmax = max_sz(statmax, (size_t)crop_stats_parts(lo.vox, u.cout) * u.cout * 3);
            rtmpmax = max_sz(rtmpmax, uvox * u.cout); rdumax = max_sz(rdumax, uvox * u.cin);
            if (p->cfg.up_resize >= 3) {   // 1x1x1 variants: centre tap of a 1x3x3 kernel on the LOW-resolution grid
                wmax = max_sz(wmax, max_sz(conv_packed_floats(CONV_K3_PLANAR, u.cin, u.cout), conv_packed_floats(CONV_K3_PLANAR, u.cout, u.cin)));
                if (training) slabmax = max_sz(slabmax, (size_t)wgrad_splits(CONV_K3_PLANAR, N, li.D, li.H, li.W, u.cin, u.cout) * 9 * (cdiv(u.cout, 32) * 32) * (cdiv(u.cin, 32) * 32));
            }
            if (training) slabmax = max_sz(slabmax, (size_t)wgrad_splits(kind, N, Ud, Uh, Uw, u.cin, u.cout) * taps * (cdiv(u.cout, 32) * 32) * (cdiv(u.cin, 32) * 32));
        } else if (u.is_up) {
            const int sd = u.planar ? 1 : 2, taps = sd * 4;
            const LevelDims& li = ND.u[k].in;
            wmax = max_sz(wmax, max_sz((size_t)pad_cols(taps * u.cout) * u.cin, (size_t)taps * pad_cols(u.cin) * u.cout));
            statmax = max_sz(statmax, (size_t)conv_stats_parts(CONV_POINT, CF_SCATTER_UP, N, li.D, li.H, li.W, sd, u.cin, taps * u.cout) * u.cout * 3);
            if (training) { slab_own[k] = (size_t)wgrad_splits(CONV_POINT, N, li.D, li.H, li.W, u.cin, u.cout) * taps * (cdiv(u.cout, 32) * 32) * (cdiv(u.cin, 32) * 32); slabmax = max_sz(slabmax, slab_own[k]); }
        } else {
            const int taps = u.planar ? 9 : 27;
            const ConvKind kind = u.planar ? CONV_K3_PLANAR : CONV_K3;
            if (u.cin >= 8) {
                wmax = max_sz(wmax, max_sz(conv_packed_floats(kind, u.cin, u.cout), conv_packed_floats(kind, u.cout, u.cin)));
                statmax = max_sz(statmax, (size_t)conv_stats_parts(kind, 0, N, ci.D, ci.H, ci.W, 2, u.cin, u.cout) * u.cout * 3);
                if (training) { slab_own[k] = (size_t)wgrad_splits(kind, N, ci.D, ci.H, ci.W, u.cin, u.cout) * taps * (cdiv(u.cout, 32) * 32) * (cdiv(u.cin, 32) * 32); slabmax = max_sz(slabmax, slab_own[k]); }
            } else {
                wmax = max_sz(wmax, conv_packed_floats(kind, u.cout, u.cin));   // only its dgrad (dx requested) packs weights
                statmax = max_sz(statmax, (size_t)conv_small_stats_parts2(N, ci.D, ci.H, ci.W, u.planar, u.cin, u.cout) * u.cout * 3);
                if (training) slabmax = max_sz(slabmax, (size_t)conv_small_wgrad_splits(N, ci.D, ci.H, ci.W, u.planar) * taps * u.cout * u.cin);
            }
        }
        if (valid && !u.is_up) {     // conv on the input grid into a temporary, crop + statistics into the unit's tensor; padded gradient back
            statmax = max_sz(statmax, (size_t)crop_stats_parts(lo.vox, u.cout) * u.cout * 3);
            rtmpmax = max_sz(rtmpmax, ci.vox * u.cout);
        }
        if (training) bnpartmax = max_sz(bnpartmax, (size_t)bn_bwd_parts(lo.vox, u.cout) * 3 * u.cout);
    }
    if (training) {
        const int C0 = p->chan(0);
        slabmax = max_sz(slabmax, (size_t)conv_final_bwd_parts(ND.Y.vox) * (p->cfg.out_channels * C0 + p->cfg.out_channels));
        // (e3_unet_backward_loss: the head's partial rows ride in the last unit's BatchNorm-backward reduce pass, one row per workgroup of THAT pass)
        slabmax = max_sz(slabmax, (size_t)bn_bwd_parts(ND.Y.vox, C0) * (p->cfg.out_channels * C0 + p->cfg.out_channels));
    }
    B.wpack = T.take(wmax);
    B.wemb = B.gemb = nullptr;
    if (p->cfg.up_resize >= 3) { const size_t e = (size_t)p->chan(nb - 1) * p->chan(nb - 1) / 2 * 27; B.wemb = T.take(e); if (training) B.gemb = T.take(e); }
    if (rtmpmax) { B.rtmp = T.take(rtmpmax); if (rdumax) B.rdu = T.take(rdumax); if (training) B.rpad = T.take(rtmpmax); }
    B.wpk_f.assign(p->units.size(), nullptr); B.wpk_d.assign(p->units.size(), nullptr);
    size_t skmax = 0;
    for (size_t k = 0; k < p->units.size(); ++k) {
        const ConvUnit& u = p->units[k];
        const LevelDims& lo = ND.u[k].in;          // (the conv grid)
        if (u.is_up || u.planar || u.cin < 8) continue;
        // split-K (training only; forward: units with batch statistics, 'same' convs): the splits count when the Winograd grid is sized
        // (a residual unit's conv writes plain accumulations next to its shortcut: no split-K in its forward)
        const bool skf = training && u.has_norm() && !valid && u.res_in < 0, skd = training && !valid;
        const int sf = skf ? conv_wino_splitk(lo.D, lo.H, lo.W, u.cin, u.cout) : 0, sd = skd ? conv_wino_splitk(lo.D, lo.H, lo.W, u.cout, u.cin) : 0;
        const bool wf = sf > 0 || conv_use_wino(CONV_K3, 0, N, lo.D, lo.H, lo.W, u.cin, u.cout);
        const bool wd = training && (sd > 0 || conv_use_wino(CONV_K3, 0, N, lo.D, lo.H, lo.W, u.cout, u.cin));
        if (wf) B.wpk_f[k] = T.take(conv_packed_floats(CONV_K3, u.cin, u.cout));
        if (wd) B.wpk_d[k] = T.take(conv_packed_floats(CONV_K3, u.cout, u.cin));
        if (training) {      // split-K partial sums (forward: [S][vox][cout], dgrad: [S][vox][cin])
            const size_t vox = (size_t)N * lo.D * lo.H * lo.W;
            if (sf > 1) skmax = max_sz(skmax, sf * vox * u.cout);
            if (sd > 1) skmax = max_sz(skmax, sd * vox * u.cin);
            if (sf > 1) statmax = max_sz(statmax, (size_t)crop_stats_parts(vox, u.cout) * u.cout * 3);
        }
    }
    if (skmax) B.skws = T.take(skmax);
    B.wpk_u.assign(p->units.size(), nullptr);
    if (!training) {
        for (size_t k = 0; k < p->units.size(); ++k) {
            const ConvUnit& u = p->units[k];
            const ConvKind kind = u.planar ? CONV_K3_PLANAR : CONV_K3;
            size_t n = 0;
            if (u.is_up == 2) n = max_sz(conv_packed_floats(kind, u.cin, u.cout), p->cfg.up_resize >= 3 ? conv_packed_floats(CONV_K3_PLANAR, u.cin, u.cout) : 0);
            else if (u.is_up) n = (size_t)pad_cols((u.planar ? 4 : 8) * u.cout) * u.cin;
            else if (u.cin >= 8 && !B.wpk_f[k]) n = conv_packed_floats(kind, u.cin, u.cout);
            if (n) B.wpk_u[k] = T.take(n);
        }
    }
    for (int j = 0; att_on && j + 1 < nb; ++j) {
        const LevelDims& lo = ND.u[up_unit(j)].out;
        statmax = max_sz(statmax, (size_t)crop_stats_parts(lo.vox, p->chan(j)) * p->chan(j) * 3);
    }
    size_t res2max = 0;
    for (size_t k = 0; k < p->units.size(); ++k) {
        const ConvUnit& u = p->units[k];
        if (u.res_in < 0) continue;
        const size_t vox = ND.u[k].out.vox;
        res2max = max_sz(res2max, 2 * vox * u.cout);
        statmax = max_sz(statmax, (size_t)crop_stats_parts(vox, u.cout) * u.cout * 3);
        if (training && u.p_pw >= 0) att_partmax = max_sz(att_partmax, pw_part_floats(vox, u.cout, p->units[u.res_in].cin));
    }
    if (res2max) B.res2 = T.take(res2max);
    B.stats = T.take(statmax);
    if (!att_on && training && att_partmax) B.att_part = T.take(att_partmax);
    if (att_on) {
        B.att_phi = T.take(att_dec);
        size_t cc = 0;      // coarse voxels x C/2, maximum over the levels
        for (int j = 0; j + 1 < nb; ++j) { const AttDims ad = att_dims(p, ND, N, j, up_unit(j)); cc = max_sz(cc, (size_t)N * (ad.d > 0 ? ad.d : 0) * ad.h * ad.w * (ad.C / 2)); }
        B.att_phires = T.take(cc);
        if (training) {
            B.att_tmp = T.take(att_cf);
            B.att_tf = T.take(att_fine); B.att_tc = T.take(2 * att_coarse); B.att_df = T.take(cc); B.att_dphi = T.take(att_dec);
            B.att_part = T.take(att_partmax);
            for (int j = 0; j + 1 < nb; ++j) {
                const LevelDims& lo = ND.u[up_unit(j)].out;
                B.att[j].bnpart = T.take((size_t)bn_bwd_parts(lo.vox, p->chan(j)) * 3 * p->chan(j));
            }
        }
    }
    B.small = T.take((size_t)5 * p->chan(nb - 1) + 64);   // [4][C] backward coefficients + [C] PReLU scratch
    B.bnred = T.take((size_t)BN_PRERED * p->chan(nb - 1) * 3);
    B.ones = T.take(p->chan(nb - 1)); B.zeros = T.take((size_t)4 * p->chan(nb - 1));
    B.bnpart_u.assign(p->units.size(), nullptr);
    B.biaspart0 = nullptr;
    if (training) {
        if (p->units[0].cin < 8)
            B.biaspart0 = T.take((size_t)conv_small_wgrad_splits(N, ND.X[0].D, ND.X[0].H, ND.X[0].W, p->units[0].planar) * p->units[0].cout);
        for (size_t k = 0; k < p->units.size(); ++k)
            B.bnpart_u[k] = T.take((size_t)bn_bwd_parts(ND.u[k].out.vox, p->units[k].cout) * 3 * p->units[k].cout);
        B.bnpart = T.take(bnpartmax);
        B.slab = T.take(slabmax);
        static const bool batch_reduce = getenv("E3_NO_REDUCE_BATCH") == nullptr;
        B.slab_u.assign(p->units.size(), nullptr);
        B.dz_u.assign(p->units.size(), nullptr);
        static const bool no_defer = getenv("E3_WGRAD_NO_DEFER") != nullptr;      // A/B switch
        int sk_tile_pairs = 0;
        for (size_t k = 0; !no_defer && k < p->units.size(); ++k) {
            const ConvUnit& u = p->units[k];
            if (valid || u.is_up || u.planar || u.cin < 8 || u.res_in >= 0 || !wgrad_use_wino(CONV_K3)) continue;
            bool feeds_res = false;      // (a ConvBlock's first conv: its data gradient takes the shortcut's along -- left on the per-layer path)
            for (size_t q = 0; q < p->units.size(); ++q) feeds_res = feeds_res || p->units[q].res_in == (int)k;
            if (feeds_res) continue;
            // Not the level-0 layers: their dZ (268 MB at cfg 2) is read by the weight gradient right behind the pass that wrote it -- partly out of the memory-side
            // cache --, deferred it comes from HBM with the staging stalls that brings.  Same box, cfg-2 step: one launch per layer 11.054 ms, layers up to 80 MB
            // (levels 1 - 3) deferred 11.006, ALL layers 11.140 (profiles/r06_wgrad_streamk.md, second table).  E3_WGRAD_DEFER_MAX_MB moves the limit.
            static const double defer_max_mb = getenv("E3_WGRAD_DEFER_MAX_MB") ? atof(getenv("E3_WGRAD_DEFER_MAX_MB")) : 80.0;
            if ((double)ND.u[k].out.vox * u.cout * 4.0 > defer_max_mb * 1048576.0) continue;
            B.dz_u[k] = T.take(ND.u[k].out.vox * u.cout);
            sk_tile_pairs += cdiv(u.cout, 32) * cdiv(u.cin, 32);
        }
        if (sk_tile_pairs) { B.wsk_floats = wgrad_wino_sk_slab_floats(sk_tile_pairs); B.wsk_slab = T.take(B.wsk_floats); }
        for (size_t k = 0; batch_reduce && k < p->units.size(); ++k)
            if (slab_own[k] && !B.dz_u[k]) B.slab_u[k] = T.take(slab_own[k]);
        for (int j = 0; j < nb; ++j) {
            const size_t n = ND.X[j].vox * p->chan(j);        // the level's input grid is its largest
            B.g1[j] = T.take(n); B.g2[j] = T.take(n);
            if (j + 1 < nb) B.dcat[j] = T.take(2 * n);
            if (res2max) B.gres[j] = T.take(n);
            if ((valid || att_on) && j + 1 < nb) B.gskip[j] = T.take(ND.E[j].vox * p->chan(j));
            // attention: gradient of the (cropped) skip as the gate sees it; 'same' convs: that IS the skip's gradient
            if (att_on && j + 1 < nb) B.att[j].gx = valid ? T.take(ND.u[up_unit(j)].out.vox * p->chan(j)) : B.gskip[j];
        }
    } else {
        B.bnpart = B.slab = nullptr;
    }
    B.scratch_bytes = T.off;
}

struct Prof {
    e3_unet_plan* p; hipStream_t s; bool on;
    Prof(e3_unet_plan* plan, hipStream_t st, int layer, int which) : p(plan), s(st) {
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
    ~Prof() { if (on) { (void)hipEventRecord(p->prof_events[p->prof_used].second, s); p->prof_used++; } }
};

#define RUN(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

}  // namespace

// scratch pointer -> what its packed weights belong to (plan_internal.h: PackedSig)
static std::mutex g_sig_mutex;
static std::map<const void*, PackedSig> g_sigs;
void packed_sig_forget(const void* scratch) { std::lock_guard<std::mutex> g(g_sig_mutex); g_sigs.erase(scratch); }
// true when the record of `scratch` equals `sig`; the record is removed either way (the caller puts it back when its call has succeeded)
static bool packed_sig_take(const void* scratch, const PackedSig& sig) {
    std::lock_guard<std::mutex> g(g_sig_mutex);
    auto it = g_sigs.find(scratch);
    const bool same = it != g_sigs.end() && it->second == sig;
    if (it != g_sigs.end()) g_sigs.erase(it);
    return same;
}
static void packed_sig_record(const void* scratch, const PackedSig& sig) { std::lock_guard<std::mutex> g(g_sig_mutex); g_sigs[scratch] = sig; }

std::vector<NeedBox> need_boxes(const e3_unet_plan* plan, const NetDims& ND, const int* roi) {
    std::vector<NeedBox> need(plan->units.size());
    if (!roi) return need;
    NeedBox b; b.on = true;
    const int yd[3] = {ND.Y.D, ND.Y.H, ND.Y.W};
    for (int i = 0; i < 3; ++i) { b.lo[i] = roi[i] < yd[i] ? roi[i] : yd[i] - 1; b.hi[i] = roi[3 + i] < yd[i] ? roi[3 + i] : yd[i]; }
    for (size_t k = plan->units.size(); k-- > 0 && b.on;) {
        const ConvUnit& u = plan->units[k];
        if (u.is_down || u.res_in >= 0) break;
        need[k] = b;
        const int di[3] = {ND.u[k].in.D, ND.u[k].in.H, ND.u[k].in.W};
        if (u.is_up == 1) {            // transposed conv, kernel = stride: output voxel o reads input voxel o / stride
            const int st[3] = {u.planar ? 1 : 2, 2, 2};
            for (int i = 0; i < 3; ++i) { b.lo[i] = b.lo[i] / st[i]; b.hi[i] = (b.hi[i] + st[i] - 1) / st[i]; }
        } else if (u.is_up) {
            b.on = false;              // (ResizeConv: everything in front of it is computed in full)
        } else {
            const int r[3] = {u.planar ? 0 : 1, 1, 1};
            // F(2x2x4) Winograd tiles (conv_wino4.hip): an F(2,3) output never touches the tile's far input -- structural zeros of the transforms -- but
            // the four outputs of an F(4,3) tile meet all six of its inputs and are independent of the far ones only up to rounding: every input of
            // every W tile that holds a needed output must be a computed value, so the box grows to the tile grid along W first
            if (!u.planar && u.cin >= 8 && conv_use_wino(CONV_K3, 0, 1, di[0], di[1], di[2], u.cin, u.cout) &&
                conv_wino_layout(CF_WINO4, di[0], di[1], di[2], u.cin, u.cout, 1) == 2) { b.lo[2] = b.lo[2] / 4 * 4; b.hi[2] = (b.hi[2] + 3) / 4 * 4; }
            for (int i = 0; i < 3; ++i) { b.lo[i] = b.lo[i] - r[i] < 0 ? 0 : b.lo[i] - r[i]; b.hi[i] = b.hi[i] + r[i] > di[i] ? di[i] : b.hi[i] + r[i]; }
        }
        for (int i = 0; i < 3; ++i) if (b.hi[i] > di[i]) b.hi[i] = di[i];
    }
    return need;
}

extern "C" {

int e3_unet_plan_create(const e3_unet_cfg* cfg, e3_unet_plan** out) {
    E3_REQUIRE(cfg && out, E3_ERR_INVALID, "null argument");
    E3_REQUIRE(cfg->n_blocks >= 1 && cfg->n_blocks <= 8, E3_ERR_INVALID, "n_blocks must be in 1..8");
    E3_REQUIRE(cfg->in_channels >= 1, E3_ERR_INVALID, "in_channels must be >= 1");
    E3_REQUIRE(cfg->in_channels < 8 || cfg->in_channels % 8 == 0, E3_ERR_UNSUPPORTED, "in_channels must be < 8 or a multiple of 8");
    E3_REQUIRE(cfg->out_channels >= 1 && cfg->out_channels <= 16, E3_ERR_UNSUPPORTED, "out_channels must be in 1..16 on the HIP path");
    E3_REQUIRE(cfg->start_filts >= 8 && cfg->start_filts % 8 == 0, E3_ERR_UNSUPPORTED, "start_filts must be a multiple of 8 on the HIP path");
    E3_REQUIRE((cfg->start_filts << (cfg->n_blocks - 1)) <= 1024, E3_ERR_UNSUPPORTED, "more than 1024 channels at the bottom level");
    E3_REQUIRE((cfg->act_slope >= 0.f && cfg->act_slope <= 1.f) || cfg->act_slope == ACT_SILU || cfg->act_slope == ACT_PRELU, E3_ERR_INVALID,
               "act_slope must be in [0, 1] (0 ReLU, 0.1 LeakyReLU, 1 identity), 2 (SiLU) or 3 (PReLU: learnable slopes in the parameter table)");
    E3_REQUIRE(cfg->normalization >= 0 && cfg->normalization <= 2, E3_ERR_UNSUPPORTED, "normalization must be 0 (none), 1 (batch) or 2 (group)");
    if (cfg->normalization == 2)
        E3_REQUIRE(cfg->num_groups >= 1 && cfg->start_filts % cfg->num_groups == 0, E3_ERR_INVALID, "num_groups must divide every channel count");
    E3_REQUIRE(cfg->attention >= 0 && cfg->attention <= 2, E3_ERR_INVALID, "attention must be 0 (off), 1 (dim=3) or 2 (dim=2)");
    E3_REQUIRE(cfg->resunet == 0 || cfg->resunet == 1, E3_ERR_INVALID, "resunet must be 0 or 1");
    E3_REQUIRE(cfg->resunet || (cfg->enc_res_blocks == 0 && cfg->dec_res_blocks == 0), E3_ERR_INVALID, "res_blocks belong to the ResUNet");
    E3_REQUIRE(cfg->enc_res_blocks >= 0 && cfg->enc_res_blocks <= 8 && cfg->dec_res_blocks >= 0 && cfg->dec_res_blocks <= 8, E3_ERR_INVALID, "res_blocks must be in 0..8");
    E3_REQUIRE(!(cfg->conv_valid && (cfg->enc_res_blocks || cfg->dec_res_blocks)), E3_ERR_UNSUPPORTED,
               "residual shortcuts need conv_mode='same' (a 'valid' conv2 is smaller than the block input it would be added to)");
    const bool last_norm = cfg->normalization != 0, all_norm = last_norm && cfg->full_norm != 0;
    e3_unet_plan* p = new e3_unet_plan();
    static std::atomic<unsigned> next_uid{1};
    p->uid = next_uid.fetch_add(1);
    p->cfg = *cfg;
    p->n_bn = 0;
    const int nb = cfg->n_blocks;
    p->att.assign(cfg->attention ? nb : 0, AttUnit{});
    p->enc_last_unit.assign(nb, -1); p->up_unit.assign(nb, -1);
    auto attention_params = [&](const std::string& b, int j, int ins, int outs) {
        // GridAttention(in_channels=outs, gating_channels=ins) (unet.py:376-379,452-505); inter_channels = outs / 2
        const std::string a = b + "attention.";
        const int T = cfg->attention == 2 ? 4 : 8, ci = outs / 2;
        AttUnit& au = p->att[j];
        au.p_ww = add_param(p, a + "w.0.weight", (int64_t)outs * outs, 0);
        au.p_wb = add_param(p, a + "w.0.bias", outs, 0);
        au.p_g = add_param(p, a + "w.1.weight", outs, 0);
        au.p_be = add_param(p, a + "w.1.bias", outs, 0);
        au.p_rm = add_param(p, a + "w.1.running_mean", outs, 1);
        au.p_rv = add_param(p, a + "w.1.running_var", outs, 1);
        au.bn_index = p->n_bn++;
        au.p_theta = add_param(p, a + "theta.weight", (int64_t)ci * outs * T, 0);
        au.p_phi_w = add_param(p, a + "phi.weight", (int64_t)ci * ins, 0);
        au.p_phi_b = add_param(p, a + "phi.bias", ci, 0);
        au.p_psi_w = add_param(p, a + "psi.weight", ci, 0);
        au.p_psi_b = add_param(p, a + "psi.bias", 1, 0);
    };
    if (!cfg->resunet) {
        for (int i = 0; i < nb; ++i) {   // unet.py:832-850
            const std::string b = "down_convs." + std::to_string(i) + ".";
            const int ins = i == 0 ? cfg->in_channels : p->chan(i - 1), outs = p->chan(i);
            add_unit(p, b + "conv1", b + "norm0", ins, outs, i, p->planar(i), 0, all_norm, b + "act1");     // unet.py:235-242
            p->units.back().is_down = true;
            add_unit(p, b + "conv2", b + "norm1", outs, outs, i, p->planar(i), 0, last_norm, b + "act2");
            p->units.back().is_down = true; p->units.back().enc_last = true;
            p->enc_last_unit[i] = (int)p->units.size() - 1;
        }
        for (int k = 0; k + 1 < nb; ++k) {   // unet.py:854-879: block k works at level nb-2-k
            const int j = nb - 2 - k;
            const std::string b = "up_convs." + std::to_string(k) + ".";
            const int ins = p->chan(j + 1), outs = p->chan(j);
            add_unit(p, b + (cfg->up_resize ? "upconv.conv" : "upconv"), b + "norm0", ins, outs, j, p->planar(j), cfg->up_resize ? 2 : 1, all_norm, b + "act0");   // unet.py:152-176,365-375
            p->up_unit[j] = (int)p->units.size() - 1;
            if (cfg->attention) attention_params(b, j, ins, outs);
            add_unit(p, b + "conv1", b + "norm1", cfg->merge_add ? outs : 2 * outs, outs, j, p->planar(j), 0, all_norm, b + "act1");   // unet.py:352-360
            p->units.back().to_cat = true;
            add_unit(p, b + "conv2", b + "norm2", outs, outs, j, p->planar(j), 0, last_norm, b + "act2");
        }
    } else {
        // elektronn3.models.resunet.UNet (resunet.py:888-934): DownBlock / UpBlock = a Sequential of ConvBlocks (conv1-norm1-act1-conv2-[+ proj(inp)]-
        // norm2-act2, resunet.py:212-262), max(1, res_blocks) of them; res_blocks >= 1 turns the shortcuts on, except from the input image
        // (skip_first_residual, resunet.py:906).  A norm follows every conv (full_norm only reaches norm0, which is always built: resunet.py:411).
        p->enc_convs = 2 * (cfg->enc_res_blocks > 1 ? cfg->enc_res_blocks : 1);
        p->dec_convs = 2 * (cfg->dec_res_blocks > 1 ? cfg->dec_res_blocks : 1);
        auto conv_block = [&](const std::string& cb, int ins, int outs, int level, bool residual, bool down, bool first_of_up) {
            add_unit(p, cb + "conv1", cb + "norm1", ins, outs, level, p->planar(level), 0, last_norm, cb + "act1");
            p->units.back().is_down = down; p->units.back().to_cat = first_of_up;
            const int first = (int)p->units.size() - 1;
            add_unit(p, cb + "conv2", cb + "norm2", outs, outs, level, p->planar(level), 0, last_norm, cb + "act2");
            ConvUnit& u2 = p->units.back();
            u2.is_down = down;
            if (residual) {
                u2.res_in = first;
                if (ins != outs) {       // "projection" to match the channel counts (resunet.py:247-251)
                    const int pw = add_param(p, cb + "proj.weight", (int64_t)ins * outs, 0), pb = add_param(p, cb + "proj.bias", outs, 0);
                    p->units.back().p_pw = pw; p->units.back().p_pb = pb;
                }
            }
        };
        for (int i = 0; i < nb; ++i) {
            const std::string b = "down_convs." + std::to_string(i) + ".convs.";
            const int ins = i == 0 ? cfg->in_channels : p->chan(i - 1), outs = p->chan(i);
            const bool res = cfg->enc_res_blocks >= 1;
            for (int c = 0; c < p->enc_convs / 2; ++c)
                conv_block(b + std::to_string(c) + ".", c == 0 ? ins : outs, outs, i, res && !(c == 0 && i == 0), true, false);
            p->units.back().enc_last = true;
            p->enc_last_unit[i] = (int)p->units.size() - 1;
        }
        for (int k = 0; k + 1 < nb; ++k) {
            const int j = nb - 2 - k;
            const std::string b = "up_convs." + std::to_string(k) + ".";
            const int ins = p->chan(j + 1), outs = p->chan(j);
            add_unit(p, b + (cfg->up_resize ? "upconv.conv" : "upconv"), b + "norm0", ins, outs, j, p->planar(j), cfg->up_resize ? 2 : 1, last_norm, b + "act0");
            p->up_unit[j] = (int)p->units.size() - 1;
            if (cfg->attention) attention_params(b, j, ins, outs);
            const bool res = cfg->dec_res_blocks >= 1;
            for (int c = 0; c < p->dec_convs / 2; ++c)
                conv_block(b + "convs." + std::to_string(c) + ".", c == 0 ? (cfg->merge_add ? outs : 2 * outs) : outs, outs, j, res, false, c == 0);
        }
    }
    p->p_final_w = add_param(p, "conv_final.weight", (int64_t)cfg->out_channels * p->chan(0), 0);
    p->p_final_b = add_param(p, "conv_final.bias", cfg->out_channels, 0);
    *out = p;
    return E3_OK;
}

void e3_unet_plan_destroy(e3_unet_plan* plan) {
    if (!plan) return;
    for (auto& e : plan->prof_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    delete plan;
}

int e3_unet_param_count(const e3_unet_plan* plan) { return plan ? (int)plan->params.size() : 0; }
int e3_unet_bn_count(const e3_unet_plan* plan) { return plan ? plan->n_bn : 0; }

int e3_unet_param_info(const e3_unet_plan* plan, int index, char* name, int name_len, int64_t* numel, int* kind) {
    E3_REQUIRE(plan && index >= 0 && index < (int)plan->params.size(), E3_ERR_INVALID, "bad param index");
    const ParamSlot& s = plan->params[index];
    if (name && name_len > 0) snprintf(name, name_len, "%s", s.name.c_str());
    if (numel) *numel = s.numel;
    if (kind) *kind = s.kind;
    return E3_OK;
}

int e3_unet_conv_count(const e3_unet_plan* plan) { return plan ? (int)plan->units.size() + 1 : 0; }

int e3_unet_conv_info(const e3_unet_plan* plan, int layer, char* name, int name_len, int* cin, int* cout, int* taps, int* level) {
    E3_REQUIRE(plan && layer >= 0 && layer <= (int)plan->units.size(), E3_ERR_INVALID, "bad layer index");
    if (layer == (int)plan->units.size()) {
        if (name && name_len > 0) snprintf(name, name_len, "conv_final");
        if (cin) *cin = plan->chan(0);
        if (cout) *cout = plan->cfg.out_channels;
        if (taps) *taps = 1;
        if (level) *level = 0;
        return E3_OK;
    }
    const ConvUnit& u = plan->units[layer];
    if (name && name_len > 0) snprintf(name, name_len, "%s", u.name.c_str());
    if (cin) *cin = u.cin;
    if (cout) *cout = u.cout;
    if (taps) *taps = u.is_up == 1 ? (u.planar ? 4 : 8) : (u.planar ? 9 : 27);
    if (level) *level = u.level;
    return E3_OK;
}

int e3_unet_set_rrelu(e3_unet_plan* plan, double lower, double upper, unsigned seed) {
    E3_REQUIRE(plan, E3_ERR_INVALID, "null plan");
    E3_REQUIRE(seed == 0 || (lower >= 0.0 && lower <= upper && upper <= 1.0), E3_ERR_INVALID, "RReLU needs 0 <= lower <= upper <= 1");
    e3_unet_plan::RRelu& r = plan->rrelu_state();       // per calling thread: the forward / backward call that follows on this thread reads it
    r.seed = seed; r.lo = (float)lower; r.hi = (float)upper;
    return E3_OK;
}

int e3_unet_attention_map(const e3_unet_plan* plan, void* stream, int N, int D, int H, int W, int training, void* saved, void* scratch,
                          int block, float* out, int* Do, int* Ho, int* Wo) {
    E3_REQUIRE(plan && N > 0 && D > 0 && H > 0 && W > 0, E3_ERR_INVALID, "bad argument");
    E3_REQUIRE(plan->cfg.attention, E3_ERR_INVALID, "the plan was built with attention = 0");
    const int nb = plan->cfg.n_blocks;
    E3_REQUIRE(block >= 0 && block < nb - 1, E3_ERR_INVALID, "block must index a decoder block");
    NetDims ND; net_dims(plan, N, D, H, W, ND);
    E3_REQUIRE(ND.ok, E3_ERR_INVALID, "input too small for this network");
    const int j = nb - 2 - block;                                  // up_convs[block] works at level nb - 2 - block
    const LevelDims& lo = ND.u[(size_t)plan->up_unit[j]].out;
    if (Do) *Do = lo.D;
    if (Ho) *Ho = lo.H;
    if (Wo) *Wo = lo.W;
    if (!out) return E3_OK;
    E3_REQUIRE(scratch && (saved || !training), E3_ERR_INVALID, "the forward's workspaces are needed");
    Buffers B;
    plan_buffers(plan, N, D, H, W, training != 0, saved, scratch, B);
    E3_CHECK_HIP(hipMemcpyAsync(out, B.att[j].att, lo.vox * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return E3_OK;
}

int e3_unet_profile_select(e3_unet_plan* plan, int layer, int which) {
    E3_REQUIRE(plan, E3_ERR_INVALID, "null plan");
    plan->prof_layer = layer; plan->prof_which = which; plan->prof_used = 0;
    return E3_OK;
}

int e3_unet_profile_read(e3_unet_plan* plan, double* mean_ms, int* launches) {
    E3_REQUIRE(plan && mean_ms && launches, E3_ERR_INVALID, "null argument");
    double tot = 0.0; int n = 0;
    for (size_t i = 0; i < plan->prof_used; ++i) {
        float ms = 0.f;
        E3_CHECK_HIP(hipEventSynchronize(plan->prof_events[i].second));
        E3_CHECK_HIP(hipEventElapsedTime(&ms, plan->prof_events[i].first, plan->prof_events[i].second));
        tot += ms; ++n;
    }
    *mean_ms = n ? tot / n : 0.0; *launches = n;
    plan->prof_used = 0;
    return E3_OK;
}

int e3_unet_out_dims(const e3_unet_plan* plan, int D, int H, int W, int* Do, int* Ho, int* Wo) {
    E3_REQUIRE(plan && D > 0 && H > 0 && W > 0 && Do && Ho && Wo, E3_ERR_INVALID, "bad argument");
    NetDims nd; net_dims(plan, 1, D, H, W, nd);
    E3_REQUIRE(nd.ok, E3_ERR_INVALID, "input too small for this network (conv_mode='valid' shrinks every conv by 2)");
    *Do = nd.Y.D; *Ho = nd.Y.H; *Wo = nd.Y.W;
    return E3_OK;
}

int e3_unet_sizes(const e3_unet_plan* plan, int N, int D, int H, int W, int training, size_t* saved_bytes, size_t* scratch_bytes) {
    E3_REQUIRE(plan && N > 0 && D > 0 && H > 0 && W > 0, E3_ERR_INVALID, "bad shape");
    { NetDims nd; net_dims(plan, N, D, H, W, nd);
      E3_REQUIRE(nd.ok, E3_ERR_INVALID, "input too small for this network (conv_mode='valid' shrinks every conv by 2)"); }
    Buffers B;
    plan_buffers(plan, N, D, H, W, training != 0, nullptr, nullptr, B);
    if (saved_bytes) *saved_bytes = B.saved_bytes;
    if (scratch_bytes) *scratch_bytes = B.scratch_bytes;
    return E3_OK;
}

static int unet_forward_impl(e3_unet_plan* plan, void* stream, const float* x, int N, int D, int H, int W,
                             void* const* params, const float* momenta, float* y,
                             void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, uint32_t flags, const e3_ce_dice_args* la,
                             const int* roi = nullptr, const e3_tile_view* view = nullptr);

int e3_unet_forward(e3_unet_plan* plan, void* stream, const float* x, int N, int D, int H, int W,
                    void* const* params, const float* momenta, float* y,
                    void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, uint32_t flags) {
    return unet_forward_impl(plan, stream, x, N, D, H, W, params, momenta, y, saved, saved_bytes, scratch, scratch_bytes, flags, nullptr);
}

int e3_unet_forward_loss(e3_unet_plan* plan, void* stream, const float* x, int N, int D, int H, int W,
                         void* const* params, const float* momenta, float* y,
                         void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, uint32_t flags, const e3_ce_dice_args* loss) {
    E3_REQUIRE(plan && loss && loss->target && loss->workspace && (loss->loss_out || loss->sums_out), E3_ERR_INVALID, "forward_loss: null criterion argument");
    E3_REQUIRE(!(flags & E3_FWD_SOFTMAX), E3_ERR_INVALID, "forward_loss: the criterion takes logits (no E3_FWD_SOFTMAX)");
    E3_REQUIRE(plan->cfg.out_channels >= 2, E3_ERR_INVALID, "forward_loss: the criterion needs at least two classes");
    E3_REQUIRE(loss->workspace_bytes >= ce_dice_workspace_floats(plan->cfg.out_channels) * sizeof(float), E3_ERR_WORKSPACE, "ce_dice workspace too small");
    return unet_forward_impl(plan, stream, x, N, D, H, W, params, momenta, y, saved, saved_bytes, scratch, scratch_bytes, flags, loss);
}

int e3_unet_forward_roi(e3_unet_plan* plan, void* stream, const float* x, int N, int D, int H, int W,
                        void* const* params, float* y, void* scratch, size_t scratch_bytes, uint32_t flags, const int roi[6]) {
    E3_REQUIRE(roi, E3_ERR_INVALID, "forward_roi: null region");
    E3_REQUIRE(!(flags & (E3_FWD_TRAINING | E3_FWD_FROZEN_BN)), E3_ERR_INVALID, "forward_roi: inference only");
    for (int i = 0; i < 3; ++i) E3_REQUIRE(roi[i] >= 0 && roi[3 + i] > roi[i], E3_ERR_INVALID, "forward_roi: empty or negative region");
    return unet_forward_impl(plan, stream, x, N, D, H, W, params, nullptr, y, nullptr, 0, scratch, scratch_bytes, flags, nullptr, roi);
}

int e3_unet_forward_tile(e3_unet_plan* plan, void* stream, const e3_tile_view* view, int N, int D, int H, int W,
                         void* const* params, void* scratch, size_t scratch_bytes, uint32_t flags, const int roi[6]) {
    E3_REQUIRE(plan && view && view->x && view->y && roi, E3_ERR_INVALID, "forward_tile: null argument");
    E3_REQUIRE(!(flags & (E3_FWD_TRAINING | E3_FWD_FROZEN_BN)), E3_ERR_INVALID, "forward_tile: inference only");
    E3_REQUIRE(plan->cfg.in_channels == 1 && !plan->cfg.conv_valid && plan->units.front().cin < 8, E3_ERR_UNSUPPORTED,
               "forward_tile: one input channel and 'same' convolutions (copy the tile and take e3_unet_forward_roi)");
    for (int i = 0; i < 3; ++i) E3_REQUIRE(roi[i] >= 0 && roi[3 + i] > roi[i], E3_ERR_INVALID, "forward_tile: empty or negative region");
    E3_REQUIRE(roi[3] <= D && roi[4] <= H && roi[5] <= W, E3_ERR_INVALID, "forward_tile: region outside the tile");
    E3_REQUIRE(view->x_stride[1] > 0 && view->x_stride[2] > 0, E3_ERR_INVALID, "forward_tile: bad input strides");
    return unet_forward_impl(plan, stream, view->x, N, D, H, W, params, nullptr, view->y, nullptr, 0, scratch, scratch_bytes, flags, nullptr, roi, view);
}

static int unet_forward_impl(e3_unet_plan* plan, void* stream, const float* x, int N, int D, int H, int W,
                             void* const* params, const float* momenta, float* y,
                             void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, uint32_t flags, const e3_ce_dice_args* la,
                             const int* roi, const e3_tile_view* view) {
    E3_REQUIRE(plan && x && y && params && scratch, E3_ERR_INVALID, "null argument");
    hipStream_t s = (hipStream_t)stream;
    const bool training = (flags & E3_FWD_TRAINING) != 0;
    // frozen BatchNorm: the training-mode data flow (raw tensors and activations are saved for a backward) with the RUNNING statistics
    // in place of batch statistics and no update of them -- what autograd does for a module in eval mode
    const bool frozen = training && (flags & E3_FWD_FROZEN_BN) != 0 && (plan->cfg.normalization == 1 || plan->cfg.attention);
    const e3_unet_cfg& cfg = plan->cfg;
    const int nb = cfg.n_blocks;
    { NetDims nd0; net_dims(plan, N, D, H, W, nd0);
      E3_REQUIRE(N > 0 && nd0.ok, E3_ERR_INVALID, "input too small for this network (conv_mode='valid' shrinks every conv by 2)"); }
    E3_REQUIRE(!training || (saved && momenta), E3_ERR_INVALID, "training forward needs `saved` and `momenta`");
    E3_REQUIRE(training || cfg.normalization != 2, E3_ERR_INVALID, "GroupNorm has no running statistics: pass E3_FWD_TRAINING also for inference");
    Buffers B;
    plan_buffers(plan, N, D, H, W, training, saved, scratch, B);
    E3_REQUIRE(!training || saved_bytes >= B.saved_bytes, E3_ERR_WORKSPACE, "`saved` buffer too small");
    E3_REQUIRE(scratch_bytes >= B.scratch_bytes, E3_ERR_WORKSPACE, "`scratch` buffer too small");
    NetDims ND; net_dims(plan, N, D, H, W, ND);
    E3_REQUIRE(ND.ok, E3_ERR_INVALID, "input too small for this network (conv_mode='valid' shrinks every conv by 2)");
    const bool valid = cfg.conv_valid != 0;
    auto P = [&](int i) { return (float*)params[i]; };

    // needed regions (e3_unet_forward_roi): the caller keeps only the voxels [roi[0..2], roi[3..5]) of y (the Predictor's central crop of an
    // overlapping tile, inference.py:496-525), so a decoder conv has to produce only what the layers behind it read for those voxels: the
    // box grows by the 3x3x3 reach per conv and halves per transposed conv on the way back through the decoder, and stops mattering where
    // it covers the tensor (the encoder is needed in full: the bottom level sees all of it).  The boxes are handed to the conv launchers;
    // kernels without the facility compute the whole tensor.  Outside the boxes the buffers keep whatever they held.
    const std::vector<NeedBox> need = need_boxes(plan, ND, (roi && !training && !valid && !cfg.attention) ? roi : nullptr);

    // split-K of a bottom-level conv (conv_wino_splitk): training forward of units with batch statistics only (their raw output and its
    // statistics come from the reduction pass; the eval / no-norm paths keep the fused epilogues)
    auto fwd_split = [&](size_t k) -> int {
        const ConvUnit& u = plan->units[k];
        if (!(training && u.has_norm()) || valid || !B.wpk_f[k] || !B.skws || u.res_in >= 0) return 1;
        const int S = conv_wino_splitk(ND.u[k].in.D, ND.u[k].in.H, ND.u[k].in.W, u.cin, u.cout);
        return S > 1 ? S : 1;
    };
    // eval-mode forwards may take the F(2x2x4) Winograd tiles (conv_wino4.hip); a training forward -- whose ReLU / arg-max decisions the gradients
    // hang on -- keeps F(2x2x2) (profiles/r05_f224_emulation.md)
    const int w4f = training ? 0 : CF_WINO4;
    // inference: the caller states that the packed / folded weights of the previous call (same plan, same scratch, same N, D, H, W, same parameter values) are
    // still in place -- the tile loop of a Predictor re-packed 22 MB of weights 726 times.  (Not with the shared `wemb` of the 1x1x1 ResizeConv variants.)
    // The claim is checked against what the last successful inference forward on this scratch buffer recorded (plan, shape, the parameter table's addresses,
    // roi / tile form): a caller that sets the flag wrongly gets a fresh pack, not stale weights.  Every call forgets the record first, so a call that fails
    // half-way leaves none behind.
    PackedSig sig; sig.plan = plan; sig.N = N; sig.D = D; sig.H = H; sig.W = W; sig.mode = (roi ? 1u : 0u) | (view ? 2u : 0u);
    sig.params = 1469598103934665603ull;
    for (size_t i = 0; i < plan->params.size(); ++i) sig.params = (sig.params ^ (uint64_t)(uintptr_t)params[i]) * 1099511628211ull;
    const bool sig_match = packed_sig_take(scratch, sig);
    const bool reuse = !training && (flags & E3_FWD_REUSE_PACKED) != 0 && cfg.up_resize < 3 && sig_match;
    auto wp_of = [&](size_t k) { return (!training && B.wpk_u[k]) ? B.wpk_u[k] : B.wpack; };
    if (!reuse)
    {   // Winograd weight transforms of every layer that uses them, in one launch
        std::vector<WinoPackJob> jobs;
        for (size_t k = 0; k < plan->units.size(); ++k) {
            if (!B.wpk_f[k]) continue;
            const ConvUnit& u = plan->units[k];
            const int S = fwd_split(k);
            if (S == 1) { jobs.push_back({P(u.p_w), B.wpk_f[k], u.cout, u.cin, 0, 0, 0, conv_wino_layout(w4f, ND.u[k].in.D, ND.u[k].in.H, ND.u[k].in.W, u.cin, u.cout, 1)}); continue; }
            for (int sp = 0; sp < S; ++sp)     // one packed weight set per share of the input channels
                jobs.push_back({P(u.p_w), B.wpk_f[k] + sp * conv_packed_floats(CONV_K3, u.cin / S, u.cout), u.cout, u.cin, 0, sp * (u.cin / S), u.cin / S});
        }
        if (!jobs.empty()) RUN(launch_wino_pack_multi(jobs.data(), (int)jobs.size(), s));
    }
    if (!reuse)
    {   // epilogue constants of every unit that has them (eval-mode BN fold; bias fold of units without a norm), in one launch
        std::vector<FoldJob> jobs;
        for (size_t k = 0; k < plan->units.size(); ++k) {
            const ConvUnit& u = plan->units[k];
            if (!u.has_norm()) jobs.push_back({nullptr, nullptr, nullptr, nullptr, P(u.p_b), B.ub[k].scale, B.ub[k].shift, u.cout});
            else if (!training) jobs.push_back({P(u.p_g), P(u.p_be), P(u.p_rm), P(u.p_rv), P(u.p_b), B.ub[k].scale, B.ub[k].shift, u.cout});
        }
        if (!jobs.empty()) RUN(launch_fold_multi(jobs.data(), (int)jobs.size(), cfg.bn_eps, s));
    }
    const float* cur = x; int cur_ldc = cfg.in_channels;
    if (cfg.in_channels > 1) { RUN(launch_ncdhw_to_ndhwc(x, B.xin, N, cfg.in_channels, ND.X[0].vox / N, s)); cur = B.xin; }

    std::vector<std::pair<const float*, int>> unit_ins(plan->units.size());      // input view of every unit (shortcut source of the ResUNet's ConvBlocks)
    int head_fused = 0;          // the last conv's epilogue took the 1x1x1 head along (ConvArgs::head_*)
    // inference: a conv1 -> conv2 chain whose two launches take the F(2x2x4) kernel hands its tensor over channel-chunked ([C / 8][voxel][8], ConvArgs::y_chunk ->
    // x_chunk): conv2 stages 8-channel chunks, and a halo row of a chunk plane is one contiguous run instead of 32 bytes out of every voxel's row
    // (profiles/r05_w4_phases.md section 5: the staging's line efficiency).  Only where conv2 is the tensor's ONLY reader.  E3_NO_CHUNKED_FWD=1: A/B switch.
    static const bool no_chunked_fwd = getenv("E3_NO_CHUNKED_FWD") != nullptr;
    size_t cur_chunk = 0;        // != 0: `cur` is channel-chunked with this many floats between chunk planes
    // ... and the concat buffer of a decoder level (B.cat[j]: the transposed conv's output | the encoder's skip activation, read by the level's first conv
    // alone) likewise: the two halves are the plane groups [0, C / 8) and [C / 8, 2 C / 8) of ONE chunked tensor of 2 C channels.  Its writers are the
    // transposed-conv GEMM kernels and the encoder conv's F(2x2x4) epilogue (plain or with the fused pool, whose pooled output stays in rows).
    static const bool no_pool_fuse = getenv("E3_WINO_NO_POOL") != nullptr;      // (the separate pool pass reads rows)
    auto cat_chunked = [&](int j) -> size_t {
        if (no_chunked_fwd || no_pool_fuse || training || valid || cfg.attention || cfg.merge_add || cfg.up_resize || cfg.act_slope != 0.f || j >= nb - 1 || !B.cat[j]) return 0;
        const size_t ke = (size_t)plan->enc_last_unit[j];
        size_t ku = 0;
        for (; ku < plan->units.size(); ++ku) if (plan->units[ku].is_up && plan->units[ku].level == j) break;
        if (ku + 1 >= plan->units.size()) return 0;
        const ConvUnit& e = plan->units[ke]; const ConvUnit& up = plan->units[ku]; const ConvUnit& c1 = plan->units[ku + 1];
        const int C = up.cout;
        if (up.is_up != 1 || up.planar || e.planar || c1.planar || e.cout != C || c1.cin != 2 * C || (C & 31) || e.cin < 8 || e.res_in >= 0 || c1.res_in >= 0 || c1.is_up) return 0;
        if (!B.wpk_f[ke] || !B.wpk_f[ku + 1] || B.ub[ke].act != B.cat[j] + C || B.ub[ku].act != B.cat[j] || need[ke].on) return 0;
        for (size_t q = 0; q < plan->units.size(); ++q)
            if (plan->units[q].res_in == (int)ku + 1 || plan->units[q].res_in == (int)ke + 1) return 0;           // (a ResUNet shortcut reads a half as well)
        const LevelDims& ei = ND.u[ke].in; const LevelDims& eo = ND.u[ke].out; const LevelDims& ci1 = ND.u[ku + 1].in; const LevelDims& uo = ND.u[ku].out;
        if (eo.D != uo.D || eo.H != uo.H || eo.W != uo.W || ci1.D != uo.D || ci1.H != uo.H || ci1.W != uo.W || eo.vox != uo.vox) return 0;
        if (!chunked_layout_ok(uo.vox, 2 * C) || !upconv_gemm_ok(CF_SCATTER_UP, up.cin, C, 4 * (up.planar ? 1 : 2) * C)) return 0;
        if (conv_wino_layout(w4f, ei.D, ei.H, ei.W, e.cin, e.cout, 1) != 2 || conv_wino_layout(w4f, ci1.D, ci1.H, ci1.W, c1.cin, c1.cout, 1) != 2) return 0;
        return uo.vox * 8;
    };
    // unit k + 1 is a plain 3x3x3 conv on the F(2x2x4) kernel that reads a packed C-channel tensor on the grid `g` -- and nobody else reads that tensor
    auto next_reads_chunks = [&](size_t k, int C, const LevelDims& g) -> bool {
        if (no_chunked_fwd || training || valid || k + 1 >= plan->units.size()) return false;
        const ConvUnit& v = plan->units[k + 1];
        if (v.is_up || v.planar || v.cin != C || v.res_in >= 0 || !B.wpk_f[k + 1]) return false;
        for (size_t j = 0; j < plan->units.size(); ++j)
            if (plan->units[j].res_in == (int)k + 1) return false;           // (a ResUNet shortcut reads the tensor as well)
        const LevelDims& a1 = ND.u[k + 1].in;
        return chunked_layout_ok(g.vox, C) && conv_wino_layout(w4f, a1.D, a1.H, a1.W, v.cin, v.cout, 1) == 2 && a1.D == g.D && a1.H == g.H && a1.W == g.W && a1.vox == g.vox;
    };
    auto chain_chunked = [&](size_t k, bool plain_epilogue) -> bool {
        if (!plain_epilogue) return false;
        const ConvUnit& u = plan->units[k];
        if (u.is_up || u.planar || u.cin < 8 || u.enc_last || u.res_in >= 0 || !B.wpk_f[k] || B.ub[k].act_ldc != u.cout) return false;
        const LevelDims& a0 = ND.u[k].in;
        return conv_wino_layout(w4f, a0.D, a0.H, a0.W, u.cin, u.cout, 1) == 2 && next_reads_chunks(k, u.cout, ND.u[k].out);
    };
    for (size_t k = 0; k < plan->units.size(); ++k) {
        const ConvUnit& u = plan->units[k];
        UnitBufs& b = B.ub[k];
        const LevelDims& lo = ND.u[k].out;         // the unit's output tensor
        const LevelDims& ci = ND.u[k].in;          // plain convs: the grid the conv kernel runs on (== lo unless conv_mode='valid')
        const bool vcrop = valid && !u.is_up;      // 'valid' conv = the 'same' conv on the input grid, cropped by the padding
        const float* const unit_in = cur; const int unit_in_ldc = cur_ldc;      // (the gating signal of the block's GridAttention)
        E3_REQUIRE(!cur_chunk || (!u.is_up && u.cin >= 8 && !u.planar && B.wpk_f[k]), E3_ERR_INVALID, "channel-chunked tensor handed to a unit that reads rows");
        unit_ins[k] = {cur, cur_ldc};
        const bool is_enc_conv2 = u.enc_last;
        const bool pool_after = is_enc_conv2 && u.level < nb - 1;
        int pool_fused = 0;          // the conv's epilogue took the max-pool along (ConvArgs::pool_out)
        size_t pool_chunk_out = 0;   // ... and wrote the pooled tensor channel-chunked
        const int kd = u.planar ? 1 : 2;
        const float slope = cfg.act_slope;
        ActArg act = u.p_a >= 0 ? ActArg(0.f, P(u.p_a)) : ActArg(slope);
        if (training && !frozen) act = plan->rrelu_of(act, (int)k);      // train-mode RReLU: random slopes (the backward recomputes them)
        const bool bn_train = training && u.has_norm();   // batch statistics needed: conv writes the raw output, BN+ReLU is a second pass
        float* const stat_buf = frozen ? nullptr : B.stats;   // (frozen statistics: nothing to measure)
        const bool residual = u.res_in >= 0;        // y = conv2(..) + proj(inp) before the norm (resunet.py:254-262)
        const bool two_pass = bn_train || slope != 0.f || u.is_up == 2 || vcrop || residual;   // (non-ReLU activations are not in the conv epilogues; the
                                                                           // ResizeConv output may need the autocrop before the norm)
        float* dst = two_pass ? b.raw : b.act;            // otherwise the conv writes the activation directly
        const int dst_ldc = two_pass ? u.cout : b.act_ldc;
        const float* es = nullptr; const float* eh = nullptr;
        if ((!u.has_norm() || !training) && !two_pass) {   // nn.Identity: y = relu(acc + bias) always; eval-mode BN folded into the conv
            es = b.scale; eh = b.shift;                    // epilogue (running stats) -- constants from the launch above
        }
        int parts = 0;
        if (u.is_up == 2 && cfg.up_resize >= 3) {
            // ResizeConv(kernel_size=1): a 1x1x1 conv commutes with the up-sampling (nearest copies voxels; the linear weights sum to 1, so
            // the bias passes through), so it runs at LOW resolution -- as the centre tap of a 1x3x3 kernel on the planar conv kernels
            // (9 instead of 8*27 multiplies per low-resolution voxel) -- and its Cout-channel result is up-sampled, cropped, measured.
            const LevelDims& li = ND.u[k].in;
            const int sd = u.planar ? 1 : 2, Ud = li.D * sd, Uh = li.H * 2, Uw = li.W * 2, NPad = pad_cols(u.cout);
            const bool lin = cfg.up_resize == 4;
            RUN(launch_embed_center_tap(P(u.p_w), B.wemb, (size_t)u.cout * u.cin, 9, s));
            RUN(launch_pack_conv_auto(CONV_K3_PLANAR, 0, B.wemb, wp_of(k), u.cout, u.cin, N, li.D, li.H, li.W, 0, s));
            ConvArgs a{};
            a.x = cur; a.x_ldc = cur_ldc; a.Cin = u.cin; a.wt = wp_of(k); a.bias = bn_train ? P(u.p_b) : nullptr;
            a.y = B.rdu; a.y_ldc = u.cout; a.N = N; a.D = li.D; a.H = li.H; a.W = li.W; a.sd = 2;
            a.Cout = u.cout; a.Ncols = u.cout; a.NPad = NPad; a.G = 1; a.flags = 0; a.stats = nullptr;
            { Prof pr(plan, s, (int)k, 0); RUN(launch_conv_mfma(CONV_K3_PLANAR, a, s)); }
            RUN(launch_upsample_nearest(B.rdu, u.cout, B.rtmp, u.cout, N, li.D, li.H, li.W, sd, s, lin));
            RUN(launch_crop_stats(B.rtmp, b.raw, u.cout, N, Ud, Uh, Uw, lo.D, lo.H, lo.W, B.stats, s));
            parts = crop_stats_parts(lo.vox, u.cout);
        } else if (u.is_up == 2) {      // ResizeConv (unet.py:411-449): nn.Upsample(nearest) then conv3 on the up-sampled grid, autocrop afterwards
            const LevelDims& li = ND.u[k].in;
            const int sd = u.planar ? 1 : 2, Ud = li.D * sd, Uh = li.H * 2, Uw = li.W * 2, NPad = pad_cols(u.cout);
            const ConvKind kind = u.planar ? CONV_K3_PLANAR : CONV_K3;
            const bool same = Ud == lo.D && Uh == lo.H && Uw == lo.W;
            RUN(launch_upsample_nearest(cur, cur_ldc, B.ups[k], u.cin, N, li.D, li.H, li.W, sd, s, cfg.up_resize == 2));
            if (!reuse) RUN(launch_pack_conv_auto(kind, 0, P(u.p_w), wp_of(k), u.cout, u.cin, N, Ud, Uh, Uw, 0, s));
            ConvArgs a{};
            a.x = B.ups[k]; a.x_ldc = u.cin; a.Cin = u.cin; a.wt = wp_of(k); a.bias = bn_train ? P(u.p_b) : nullptr;
            a.y = same ? b.raw : B.rtmp; a.y_ldc = u.cout; a.N = N; a.D = Ud; a.H = Uh; a.W = Uw; a.sd = 2;
            a.Cout = u.cout; a.Ncols = u.cout; a.NPad = NPad; a.G = 1; a.flags = 0;
            a.stats = (bn_train && same) ? stat_buf : nullptr;
            parts = conv_stats_parts(kind, 0, N, Ud, Uh, Uw, 2, u.cin, u.cout);
            { Prof pr(plan, s, (int)k, 0); RUN(launch_conv_mfma(kind, a, s)); }
            if (!same) {             // crop one voxel at the high end where the skip has an odd size (unet.py:289-299) + statistics
                RUN(launch_crop_stats(B.rtmp, b.raw, u.cout, N, Ud, Uh, Uw, lo.D, lo.H, lo.W, B.stats, s));
                parts = crop_stats_parts(lo.vox, u.cout);
            }
        } else if (u.is_up) {
            const LevelDims& li = ND.u[k].in;
            const int sd = u.planar ? 1 : 2, taps = sd * 4, NPad = pad_cols(taps * u.cout);
            if (!reuse) RUN(launch_pack_weights(PACK_UP_FWD, P(u.p_w), wp_of(k), u.cout, u.cin, taps, NPad, s));
            ConvArgs a{};
            a.x = cur; a.x_ldc = cur_ldc; a.Cin = u.cin; a.wt = wp_of(k); a.bias = bn_train ? P(u.p_b) : nullptr;
            a.y = dst; a.y_ldc = dst_ldc; a.N = N; a.D = li.D; a.H = li.H; a.W = li.W;
            a.Do = lo.D; a.Ho = lo.H; a.Wo = lo.W; a.sd = sd;   // autocrop of the up-convolved tensor (unet.py:289-299)
            a.Cout = u.cout; a.Ncols = taps * u.cout; a.NPad = NPad; a.epi_scale = es; a.epi_shift = eh;
            a.stats = bn_train ? stat_buf : nullptr; a.G = 1; a.flags = CF_SCATTER_UP;
            const size_t upck = cat_chunked(u.level);
            E3_REQUIRE(!upck || (!two_pass && es && !bn_train), E3_ERR_INVALID, "channel-chunked concat buffer: the transposed conv cannot write its half");
            if (upck) { a.y_chunk = upck; a.y_ldc = 8; }
            if (need[k].on && N == 1 && !bn_train && !two_pass) {
                // needed region along D only (a transposed conv with kernel = stride has no halo, so a range of input planes is simply a smaller
                // tensor: pointers and depths move, the kernels do not change); one sample, because the sample stride is implied by the dims
                const int p0 = need[k].lo[0] / sd, p1 = (need[k].hi[0] + sd - 1) / sd < li.D ? (need[k].hi[0] + sd - 1) / sd : li.D;
                if (p1 > p0 && (p0 > 0 || p1 < li.D)) {
                    const int o0 = p0 * sd, o1 = p1 * sd < lo.D ? p1 * sd : lo.D;
                    a.x = cur + (size_t)p0 * li.H * li.W * cur_ldc; a.D = p1 - p0;
                    a.y = dst + (size_t)o0 * lo.H * lo.W * (upck ? 8 : dst_ldc); a.Do = o1 - o0;
                }
                // ... and a box of input rows / columns for the kernels that can walk one (ConvArgs::box_* in INPUT voxels; E3_NO_STORE_BOX=1: whole planes)
                static const bool no_up_box = getenv("E3_NO_STORE_BOX") != nullptr;
                if (!no_up_box) {
                    a.box_lo[0] = 0; a.box_hi[0] = a.D;
                    a.box_lo[1] = need[k].lo[1] / 2; a.box_hi[1] = (need[k].hi[1] + 1) / 2 < li.H ? (need[k].hi[1] + 1) / 2 : li.H;
                    a.box_lo[2] = need[k].lo[2] / 2; a.box_hi[2] = (need[k].hi[2] + 1) / 2 < li.W ? (need[k].hi[2] + 1) / 2 : li.W;
                }
            }
            parts = conv_stats_parts(CONV_POINT, CF_SCATTER_UP, N, li.D, li.H, li.W, sd, u.cin, taps * u.cout);
            { Prof pr(plan, s, (int)k, 0); RUN(launch_conv_mfma(CONV_POINT, a, s)); }
        } else if (u.cin < 8) {
            ConvSmallArgs a{};
            a.x = cur; a.Cin = u.cin; a.w = P(u.p_w); a.bias = bn_train ? P(u.p_b) : nullptr; a.y = vcrop ? B.rtmp : dst; a.y_ldc = dst_ldc;
            a.N = N; a.D = ci.D; a.H = ci.H; a.W = ci.W; a.Cout = u.cout; a.planar = u.planar;
            a.epi_scale = es; a.epi_shift = eh; a.stats = (bn_train && !vcrop) ? stat_buf : nullptr;
            if (view && k == 0) { a.xs_n = view->x_stride[0]; a.xs_d = view->x_stride[1]; a.xs_h = view->x_stride[2]; }      // (the tile is read in place)
            if (es && !two_pass && !vcrop && !u.enc_last && dst_ldc == u.cout && conv_first_chunk_ok(N, ci.D, ci.H, ci.W, u.planar, u.cin, u.cout) &&
                next_reads_chunks(k, u.cout, lo)) { a.y_chunk = lo.vox * 8; cur_chunk = a.y_chunk; }
            parts = conv_small_stats_parts2(N, ci.D, ci.H, ci.W, u.planar, u.cin, u.cout);
            { Prof pr(plan, s, (int)k, 0); RUN(launch_conv_small_fwd(a, s)); }
        } else {
            const int taps = u.planar ? 9 : 27, NPad = pad_cols(u.cout);
            const ConvKind kind = u.planar ? CONV_K3_PLANAR : CONV_K3;
            (void)taps;
            if (!B.wpk_f[k] && !reuse) RUN(launch_pack_conv_auto(kind, 0, P(u.p_w), wp_of(k), u.cout, u.cin, N, ci.D, ci.H, ci.W, w4f, s));
            ConvArgs a{};
            a.x = cur; a.x_ldc = cur_ldc; a.Cin = u.cin; a.wt = B.wpk_f[k] ? B.wpk_f[k] : wp_of(k); a.bias = bn_train ? P(u.p_b) : nullptr;
            a.y = vcrop ? B.rtmp : dst; a.y_ldc = dst_ldc; a.N = N; a.D = ci.D; a.H = ci.H; a.W = ci.W; a.sd = 2;
            a.Cout = u.cout; a.Ncols = u.cout; a.NPad = NPad; a.epi_scale = es; a.epi_shift = eh;
            a.stats = (bn_train && !vcrop) ? stat_buf : nullptr; a.G = 1; a.flags = w4f;
            if (residual) { a.y = B.res2; a.y_ldc = u.cout; a.bias = nullptr; a.stats = nullptr; }      // pure accumulations; bias + shortcut + statistics below
            parts = conv_stats_parts(kind, w4f, N, ci.D, ci.H, ci.W, 2, u.cin, u.cout);
            const int S = (kind == CONV_K3) ? fwd_split(k) : 1;
            a.x_chunk = cur_chunk; cur_chunk = 0;
            if (kind == CONV_K3 && S == 1 && es && !two_pass && !vcrop && !residual && !pool_after && chain_chunked(k, true)) { a.y_chunk = lo.vox * 8; cur_chunk = a.y_chunk; }
            // inference: the ceil-mode max-pool behind an encoder block rides in the conv's epilogue where the kernel can take it (a Winograd tile is a window)
            if (pool_after && !training && !two_pass && kd == 2 && kind == CONV_K3 && es && !vcrop && !residual) { a.pool_out = B.pooled[u.level]; a.pool_done = &pool_fused; }
            // inference: the 1x1x1 head (+ softmax) rides in the epilogue of the network's LAST conv where the kernel can take it (ConvArgs::head_*): the
            // last activation tensor is neither written nor read
            if (k + 1 == plan->units.size() && !training && !la && !two_pass && kind == CONV_K3 && es && !vcrop && !residual && S == 1 && u.cout == 32 &&
                plan->chan(0) == 32 && cfg.out_channels <= 4 && (!roi || view || (roi[3] <= ND.Y.D && roi[4] <= ND.Y.H && roi[5] <= ND.Y.W))) {
                a.head_w = P(plan->p_final_w); a.head_b = P(plan->p_final_b); a.head_cout = cfg.out_channels; a.head_softmax = (flags & E3_FWD_SOFTMAX) ? 1 : 0;
                a.head_done = &head_fused;
                const long long S1 = (long long)(ND.Y.vox / N);
                if (view) {           // the kept region, straight into its place in the output volume
                    a.head_y = y; for (int i = 0; i < 4; ++i) a.head_ys[i] = view->y_stride[i];
                    for (int i = 0; i < 3; ++i) { a.head_lo[i] = roi[i]; a.head_hi[i] = roi[3 + i]; }
                } else {
                    a.head_ys[0] = (long long)cfg.out_channels * S1; a.head_ys[1] = S1; a.head_ys[2] = (long long)ND.Y.H * ND.Y.W; a.head_ys[3] = ND.Y.W;
                    if (roi) {        // e3_unet_forward_roi: the kept region only, in the tile's own layout
                        for (int i = 0; i < 3; ++i) { a.head_lo[i] = roi[i]; a.head_hi[i] = roi[3 + i]; }
                        a.head_y = y + (long long)roi[0] * a.head_ys[2] + (long long)roi[1] * a.head_ys[3] + roi[2];
                    } else {
                        a.head_y = y; a.head_lo[0] = a.head_lo[1] = a.head_lo[2] = 0; a.head_hi[0] = ND.Y.D; a.head_hi[1] = ND.Y.H; a.head_hi[2] = ND.Y.W;
                    }
                }
            }
            if (need[k].on && kind == CONV_K3 && S == 1 && !a.stats)
                for (int i = 0; i < 3; ++i) { a.box_lo[i] = need[k].lo[i]; a.box_hi[i] = need[k].hi[i]; }
            if (S > 1) {         // partial sums per share of the input channels, then sum + bias + statistics in one small pass
                a.splitk = S; a.sk_x = u.cin / S; a.Cin = u.cin / S; a.sk_w = (unsigned)conv_packed_floats(CONV_K3, u.cin / S, u.cout);
                a.sk_y = lo.vox * u.cout; a.y = B.skws; a.y_ldc = u.cout; a.bias = nullptr; a.stats = nullptr;
            }
            // the encoder's skip activation: second half of the level's concat buffer = its plane groups [C / 8, 2 C / 8).  (Only with the pool in
            // the epilogue: the separate pool pass reads rows.)
            const size_t skck = (is_enc_conv2 && pool_after) ? cat_chunked(u.level) : 0;
            E3_REQUIRE(!skck || (kind == CONV_K3 && S == 1 && es && !two_pass && !vcrop && !residual && dst == B.cat[u.level] + u.cout && a.box_hi[0] <= 0 && a.pool_out),
                       E3_ERR_INVALID, "channel-chunked concat buffer: the encoder conv cannot write its half");
            if (skck) { a.y = B.cat[u.level] + (size_t)u.cout * lo.vox; a.y_chunk = skck; }
            // needed region: of the skip activation the decoder reads only what the level's transposed conv has to produce (need[] of the up unit = the box of the concat
            // buffer's voxels that the block's first conv reads) -- the rest is computed for the pool but not stored (E3_NO_STORE_BOX=1: A/B switch)
            static const bool no_store_box = getenv("E3_NO_STORE_BOX") != nullptr;
            if (!no_store_box && is_enc_conv2 && pool_after && a.pool_out && !no_pool_fuse && a.box_hi[0] <= 0 && (u.cout & 31) == 0 && es && S == 1 && B.wpk_f[k] &&
                conv_wino_layout(w4f, ci.D, ci.H, ci.W, u.cin, u.cout, 1) == 2 && !training && !vcrop && !residual && !cfg.attention && !cfg.merge_add && dst == B.cat[u.level] + u.cout) {
                size_t ku = 0;
                for (; ku < plan->units.size(); ++ku) if (plan->units[ku].is_up && plan->units[ku].level == u.level) break;
                if (ku < plan->units.size() && need[ku].on && ND.u[ku].out.D == lo.D && ND.u[ku].out.H == lo.H && ND.u[ku].out.W == lo.W)
                    for (int i = 0; i < 3; ++i) { a.sbox_lo[i] = need[ku].lo[i]; a.sbox_hi[i] = need[ku].hi[i]; }
            }
            // ... and the pooled tensor goes to the next level's first conv chunked as well (only out of the conv epilogue: the separate pool pass writes rows)
            size_t plck = 0;
            if (a.pool_out && !no_pool_fuse && a.box_hi[0] <= 0 && (u.cout & 31) == 0 && B.wpk_f[k] && S == 1 && conv_wino_layout(w4f, ci.D, ci.H, ci.W, u.cin, u.cout, 1) == 2 && es) {
                LevelDims pg = lo; pg.D = (lo.D + 1) / 2; pg.H = (lo.H + 1) / 2; pg.W = (lo.W + 1) / 2; pg.vox = (size_t)N * pg.D * pg.H * pg.W;
                if (next_reads_chunks(k, u.cout, pg)) { plck = pg.vox * 8; a.pool_chunk = plck; }
            }
            { Prof pr(plan, s, (int)k, 0); RUN(launch_conv_mfma(kind, a, s)); }
            E3_REQUIRE((!skck && !plck && a.sbox_hi[0] <= 0) || !pool_after || pool_fused, E3_ERR_INVALID, "channel-chunked / partly stored skip tensor without the pool in the conv's epilogue");
            pool_chunk_out = plck;
            if (S > 1) {
                RUN(launch_splitk_reduce(B.skws, S, lo.vox * u.cout, P(u.p_b), dst, dst_ldc, u.cout, lo.vox, stat_buf, s));
                parts = crop_stats_parts(lo.vox, u.cout);
            }
        }
        if (residual) {
            E3_REQUIRE(!u.is_up && u.cin >= 8 && !vcrop, E3_ERR_INVALID, "residual unit: plain 'same' conv expected");
            const std::pair<const float*, int>& in = unit_ins[(size_t)u.res_in];
            const int cin0 = plan->units[(size_t)u.res_in].cin;
            float* const slot1 = B.res2 + lo.vox * u.cout;
            if (u.p_pw >= 0) RUN(launch_pw_fwd(in.first, in.second, cin0, P(u.p_pw), P(u.p_pb), slot1, u.cout, u.cout, lo.vox, s));
            else {
                E3_REQUIRE(in.second == u.cout && cin0 == u.cout, E3_ERR_INVALID, "identity shortcut: packed input with the block's channel count expected");
                E3_CHECK_HIP(hipMemcpyAsync(slot1, in.first, lo.vox * u.cout * sizeof(float), hipMemcpyDeviceToDevice, s));
            }
            RUN(launch_splitk_reduce(B.res2, 2, lo.vox * u.cout, bn_train ? P(u.p_b) : nullptr, b.raw, u.cout, u.cout, lo.vox,
                                     (bn_train && !frozen) ? B.stats : nullptr, s));
            parts = crop_stats_parts(lo.vox, u.cout);
        }
        if (vcrop) {        // the interior of the 'same' result is the 'valid' result (no tap of an interior voxel touches the padding)
            RUN(launch_crop_stats(B.rtmp, b.raw, u.cout, N, ci.D, ci.H, ci.W, lo.D, lo.H, lo.W, B.stats, s, ND.u[k].od, ND.u[k].oh, ND.u[k].ow));
            parts = crop_stats_parts(lo.vox, u.cout);
        }
        if (bn_train && frozen) {
            RUN(launch_bn_frozen(P(u.p_g), P(u.p_be), P(u.p_rm), P(u.p_rv), cfg.bn_eps, b.mean, b.invstd, b.scale, b.shift, u.cout, s));
            if (k + 1 < plan->units.size())
                RUN(launch_bn_relu_apply(b.raw, u.cout, b.scale, b.shift, b.act, b.act_ldc, pool_after ? B.pooled[u.level] : nullptr, kd,
                                         N, lo.D, lo.H, lo.W, u.cout, s, act));
        } else if (bn_train) {
            BnFinalizeArgs f{};
            f.stats = B.stats; f.parts = parts; f.C = u.cout; f.gamma = P(u.p_g); f.beta = P(u.p_be);
            const bool group = cfg.normalization == 2;
            if (group) E3_REQUIRE(N == 1, E3_ERR_INVALID, "GroupNorm statistics are per sample: call with N = 1");
            f.group = group ? u.cout / cfg.num_groups : 1;
            f.running_mean = group ? nullptr : P(u.p_rm); f.running_var = group ? nullptr : P(u.p_rv);
            f.momentum = momenta[u.bn_index]; f.eps = cfg.bn_eps;
            f.mean = b.mean; f.invstd = b.invstd; f.scale = b.scale; f.shift = b.shift; f.scratch = B.bnred;
            RUN(launch_bn_finalize(f, s));
            // the last activation of the network only feeds the 1x1x1 head, which applies BN + ReLU while loading the raw tensor
            // (forward and backward): no apply pass, no activation tensor
            if (k + 1 < plan->units.size())
                RUN(launch_bn_relu_apply(b.raw, u.cout, b.scale, b.shift, b.act, b.act_ldc, pool_after ? B.pooled[u.level] : nullptr, kd,
                                         N, lo.D, lo.H, lo.W, u.cout, s, act));
        } else if (two_pass) {      // raw = pure accumulations; (scale, shift) = folded eval-mode BN, or (1, conv bias) without a norm
            RUN(launch_bn_relu_apply(b.raw, u.cout, b.scale, b.shift, b.act, b.act_ldc, pool_after ? B.pooled[u.level] : nullptr, kd,
                                     N, lo.D, lo.H, lo.W, u.cout, s, act));
        } else if (pool_after && !pool_fused) {
            RUN(launch_maxpool(b.act, b.act_ldc, B.pooled[u.level], kd, N, lo.D, lo.H, lo.W, u.cout, s));
        }
        if (u.is_up && cfg.attention) {
            // genc = GridAttention(enc, dec) (unet.py:393): the (cropped) skip, gated by the block's INPUT, goes through W + BatchNorm into the
            // second half of the concat buffer
            const int j = u.level, C = u.cout;
            const LevelDims& e = ND.E[j];
            const UnitBufs& eb = B.ub[(size_t)plan->enc_last_unit[j]];
            const AttUnit& au = plan->att[j];
            Buffers::AttBufs& ab = B.att[j];
            const AttDims ad = att_dims(plan, ND, N, j, k);
            E3_REQUIRE(ad.d >= 1 && ad.h >= 1 && ad.w >= 1, E3_ERR_INVALID, "attention: a skip tensor is smaller than theta's 2x2x2 kernel");
            const AttParams ap = att_params(params, au);
            const float* xs = eb.act; int ldx = eb.act_ldc;
            if (valid) {
                RUN(launch_crop_copy(eb.act, ab.x, C, C, N, e.D, e.H, e.W, lo.D, lo.H, lo.W, ND.sd_[j], ND.sh_[j], ND.sw_[j], s));
                xs = ab.x; ldx = C;
            }
            RUN(launch_att_gate_fwd(ad, xs, ldx, unit_in, unit_in_ldc, ap, ab.f, ab.sgm, ab.att, B.att_phi, B.att_phires, s));
            float* const half = B.cat[j] + C;
            if (!training) {      // eval: the BatchNorm (running statistics) folds into W's epilogue
                RUN(launch_bn_fold(P(au.p_g), P(au.p_be), P(au.p_rm), P(au.p_rv), P(au.p_wb), cfg.bn_eps, ab.scale, ab.shift, C, s));
                RUN(launch_att_out_fwd(ad, xs, ldx, ab.att, ap, ab.scale, ab.shift, half, 2 * C, s));
            } else {
                if (frozen) {
                    RUN(launch_att_out_fwd(ad, xs, ldx, ab.att, ap, nullptr, nullptr, ab.raw, C, s));
                    RUN(launch_bn_frozen(P(au.p_g), P(au.p_be), P(au.p_rm), P(au.p_rv), cfg.bn_eps, ab.mean, ab.invstd, ab.scale, ab.shift, C, s));
                } else {
                    RUN(launch_att_out_fwd(ad, xs, ldx, ab.att, ap, nullptr, nullptr, B.att_tmp, C, s));
                    RUN(launch_splitk_reduce(B.att_tmp, 1, 0, nullptr, ab.raw, C, C, lo.vox, B.stats, s));      // copy + batch statistics
                    BnFinalizeArgs f{};
                    f.stats = B.stats; f.parts = crop_stats_parts(lo.vox, C); f.C = C; f.gamma = P(au.p_g); f.beta = P(au.p_be); f.group = 1;
                    f.running_mean = P(au.p_rm); f.running_var = P(au.p_rv); f.momentum = momenta[au.bn_index]; f.eps = cfg.bn_eps;
                    f.mean = ab.mean; f.invstd = ab.invstd; f.scale = ab.scale; f.shift = ab.shift; f.scratch = B.bnred;
                    RUN(launch_bn_finalize(f, s));
                }
                RUN(launch_bn_relu_apply(ab.raw, C, ab.scale, ab.shift, half, 2 * C, nullptr, 2, N, lo.D, lo.H, lo.W, C, s, ActArg(1.f)));   // (no activation)
            }
        } else if (u.is_up && valid) {      // centre-crop the encoder's skip activation into the second half of the concat buffer (unet.py:300-325)
            const int j = u.level;
            const LevelDims& e = ND.E[j];
            const UnitBufs& eb = B.ub[(size_t)plan->enc_last_unit[j]];
            RUN(launch_crop_copy(eb.act, B.cat[j] + u.cout, 2 * u.cout, u.cout, N, e.D, e.H, e.W, lo.D, lo.H, lo.W, ND.sd_[j], ND.sh_[j], ND.sw_[j], s));
        }
        // input of the next unit
        if (pool_after) { cur = B.pooled[u.level]; cur_ldc = u.cout; cur_chunk = pool_chunk_out; }
        else if (u.is_up && cfg.merge_add) {   // mrg = updec + genc (unet.py:400-401): the two halves of the buffer summed
            RUN(launch_add_views(B.cat[u.level], 2 * u.cout, B.cat[u.level] + u.cout, 2 * u.cout, B.sum[u.level], u.cout, lo.vox, u.cout, s));
            cur = B.sum[u.level]; cur_ldc = u.cout;
        }
        else if (u.is_up) { cur = B.cat[u.level]; cur_ldc = 2 * u.cout; cur_chunk = cat_chunked(u.level); }   // conv1 of the UpConv reads the whole concat buffer
        else { cur = b.act; cur_ldc = b.act_ldc; }
    }
    if (!head_fused) {
        const ConvUnit& lu = plan->units.back();
        const UnitBufs& lb = B.ub.back();
        const bool fused = training && lu.has_norm();     // see above: head reads the raw conv output + (scale, shift)
        Prof pr(plan, s, (int)plan->units.size(), 0);
        const ActArg head_act = (training && !frozen) ? plan->rrelu_of(lu.p_a >= 0 ? ActArg(0.f, P(lu.p_a)) : ActArg(cfg.act_slope), (int)plan->units.size() - 1)
                                                      : (lu.p_a >= 0 ? ActArg(0.f, P(lu.p_a)) : ActArg(cfg.act_slope));
        if (la) {       // the criterion's sums are taken from the logits while the head has them in registers (e3_unet_forward_loss)
            int rows = 0;
            RUN(launch_conv_final_fwd_loss(fused ? lb.raw : cur, fused ? lu.cout : cur_ldc, plan->chan(0), P(plan->p_final_w), P(plan->p_final_b), y,
                                           cfg.out_channels, ND.Y.vox / N, N, s, fused ? lb.scale : nullptr, fused ? lb.shift : nullptr, head_act,
                                           la->target, la->class_weight, (float*)la->workspace, CE_DICE_MAX_ROWS, &rows));
            if (la->sums_out)       // sharded minibatch: the caller sums these over the ranks and finishes with e3_ce_dice_from_sums
                RUN(launch_ce_dice_sums_rows(cfg.out_channels, rows, (const float*)la->workspace, la->sums_out, s));
            else
                RUN(launch_ce_dice_finalize(la->class_weight, cfg.out_channels, rows, la->ce_weight, la->dice_weight, la->eps, la->smooth,
                                            (float*)la->workspace, la->loss_out, s));
        } else if (view) {      // only the kept region, straight into its place in the output volume
            const int lo[3] = {roi[0], roi[1], roi[2]}, size[3] = {roi[3] - roi[0], roi[4] - roi[1], roi[5] - roi[2]};
            RUN(launch_conv_final_fwd_box(cur, cur_ldc, plan->chan(0), P(plan->p_final_w), P(plan->p_final_b), y, cfg.out_channels, N, ND.Y.D, ND.Y.H, ND.Y.W,
                                          lo, size, view->y_stride, (flags & E3_FWD_SOFTMAX) ? 1 : 0, s, nullptr, nullptr, head_act));
        } else if (roi && !training && !fused && roi[3] <= ND.Y.D && roi[4] <= ND.Y.H && roi[5] <= ND.Y.W) {
            // e3_unet_forward_roi: the head reads and writes the kept region only (y keeps the tile's own layout)
            const int lo[3] = {roi[0], roi[1], roi[2]}, size[3] = {roi[3] - roi[0], roi[4] - roi[1], roi[5] - roi[2]};
            const long long S1 = (long long)(ND.Y.vox / N), ys[4] = {(long long)cfg.out_channels * S1, S1, (long long)ND.Y.H * ND.Y.W, (long long)ND.Y.W};
            RUN(launch_conv_final_fwd_box(cur, cur_ldc, plan->chan(0), P(plan->p_final_w), P(plan->p_final_b),
                                          y + (long long)lo[0] * ys[2] + (long long)lo[1] * ys[3] + lo[2], cfg.out_channels, N, ND.Y.D, ND.Y.H, ND.Y.W,
                                          lo, size, ys, (flags & E3_FWD_SOFTMAX) ? 1 : 0, s, nullptr, nullptr, head_act));
        } else {
            RUN(launch_conv_final_fwd(fused ? lb.raw : cur, fused ? lu.cout : cur_ldc, plan->chan(0), P(plan->p_final_w), P(plan->p_final_b), y,
                                      cfg.out_channels, ND.Y.vox / N, N, (flags & E3_FWD_SOFTMAX) ? 1 : 0, s,
                                      fused ? lb.scale : nullptr, fused ? lb.shift : nullptr, head_act));
        }
    }
    if (!training && cfg.up_resize < 3) packed_sig_record(scratch, sig);
    return E3_OK;
}

int e3_unet_backward(e3_unet_plan* plan, void* stream, const float* dy, const float* x, int N, int D, int H, int W,
                     void* const* params, void* const* grads, float* dx,
                     void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                     void* bucket_event, int bucket_after_down_block) {
    return e3_unet_backward2(plan, stream, dy, x, N, D, H, W, params, grads, dx, saved, saved_bytes, scratch, scratch_bytes, bucket_event,
                             bucket_after_down_block, 0);
}

// the criterion's request of e3_unet_backward_loss: dLoss/dlogits is formed inside the head's backward kernels
struct HeadLossReq { const float* logits; const long long* target; const float* cw; const float* coef; const float* gout; };
static int backward_impl(e3_unet_plan* plan, void* stream, const float* dy, const HeadLossReq* hl, const float* x, int N, int D, int H, int W,
                         void* const* params, void* const* grads, float* dx,
                         void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                         void* bucket_event, int bucket_after_down_block, uint32_t flags);

int e3_unet_backward2(e3_unet_plan* plan, void* stream, const float* dy, const float* x, int N, int D, int H, int W,
                      void* const* params, void* const* grads, float* dx,
                      void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                      void* bucket_event, int bucket_after_down_block, uint32_t flags) {
    E3_REQUIRE(dy, E3_ERR_INVALID, "null argument");
    return backward_impl(plan, stream, dy, nullptr, x, N, D, H, W, params, grads, dx, saved, saved_bytes, scratch, scratch_bytes, bucket_event,
                         bucket_after_down_block, flags);
}

int e3_unet_backward_loss(e3_unet_plan* plan, void* stream, const float* y, const e3_ce_dice_args* loss, const float* gout, const float* x,
                          int N, int D, int H, int W, void* const* params, void* const* grads, float* dx,
                          void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                          void* bucket_event, int bucket_after_down_block, uint32_t flags) {
    E3_REQUIRE(plan && y && loss && loss->target && loss->workspace, E3_ERR_INVALID, "null argument");
    const int C = plan->cfg.out_channels;
    E3_REQUIRE(loss->workspace_bytes >= e3_ce_dice_workspace_bytes(C), E3_ERR_WORKSPACE, "ce_dice workspace too small");
    // the fused form lives in the BatchNorm backward of the network's last unit (head form: at most 4 classes in registers)
    // (BatchNorm only -- what the Python caller takes this entry point for and what the tests cover: the per-sample norms run one call per sample)
    if (!(C >= 2 && C <= 4 && plan->units.back().has_norm() && plan->cfg.normalization == 1)) { e3_set_error("e3_unet_backward_loss: needs 2..4 classes and a BatchNorm in front of the head"); return E3_ERR_UNSUPPORTED; }
    const HeadLossReq hl{y, loss->target, loss->class_weight, (const float*)loss->workspace + (size_t)CE_DICE_MAX_ROWS * (2 + 3 * C), gout};
    return backward_impl(plan, stream, nullptr, &hl, x, N, D, H, W, params, grads, dx, saved, saved_bytes, scratch, scratch_bytes, bucket_event,
                         bucket_after_down_block, flags);
}

static int backward_impl(e3_unet_plan* plan, void* stream, const float* dy, const HeadLossReq* hl, const float* x, int N, int D, int H, int W,
                         void* const* params, void* const* grads, float* dx,
                         void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                         void* bucket_event, int bucket_after_down_block, uint32_t flags) {
    E3_REQUIRE(plan && (dy || hl) && x && params && grads && saved && scratch, E3_ERR_INVALID, "null argument");
    packed_sig_forget(scratch);
    // the forward ran with E3_FWD_FROZEN_BN: the statistics are constants, so dx = gamma * invstd * dz (no mean / variance terms)
    const bool frozen = (flags & E3_BWD_FROZEN_BN) != 0 && (plan->cfg.normalization == 1 || plan->cfg.attention);
    hipStream_t s = (hipStream_t)stream;
    const e3_unet_cfg& cfg = plan->cfg;
    const int nb = cfg.n_blocks;
    { NetDims nd0; net_dims(plan, N, D, H, W, nd0); E3_REQUIRE(N > 0 && nd0.ok, E3_ERR_INVALID, "input too small for this network"); }
    Buffers B;
    plan_buffers(plan, N, D, H, W, true, saved, scratch, B);
    E3_REQUIRE(saved_bytes >= B.saved_bytes, E3_ERR_WORKSPACE, "`saved` buffer too small");
    E3_REQUIRE(scratch_bytes >= B.scratch_bytes, E3_ERR_WORKSPACE, "`scratch` buffer too small");
    NetDims ND; net_dims(plan, N, D, H, W, ND);
    E3_REQUIRE(ND.ok, E3_ERR_INVALID, "input too small for this network");
    const bool valid = cfg.conv_valid != 0;
    auto P = [&](int i) { return (float*)params[i]; };
    auto G = [&](int i) { return (float*)grads[i]; };
    const int C0 = plan->chan(0);
    const int nunits = (int)plan->units.size();

    // ---- conv_final (unet.py:912): da, dW, db
    const UnitBufs& last = B.ub[nunits - 1 - 0];   // last unit of the forward feeds conv_final
    // (e3_unet_backward_loss: no pass of its own -- the REDUCE pass of the last unit's BatchNorm backward, which reads the same tensor, also
    // takes the head's weight / bias gradient sums; see below)
    if (!hl) {
        const int parts = conv_final_bwd_parts(ND.Y.vox);
        const int ps = cfg.out_channels * C0 + cfg.out_channels;
        { Prof pr(plan, s, nunits, 1);
          const bool fused = plan->units.back().has_norm();
          // the gradient w.r.t. the last activation is not written: the BN backward of the last unit recomputes it from dy and the head's
          // weights (2 fma per element instead of one 4-byte write and two 4-byte reads)
          RUN(launch_conv_final_bwd(fused ? last.raw : last.act, fused ? C0 : last.act_ldc, C0, P(plan->p_final_w), dy, nullptr, C0, B.slab,
                                    cfg.out_channels, ND.Y.vox / N, N, s, fused ? last.scale : nullptr, fused ? last.shift : nullptr,
                                    plan->rrelu_of(plan->units.back().p_a >= 0 ? ActArg(0.f, P(plan->units.back().p_a)) : ActArg(cfg.act_slope), nunits - 1))); }
        RUN(launch_colsum_finalize(B.slab, parts, ps, 0, cfg.out_channels * C0, G(plan->p_final_w), s));
        RUN(launch_colsum_finalize(B.slab, parts, ps, cfg.out_channels * C0, cfg.out_channels, G(plan->p_final_b), s));
    }

    auto bwd_split = [&](int k) -> int {      // split-K of the data-gradient conv of a bottom-level unit
        const ConvUnit& u = plan->units[k];
        if (cfg.conv_valid || !B.wpk_d[k] || !B.skws) return 1;
        const int S = conv_wino_splitk(ND.u[k].in.D, ND.u[k].in.H, ND.u[k].in.W, u.cout, u.cin);
        return S > 1 ? S : 1;
    };
    const bool w4d_ok = !(bucket_event != nullptr && ((flags >> 8) & 0x1fu) == 0);
    // REDUCE pass of a BatchNorm backward inside the data-gradient launch that produces its dA (ConvArgs::br_*, conv_wino4.hip): unit k1 is a plain
    // 3x3x3 conv whose input is the activation of the plain BatchNorm unit k1 - 1 -- the second conv of every encoder / decoder block.
    // bnred_parts[k1] > 0: that launch writes the partial rows of unit k1 - 1, whose own reduce pass (a read of dA and x) is skipped.
    std::vector<int> bnred_parts((size_t)nunits, 0);
    {
        // ON where the data gradient runs on the F(2x2x4) kernel of conv_wino4.hip, whose store phase holds dA as whole voxel rows (E3_NO_BNRED_FUSE=1: A/B
        // switch; round 4 had the fusion in a 16-tile F(2x2x2) kernel that was slower than the persistent kernel by more than the pass it saved)
        static const bool off = getenv("E3_NO_BNRED_FUSE") != nullptr;
        const int reserve_req0 = bucket_event ? (int)((flags >> 8) & 0x1fu) * 8 : 0;
        for (int k1 = 1; k1 < nunits && !off && !valid && !cfg.attention && !cfg.resunet && cfg.normalization == 1 && reserve_req0 == 0; ++k1) {
            const ConvUnit& u1 = plan->units[k1];
            const ConvUnit& u0 = plan->units[k1 - 1];
            // ... or the FIRST conv of a decoder block behind the transposed conv's BatchNorm: its data gradient is that of the whole concat buffer, whose first
            // half is dA of that unit (ConvArgs::br_cols; the second half belongs to the encoder's pooled unit, whose dA also takes the pool's gradient)
            const bool cat_pair = u1.to_cat && u0.is_up == 1 && !cfg.merge_add && 2 * u0.cout == u1.cin && u0.cout % 32 == 0;
            if (u1.is_up || u1.planar || u1.cin < 8 || (u1.to_cat && !cat_pair) || u1.res_in >= 0 || !B.wpk_d[k1] || bwd_split(k1) != 1) continue;
            if ((u0.is_up && !cat_pair) || u0.level != u1.level || !u0.has_norm() || u0.p_a >= 0 || u0.res_in >= 0 || (!cat_pair && u0.cout != u1.cin)) continue;
            if (u0.enc_last && u0.level < nb - 1) continue;      // (pooled unit: its dA is pool gradient + skip gradient)
            if (plan->rrelu_of(ActArg(cfg.act_slope), k1 - 1).seed != 0u || !(cfg.act_slope >= 0.f && cfg.act_slope <= 1.f)) continue;      // (constant-slope activations: ReLU, LeakyReLU, identity)
            const LevelDims& c1 = ND.u[k1].in;
            const LevelDims& o0 = ND.u[k1 - 1].out;
            if (c1.D != o0.D || c1.H != o0.H || c1.W != o0.W) continue;
            // (measured at cfg 2: the launch costs +17..23 % with the reduction on board, the pass it replaces ~10 us at level 2, 30 at level 1, 105 at
            // level 0 -- below 32 MB the separate pass is cheaper)
            static const size_t min_mb = getenv("E3_BNRED_MIN_MB") ? (size_t)atol(getenv("E3_BNRED_MIN_MB")) : 32;      // (tests: 0 = wherever the grid allows)
            if (o0.vox * (size_t)u0.cout * 4 < (min_mb << 20)) continue;
            const int parts = ((w4d_ok && conv_wino_layout(CF_WINO4, c1.D, c1.H, c1.W, u1.cout, u1.cin, 1) == 2) ? conv_wino4_bnred_parts(N, c1.D, c1.H, c1.W, u1.cout, u1.cin) : 0);
            if (parts > 0 && parts <= bn_bwd_parts(o0.vox, u0.cout)) bnred_parts[(size_t)k1] = parts;
        }
    }
    // a data gradient takes no ReLU / arg-max decision: F(2x2x4) Winograd tiles (conv_wino4.hip) -- except in the overlapped data-parallel mode without a CU
    // reserve, whose launches behind the bucket event must be one-brick kernels (CF_NO_PERSIST)
    const int w4d = (bucket_event != nullptr && ((flags >> 8) & 0x1fu) == 0) ? 0 : CF_WINO4;
    {   // dgrad form of the Winograd weights of every layer, in one launch
        std::vector<WinoPackJob> jobs;
        for (int k = 0; k < nunits; ++k) {
            if (!(B.wpk_d[k] && (k > 0 || dx))) continue;
            const ConvUnit& u = plan->units[k];
            const int S = bwd_split(k);
            if (S == 1) { jobs.push_back({P(u.p_w), B.wpk_d[k], u.cout, u.cin, 1, 0, 0, conv_wino_layout(bnred_parts[(size_t)k] ? (CF_BNRED | CF_WINO4) : w4d, ND.u[k].in.D, ND.u[k].in.H, ND.u[k].in.W, u.cout, u.cin, 1)}); continue; }
            for (int sp = 0; sp < S; ++sp)     // dgrad: the GEMM-K channels are the conv's OUTPUT channels
                jobs.push_back({P(u.p_w), B.wpk_d[k] + sp * conv_packed_floats(CONV_K3, u.cout / S, u.cin), u.cout, u.cin, 1, sp * (u.cout / S), u.cout / S});
        }
        if (!jobs.empty()) RUN(launch_wino_pack_multi(jobs.data(), (int)jobs.size(), s));
    }
    RUN(launch_fill(B.ones, 1.f, (size_t)plan->chan(nb - 1), s));
    RUN(launch_fill(B.zeros, 0.f, (size_t)4 * plan->chan(nb - 1), s));
    // ---- walk the units backwards.  `g` = gradient w.r.t. the current unit's activation.
    const float* g = B.g1[0]; int g_ldc = C0;
    bool event_done = bucket_event == nullptr;
    // after the bucket event a collective may be resident on some CUs: the one-round kernels leave `reserve` CUs alone (E3_BWD_CU_RESERVE)
    const int reserve_req = bucket_event ? (int)((flags >> 8) & 0x1fu) * 8 : 0;
    auto reserve = [&]() { return event_done ? reserve_req : 0; };
    // Winograd weight gradients deferred to ONE stream-K launch behind the loop (Buffers::dz_u) -- unless the caller wants gradients of a bucket early (the
    // overlapped all-reduce) or the per-layer profile of a weight gradient is being taken
    const bool defer_wgrad = bucket_event == nullptr && !(plan->prof_layer >= 0 && plan->prof_which == 2);
    std::vector<WgradSkLayer> wsk_layers;
    std::vector<WgradReduceJob> wred_jobs;   // slab reductions of the weight gradients (units with their own slab), several per launch
    size_t wred_bytes = 0;                   // pending slab bytes (flushing every 48 / 96 / 160 MB was measured: no better than one launch)
    const size_t wred_limit = ~(size_t)0;
    auto wred_push = [&](const WgradReduceJob& j) -> int {
        wred_jobs.push_back(j);
        wred_bytes += (size_t)j.splits * j.T * j.RPad * j.CPad * 4;
        if (wred_bytes < wred_limit) return E3_OK;
        const int rc = launch_wgrad_reduce_multi(wred_jobs.data(), (int)wred_jobs.size(), s);
        wred_jobs.clear(); wred_bytes = 0;
        return rc;
    };
    std::vector<ColsumJob> bias_jobs;     // conv-bias gradients (column sums of the apply pass' partials), flushed in one launch
    int pending_res = -1;                  // residual unit whose shortcut gradient still has to reach its ConvBlock's input
    for (int k = nunits - 1; k >= 0; --k) {
        const ConvUnit& u = plan->units[k];
        const UnitBufs& b = B.ub[k];
        const LevelDims& lo = ND.u[k].out;         // the unit's output tensor (BatchNorm / activation / pool work on it)
        const LevelDims& ci = ND.u[k].in;          // plain convs: the grid the conv kernels run on
        const bool vcrop = valid && !u.is_up;
        const int j = u.level;
        const bool is_down = u.is_down;
        const bool is_enc_conv2 = u.enc_last;
        const bool pooled_unit = is_enc_conv2 && j < nb - 1;
        const int kd = u.planar ? 1 : 2;
        if (!event_done && is_down) {
            const int blk = j;   // encoder block index == level
            if (is_enc_conv2 && blk == bucket_after_down_block - 1) {
                if (!bias_jobs.empty()) { RUN(launch_colsum_multi(bias_jobs.data(), (int)bias_jobs.size(), s)); bias_jobs.clear(); }   // the bucket's gradients must be final
                if (!wred_jobs.empty()) { RUN(launch_wgrad_reduce_multi(wred_jobs.data(), (int)wred_jobs.size(), s)); wred_jobs.clear(); wred_bytes = 0; }
                E3_CHECK_HIP(hipEventRecord((hipEvent_t)bucket_event, s)); event_done = true;
            }
        }
        AttDims att_d{}; AttParams att_p{};
        if (u.is_up && cfg.attention) {
            // GridAttention backward (unet.py:393,509-530): the gradient of the concat buffer's second half -> BatchNorm -> W -> gate; the
            // gradients of x (the skip) and, after this unit's data gradient below, of the gating signal
            const int C = u.cout;
            const AttUnit& au = plan->att[j];
            Buffers::AttBufs& ab = B.att[j];
            att_d = att_dims(plan, ND, N, j, (size_t)k); att_p = att_params(params, au);
            const AttParams ag = att_params(grads, au);
            const UnitBufs& eb = B.ub[(size_t)plan->enc_last_unit[j]];
            const float* xs = valid ? ab.x : eb.act; const int ldx = valid ? C : eb.act_ldc;
            float* dz = B.g1[j];                 // (free: the gradient of this unit's activation sits in dcat)
            BnBwdArgs a{};
            a.x = ab.raw; a.x_ldc = C; a.mean = ab.mean; a.invstd = ab.invstd; a.scale = ab.scale; a.shift = ab.shift; a.gamma = P(au.p_g);
            a.act = ActArg(1.f);
            a.g1 = cfg.merge_add ? B.dcat[j] : B.dcat[j] + C; a.g1_ldc = cfg.merge_add ? C : 2 * C;
            a.kd = 2; a.N = N; a.D = lo.D; a.H = lo.H; a.W = lo.W; a.C = C;
            a.parts = bn_bwd_parts(lo.vox, C); a.part = ab.bnpart; a.coef = B.small; a.dx = dz; a.dx_ldc = C;
            RUN(launch_bn_bwd_reduce(a, s));
            RUN(launch_bn_bwd_finalize(a.part, a.parts, C, (float)(1.0 / (double)lo.vox), G(au.p_g), G(au.p_be), B.small, s));
            if (frozen) a.coef = B.zeros;
            RUN(launch_bn_bwd_apply(a, s));
            bias_jobs.push_back({a.part, a.parts, 3 * C, 2 * C, C, G(au.p_wb)});
            RUN(launch_att_bwd(att_d, dz, xs, ldx, B.ub[k - 1].act, B.ub[k - 1].act_ldc, ab.f, ab.sgm, ab.att, att_p, ag, ab.gx, B.att_dphi, B.att_tf,
                               B.att_tc, B.att_df, B.att_part, s));
        }
        // -- BN + ReLU (+ pool, + skip) backward -> dxr = gradient w.r.t. the raw conv output
        // (residual unit: the raw tensor is conv2(..) + shortcut; its gradient also feeds the shortcut and must outlive the ConvBlock's first conv)
        float* dxr = u.res_in >= 0 ? B.gres[j] : ((defer_wgrad && B.dz_u[k]) ? B.dz_u[k] : B.g2[j]);
        // Channel-chunked dxr ([Cout / 8][voxel][8]) where BOTH its consumers stage 8-channel chunks of it -- the F(2x2x4) data gradient and the Winograd
        // weight gradient: a halo row of a chunk is then one contiguous run instead of 32 bytes out of every voxel's row (measured on the staging of
        // conv3_wino4_kernel: profiles/r05_w4_phases.md section 5).  The APPLY pass below writes it that way at no cost.  E3_NO_CHUNKED=1: A/B switch.
        static const bool no_chunked = getenv("E3_NO_CHUNKED") != nullptr;
        const int dgrad_flags = ((bucket_event != nullptr && event_done && reserve() == 0) ? CF_NO_PERSIST : 0) | w4d | (bnred_parts[(size_t)k] ? (CF_BNRED | CF_WINO4) : 0);
        const bool dz_chunked = !no_chunked && !u.is_up && !u.planar && u.cin >= 8 && u.res_in < 0 && !vcrop && (k > 0 || dx) && B.wpk_d[k] && bwd_split(k) == 1 &&
                                wgrad_use_wino(CONV_K3) && chunked_layout_ok(lo.vox, u.cout) && conv_wino_layout(dgrad_flags, ci.D, ci.H, ci.W, u.cout, u.cin, 1) == 2;
        const size_t dz_chunk = dz_chunked ? lo.vox * 8 : 0;
        bool fuse_first = false; SmallWgradFuse first_fuse{};
        {
            BnBwdArgs a{};
            a.x = b.raw; a.x_ldc = u.cout; a.mean = b.mean; a.invstd = b.invstd; a.scale = b.scale; a.shift = b.shift;
            a.act = plan->rrelu_of(u.p_a >= 0 ? ActArg(0.f, P(u.p_a)) : ActArg(cfg.act_slope), k);
            if (u.has_norm()) a.gamma = P(u.p_g);
            else {   // nn.Identity + activation: dz = dA * act'(z) is the APPLY pass with the constants of an identity "norm":
                     // mean 0, invstd 1, gamma 1, c = k = 0  =>  dx = dz, sum dx = conv-bias gradient.  ReLU: x := a (mask 1*a + 0 > 0);
                     // other activations kept the raw accumulations: z = raw*1 + bias (scale, shift of the forward's bias fold)
                a.mean = B.zeros; a.invstd = B.ones; a.gamma = B.ones;
                if (cfg.act_slope == 0.f) { a.x = b.act; a.x_ldc = b.act_ldc; a.scale = B.ones; a.shift = B.zeros; }
            }
            if (k == nunits - 1) {      // incoming gradient = that of the 1x1x1 head, recomputed on the fly
                a.g1 = nullptr; a.head_dy = dy; a.head_w = P(plan->p_final_w); a.head_cout = cfg.out_channels; a.head_S = ND.Y.vox / N;
                if (hl) { a.hl_logits = hl->logits; a.hl_target = hl->target; a.hl_cw = hl->cw; a.hl_coef = hl->coef; a.hl_gout = hl->gout; a.head_part = B.slab; }
            }
            else if (pooled_unit && cfg.attention && !valid) {   // the GridAttention's backward wrote the skip's gradient
                a.g1 = B.gskip[j]; a.g1_ldc = u.cout; a.gpool = g; a.a = b.act; a.a_ldc = b.act_ldc; a.pooled = B.pooled[j];
            }
            else if (pooled_unit && valid) {   // the skip was centre-cropped: its gradient is zero outside that box
                const LevelDims& dc = ND.u[(size_t)plan->up_unit[j]].out;
                if (cfg.attention) RUN(launch_pad_box(B.att[j].gx, B.gskip[j], u.cout, N, dc.D, dc.H, dc.W, lo.D, lo.H, lo.W, s, ND.sd_[j], ND.sh_[j], ND.sw_[j], u.cout));
                else
                RUN(launch_pad_box(cfg.merge_add ? B.dcat[j] : B.dcat[j] + u.cout, B.gskip[j], u.cout, N, dc.D, dc.H, dc.W, lo.D, lo.H, lo.W, s,
                                   ND.sd_[j], ND.sh_[j], ND.sw_[j], cfg.merge_add ? u.cout : 2 * u.cout));
                a.g1 = B.gskip[j]; a.g1_ldc = u.cout; a.gpool = g; a.a = b.act; a.a_ldc = b.act_ldc; a.pooled = B.pooled[j];
            }
            else if (pooled_unit) { a.g1 = cfg.merge_add ? B.dcat[j] : B.dcat[j] + u.cout; a.g1_ldc = cfg.merge_add ? u.cout : 2 * u.cout; a.gpool = g; a.a = b.act; a.a_ldc = b.act_ldc; a.pooled = B.pooled[j]; }
            else { a.g1 = g; a.g1_ldc = g_ldc; }
            a.kd = kd; a.N = N; a.D = lo.D; a.H = lo.H; a.W = lo.W; a.C = u.cout;
            a.parts = bn_bwd_parts(lo.vox, u.cout); a.part = B.bnpart_u[k]; a.coef = B.small; a.dx = dxr; a.dx_ldc = u.cout; a.dx_chunk = dz_chunk;
            if (!u.has_norm() && u.p_a >= 0) {   // no norm, but the PReLU slope gradient needs the REDUCE pass (its row 2)
                RUN(launch_bn_bwd_reduce(a, s));
                RUN(launch_prelu_dslope(a.part, a.parts, u.cout, B.small + 4 * u.cout, G(u.p_a), s));
            }
            const int fused_red = (k + 1 < nunits) ? bnred_parts[(size_t)k + 1] : 0;      // the data gradient of unit k + 1 took the sums along
            if (u.has_norm() && fused_red) {
                RUN(launch_bn_bwd_finalize(a.part, fused_red, u.cout, (float)(1.0 / (double)lo.vox), G(u.p_g), G(u.p_be), B.small, s));
                if (frozen) a.coef = B.zeros;
            } else if (u.has_norm()) {
                { Prof pr(plan, s, nunits, hl && k == nunits - 1 ? 1 : -1); RUN(launch_bn_bwd_reduce(a, s)); }
                if (a.head_part) {      // the head's gradients from the partial sums of that pass
                    const int ps = cfg.out_channels * C0 + cfg.out_channels;
                    RUN(launch_colsum_finalize(B.slab, a.parts, ps, 0, cfg.out_channels * C0, G(plan->p_final_w), s));
                    RUN(launch_colsum_finalize(B.slab, a.parts, ps, cfg.out_channels * C0, cfg.out_channels, G(plan->p_final_b), s));
                    a.head_part = nullptr;
                }
                if (u.p_a >= 0) RUN(launch_prelu_dslope(a.part, a.parts, u.cout, B.small + 4 * u.cout, G(u.p_a), s));
                RUN(launch_bn_bwd_finalize(a.part, a.parts, u.cout, (float)(1.0 / (double)lo.vox), G(u.p_g), G(u.p_be), B.small, s));
                if (frozen) a.coef = B.zeros;      // (dgamma = sum dz*xhat and dbeta = sum dz stand; the correction terms vanish)
                if (cfg.normalization == 2)
                    RUN(launch_gn_bwd_coef(G(u.p_g), G(u.p_be), P(u.p_g), b.invstd, u.cout, u.cout / cfg.num_groups, (float)(1.0 / (double)lo.vox), B.small, s));
            } else a.coef = B.zeros;
            // the first conv without a requested input gradient: dxr has a single consumer (its wgrad), which computes it on the fly
            static const bool no_first_fuse = getenv("E3_NO_FIRST_FUSE") != nullptr;     // A/B switch
            fuse_first = k == 0 && !dx && u.cin < 8 && !u.is_up && !no_first_fuse && cfg.normalization != 2 && !valid;   // (the fused staging has no group terms)
            if (fuse_first) {
                first_fuse = SmallWgradFuse{a.x, a.x_ldc, a.g1, a.g1_ldc, a.scale, a.shift, a.mean, a.invstd, a.gamma, a.coef, B.biaspart0, a.act};
                bias_jobs.push_back({B.biaspart0, conv_small_wgrad_splits(N, ci.D, ci.H, ci.W, u.planar), u.cout, 0, u.cout, G(u.p_b)});
            } else {
                RUN(launch_bn_bwd_apply(a, s));
                bias_jobs.push_back({a.part, a.parts, 3 * u.cout, 2 * u.cout, u.cout, G(u.p_b)});
            }
        }
        // -- input activation of this conv
        auto input_of = [&](int q, const float*& xi, int& xi_ldc) {
            const ConvUnit& uq = plan->units[q];
            if (q == 0) { xi = cfg.in_channels > 1 ? B.xin : x; xi_ldc = cfg.in_channels; return; }
            const ConvUnit& pu = plan->units[q - 1];
            const bool prev_pooled = pu.enc_last && pu.level < nb - 1 && uq.is_down;
            if (prev_pooled) { xi = B.pooled[pu.level]; xi_ldc = pu.cout; }
            else if (pu.is_up && cfg.merge_add) { xi = B.sum[pu.level]; xi_ldc = pu.cout; }
            else if (pu.is_up) { xi = B.cat[pu.level]; xi_ldc = 2 * pu.cout; }
            else { xi = B.ub[q - 1].act; xi_ldc = B.ub[q - 1].act_ldc; }
            if (uq.is_up) {   // input of upconv k-th: activation at level j+1 = output of the previous unit (packed or cat-skip of bottom block)
                xi = B.ub[q - 1].act; xi_ldc = B.ub[q - 1].act_ldc;
            }
        };
        const float* xin; int xin_ldc;
        input_of(k, xin, xin_ldc);
        if (u.res_in >= 0) {      // shortcut: projection gradients now (dxr and the ConvBlock's input are at hand); its input gradient joins the
                                  // data gradient of the ConvBlock's first conv below
            if (u.p_pw >= 0) {
                const float* xr; int xr_ldc;
                input_of(u.res_in, xr, xr_ldc);
                RUN(launch_pw_wgrad(dxr, u.cout, u.cout, xr, xr_ldc, plan->units[u.res_in].cin, B.att_part, G(u.p_pw), G(u.p_pb), lo.vox, s));
            }
            pending_res = k;
        }
        // -- weight gradient
        const float* dyu = dxr;      // gradient of the conv output on the grid the conv kernels run on: zero in cropped-away voxels
        if (vcrop) {                 // ('valid' plain conv: dxr sits at (od, oh, ow) inside the input grid)
            RUN(launch_pad_box(dxr, B.rpad, u.cout, N, lo.D, lo.H, lo.W, ci.D, ci.H, ci.W, s, ND.u[k].od, ND.u[k].oh, ND.u[k].ow));
            dyu = B.rpad;
        }
        if (u.is_up == 2 && cfg.up_resize >= 3) {
            // backward of (1x1x1 conv at low resolution -> up-sampling -> crop): pad, transpose of the up-sampling, then the conv's gradients
            const LevelDims& li = ND.u[k].in;
            const int sd = u.planar ? 1 : 2, Ud = li.D * sd, Uh = li.H * 2, Uw = li.W * 2;
            if (!(Ud == lo.D && Uh == lo.H && Uw == lo.W)) {
                RUN(launch_pad_box(dxr, B.rpad, u.cout, N, lo.D, lo.H, lo.W, Ud, Uh, Uw, s));
                dyu = B.rpad;
            }
            RUN(launch_downsample_sum(dyu, B.rdu, u.cout, u.cout, N, li.D, li.H, li.W, sd, s, cfg.up_resize == 4));
            dyu = B.rdu;             // gradient of the low-resolution conv output [N, li, Cout]
            WgradArgs a{};
            a.x = xin; a.x_ldc = xin_ldc; a.Cin = u.cin; a.dy = dyu; a.dy_ldc = u.cout; a.Cout = u.cout; a.part = B.slab;
            a.N = N; a.D = li.D; a.H = li.H; a.W = li.W; a.CoPad = cdiv(u.cout, 32) * 32; a.CiPad = cdiv(u.cin, 32) * 32;
            a.splits = wgrad_splits(CONV_K3_PLANAR, N, li.D, li.H, li.W, u.cin, u.cout);
            { Prof pr(plan, s, k, 2); RUN(launch_wgrad_mfma(CONV_K3_PLANAR, a, s)); }
            RUN(launch_wgrad_reduce(B.slab, B.gemb, a.splits, 9, a.CoPad, a.CiPad, u.cout, u.cin, s));
            RUN(launch_extract_center_tap(B.gemb, G(u.p_w), (size_t)u.cout * u.cin, 9, s));
        } else if (u.is_up == 2) {
            const LevelDims& li = ND.u[k].in;
            const int Ud = li.D * (u.planar ? 1 : 2), Uh = li.H * 2, Uw = li.W * 2, taps = u.planar ? 9 : 27;
            const ConvKind kind = u.planar ? CONV_K3_PLANAR : CONV_K3;
            if (!(Ud == lo.D && Uh == lo.H && Uw == lo.W)) {
                RUN(launch_pad_box(dxr, B.rpad, u.cout, N, lo.D, lo.H, lo.W, Ud, Uh, Uw, s));
                dyu = B.rpad;
            }
            WgradArgs a{};
            a.x = B.ups[k]; a.x_ldc = u.cin; a.Cin = u.cin; a.dy = dyu; a.dy_ldc = u.cout; a.Cout = u.cout; a.part = B.slab;
            a.N = N; a.D = Ud; a.H = Uh; a.W = Uw; a.CoPad = cdiv(u.cout, 32) * 32; a.CiPad = cdiv(u.cin, 32) * 32;
            a.splits = wgrad_splits(kind, N, Ud, Uh, Uw, u.cin, u.cout);
            { Prof pr(plan, s, k, 2); RUN(launch_wgrad_mfma(kind, a, s)); }
            RUN(launch_wgrad_reduce(B.slab, G(u.p_w), a.splits, taps, a.CoPad, a.CiPad, u.cout, u.cin, s));
        } else if (u.is_up) {
            const LevelDims& li = ND.u[k].in;
            const int sd = u.planar ? 1 : 2;
            WgradArgs a{};
            a.x = xin; a.x_ldc = xin_ldc; a.Cin = u.cin; a.dy = dxr; a.dy_ldc = u.cout; a.Cout = u.cout; a.part = B.slab_u[k] ? B.slab_u[k] : B.slab;
            a.N = N; a.D = li.D; a.H = li.H; a.W = li.W; a.Do = lo.D; a.Ho = lo.H; a.Wo = lo.W; a.sd = sd;
            a.CoPad = cdiv(u.cout, 32) * 32; a.CiPad = cdiv(u.cin, 32) * 32;
            a.splits = wgrad_splits(CONV_POINT, N, li.D, li.H, li.W, u.cin, u.cout);
            { Prof pr(plan, s, k, 2); RUN(launch_wgrad_mfma(CONV_POINT, a, s)); }
            if (B.slab_u[k]) RUN(wred_push({a.part, G(u.p_w), a.splits, sd * 4, a.CiPad, a.CoPad, u.cin, u.cout}));
            else RUN(launch_wgrad_reduce(B.slab, G(u.p_w), a.splits, sd * 4, a.CiPad, a.CoPad, u.cin, u.cout, s));
        } else if (u.cin < 8) {
            const int taps = u.planar ? 9 : 27;
            const int splits = conv_small_wgrad_splits(N, ci.D, ci.H, ci.W, u.planar);
            { Prof pr(plan, s, k, 2); RUN(launch_conv_small_wgrad(xin, u.cin, dyu, u.cout, B.slab, N, ci.D, ci.H, ci.W, u.cout, u.planar, s, fuse_first ? &first_fuse : nullptr)); }
            RUN(launch_wgrad_reduce(B.slab, G(u.p_w), splits, taps, u.cout, u.cin, u.cout, u.cin, s));
        } else {
            const int taps = u.planar ? 9 : 27;
            const ConvKind kind = u.planar ? CONV_K3_PLANAR : CONV_K3;
            if (defer_wgrad && B.dz_u[k] && kind == CONV_K3) {      // one stream-K launch for all of them behind the loop
                wsk_layers.push_back(WgradSkLayer{xin, xin_ldc, u.cin, dyu, u.cout, u.cout, dz_chunk, N, ci.D, ci.H, ci.W, G(u.p_w)});
            } else {
                WgradArgs a{};
                a.x = xin; a.x_ldc = xin_ldc; a.Cin = u.cin; a.dy = dyu; a.dy_ldc = u.cout; a.dy_chunk = dz_chunk; a.Cout = u.cout; a.part = B.slab_u[k] ? B.slab_u[k] : B.slab;
                a.N = N; a.D = ci.D; a.H = ci.H; a.W = ci.W; a.CoPad = cdiv(u.cout, 32) * 32; a.CiPad = cdiv(u.cin, 32) * 32;
                a.cu_reserve = reserve();      // (fewer, longer splits: the slab sized for the full chip is large enough)
                a.splits = wgrad_splits(kind, N, ci.D, ci.H, ci.W, u.cin, u.cout, a.cu_reserve);
                { Prof pr(plan, s, k, 2); RUN(launch_wgrad_mfma(kind, a, s)); }
                if (B.slab_u[k]) RUN(wred_push({a.part, G(u.p_w), a.splits, taps, a.CoPad, a.CiPad, u.cout, u.cin}));
                else RUN(launch_wgrad_reduce(B.slab, G(u.p_w), a.splits, taps, a.CoPad, a.CiPad, u.cout, u.cin, s));
            }
        }
        // -- data gradient -> g for the previous unit
        if (k == 0 && !dx) break;
        if (u.is_up == 2 && cfg.up_resize >= 3) {      // dgrad of the low-resolution 1x1x1 conv: directly the gradient of the previous unit's output
            const LevelDims& li = ND.u[k].in;
            const int NPad = pad_cols(u.cin);
            RUN(launch_embed_center_tap(P(u.p_w), B.wemb, (size_t)u.cout * u.cin, 9, s));
            RUN(launch_pack_conv_auto(CONV_K3_PLANAR, 1, B.wemb, B.wpack, u.cout, u.cin, N, li.D, li.H, li.W, 0, s));
            ConvArgs a{};
            a.x = dyu; a.x_ldc = u.cout; a.Cin = u.cout; a.wt = B.wpack; a.y = B.g1[j + 1]; a.y_ldc = u.cin;
            a.N = N; a.D = li.D; a.H = li.H; a.W = li.W; a.sd = 2;
            a.Cout = u.cin; a.Ncols = u.cin; a.NPad = NPad; a.G = 1;
            a.flags = (bucket_event != nullptr && event_done) ? CF_NO_PERSIST : 0;
            { Prof pr(plan, s, k, 1); RUN(launch_conv_mfma(CONV_K3_PLANAR, a, s)); }
            g = B.g1[j + 1]; g_ldc = u.cin;
        } else if (u.is_up == 2) {      // conv dgrad on the up-sampled grid, then the sum over each (sd x 2 x 2) block = backward of nn.Upsample(nearest)
            const LevelDims& li = ND.u[k].in;
            const int sd = u.planar ? 1 : 2, Ud = li.D * sd, Uh = li.H * 2, Uw = li.W * 2, NPad = pad_cols(u.cin);
            const ConvKind kind = u.planar ? CONV_K3_PLANAR : CONV_K3;
            RUN(launch_pack_conv_auto(kind, 1, P(u.p_w), B.wpack, u.cout, u.cin, N, Ud, Uh, Uw, 0, s));
            ConvArgs a{};
            a.x = dyu; a.x_ldc = u.cout; a.Cin = u.cout; a.wt = B.wpack; a.y = B.rdu; a.y_ldc = u.cin;
            a.N = N; a.D = Ud; a.H = Uh; a.W = Uw; a.sd = 2;
            a.Cout = u.cin; a.Ncols = u.cin; a.NPad = NPad; a.G = 1;
            a.flags = (bucket_event != nullptr && event_done) ? CF_NO_PERSIST : 0;
            { Prof pr(plan, s, k, 1); RUN(launch_conv_mfma(kind, a, s)); }
            RUN(launch_downsample_sum(B.rdu, B.g1[j + 1], u.cin, u.cin, N, li.D, li.H, li.W, sd, s, cfg.up_resize == 2));
            g = B.g1[j + 1]; g_ldc = u.cin;
        } else if (u.is_up) {
            const LevelDims& li = ND.u[k].in;
            const int sd = u.planar ? 1 : 2, taps = sd * 4, NPad = pad_cols(u.cin);
            RUN(launch_pack_weights(PACK_UP_DGRAD, P(u.p_w), B.wpack, u.cout, u.cin, taps, NPad, s));
            ConvArgs a{};
            a.x = dxr; a.x_ldc = u.cout; a.Cin = u.cout; a.wt = B.wpack; a.y = B.g1[j + 1]; a.y_ldc = u.cin;
            a.N = N; a.D = li.D; a.H = li.H; a.W = li.W; a.Do = lo.D; a.Ho = lo.H; a.Wo = lo.W; a.sd = sd;
            a.Cout = u.cin; a.Ncols = u.cin; a.NPad = NPad; a.G = taps; a.flags = CF_GATHER_UP;
            { Prof pr(plan, s, k, 1); RUN(launch_conv_mfma(CONV_POINT, a, s)); }
            g = B.g1[j + 1]; g_ldc = u.cin;
        } else {
            const int taps = u.planar ? 9 : 27, NPad = pad_cols(u.cin);
            const ConvKind kind = u.planar ? CONV_K3_PLANAR : CONV_K3;
            (void)taps;
            if (!B.wpk_d[k]) RUN(launch_pack_conv_auto(kind, 1, P(u.p_w), B.wpack, u.cout, u.cin, N, ci.D, ci.H, ci.W, w4d, s));
            const bool to_cat = u.to_cat;   // UpConv.conv1: gradient of the concat buffer
            float* out; int out_ldc = u.cin;
            if (k == 0) out = (cfg.in_channels > 1) ? B.g1[0] : dx;   // g1[0] is free by now (C0 >= in_channels)
            else if (to_cat) out = B.dcat[j];
            else out = B.g1[j];
            ConvArgs a{};
            a.x = dyu; a.x_ldc = u.cout; a.Cin = u.cout; a.wt = B.wpk_d[k] ? B.wpk_d[k] : B.wpack; a.y = out; a.y_ldc = out_ldc;
            a.N = N; a.D = ci.D; a.H = ci.H; a.W = ci.W; a.sd = 2;
            a.Cout = u.cin; a.Ncols = u.cin; a.NPad = NPad; a.G = 1;
            // the gradient all-reduce may be running on some CUs: with a reserve the persistent kernel leaves them alone, without one the
            // one-brick-per-workgroup kernel degrades by the fraction of CUs taken instead of needing a second round
            a.cu_reserve = reserve();
            a.flags = ((bucket_event != nullptr && event_done && a.cu_reserve == 0) ? CF_NO_PERSIST : 0) | w4d;
            a.x_chunk = dz_chunk;
            if (bnred_parts[(size_t)k]) {      // this launch also takes the REDUCE sums of unit k - 1's BatchNorm backward
                const UnitBufs& b0 = B.ub[k - 1];
                a.flags |= CF_BNRED | CF_WINO4;
                a.br_x = b0.raw; a.br_ldc = plan->units[k - 1].cout; a.br_scale = b0.scale; a.br_shift = b0.shift; a.br_mean = b0.mean; a.br_invstd = b0.invstd;
                a.br_slope = cfg.act_slope; a.br_part = B.bnpart_u[k - 1]; a.br_cols = u.to_cat ? plan->units[k - 1].cout : 0;
            }
            const int S = (kind == CONV_K3) ? bwd_split(k) : 1;
            const size_t gvox = (size_t)N * ci.D * ci.H * ci.W;
            if (S > 1) {
                a.splitk = S; a.sk_x = u.cout / S; a.Cin = u.cout / S; a.sk_w = (unsigned)conv_packed_floats(CONV_K3, u.cout / S, u.cin);
                a.sk_y = gvox * u.cin; a.y = B.skws; a.y_ldc = u.cin;
            }
            { Prof pr(plan, s, k, 1); RUN(launch_conv_mfma(kind, a, s)); }
            if (S > 1) RUN(launch_splitk_reduce(B.skws, S, gvox * u.cin, nullptr, out, out_ldc, u.cin, gvox, nullptr, s));
            if (k == 0) { if (cfg.in_channels > 1) RUN(launch_ndhwc_to_ncdhw(B.g1[0], cfg.in_channels, dx, N, cfg.in_channels, ND.X[0].vox / N, s)); }
            if (pending_res >= 0 && plan->units[pending_res].res_in == k) {      // + the shortcut's share: d inp += proj^T dsum (or dsum itself)
                const ConvUnit& ru = plan->units[pending_res];
                const size_t rvox = ND.u[pending_res].out.vox;
                if (ru.p_pw >= 0) RUN(launch_pw_dgrad_acc(B.gres[j], ru.cout, ru.cout, P(ru.p_pw), out, out_ldc, u.cin, rvox, s));
                else RUN(launch_add_views(out, out_ldc, B.gres[j], ru.cout, out, out_ldc, rvox, u.cin, s));
                pending_res = -1;
            }
            g = out; g_ldc = (to_cat && !cfg.merge_add) ? 2 * u.cout : u.cin;   // concat: the next unit (upconv) reads the first half, ldc = 2*C; add: d(up + skip) goes to both
        }
        if (u.is_up && cfg.attention)      // the block's input is also the gate's gating signal: + dphi . phi_w
            RUN(launch_att_bwd_gate_input(att_d, att_resized(att_d) ? B.att_dphi : B.att_df, att_p, B.g1[j + 1], u.cin, s));
    }
    if (!bias_jobs.empty()) RUN(launch_colsum_multi(bias_jobs.data(), (int)bias_jobs.size(), s));
    if (!wsk_layers.empty()) RUN(launch_wgrad_wino_sk(wsk_layers.data(), (int)wsk_layers.size(), B.wsk_slab, B.wsk_floats, s));
    if (!wred_jobs.empty()) RUN(launch_wgrad_reduce_multi(wred_jobs.data(), (int)wred_jobs.size(), s));
    if (!event_done) E3_CHECK_HIP(hipEventRecord((hipEvent_t)bucket_event, s));
    return E3_OK;
}

}  // extern "C"
