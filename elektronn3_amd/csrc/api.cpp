// Per-op entry points of the C ABI (include/e3unet.h): argument checking + launches.  Host code only.
#include "../../include/e3unet.h"
#include <stdlib.h>

#include "kernels.h"

static thread_local std::string g_err;
void e3_set_error(const std::string& msg) { g_err = msg; }

extern "C" {

const char* e3_last_error(void) { return g_err.c_str(); }
const char* e3_version(void) { return "e3unet 0.1 gfx950"; }

static inline ConvKind kind_of(int planar) { return planar ? CONV_K3_PLANAR : CONV_K3; }
static inline int pad_cols(int n) { const int t = conv_col_tile(n); return cdiv(n, t) * t; }

// ---------------------------------------------------------------------------------------------- conv 3x3x3
size_t e3_conv3d_workspace_bytes(int Cin, int Cout, int planar) {
    const int T = planar ? 9 : 27;
    (void)T;
    const size_t a = conv_packed_floats(kind_of(planar), Cin, Cout), b = conv_packed_floats(kind_of(planar), Cout, Cin);
    return align_up((a > b ? a : b) * sizeof(float), 256);
}

int e3_conv3d_stats_parts(int Cin, int Cout, int N, int D, int H, int W, int planar) {
    if (Cin < 8) return conv_small_stats_parts2(N, D, H, W, planar, Cin, Cout);
    // (a call with a BN prologue uses the direct kernel: report the larger of the two record counts, empty records are neutral)
    const int p0 = conv_stats_parts(kind_of(planar), 0, N, D, H, W, 2, Cin, Cout), p1 = conv_stats_parts(kind_of(planar), CF_NO_WINO | CF_NO_KSPLIT, N, D, H, W, 2, Cin, Cout);
    return p0 > p1 ? p0 : p1;
}

int e3_conv3d_fwd(void* stream, const float* x, int x_ldc, int Cin, const float* w, const float* bias,
                  float* y, int y_ldc, int Cout, int N, int D, int H, int W, int planar,
                  const float* pro_scale, const float* pro_shift, const float* epi_scale, const float* epi_shift,
                  float* stats, void* workspace, size_t workspace_bytes) {
    hipStream_t s = (hipStream_t)stream;
    if (Cin < 8) {
        E3_REQUIRE(!pro_scale, E3_ERR_UNSUPPORTED, "direct first-layer conv has no prologue");
        E3_REQUIRE(x_ldc == Cin, E3_ERR_UNSUPPORTED, "direct first-layer conv needs a packed input");
        ConvSmallArgs a{};
        a.x = x; a.Cin = Cin; a.w = w; a.bias = epi_scale ? nullptr : bias; a.y = y; a.y_ldc = y_ldc;
        a.N = N; a.D = D; a.H = H; a.W = W; a.Cout = Cout; a.planar = planar;
        a.epi_scale = epi_scale; a.epi_shift = epi_shift; a.stats = stats;
        return launch_conv_small_fwd(a, s);
    }
    E3_REQUIRE(workspace_bytes >= e3_conv3d_workspace_bytes(Cin, Cout, planar), E3_ERR_WORKSPACE, "conv3d workspace too small");
    const int T = planar ? 9 : 27, NPad = pad_cols(Cout);
    (void)T;
    // (the Winograd kernels have no BN prologue; the folded epilogue is the eval-mode forward, which may take the F(2x2x4) tiles of conv_wino4.hip)
    const int flags0 = pro_scale ? (CF_NO_WINO | CF_NO_KSPLIT) : ((epi_scale && !stats) ? CF_WINO4 : 0);
    int rc = launch_pack_conv_auto(kind_of(planar), 0, w, (float*)workspace, Cout, Cin, N, D, H, W, flags0, s);
    if (rc) return rc;
    ConvArgs a{};
    a.x = x; a.x_ldc = x_ldc; a.Cin = Cin; a.wt = (const float*)workspace; a.bias = epi_scale ? nullptr : bias;
    a.y = y; a.y_ldc = y_ldc; a.N = N; a.D = D; a.H = H; a.W = W; a.sd = 2;
    a.Cout = Cout; a.Ncols = Cout; a.NPad = NPad;
    a.pro_scale = pro_scale; a.pro_shift = pro_shift; a.epi_scale = epi_scale; a.epi_shift = epi_shift;
    a.stats = stats; a.G = 1; a.flags = flags0;
#ifdef E3_TIMING
    if (const char* dbg = getenv("E3_CONV_ABLATE")) a.flags |= atoi(dbg) & (256 | 512 | 1024);   // timing experiments (developer builds only)
#endif
    return launch_conv_mfma(kind_of(planar), a, s);
}

int e3_conv3d_dgrad(void* stream, const float* dy, int dy_ldc, int Cout, const float* w, float* dx, int dx_ldc, int Cin,
                    int N, int D, int H, int W, int planar, void* workspace, size_t workspace_bytes) {
    hipStream_t s = (hipStream_t)stream;
    E3_REQUIRE(workspace_bytes >= e3_conv3d_workspace_bytes(Cin, Cout, planar), E3_ERR_WORKSPACE, "conv3d workspace too small");
    const int T = planar ? 9 : 27, NPad = pad_cols(Cin);
    (void)T;
    int rc = launch_pack_conv_auto(kind_of(planar), 1, w, (float*)workspace, Cout, Cin, N, D, H, W, CF_WINO4, s);
    if (rc) return rc;
    ConvArgs a{};
    a.x = dy; a.x_ldc = dy_ldc; a.Cin = Cout; a.wt = (const float*)workspace; a.bias = nullptr;
    a.y = dx; a.y_ldc = dx_ldc; a.N = N; a.D = D; a.H = H; a.W = W; a.sd = 2;
    a.Cout = Cin; a.Ncols = Cin; a.NPad = NPad; a.G = 1; a.flags = CF_WINO4;
    return launch_conv_mfma(kind_of(planar), a, s);
}

size_t e3_conv3d_wgrad_workspace_bytes(int Cin, int Cout, int N, int D, int H, int W, int planar) {
    const int T = planar ? 9 : 27;
    if (Cin < 8) return align_up((size_t)conv_small_wgrad_splits(N, D, H, W, planar) * T * Cout * Cin * sizeof(float), 256);
    const int splits = wgrad_splits(kind_of(planar), N, D, H, W, Cin, Cout);
    return align_up((size_t)splits * T * (cdiv(Cout, 32) * 32) * (cdiv(Cin, 32) * 32) * sizeof(float), 256);
}

int e3_conv3d_wgrad(void* stream, const float* x, int x_ldc, int Cin, const float* dy, int dy_ldc, int Cout, float* dw,
                    int N, int D, int H, int W, int planar, void* workspace, size_t workspace_bytes) {
    hipStream_t s = (hipStream_t)stream;
    E3_REQUIRE(workspace_bytes >= e3_conv3d_wgrad_workspace_bytes(Cin, Cout, N, D, H, W, planar), E3_ERR_WORKSPACE, "wgrad workspace too small");
    const int T = planar ? 9 : 27;
    if (Cin < 8) {
        E3_REQUIRE(x_ldc == Cin, E3_ERR_UNSUPPORTED, "direct first-layer wgrad needs a packed input");
        const int splits = conv_small_wgrad_splits(N, D, H, W, planar);
        int rc = launch_conv_small_wgrad(x, Cin, dy, dy_ldc, (float*)workspace, N, D, H, W, Cout, planar, s);
        if (rc) return rc;
        return launch_wgrad_reduce((const float*)workspace, dw, splits, T, Cout, Cin, Cout, Cin, s);
    }
    WgradArgs a{};
    a.x = x; a.x_ldc = x_ldc; a.Cin = Cin; a.dy = dy; a.dy_ldc = dy_ldc; a.Cout = Cout; a.part = (float*)workspace;
    a.N = N; a.D = D; a.H = H; a.W = W; a.CoPad = cdiv(Cout, 32) * 32; a.CiPad = cdiv(Cin, 32) * 32;
    a.splits = wgrad_splits(kind_of(planar), N, D, H, W, Cin, Cout);
    int rc = launch_wgrad_mfma(kind_of(planar), a, s);
    if (rc) return rc;
    return launch_wgrad_reduce(a.part, dw, a.splits, T, a.CoPad, a.CiPad, Cout, Cin, s);
}

// ---------------------------------------------------------------------------------------------- transposed conv
size_t e3_convT_workspace_bytes(int Cin, int Cout, int sd) {
    const int T = sd * 4;
    const size_t a = (size_t)pad_cols(T * Cout) * Cin, b = (size_t)T * pad_cols(Cin) * Cout;
    return align_up((a > b ? a : b) * sizeof(float), 256);
}
int e3_convT_stats_parts(int Cin, int Cout, int N, int D, int H, int W, int sd) {
    return conv_stats_parts(CONV_POINT, CF_SCATTER_UP, N, D, H, W, sd, Cin, sd * 4 * Cout);
}

int e3_convT_fwd(void* stream, const float* x, int x_ldc, int Cin, const float* w, const float* bias, float* y, int y_ldc,
                 int Cout, int N, int D, int H, int W, int sd, int Do, int Ho, int Wo, float* stats,
                 void* workspace, size_t workspace_bytes) {
    hipStream_t s = (hipStream_t)stream;
    E3_REQUIRE(sd == 1 || sd == 2, E3_ERR_INVALID, "sd must be 1 or 2");
    E3_REQUIRE(workspace_bytes >= e3_convT_workspace_bytes(Cin, Cout, sd), E3_ERR_WORKSPACE, "convT workspace too small");
    const int T = sd * 4, NPad = pad_cols(T * Cout);
    int rc = launch_pack_weights(PACK_UP_FWD, w, (float*)workspace, Cout, Cin, T, NPad, s);
    if (rc) return rc;
    ConvArgs a{};
    a.x = x; a.x_ldc = x_ldc; a.Cin = Cin; a.wt = (const float*)workspace; a.bias = bias; a.y = y; a.y_ldc = y_ldc;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Do = Do; a.Ho = Ho; a.Wo = Wo; a.sd = sd;
    a.Cout = Cout; a.Ncols = T * Cout; a.NPad = NPad; a.stats = stats; a.G = 1; a.flags = CF_SCATTER_UP;
    return launch_conv_mfma(CONV_POINT, a, s);
}

int e3_convT_dgrad(void* stream, const float* dy, int dy_ldc, int Cout, const float* w, float* dx, int dx_ldc, int Cin,
                   int N, int D, int H, int W, int sd, int Do, int Ho, int Wo, void* workspace, size_t workspace_bytes) {
    hipStream_t s = (hipStream_t)stream;
    E3_REQUIRE(sd == 1 || sd == 2, E3_ERR_INVALID, "sd must be 1 or 2");
    E3_REQUIRE(workspace_bytes >= e3_convT_workspace_bytes(Cin, Cout, sd), E3_ERR_WORKSPACE, "convT workspace too small");
    const int T = sd * 4, NPad = pad_cols(Cin);
    int rc = launch_pack_weights(PACK_UP_DGRAD, w, (float*)workspace, Cout, Cin, T, NPad, s);
    if (rc) return rc;
    ConvArgs a{};
    a.x = dy; a.x_ldc = dy_ldc; a.Cin = Cout; a.wt = (const float*)workspace; a.y = dx; a.y_ldc = dx_ldc;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Do = Do; a.Ho = Ho; a.Wo = Wo; a.sd = sd;
    a.Cout = Cin; a.Ncols = Cin; a.NPad = NPad; a.G = T; a.flags = CF_GATHER_UP;
    return launch_conv_mfma(CONV_POINT, a, s);
}

size_t e3_convT_wgrad_workspace_bytes(int Cin, int Cout, int N, int D, int H, int W, int sd) {
    const int splits = wgrad_splits(CONV_POINT, N, D, H, W, Cin, Cout);
    return align_up((size_t)splits * sd * 4 * (cdiv(Cout, 32) * 32) * (cdiv(Cin, 32) * 32) * sizeof(float), 256);
}

int e3_convT_wgrad(void* stream, const float* x, int x_ldc, int Cin, const float* dy, int dy_ldc, int Cout, float* dw,
                   int N, int D, int H, int W, int sd, int Do, int Ho, int Wo, void* workspace, size_t workspace_bytes) {
    hipStream_t s = (hipStream_t)stream;
    E3_REQUIRE(workspace_bytes >= e3_convT_wgrad_workspace_bytes(Cin, Cout, N, D, H, W, sd), E3_ERR_WORKSPACE, "convT wgrad workspace too small");
    WgradArgs a{};
    a.x = x; a.x_ldc = x_ldc; a.Cin = Cin; a.dy = dy; a.dy_ldc = dy_ldc; a.Cout = Cout; a.part = (float*)workspace;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Do = Do; a.Ho = Ho; a.Wo = Wo; a.sd = sd;
    a.CoPad = cdiv(Cout, 32) * 32; a.CiPad = cdiv(Cin, 32) * 32;
    a.splits = wgrad_splits(CONV_POINT, N, D, H, W, Cin, Cout);
    int rc = launch_wgrad_mfma(CONV_POINT, a, s);
    if (rc) return rc;
    return launch_wgrad_reduce(a.part, dw, a.splits, sd * 4, a.CiPad, a.CoPad, Cin, Cout, s);
}

// ---------------------------------------------------------------------------------------------- BN / ReLU / pool
int e3_bn_finalize(void* stream, const float* stats, int parts, int C, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps,
                   float* mean, float* invstd, float* scale, float* shift) {
    BnFinalizeArgs a{};
    a.stats = stats; a.parts = parts; a.C = C; a.gamma = gamma; a.beta = beta; a.running_mean = running_mean;
    a.running_var = running_var; a.momentum = momentum; a.eps = eps; a.mean = mean; a.invstd = invstd; a.scale = scale; a.shift = shift;
    return launch_bn_finalize(a, (hipStream_t)stream);
}

int e3_bn_relu_apply(void* stream, const float* x, int x_ldc, const float* scale, const float* shift, float* a, int a_ldc,
                     float* pooled, int kd, int N, int D, int H, int W, int C) {
    return launch_bn_relu_apply(x, x_ldc, scale, shift, a, a_ldc, pooled, kd, N, D, H, W, C, (hipStream_t)stream);
}

int e3_maxpool(void* stream, const float* a, int a_ldc, float* pooled, int kd, int N, int D, int H, int W, int C) {
    return launch_maxpool(a, a_ldc, pooled, kd, N, D, H, W, C, (hipStream_t)stream);
}

size_t e3_bn_bwd_workspace_bytes(int N, int D, int H, int W, int C) {
    const int parts = bn_bwd_parts((size_t)N * D * H * W, C);
    return align_up(((size_t)parts * 3 * C + 4 * C) * sizeof(float), 256);
}

int e3_bn_relu_bwd(void* stream, const float* x, int x_ldc, const float* mean, const float* invstd, const float* gamma,
                   const float* scale, const float* shift, const float* g1, int g1_ldc, const float* gpool,
                   const float* a, int a_ldc, const float* pooled, int kd, int N, int D, int H, int W, int C,
                   float* dx, int dx_ldc, float* dgamma, float* dbeta, float* dxsum, void* workspace, size_t workspace_bytes) {
    hipStream_t s = (hipStream_t)stream;
    E3_REQUIRE(workspace_bytes >= e3_bn_bwd_workspace_bytes(N, D, H, W, C), E3_ERR_WORKSPACE, "bn backward workspace too small");
    E3_REQUIRE(g1 || gpool, E3_ERR_INVALID, "bn backward needs a gradient source");
    E3_REQUIRE(!gpool || (a && pooled), E3_ERR_INVALID, "pooled gradient needs a and pooled");
    BnBwdArgs b{};
    b.x = x; b.x_ldc = x_ldc; b.mean = mean; b.invstd = invstd; b.gamma = gamma; b.scale = scale; b.shift = shift;
    b.g1 = g1; b.g1_ldc = g1_ldc; b.gpool = gpool; b.a = a; b.a_ldc = a_ldc; b.pooled = pooled; b.kd = kd;
    b.N = N; b.D = D; b.H = H; b.W = W; b.C = C;
    b.parts = bn_bwd_parts((size_t)N * D * H * W, C);
    b.part = (float*)workspace;
    float* coef = b.part + (size_t)b.parts * 3 * C;
    b.coef = coef; b.dx = dx; b.dx_ldc = dx_ldc;
    int rc = launch_bn_bwd_reduce(b, s);
    if (rc) return rc;
    rc = launch_bn_bwd_finalize(b.part, b.parts, C, 1.0f / (float)((double)N * D * H * W), dgamma, dbeta, coef, s);
    if (rc) return rc;
    rc = launch_bn_bwd_apply(b, s);
    if (rc) return rc;
    if (dxsum) rc = launch_colsum_finalize(b.part, b.parts, 3 * C, 2 * C, C, dxsum, s);
    return rc;
}

// ---------------------------------------------------------------------------------------------- criterion
size_t e3_ce_dice_workspace_bytes(int C) { return align_up(ce_dice_workspace_floats(C) * sizeof(float), 256); }

int e3_ce_dice_fwd(void* stream, const float* logits, const long long* target, const float* w, int C, int N, int D, int H, int W,
                   float ce_weight, float dice_weight, float eps, float smooth, void* workspace, size_t workspace_bytes, float* loss_out) {
    E3_REQUIRE(workspace_bytes >= e3_ce_dice_workspace_bytes(C), E3_ERR_WORKSPACE, "ce_dice workspace too small");
    return launch_ce_dice_fwd(logits, target, w, C, N, (size_t)D * H * W, ce_weight, dice_weight, eps, smooth, (float*)workspace, loss_out, (hipStream_t)stream);
}

int e3_ce_dice_sums(void* stream, const float* logits, const long long* target, const float* w, int C, int N, int D, int H, int W,
                    void* workspace, size_t workspace_bytes, double* sums) {
    E3_REQUIRE(workspace_bytes >= e3_ce_dice_workspace_bytes(C), E3_ERR_WORKSPACE, "ce_dice workspace too small");
    E3_REQUIRE(sums != nullptr, E3_ERR_INVALID, "ce_dice_sums: sums is NULL");
    return launch_ce_dice_sums(logits, target, w, C, N, (size_t)D * H * W, (float*)workspace, sums, (hipStream_t)stream);
}

int e3_ce_dice_from_sums(void* stream, const double* sums, const float* w, int C, float ce_weight, float dice_weight, float eps, float smooth,
                         void* workspace, size_t workspace_bytes, float* loss_out) {
    E3_REQUIRE(workspace_bytes >= e3_ce_dice_workspace_bytes(C), E3_ERR_WORKSPACE, "ce_dice workspace too small");
    E3_REQUIRE(sums != nullptr, E3_ERR_INVALID, "ce_dice_from_sums: sums is NULL");
    return launch_ce_dice_from_sums(sums, w, C, ce_weight, dice_weight, eps, smooth, (float*)workspace, loss_out, (hipStream_t)stream);
}

int e3_ce_dice_bwd(void* stream, const float* logits, const long long* target, const float* w, int C, int N, int D, int H, int W,
                   const void* workspace, size_t workspace_bytes, const float* gout, float* dlogits) {
    E3_REQUIRE(workspace_bytes >= e3_ce_dice_workspace_bytes(C), E3_ERR_WORKSPACE, "ce_dice workspace too small");
    return launch_ce_dice_bwd(logits, target, w, C, N, (size_t)D * H * W, (const float*)workspace, gout, dlogits, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- optimizer
int e3_swa_update(void* stream, int n_tensors, void* const* params, void* const* swa_buffers, const long long* numels, long long n_avg) {
    E3_REQUIRE(n_tensors >= 0 && n_avg >= 0 && (n_tensors == 0 || (params && swa_buffers && numels)), E3_ERR_INVALID, "swa_update: bad arguments");
    return launch_swa(n_tensors, params, swa_buffers, numels, 1.0 / (double)(n_avg + 1), 0, (hipStream_t)stream);
}

int e3_swa_swap(void* stream, int n_tensors, void* const* params, void* const* swa_buffers, const long long* numels) {
    E3_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (params && swa_buffers && numels)), E3_ERR_INVALID, "swa_swap: bad arguments");
    return launch_swa(n_tensors, params, swa_buffers, numels, 0.0, 1, (hipStream_t)stream);
}

size_t e3_adamw_state_floats(int n_tensors, const long long* numels) {
    return (n_tensors > 0 && numels) ? adamw_state_floats(n_tensors, numels) : 0;
}
size_t e3_adamw_state_offset(int n_tensors, const long long* numels, int tensor) {
    return (n_tensors > 0 && numels) ? adamw_state_offset(n_tensors, numels, tensor) : 0;
}

int e3_adamw_step(void* stream, int n_tensors, void* const* params, void* const* grads, const long long* numels,
                  float* exp_avg, float* exp_avg_sq, float* step, float* coef,
                  double lr, double beta1, double beta2, double eps, double weight_decay,
                  const float* grad_scale, const float* found_inf) {
    E3_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (params && grads && numels)), E3_ERR_INVALID, "null tensor table");
    E3_REQUIRE(exp_avg && exp_avg_sq && step && coef, E3_ERR_INVALID, "null optimizer state");
    E3_REQUIRE(lr >= 0.0 && eps >= 0.0 && weight_decay >= 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0,
               E3_ERR_INVALID, "invalid AdamW hyper-parameter");
    return launch_adamw(n_tensors, params, grads, numels, exp_avg, exp_avg_sq, step, coef, lr, beta1, beta2, eps, weight_decay,
                        grad_scale, found_inf, (hipStream_t)stream);
}

int e3_adamw_step_bf16(void* stream, int n_tensors, void* const* params, void* const* grads, const long long* numels,
                       float* exp_avg, float* exp_avg_sq, float* step, float* coef,
                       double lr, double beta1, double beta2, double eps, double weight_decay,
                       const float* grad_scale, const float* found_inf) {
    E3_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (params && grads && numels)), E3_ERR_INVALID, "null tensor table");
    E3_REQUIRE(exp_avg && exp_avg_sq && step && coef, E3_ERR_INVALID, "null optimizer state");
    E3_REQUIRE(lr >= 0.0 && eps >= 0.0 && weight_decay >= 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0,
               E3_ERR_INVALID, "invalid AdamW hyper-parameter");
    return launch_adamw(n_tensors, params, grads, numels, exp_avg, exp_avg_sq, step, coef, lr, beta1, beta2, eps, weight_decay,
                        grad_scale, found_inf, (hipStream_t)stream, 1);
}

// ---------------------------------------------------------------------------------------------- final 1x1x1 conv
int e3_conv1_fwd(void* stream, const float* a, int a_ldc, int C, const float* w, const float* bias, float* y_ncdhw,
                 int Cout, int N, int D, int H, int W, int softmax) {
    return launch_conv_final_fwd(a, a_ldc, C, w, bias, y_ncdhw, Cout, (size_t)D * H * W, N, softmax, (hipStream_t)stream);
}

size_t e3_conv1_bwd_workspace_bytes(int C, int Cout, int N, int D, int H, int W) {
    return align_up((size_t)conv_final_bwd_parts((size_t)N * D * H * W) * (Cout * C + Cout) * sizeof(float), 256);
}

int e3_conv1_bwd(void* stream, const float* a, int a_ldc, int C, const float* w, const float* dy_ncdhw, float* da, int da_ldc,
                 float* dw, float* db, int Cout, int N, int D, int H, int W, void* workspace, size_t workspace_bytes) {
    hipStream_t s = (hipStream_t)stream;
    E3_REQUIRE(workspace_bytes >= e3_conv1_bwd_workspace_bytes(C, Cout, N, D, H, W), E3_ERR_WORKSPACE, "conv1 backward workspace too small");
    const int parts = conv_final_bwd_parts((size_t)N * D * H * W);
    int rc = launch_conv_final_bwd(a, a_ldc, C, w, dy_ncdhw, da, da_ldc, (float*)workspace, Cout, (size_t)D * H * W, N, s);
    if (rc) return rc;
    const int ps = Cout * C + Cout;
    rc = launch_colsum_finalize((const float*)workspace, parts, ps, 0, Cout * C, dw, s);
    if (rc) return rc;
    return launch_colsum_finalize((const float*)workspace, parts, ps, Cout * C, Cout, db, s);
}

int e3_ncdhw_to_ndhwc(void* stream, const float* src, float* dst, int N, int C, int D, int H, int W) {
    return launch_ncdhw_to_ndhwc(src, dst, N, C, (size_t)D * H * W, (hipStream_t)stream);
}
int e3_ndhwc_to_ncdhw(void* stream, const float* src, int src_ldc, float* dst, int N, int C, int D, int H, int W) {
    return launch_ndhwc_to_ncdhw(src, src_ldc, dst, N, C, (size_t)D * H * W, (hipStream_t)stream);
}

}  // extern "C"
