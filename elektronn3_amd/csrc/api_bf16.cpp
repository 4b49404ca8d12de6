// Per-op entry points of the bf16 path (include/e3unet.h): argument checking + launches.  Host code only.
#include "../../include/e3unet.h"

#include "bf16.h"
#include "kernels.h"

extern "C" {

size_t e3_conv3d_workspace_bytes_bf16(int Cin, int Cout, int N, int D, int H, int W, int planar) {
    const int T = planar ? 9 : 27;
    if (Cin < 8) return align_up((size_t)conv_small_b16_wgrad_splits(N, D, H, W, planar) * T * Cout * Cin * 4 + 256, 256);      // first conv: only the wgrad slab
    const size_t pack = align_up(conv_b16_packed_elems(Cin, Cout, planar) * 2, 256);
    const size_t slab = Cin % 32 == 0 && Cout % 32 == 0 ? (size_t)wgrad_b16_splits(N, D, H, W, Cin, Cout, planar) * T * Cin * Cout * 4 : 0;
    // forward / dgrad: packed weights, then (low-resolution shapes) the split-K partial sums of either direction
    const size_t pf = conv_b16_partial_floats(N, D, H, W, Cin, Cout, planar), pd = conv_b16_partial_floats(N, D, H, W, Cout, Cin, planar);
    const size_t fwd = pack + align_up((pf > pd ? pf : pd) * 4, 256);
    return fwd > slab ? fwd : align_up(slab, 256);
}

int e3_conv3d_stats_parts_bf16(int Cin, int Cout, int N, int D, int H, int W, int planar) {
    return Cin < 8 ? conv_small_b16_stats_parts2(N, D, H, W, planar, Cin, Cout) : conv_b16_stats_parts(N, D, H, W, Cin, Cout, planar);
}

int e3_conv3d_fwd_bf16(void* stream, const void* x, int x_ldc, int Cin, const float* w, const float* bias, void* y, int y_ldc, int Cout,
                       int N, int D, int H, int W, int planar, const float* epi_scale, const float* epi_shift, float* stats,
                       void* workspace, size_t workspace_bytes) {
    hipStream_t s = (hipStream_t)stream;
    E3_REQUIRE(x && w && y && workspace, E3_ERR_INVALID, "null argument");
    E3_REQUIRE(workspace_bytes >= e3_conv3d_workspace_bytes_bf16(Cin, Cout, N, D, H, W, planar), E3_ERR_WORKSPACE, "conv3d bf16 workspace too small");
    if (Cin < 8)        // the network's first conv (x dense [voxel][Cin]): matrix-core kernel for one input channel, VALU kernel otherwise
        return launch_conv_small_b16_fwd((const bf16_t*)x, Cin, w, epi_scale ? nullptr : bias, (bf16_t*)y, y_ldc, N, D, H, W, Cout, planar, epi_scale, epi_shift, stats, s);
    int rc = launch_pack_conv_b16(w, (bf16_t*)workspace, Cout, Cin, planar, 0, s);
    if (rc) return rc;
    ConvB16Args a{};
    a.x = (const bf16_t*)x; a.x_ldc = x_ldc; a.Cin = Cin; a.wt = (const bf16_t*)workspace; a.bias = epi_scale ? nullptr : bias;
    a.y = (bf16_t*)y; a.y_ldc = y_ldc; a.N = N; a.D = D; a.H = H; a.W = W; a.Cout = Cout; a.planar = planar;
    a.epi_scale = epi_scale; a.epi_shift = epi_shift; a.stats = stats;
    a.partial = (float*)((char*)workspace + align_up(conv_b16_packed_elems(Cin, Cout, planar) * 2, 256));
    return launch_conv_b16(a, s);
}

int e3_conv3d_dgrad_bf16(void* stream, const void* dy, int dy_ldc, int Cout, const float* w, void* dx, int dx_ldc, int Cin,
                         int N, int D, int H, int W, int planar, void* workspace, size_t workspace_bytes) {
    hipStream_t s = (hipStream_t)stream;
    E3_REQUIRE(dy && w && dx && workspace, E3_ERR_INVALID, "null argument");
    E3_REQUIRE(workspace_bytes >= e3_conv3d_workspace_bytes_bf16(Cin, Cout, N, D, H, W, planar), E3_ERR_WORKSPACE, "conv3d bf16 workspace too small");
    int rc = launch_pack_conv_b16(w, (bf16_t*)workspace, Cout, Cin, planar, 1, s);
    if (rc) return rc;
    ConvB16Args a{};
    a.x = (const bf16_t*)dy; a.x_ldc = dy_ldc; a.Cin = Cout; a.wt = (const bf16_t*)workspace; a.bias = nullptr;
    a.y = (bf16_t*)dx; a.y_ldc = dx_ldc; a.N = N; a.D = D; a.H = H; a.W = W; a.Cout = Cin; a.planar = planar;
    a.partial = (float*)((char*)workspace + align_up(conv_b16_packed_elems(Cin, Cout, planar) * 2, 256));
    return launch_conv_b16(a, s);
}

int e3_conv3d_wgrad_bf16(void* stream, const void* x, int x_ldc, int Cin, const void* dy, int dy_ldc, int Cout, float* dw,
                         int N, int D, int H, int W, int planar, void* workspace, size_t workspace_bytes) {
    hipStream_t s = (hipStream_t)stream;
    E3_REQUIRE(x && dy && dw && workspace, E3_ERR_INVALID, "null argument");
    const int T = planar ? 9 : 27;
    if (Cin < 8) {
        const int splits = conv_small_b16_wgrad_splits(N, D, H, W, planar);
        E3_REQUIRE(workspace_bytes >= (size_t)splits * T * Cout * Cin * 4, E3_ERR_WORKSPACE, "wgrad bf16 workspace too small");
        int rc1 = launch_conv_small_b16_wgrad((const bf16_t*)x, Cin, (const bf16_t*)dy, dy_ldc, (float*)workspace, N, D, H, W, Cout, planar, s);
        if (rc1) return rc1;
        return launch_wgrad_reduce((float*)workspace, dw, splits, T, Cout, Cin, Cout, Cin, s);
    }
    E3_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, E3_ERR_UNSUPPORTED, "bf16 wgrad: channel counts must be multiples of 32");
    WgradB16Args a{};
    a.x = (const bf16_t*)x; a.x_ldc = x_ldc; a.Cin = Cin; a.dy = (const bf16_t*)dy; a.dy_ldc = dy_ldc; a.Cout = Cout; a.part = (float*)workspace;
    a.N = N; a.D = D; a.H = H; a.W = W; a.planar = planar;
    a.splits = wgrad_b16_splits(N, D, H, W, Cin, Cout, planar);
    E3_REQUIRE(workspace_bytes >= (size_t)a.splits * T * Cin * Cout * 4, E3_ERR_WORKSPACE, "wgrad bf16 workspace too small");
    int rc = launch_wgrad_b16(a, s);
    if (rc) return rc;
    return launch_wgrad_reduce(a.part, dw, a.splits, T, Cout, Cin, Cout, Cin, s);
}

size_t e3_convT_workspace_bytes_bf16(int Cin, int Cout, int N, int D, int H, int W) {
    const size_t pack = align_up(upconv_b16_packed_elems(Cin, Cout, 2) * 2, 256);
    const size_t slab = (size_t)upconv_b16_wgrad_splits(N, D, H, W) * 8 * Cin * Cout * 4;
    return pack > slab ? pack : align_up(slab, 256);
}
int e3_convT_stats_parts_bf16(int Cin, int N, int D, int H, int W) { return upconv_b16_stats_parts(N, D, H, W, 2, Cin); }

int e3_convT_fwd_bf16(void* stream, const void* x, int x_ldc, int Cin, const float* w, const float* bias, void* y, int y_ldc, int Cout,
                      int N, int D, int H, int W, int Do, int Ho, int Wo, float* stats, void* workspace, size_t workspace_bytes) {
    hipStream_t s = (hipStream_t)stream;
    E3_REQUIRE(x && w && y && workspace, E3_ERR_INVALID, "null argument");
    E3_REQUIRE(workspace_bytes >= upconv_b16_packed_elems(Cin, Cout, 2) * 2, E3_ERR_WORKSPACE, "convT bf16 workspace too small");
    int rc = launch_pack_upconv_b16(w, (bf16_t*)workspace, Cin, Cout, 2, 0, s);
    if (rc) return rc;
    UpconvB16Args a{};
    a.x = (const bf16_t*)x; a.x_ldc = x_ldc; a.Cin = Cin; a.y = (bf16_t*)y; a.y_ldc = y_ldc; a.Cout = Cout; a.wt = (const bf16_t*)workspace;
    a.bias = bias; a.N = N; a.D = D; a.H = H; a.W = W; a.Do = Do; a.Ho = Ho; a.Wo = Wo; a.sd = 2; a.stats = stats;
    return launch_upconv_b16_fwd(a, s);
}

int e3_convT_dgrad_bf16(void* stream, const void* dy, int dy_ldc, int Cout, const float* w, void* dx, int dx_ldc, int Cin,
                        int N, int D, int H, int W, int Do, int Ho, int Wo, void* workspace, size_t workspace_bytes) {
    hipStream_t s = (hipStream_t)stream;
    E3_REQUIRE(dy && w && dx && workspace, E3_ERR_INVALID, "null argument");
    E3_REQUIRE(workspace_bytes >= upconv_b16_packed_elems(Cin, Cout, 2) * 2, E3_ERR_WORKSPACE, "convT bf16 workspace too small");
    int rc = launch_pack_upconv_b16(w, (bf16_t*)workspace, Cin, Cout, 2, 1, s);
    if (rc) return rc;
    UpconvB16Args a{};
    a.x = (const bf16_t*)dx; a.x_ldc = dx_ldc; a.Cin = Cin; a.y = (bf16_t*)const_cast<void*>(dy); a.y_ldc = dy_ldc; a.Cout = Cout;
    a.wt = (const bf16_t*)workspace; a.N = N; a.D = D; a.H = H; a.W = W; a.Do = Do; a.Ho = Ho; a.Wo = Wo; a.sd = 2;
    return launch_upconv_b16_dgrad(a, s);
}

int e3_convT_wgrad_bf16(void* stream, const void* x, int x_ldc, int Cin, const void* dy, int dy_ldc, int Cout, float* dw,
                        int N, int D, int H, int W, int Do, int Ho, int Wo, void* workspace, size_t workspace_bytes) {
    hipStream_t s = (hipStream_t)stream;
    E3_REQUIRE(x && dy && dw && workspace, E3_ERR_INVALID, "null argument");
    const int splits = upconv_b16_wgrad_splits(N, D, H, W);
    E3_REQUIRE(workspace_bytes >= (size_t)splits * 8 * Cin * Cout * 4, E3_ERR_WORKSPACE, "convT wgrad bf16 workspace too small");
    int rc = launch_upconv_b16_wgrad((const bf16_t*)x, x_ldc, Cin, (const bf16_t*)dy, dy_ldc, Cout, (float*)workspace, N, D, H, W, Do, Ho, Wo, 2, splits, s);
    if (rc) return rc;
    return launch_wgrad_reduce((const float*)workspace, dw, splits, 8, Cin, Cout, Cin, Cout, s);
}

}  // extern "C"
