"""Thin torch-tensor wrappers over the per-op C ABI (include/e3unet.h).

Tensors here are fp32 NDHWC ("channels-last-3d") CUDA tensors of shape (N, D, H, W, C); a trailing-dim slice of a
wider buffer is a valid "view" (ldc = stride of the W axis).  These wrappers exist for unit parity tests and for
callers that want single ops; UNet.forward uses the whole-network entry points instead (one native call per pass).
"""
import torch

from . import _lib
from ._lib import c_size_t, check, ptr, stream_ptr


def _ldc(t):
    """Floats between consecutive voxels of an (N,D,H,W,C) NDHWC view (strides of size-1 dims are ignored)."""
    N, D, H, W, C = t.shape
    sN, sD, sH, sW, sC = t.stride()
    if C > 1 and sC != 1:
        raise ValueError('not an NDHWC view: channel stride must be 1')
    if W > 1:
        ldc = sW
    elif H > 1:
        ldc = sH
    elif D > 1:
        ldc = sD
    elif N > 1:
        ldc = sN
    else:
        ldc = C
    ok = ldc >= C and (H == 1 or W == 1 or sH == W * ldc) and (D == 1 or sD == H * W * ldc or (H == 1 and W == 1)) \
        and (N == 1 or sN == D * H * W * ldc or (D == 1 and H == 1 and W == 1))
    if not ok:
        raise ValueError(f'not an NDHWC view: shape {tuple(t.shape)} strides {t.stride()}')
    return ldc


def _chk(t, name):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise ValueError(f'{name}: expected a float32 CUDA tensor (the HIP path has no CPU fallback)')
    if t.dim() == 5:
        _ldc(t)
    return t


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def conv3d(x, w, bias=None, planar=False, pro=None, epi=None, want_stats=False, out=None):
    """3x3x3 (or 1x3x3) 'same' convolution. x: (N,D,H,W,Cin); w: torch layout (Cout,Cin,kd,3,3)."""
    L = _lib.load()
    _chk(x, 'x')
    N, D, H, W, Cin = x.shape
    Cout = w.shape[0]
    y = out if out is not None else torch.empty((N, D, H, W, Cout), device=x.device, dtype=torch.float32)
    ws = _ws(L.e3_conv3d_workspace_bytes(Cin, Cout, int(planar)), x.device)
    stats = None
    if want_stats:
        parts = L.e3_conv3d_stats_parts(Cin, Cout, N, D, H, W, int(planar))
        stats = torch.zeros((parts, Cout, 3), device=x.device, dtype=torch.float32)
    w = w.contiguous()
    check(L.e3_conv3d_fwd(stream_ptr(x.device), ptr(x), _ldc(x), Cin, ptr(w), ptr(bias), ptr(y), _ldc(y), Cout, N, D, H, W,
                          int(planar), ptr(pro[0]) if pro else None, ptr(pro[1]) if pro else None,
                          ptr(epi[0]) if epi else None, ptr(epi[1]) if epi else None, ptr(stats), ptr(ws), c_size_t(ws.numel())))
    return (y, stats) if want_stats else y


def conv3d_dgrad(dy, w, planar=False):
    L = _lib.load()
    _chk(dy, 'dy')
    N, D, H, W, Cout = dy.shape
    Cin = w.shape[1]
    dx = torch.empty((N, D, H, W, Cin), device=dy.device, dtype=torch.float32)
    ws = _ws(L.e3_conv3d_workspace_bytes(Cin, Cout, int(planar)), dy.device)
    w = w.contiguous()
    check(L.e3_conv3d_dgrad(stream_ptr(dy.device), ptr(dy), _ldc(dy), Cout, ptr(w), ptr(dx), Cin, Cin, N, D, H, W, int(planar),
                            ptr(ws), c_size_t(ws.numel())))
    return dx


def conv3d_wgrad(x, dy, planar=False):
    L = _lib.load()
    _chk(x, 'x'); _chk(dy, 'dy')
    N, D, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    kd = 1 if planar else 3
    dw = torch.empty((Cout, Cin, kd, 3, 3), device=x.device, dtype=torch.float32)
    ws = _ws(L.e3_conv3d_wgrad_workspace_bytes(Cin, Cout, N, D, H, W, int(planar)), x.device)
    check(L.e3_conv3d_wgrad(stream_ptr(x.device), ptr(x), _ldc(x), Cin, ptr(dy), _ldc(dy), Cout, ptr(dw), N, D, H, W, int(planar),
                            ptr(ws), c_size_t(ws.numel())))
    return dw


def convT(x, w, bias=None, out_dims=None, want_stats=False, out=None):
    """ConvTranspose3d(kernel=stride=(sd,2,2)). x: (N,D,H,W,Cin); w: (Cin,Cout,sd,2,2)."""
    L = _lib.load()
    _chk(x, 'x')
    N, D, H, W, Cin = x.shape
    Cout, sd = w.shape[1], w.shape[2]
    Do, Ho, Wo = out_dims if out_dims is not None else (sd * D, 2 * H, 2 * W)
    y = out if out is not None else torch.empty((N, Do, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    ws = _ws(L.e3_convT_workspace_bytes(Cin, Cout, sd), x.device)
    stats = None
    if want_stats:
        stats = torch.zeros((L.e3_convT_stats_parts(Cin, Cout, N, D, H, W, sd), Cout, 3), device=x.device, dtype=torch.float32)
    w = w.contiguous()
    check(L.e3_convT_fwd(stream_ptr(x.device), ptr(x), _ldc(x), Cin, ptr(w), ptr(bias), ptr(y), _ldc(y), Cout, N, D, H, W, sd,
                         Do, Ho, Wo, ptr(stats), ptr(ws), c_size_t(ws.numel())))
    return (y, stats) if want_stats else y


def convT_dgrad(dy, w, in_dims):
    L = _lib.load()
    _chk(dy, 'dy')
    N, Do, Ho, Wo, Cout = dy.shape
    Cin, sd = w.shape[0], w.shape[2]
    D, H, W = in_dims
    dx = torch.empty((N, D, H, W, Cin), device=dy.device, dtype=torch.float32)
    ws = _ws(L.e3_convT_workspace_bytes(Cin, Cout, sd), dy.device)
    w = w.contiguous()
    check(L.e3_convT_dgrad(stream_ptr(dy.device), ptr(dy), _ldc(dy), Cout, ptr(w), ptr(dx), Cin, Cin, N, D, H, W, sd, Do, Ho, Wo,
                           ptr(ws), c_size_t(ws.numel())))
    return dx


def convT_wgrad(x, dy, sd):
    L = _lib.load()
    _chk(x, 'x'); _chk(dy, 'dy')
    N, D, H, W, Cin = x.shape
    _, Do, Ho, Wo, Cout = dy.shape
    dw = torch.empty((Cin, Cout, sd, 2, 2), device=x.device, dtype=torch.float32)
    ws = _ws(L.e3_convT_wgrad_workspace_bytes(Cin, Cout, N, D, H, W, sd), x.device)
    check(L.e3_convT_wgrad(stream_ptr(x.device), ptr(x), _ldc(x), Cin, ptr(dy), _ldc(dy), Cout, ptr(dw), N, D, H, W, sd, Do, Ho, Wo,
                           ptr(ws), c_size_t(ws.numel())))
    return dw


def bn_finalize(stats, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    L = _lib.load()
    parts, C, _ = stats.shape
    mean, invstd, scale, shift = (torch.empty(C, device=stats.device, dtype=torch.float32) for _ in range(4))
    check(L.e3_bn_finalize(stream_ptr(stats.device), ptr(stats), parts, C, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                           float(momentum), float(eps), ptr(mean), ptr(invstd), ptr(scale), ptr(shift)))
    return mean, invstd, scale, shift


def bn_relu_apply(x, scale, shift, pool_kd=0, out=None):
    """a = relu(x*scale+shift); with pool_kd in (1,2) also returns MaxPool3d((kd,2,2), ceil_mode=True)(a)."""
    L = _lib.load()
    _chk(x, 'x')
    N, D, H, W, C = x.shape
    a = out if out is not None else torch.empty((N, D, H, W, C), device=x.device, dtype=torch.float32)
    pooled = None
    if pool_kd:
        pooled = torch.empty((N, -(-D // pool_kd), -(-H // 2), -(-W // 2), C), device=x.device, dtype=torch.float32)
    check(L.e3_bn_relu_apply(stream_ptr(x.device), ptr(x), _ldc(x), ptr(scale), ptr(shift), ptr(a), _ldc(a), ptr(pooled), pool_kd or 2,
                             N, D, H, W, C))
    return (a, pooled) if pool_kd else a


def maxpool(a, kd):
    L = _lib.load()
    _chk(a, 'a')
    N, D, H, W, C = a.shape
    pooled = torch.empty((N, -(-D // kd), -(-H // 2), -(-W // 2), C), device=a.device, dtype=torch.float32)
    check(L.e3_maxpool(stream_ptr(a.device), ptr(a), _ldc(a), ptr(pooled), kd, N, D, H, W, C))
    return pooled


def bn_relu_bwd(x, mean, invstd, gamma, scale, shift, g1=None, gpool=None, a=None, pooled=None, kd=2):
    """Returns dx, dgamma, dbeta, dxsum."""
    L = _lib.load()
    _chk(x, 'x')
    N, D, H, W, C = x.shape
    dx = torch.empty((N, D, H, W, C), device=x.device, dtype=torch.float32)
    dg, db, dxs = (torch.empty(C, device=x.device, dtype=torch.float32) for _ in range(3))
    ws = _ws(L.e3_bn_bwd_workspace_bytes(N, D, H, W, C), x.device)
    check(L.e3_bn_relu_bwd(stream_ptr(x.device), ptr(x), _ldc(x), ptr(mean), ptr(invstd), ptr(gamma), ptr(scale), ptr(shift),
                           ptr(g1), _ldc(g1) if g1 is not None else 0, ptr(gpool), ptr(a), _ldc(a) if a is not None else 0, ptr(pooled), kd,
                           N, D, H, W, C, ptr(dx), C, ptr(dg), ptr(db), ptr(dxs), ptr(ws), c_size_t(ws.numel())))
    return dx, dg, db, dxs


def conv1(a, w, bias, softmax=False):
    """Final 1x1x1 conv; returns NCDHW logits (or probabilities)."""
    L = _lib.load()
    _chk(a, 'a')
    N, D, H, W, C = a.shape
    Cout = w.shape[0]
    y = torch.empty((N, Cout, D, H, W), device=a.device, dtype=torch.float32)
    w = w.contiguous()
    check(L.e3_conv1_fwd(stream_ptr(a.device), ptr(a), _ldc(a), C, ptr(w), ptr(bias), ptr(y), Cout, N, D, H, W, int(softmax)))
    return y


def conv1_bwd(a, w, dy):
    L = _lib.load()
    _chk(a, 'a')
    N, D, H, W, C = a.shape
    Cout = w.shape[0]
    da = torch.empty((N, D, H, W, C), device=a.device, dtype=torch.float32)
    dw = torch.empty((Cout, C, 1, 1, 1), device=a.device, dtype=torch.float32)
    db = torch.empty(Cout, device=a.device, dtype=torch.float32)
    ws = _ws(L.e3_conv1_bwd_workspace_bytes(C, Cout, N, D, H, W), a.device)
    w = w.contiguous(); dy = dy.contiguous()
    check(L.e3_conv1_bwd(stream_ptr(a.device), ptr(a), _ldc(a), C, ptr(w), ptr(dy), ptr(da), C, ptr(dw), ptr(db), Cout, N, D, H, W,
                         ptr(ws), c_size_t(ws.numel())))
    return da, dw, db


def to_ndhwc(x_ncdhw):
    """(N,C,D,H,W) contiguous -> (N,D,H,W,C) contiguous, on device."""
    L = _lib.load()
    N, C, D, H, W = x_ncdhw.shape
    x_ncdhw = x_ncdhw.contiguous()
    out = torch.empty((N, D, H, W, C), device=x_ncdhw.device, dtype=torch.float32)
    check(L.e3_ncdhw_to_ndhwc(stream_ptr(x_ncdhw.device), ptr(x_ncdhw), ptr(out), N, C, D, H, W))
    return out


def to_ncdhw(x_ndhwc):
    L = _lib.load()
    _chk(x_ndhwc, 'x')
    N, D, H, W, C = x_ndhwc.shape
    out = torch.empty((N, C, D, H, W), device=x_ndhwc.device, dtype=torch.float32)
    check(L.e3_ndhwc_to_ncdhw(stream_ptr(x_ndhwc.device), ptr(x_ndhwc), _ldc(x_ndhwc), ptr(out), N, C, D, H, W))
    return out


# ---------------------------------------------------------------------------------------------- native bf16 ops
class _L16:
    """The per-op entry points of the 16-bit path for a tensor's element type: `.e3_conv3d_fwd_bf16` resolves to the `_f16` twin for float16."""

    def __init__(self, t):
        self._lib, self._f16 = _lib.load(), t.dtype == torch.float16

    def __getattr__(self, name):
        return getattr(self._lib, name.replace('_bf16', '_f16') if self._f16 else name)


def _chk16(t, name):
    if not (t.is_cuda and t.dtype in (torch.bfloat16, torch.float16)):
        raise ValueError(f'{name}: expected a bfloat16 / float16 CUDA tensor')
    if t.dim() == 5:
        _ldc(t)
    return t


def conv3d_bf16(x, w, bias=None, planar=False, epi=None, want_stats=False, out=None):
    """bf16 3x3x3 / 1x3x3 'same' conv on the bf16 matrix cores. x: (N,D,H,W,Cin) bf16; w: fp32 torch layout (rounded to bf16 when packed)."""
    L = _L16(x)
    _chk16(x, 'x')
    N, D, H, W, Cin = x.shape
    Cout = w.shape[0]
    y = out if out is not None else torch.empty((N, D, H, W, Cout), device=x.device, dtype=x.dtype)
    ws = _ws(L.e3_conv3d_workspace_bytes_bf16(Cin, Cout, N, D, H, W, int(planar)), x.device)
    stats = None
    if want_stats:
        stats = torch.zeros((L.e3_conv3d_stats_parts_bf16(Cin, Cout, N, D, H, W, int(planar)), Cout, 3), device=x.device, dtype=torch.float32)
    w = w.float().contiguous()
    check(L.e3_conv3d_fwd_bf16(stream_ptr(x.device), ptr(x), _ldc(x), Cin, ptr(w), ptr(bias), ptr(y), _ldc(y), Cout, N, D, H, W, int(planar),
                               ptr(epi[0]) if epi else None, ptr(epi[1]) if epi else None, ptr(stats), ptr(ws), c_size_t(ws.numel())))
    return (y, stats) if want_stats else y


def conv3d_dgrad_bf16(dy, w, planar=False):
    L = _L16(dy)
    _chk16(dy, 'dy')
    N, D, H, W, Cout = dy.shape
    Cin = w.shape[1]
    dx = torch.empty((N, D, H, W, Cin), device=dy.device, dtype=dy.dtype)
    ws = _ws(L.e3_conv3d_workspace_bytes_bf16(Cin, Cout, N, D, H, W, int(planar)), dy.device)
    w = w.float().contiguous()
    check(L.e3_conv3d_dgrad_bf16(stream_ptr(dy.device), ptr(dy), _ldc(dy), Cout, ptr(w), ptr(dx), Cin, Cin, N, D, H, W, int(planar),
                                 ptr(ws), c_size_t(ws.numel())))
    return dx


def conv3d_wgrad_bf16(x, dy, planar=False):
    L = _L16(x)
    _chk16(x, 'x'); _chk16(dy, 'dy')
    N, D, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    dw = torch.empty((Cout, Cin, 1 if planar else 3, 3, 3), device=x.device, dtype=torch.float32)
    ws = _ws(L.e3_conv3d_workspace_bytes_bf16(Cin, Cout, N, D, H, W, int(planar)), x.device)
    check(L.e3_conv3d_wgrad_bf16(stream_ptr(x.device), ptr(x), _ldc(x), Cin, ptr(dy), _ldc(dy), Cout, ptr(dw), N, D, H, W, int(planar),
                                 ptr(ws), c_size_t(ws.numel())))
    return dw


def convT_bf16(x, w, bias=None, out_dims=None, want_stats=False, out=None):
    """bf16 ConvTranspose3d(kernel = stride = 2). x: (N,D,H,W,Cin) bf16; w: (Cin,Cout,2,2,2) fp32."""
    L = _L16(x)
    _chk16(x, 'x')
    N, D, H, W, Cin = x.shape
    Cout = w.shape[1]
    Do, Ho, Wo = out_dims if out_dims is not None else (2 * D, 2 * H, 2 * W)
    y = out if out is not None else torch.empty((N, Do, Ho, Wo, Cout), device=x.device, dtype=x.dtype)
    ws = _ws(L.e3_convT_workspace_bytes_bf16(Cin, Cout, N, D, H, W), x.device)
    stats = torch.zeros((L.e3_convT_stats_parts_bf16(Cin, N, D, H, W), Cout, 3), device=x.device, dtype=torch.float32) if want_stats else None
    w = w.float().contiguous()
    check(L.e3_convT_fwd_bf16(stream_ptr(x.device), ptr(x), _ldc(x), Cin, ptr(w), ptr(bias), ptr(y), _ldc(y), Cout, N, D, H, W, Do, Ho, Wo,
                              ptr(stats), ptr(ws), c_size_t(ws.numel())))
    return (y, stats) if want_stats else y


def convT_dgrad_bf16(dy, w, in_dims):
    L = _L16(dy)
    _chk16(dy, 'dy')
    N, Do, Ho, Wo, Cout = dy.shape
    Cin = w.shape[0]
    D, H, W = in_dims
    dx = torch.empty((N, D, H, W, Cin), device=dy.device, dtype=dy.dtype)
    ws = _ws(L.e3_convT_workspace_bytes_bf16(Cin, Cout, N, D, H, W), dy.device)
    w = w.float().contiguous()
    check(L.e3_convT_dgrad_bf16(stream_ptr(dy.device), ptr(dy), _ldc(dy), Cout, ptr(w), ptr(dx), Cin, Cin, N, D, H, W, Do, Ho, Wo,
                                ptr(ws), c_size_t(ws.numel())))
    return dx


def convT_wgrad_bf16(x, dy):
    L = _L16(x)
    _chk16(x, 'x'); _chk16(dy, 'dy')
    N, D, H, W, Cin = x.shape
    _, Do, Ho, Wo, Cout = dy.shape
    dw = torch.empty((Cin, Cout, 2, 2, 2), device=x.device, dtype=torch.float32)
    ws = _ws(L.e3_convT_workspace_bytes_bf16(Cin, Cout, N, D, H, W), x.device)
    check(L.e3_convT_wgrad_bf16(stream_ptr(x.device), ptr(x), _ldc(x), Cin, ptr(dy), _ldc(dy), Cout, ptr(dw), N, D, H, W, Do, Ho, Wo,
                                ptr(ws), c_size_t(ws.numel())))
    return dw
