"""CPU oracle for elektronn3's 3D U-Net hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package ``elektronn3_amd`` never does (it fails loudly without its HIP
library instead of falling back to anything in here).

What is restated (reference file:line, all under /root/reference):

* ``OracleUNet.forward``   <- ``UNet.forward``      elektronn3/models/unet.py:894-916
* ``_down``                <- ``DownConv.forward``  elektronn3/models/unet.py:244-253
* ``_up``                  <- ``UpConv.forward``    elektronn3/models/unet.py:384-408
* ``autocrop``             <- ``autocrop``          elektronn3/models/unet.py:256-325
* ``tiled_apply``          <- ``tiled_apply``       elektronn3/inference/inference.py:45-199
* ``predict_tiled``        <- ``Predictor.predict`` elektronn3/inference/inference.py:569-687 (softmax, shape padding)

The arithmetic itself (conv / convT / BN / ReLU / max-pool) lives in torch (third-party,
``torch>=1.6.0``, requirements.txt:1); it is restated in plain C in ``e3_oracle.c`` and called
through ctypes.  Pinned by tests/test_oracle_golden.py against tests/golden/*.npz, which were
generated from the imported reference by tests/golden/make_golden.py.
"""
import ctypes
import itertools
import os
import subprocess
from collections import OrderedDict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_fp = ctypes.POINTER(ctypes.c_float)
c_i64p = ctypes.POINTER(ctypes.c_int64)


def build():
    """Compile oracle/libe3oracle.so with gcc (see oracle/Makefile)."""
    subprocess.run(['make', '-C', _HERE, '-s'], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'libe3oracle.so')
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags['C_CONTIGUOUS'], (a.dtype, a.flags)
    return a.ctypes.data_as(c_fp)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ----------------------------------------------------------------------------- ops
def conv3d_fwd(x, w, b, pad):
    x, w = _f32(x), _f32(w)
    N, Cin, D, H, W = x.shape
    Cout, _, kd, kh, kw = w.shape
    y = np.empty((N, Cout, D + 2 * pad[0] - kd + 1, H + 2 * pad[1] - kh + 1, W + 2 * pad[2] - kw + 1), np.float32)
    lib().orc_conv3d_fwd(_p(x), _p(w), _p(_f32(b)) if b is not None else None, _p(y),
                         N, Cin, D, H, W, Cout, kd, kh, kw, *pad)
    return y


def conv3d_bwd(x, w, dy, pad, need_dx=True):
    x, w, dy = _f32(x), _f32(w), _f32(dy)
    N, Cin, D, H, W = x.shape
    Cout, _, kd, kh, kw = w.shape
    dx = None
    if need_dx:
        dx = np.empty_like(x)
        lib().orc_conv3d_bwd_data(_p(dy), _p(w), _p(dx), N, Cin, D, H, W, Cout, kd, kh, kw, *pad)
    dw = np.empty_like(w)
    db = np.empty((Cout,), np.float32)
    lib().orc_conv3d_bwd_weight(_p(x), _p(dy), _p(dw), _p(db), N, Cin, D, H, W, Cout, kd, kh, kw, *pad)
    return dx, dw, db


def convT_fwd(x, w, b):
    x, w = _f32(x), _f32(w)
    N, Cin, D, H, W = x.shape
    _, Cout, sd, sh, sw = w.shape
    y = np.empty((N, Cout, D * sd, H * sh, W * sw), np.float32)
    lib().orc_convT_fwd(_p(x), _p(w), _p(_f32(b)) if b is not None else None, _p(y), N, Cin, D, H, W, Cout, sd, sh, sw)
    return y


def convT_bwd(x, w, dy):
    x, w, dy = _f32(x), _f32(w), _f32(dy)
    N, Cin, D, H, W = x.shape
    _, Cout, sd, sh, sw = w.shape
    dx = np.empty_like(x)
    dw = np.empty_like(w)
    db = np.empty((Cout,), np.float32)
    lib().orc_convT_bwd_data(_p(dy), _p(w), _p(dx), N, Cin, D, H, W, Cout, sd, sh, sw)
    lib().orc_convT_bwd_weight(_p(x), _p(dy), _p(dw), _p(db), N, Cin, D, H, W, Cout, sd, sh, sw)
    return dx, dw, db


def bn_train_fwd(x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5):
    """Returns y, save_mean, save_invstd; updates running_* in place (may be None)."""
    x = _f32(x)
    N, C = x.shape[:2]
    S = int(np.prod(x.shape[2:]))
    y = np.empty_like(x)
    mean = np.empty((C,), np.float32)
    invstd = np.empty((C,), np.float32)
    lib().orc_bn_train_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(invstd),
                           _p(running_mean), _p(running_var),
                           ctypes.c_double(momentum), ctypes.c_double(eps), N, C, ctypes.c_size_t(S))
    return y, mean, invstd


def bn_eval_fwd(x, gamma, beta, running_mean, running_var, eps=1e-5):
    x = _f32(x)
    N, C = x.shape[:2]
    S = int(np.prod(x.shape[2:]))
    y = np.empty_like(x)
    lib().orc_bn_eval_fwd(_p(x), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(y),
                          ctypes.c_double(eps), N, C, ctypes.c_size_t(S))
    return y


def bn_train_bwd(dy, x, gamma, mean, invstd):
    dy, x = _f32(dy), _f32(x)
    N, C = x.shape[:2]
    S = int(np.prod(x.shape[2:]))
    dx = np.empty_like(x)
    dg = np.empty((C,), np.float32)
    db = np.empty((C,), np.float32)
    lib().orc_bn_train_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(invstd), _p(dx), _p(dg), _p(db),
                           N, C, ctypes.c_size_t(S))
    return dx, dg, db


def relu_fwd(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().orc_relu_fwd(_p(x), _p(y), ctypes.c_size_t(x.size))
    return y


def relu_bwd(dy, out):
    dy, out = _f32(dy), _f32(out)
    dx = np.empty_like(dy)
    lib().orc_relu_bwd(_p(dy), _p(out), _p(dx), ctypes.c_size_t(dy.size))
    return dx


def maxpool_fwd(x, k):
    x = _f32(x)
    N, C, D, H, W = x.shape
    Do, Ho, Wo = -(-D // k[0]), -(-H // k[1]), -(-W // k[2])
    y = np.empty((N, C, Do, Ho, Wo), np.float32)
    idx = np.empty((N, C, Do, Ho, Wo), np.int64)
    lib().orc_maxpool_fwd(_p(x), _p(y), idx.ctypes.data_as(c_i64p), N, C, D, H, W, *k)
    return y, idx


def maxpool_bwd(dy, idx, in_shape, k):
    dy = _f32(dy)
    N, C, D, H, W = in_shape
    dx = np.empty(in_shape, np.float32)
    lib().orc_maxpool_bwd(_p(dy), idx.ctypes.data_as(c_i64p), _p(dx), N, C, D, H, W, *k)
    return dx


def softmax_c(x):
    x = _f32(x)
    N, C = x.shape[:2]
    S = int(np.prod(x.shape[2:]))
    y = np.empty_like(x)
    lib().orc_softmax_c(_p(x), _p(y), N, C, ctypes.c_size_t(S))
    return y


def linear_upsample_matrix(n, scale):
    """(scale*n, n) matrix of nn.Upsample(scale_factor=scale, mode='linear'-family, align_corners=False) along one axis: source
    coordinate max((o + 0.5)/scale - 0.5, 0), taps floor / min(floor + 1, n - 1) with weights (1 - l, l).  scale 1 = identity."""
    M = np.zeros((scale * n, n), np.float64)
    for o in range(scale * n):
        src = max((o + 0.5) / scale - 0.5, 0.0)
        i0 = int(np.floor(src)); i1 = min(i0 + 1, n - 1); l = src - i0
        M[o, i0] += 1.0 - l; M[o, i1] += l
    return M


def group_norm_fwd(x, gamma, beta, num_groups, eps=1e-5):
    """nn.GroupNorm(num_groups, C) (get_normalization, unet.py:81-90): per sample, mean / biased variance over the C/G channels of a
    group and all voxels, then the per-channel affine map.  float64 arithmetic.  Returns (y, xhat, invstd[N,G])."""
    x = np.asarray(x, np.float64)
    N, C = x.shape[:2]
    xg = x.reshape(N, num_groups, -1)
    mean = xg.mean(-1, keepdims=True)
    var = xg.var(-1, keepdims=True)
    invstd = 1.0 / np.sqrt(var + eps)
    xhat = ((xg - mean) * invstd).reshape(x.shape)
    sh = (1, C) + (1,) * (x.ndim - 2)
    y = xhat * np.asarray(gamma, np.float64).reshape(sh) + np.asarray(beta, np.float64).reshape(sh)
    return y.astype(np.float32), xhat, invstd[..., 0]


def group_norm_bwd(dy, xhat, invstd, gamma, num_groups):
    """Backward of group_norm_fwd: dgamma_c = sum dy*xhat, dbeta_c = sum dy,
    dx = invstd_g * (gamma dy - mean_g(gamma dy) - xhat * mean_g(gamma dy xhat))."""
    dy = np.asarray(dy, np.float64)
    N, C = dy.shape[:2]
    red = (0,) + tuple(range(2, dy.ndim))
    dgamma, dbeta = (dy * xhat).sum(red), dy.sum(red)
    sh = (1, C) + (1,) * (dy.ndim - 2)
    gdy = dy * np.asarray(gamma, np.float64).reshape(sh)
    gg = gdy.reshape(N, num_groups, -1); xg = xhat.reshape(N, num_groups, -1)
    m1 = gg.mean(-1, keepdims=True); m2 = (gg * xg).mean(-1, keepdims=True)
    dx = (invstd[..., None] * (gg - m1 - xg * m2)).reshape(dy.shape)
    return dx.astype(np.float32), dgamma.astype(np.float32), dbeta.astype(np.float32)


def adamw_step(p, g, m, v, t, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
    """In-place torch.optim.AdamW step number t (1-based) on fp32 arrays p, m, v (examples/train_unet_neurodata.py:257-262)."""
    for a in (p, m, v):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    g = _f32(g)
    lib().orc_adamw_step(_p(p), _p(g), _p(m), _p(v), ctypes.c_size_t(p.size), int(t), ctypes.c_double(lr), ctypes.c_double(betas[0]),
                         ctypes.c_double(betas[1]), ctypes.c_double(eps), ctypes.c_double(weight_decay))


def ce_dice_sums(logits, target, w=None):
    """The 2 + 3C batch sums the example's criterion is a function of (modules/loss.py:158-189 dice_loss: intersection / denominator
    summed over batch and space; F.cross_entropy(weight): sum of w[t] * nll over sum of w[t]), in fp64:
    [0] sum w[t](-log p[t]), [1] sum w[t], [2+c] sum p_c[t=c], [2+C+c] sum p_c, [2+2C+c] sum [t=c].  logits (N,C,*sp), target (N,*sp)."""
    z = np.asarray(logits, np.float64)
    t = np.asarray(target)
    C = z.shape[1]
    w = np.ones(C) if w is None else np.asarray(w, np.float64)
    m = z.max(1, keepdims=True)
    lse = m + np.log(np.exp(z - m).sum(1, keepdims=True))
    p = np.exp(z - lse)
    onehot = (t[:, None] == np.arange(C).reshape((1, C) + (1,) * (z.ndim - 2)))
    ax = (0,) + tuple(range(2, z.ndim))
    s = np.zeros(2 + 3 * C)
    s[0] = (w[t] * -(np.log(p) * onehot).sum(1)).sum()
    s[1] = w[t].sum()
    s[2:2 + C] = (p * onehot).sum(ax)
    s[2 + C:2 + 2 * C] = p.sum(ax)
    s[2 + 2 * C:] = onehot.sum(ax)
    return s


def ce_dice_from_sums(s, w=None, ce_weight=0.5, dice_weight=0.5, eps=1e-4, smooth=0.0):
    """CombinedLoss([CrossEntropyLoss(w), DiceLoss(softmax, w)], (a, b)) (modules/loss.py:19-49,192-234) from the batch sums."""
    C = (len(s) - 2) // 3
    w = np.ones(C) if w is None else np.asarray(w, np.float64)
    ce = s[0] / s[1] if s[1] > 0 else 0.0
    num = 2 * s[2:2 + C] + smooth
    den = s[2 + C:2 + 2 * C] + s[2 + 2 * C:] + smooth + eps
    return ce_weight * ce + dice_weight * float((w * (1 - num / den)).mean())


def swa_update(buf, p, n_avg):
    """One running-average update of the reference's SWA wrapper (training/swa.py:169-175): buf + (p - buf) * (1 / (n_avg + 1)), each
    operation rounded to fp32 (the scalar is a Python float that ATen casts to fp32 for an fp32 tensor)."""
    buf, p = _f32(buf), _f32(p)
    return (buf + ((p - buf) * np.float32(1.0 / float(n_avg + 1))).astype(np.float32)).astype(np.float32)


def autocrop(from_down, from_up):
    """unet.py:256-325 -- crop decoder output by 1 where (u-d) is odd, centre-crop encoder output."""
    if from_down.shape[2:] == from_up.shape[2:]:
        return from_down, from_up, None, None
    ds, us = from_down.shape[2:], from_up.shape[2:]
    upcrop = [u - ((u - d) % 2) for d, u in zip(ds, us)]
    up_sl = (slice(None), slice(None)) + tuple(slice(0, c) for c in upcrop)
    from_up = from_up[up_sl]
    us = from_up.shape[2:]
    assert all(d >= u for d, u in zip(ds, us)), (ds, us)
    dn_sl = (slice(None), slice(None)) + tuple(slice((d - u) // 2, (d + u) // 2) for d, u in zip(ds, us))
    from_down = from_down[dn_sl]
    return from_down, from_up, dn_sl, up_sl


# ----------------------------------------------------------------------------- network
class OracleUNet:
    """Functional numpy/C restatement of ``UNet(...)`` for the configurations the HIP path supports:
    dim=3, up_mode='transpose', merge_mode='concat', activation='relu', normalization in
    {'batch', 'none'}, full_norm=True, conv_mode='same', attention=False, any planar_blocks.

    ``sd`` is a state_dict with the reference's key names (unet.py:832-881); values are numpy arrays.
    BN running statistics inside ``sd`` are updated in place by a training-mode forward.
    """

    def __init__(self, sd, n_blocks, planar_blocks=(), normalization='batch', momentum=0.1, eps=1e-5):
        self.sd = OrderedDict((k, (np.array(v, dtype=np.float32, copy=True) if np.asarray(v).dtype != np.int64
                                   else np.array(v, copy=True))) for k, v in sd.items())
        self.n_blocks = n_blocks
        self.planar = tuple(planar_blocks)
        self.norm = normalization
        self.num_groups = 8 if normalization == 'group' else (int(normalization[5:]) if normalization.startswith('group') else 0)
        self.valid = False            # conv_mode='valid' (set by the caller)
        self.up_linear = False        # up_mode='resizeconv_linear' (set by the caller)
        self.act_slope = 0.0          # 0 ReLU, 0.1 'leaky', 1.0 'lin' (set by the caller)
        self.instance_norms = ()      # names of the nn.InstanceNorm layers (they have no state_dict entries), set by the caller
        self.momentum, self.eps = momentum, eps
        self.training = True

    # -- helpers
    @staticmethod
    def _act_name(norm_name):
        """Activation module that follows a norm slot: DownConv norm0/norm1 -> act1/act2 (unet.py:244-253), UpConv norm0/1/2 -> act0/1/2
        (unet.py:396-407)."""
        block, norm = norm_name.rsplit('.', 1)
        k = int(norm[-1])
        return f'{block}.act{k + 1 if block.startswith("down_convs") else k}'

    def _conv(self, name, x, cache):
        w, b = self.sd[name + '.weight'], self.sd[name + '.bias']
        pad = tuple((k - 1) // 2 for k in w.shape[2:])
        if self.valid and (name.endswith('.conv1') or name.endswith('.conv2')):
            pad = (0, 0, 0)                     # conv_mode='valid': padding 0 in the convs of the blocks (unet.py:217,347)
        cache[name] = (x, pad)
        return conv3d_fwd(x, w, b, pad)

    def _norm_act(self, name, x, cache):
        # a norm layer that is nn.Identity (normalization='none', or full_norm=False: unet.py:238-242,369-375) has no entries in the state_dict
        if self.norm == 'instance' and name in self.instance_norms:
            # nn.InstanceNorm3d(C): affine=False, no running statistics, biased variance over (D,H,W) of each sample -- also in eval mode
            # (get_normalization, unet.py:91-97) == the train-mode batch norm of every sample on its own with gamma = 1, beta = 0
            C = x.shape[1]
            ones, zeros = np.ones(C, np.float32), np.zeros(C, np.float32)
            ys, stats = [], []
            for n in range(x.shape[0]):
                yn, mean, invstd = bn_train_fwd(x[n:n + 1], ones, zeros, zeros.copy(), ones.copy(), self.momentum, self.eps)
                ys.append(yn); stats.append((mean, invstd))
            y = np.concatenate(ys, 0)
            cache[name] = (x, stats)
        elif self.norm.startswith('group') and (name + '.weight') in self.sd:
            y, xhat, invstd = group_norm_fwd(x, self.sd[name + '.weight'], self.sd[name + '.bias'], self.num_groups, self.eps)
            cache[name] = (xhat, invstd)
        elif self.norm == 'batch' and (name + '.weight') in self.sd:
            g, b = self.sd[name + '.weight'], self.sd[name + '.bias']
            rm, rv = self.sd[name + '.running_mean'], self.sd[name + '.running_var']
            if self.training:
                y, mean, invstd = bn_train_fwd(x, g, b, rm, rv, self.momentum, self.eps)
                self.sd[name + '.num_batches_tracked'] = self.sd[name + '.num_batches_tracked'] + 1
                cache[name] = (x, mean, invstd)
            else:
                y = bn_eval_fwd(x, g, b, rm, rv, self.eps)
        else:
            y = x
        # get_activation (unet.py:183-199): 'relu', or LeakyReLU(0.1) ('leaky') / identity ('lin') via act_slope
        aname = self._act_name(name)
        if (aname + '.weight') in self.sd:      # nn.PReLU(num_parameters=1): max(y, 0) + w * min(y, 0) with this module's learnable w
            w = np.float32(self.sd[aname + '.weight'].reshape(-1)[0])
            a = np.where(y > 0, y, w * y).astype(np.float32)
            cache[name + '.pre'] = y
        elif self.act_slope == 2.0:      # nn.SiLU: y * sigmoid(y); the backward needs the pre-activation
            y64 = np.asarray(y, np.float64)
            a = (y64 / (1.0 + np.exp(-y64))).astype(np.float32)
            cache[name + '.pre'] = y64
        else:
            a = relu_fwd(y) if self.act_slope == 0.0 else np.where(y > 0, y, np.float32(self.act_slope) * y).astype(np.float32)
        cache[name + '.act'] = a
        return a

    def forward(self, x):
        """UNet.forward (unet.py:894-916). Returns logits (N, Cout, D, H, W)."""
        cache = {}
        x = _f32(x)
        enc = []
        for i in range(self.n_blocks):  # DownConv.forward, unet.py:244-253
            p = f'down_convs.{i}.'
            y = self._conv(p + 'conv1', x, cache)
            y = self._norm_act(p + 'norm0', y, cache)
            y = self._conv(p + 'conv2', y, cache)
            y = self._norm_act(p + 'norm1', y, cache)
            enc.append(y)
            if i < self.n_blocks - 1:
                k = (1, 2, 2) if i in self.planar else (2, 2, 2)
                x, idx = maxpool_fwd(y, k)
                cache[p + 'pool'] = (idx, y.shape, k)
            else:
                x = y
        for i in range(self.n_blocks - 1):  # UpConv.forward, unet.py:384-408
            p = f'up_convs.{i}.'
            before_pool = enc[-(i + 2)]
            if (p + 'upconv.conv.weight') in self.sd:
                # up_mode='resizeconv_nearest' (ResizeConv, unet.py:411-449): nn.Upsample(scale_factor, 'nearest') = every voxel
                # repeated (sd, 2, 2) times, then a 'same' 3x3x3 / 1x3x3 convolution on the up-sampled grid
                w, b = self.sd[p + 'upconv.conv.weight'], self.sd[p + 'upconv.conv.bias']
                sd_ = 1 if (self.n_blocks - 2 - i) in self.planar else 2      # planar decoder block: only (H, W) grow
                if self.up_linear:     # 'resizeconv_linear': separable tri-linear interpolation, align_corners=False
                    Ms = [linear_upsample_matrix(x.shape[2], sd_), linear_upsample_matrix(x.shape[3], 2), linear_upsample_matrix(x.shape[4], 2)]
                    xu = np.einsum('ad,bh,cw,nkdhw->nkabc', *Ms, x.astype(np.float64)).astype(np.float32)
                    xu = np.ascontiguousarray(xu)
                else:
                    xu = np.ascontiguousarray(x.repeat(sd_, axis=2).repeat(2, axis=3).repeat(2, axis=4))
                pad = tuple((k - 1) // 2 for k in w.shape[2:])
                cache[p + 'upconv'] = (xu, pad, sd_)
                up = conv3d_fwd(xu, w, b, pad)
            else:
                w, b = self.sd[p + 'upconv.weight'], self.sd[p + 'upconv.bias']
                cache[p + 'upconv'] = x
                up = convT_fwd(x, w, b)
            cache[p + 'up_full_shape'] = up.shape
            before_pool, up, dn_sl, up_sl = autocrop(before_pool, up)
            cache[p + 'crop'] = (dn_sl, up_sl, enc[-(i + 2)].shape, None)
            up = self._norm_act(p + 'norm0', np.ascontiguousarray(up), cache)
            # merge_mode (unet.py:398-401): 'concat' -> conv1 has 2C input channels, 'add' -> C
            if self.sd[p + 'conv1.weight'].shape[1] == 2 * up.shape[1]:
                mrg = np.concatenate((up, before_pool), axis=1)
            else:
                mrg = up + before_pool
            y = self._conv(p + 'conv1', mrg, cache)
            y = self._norm_act(p + 'norm1', y, cache)
            y = self._conv(p + 'conv2', y, cache)
            x = self._norm_act(p + 'norm2', y, cache)
        out = self._conv('conv_final', x, cache)
        self.cache = cache
        return out

    # -- backward
    def _norm_act_bwd(self, name, da, cache, grads):
        a = cache[name + '.act']
        aname = self._act_name(name)
        if (aname + '.weight') in self.sd:
            y = cache[name + '.pre']; w = np.float32(self.sd[aname + '.weight'].reshape(-1)[0])
            da64 = np.asarray(da, np.float64)
            grads[aname + '.weight'] = np.array([(da64 * np.minimum(np.asarray(y, np.float64), 0.0)).sum()], np.float32)
            dy = (_f32(da) * np.where(y > 0, np.float32(1), w)).astype(np.float32)
        elif self.act_slope == 2.0:
            z = cache[name + '.pre']; sg = 1.0 / (1.0 + np.exp(-z))
            dy = (np.asarray(da, np.float64) * sg * (1.0 + z * (1.0 - sg))).astype(np.float32)
        elif self.act_slope == 0.0:
            dy = relu_bwd(da, a)
        else:   # sign(a) == sign(pre-activation) for slope > 0
            dy = (_f32(da) * np.where(a > 0, np.float32(1), np.float32(self.act_slope))).astype(np.float32)
        if self.norm == 'instance' and name in self.instance_norms:
            x, stats = cache[name]
            ones = np.ones(x.shape[1], np.float32)
            return np.concatenate([bn_train_bwd(dy[n:n + 1], x[n:n + 1], ones, *stats[n])[0] for n in range(x.shape[0])], 0)
        if self.norm.startswith('group') and (name + '.weight') in self.sd:
            xhat, invstd = cache[name]
            dx, dg, db = group_norm_bwd(dy, xhat, invstd, self.sd[name + '.weight'], self.num_groups)
            grads[name + '.weight'], grads[name + '.bias'] = dg, db
            return dx
        if self.norm == 'batch' and (name + '.weight') in self.sd:
            x, mean, invstd = cache[name]
            dx, dg, db = bn_train_bwd(dy, x, self.sd[name + '.weight'], mean, invstd)
            grads[name + '.weight'], grads[name + '.bias'] = dg, db
            return dx
        return dy

    def _conv_bwd(self, name, dy, cache, grads, need_dx=True):
        x, pad = cache[name]
        dx, dw, db = conv3d_bwd(x, self.sd[name + '.weight'], dy, pad, need_dx)
        grads[name + '.weight'], grads[name + '.bias'] = dw, db
        return dx

    def backward(self, dout, need_dx=False):
        """Autograd twin of forward. Returns (grads: name->array, dx or None)."""
        assert self.training
        cache, grads = self.cache, OrderedDict()
        d = self._conv_bwd('conv_final', _f32(dout), cache, grads)
        enc_grads = [None] * self.n_blocks
        for i in reversed(range(self.n_blocks - 1)):
            p = f'up_convs.{i}.'
            d = self._norm_act_bwd(p + 'norm2', d, cache, grads)
            d = self._conv_bwd(p + 'conv2', d, cache, grads)
            d = self._norm_act_bwd(p + 'norm1', d, cache, grads)
            dmrg = self._conv_bwd(p + 'conv1', d, cache, grads)
            if self.sd[p + 'conv1.weight'].shape[1] == 2 * self.sd[p + 'conv1.weight'].shape[0]:
                C = dmrg.shape[1] // 2
                dup, dskip = np.ascontiguousarray(dmrg[:, :C]), np.ascontiguousarray(dmrg[:, C:])
            else:       # 'add': the gradient of the sum goes to both operands
                dup, dskip = dmrg, dmrg.copy()
            dn_sl, up_sl, enc_shape, _ = cache[p + 'crop']
            j = self.n_blocks - 2 - i  # encoder block whose before_pool was merged: encoder_outs[-(i+2)]
            if dn_sl is not None:
                full = np.zeros(enc_shape, np.float32)
                full[dn_sl] = dskip
                dskip = full
            enc_grads[j] = dskip
            dup = self._norm_act_bwd(p + 'norm0', dup, cache, grads)
            if up_sl is not None:
                full = np.zeros(cache[p + 'up_full_shape'], np.float32)
                full[up_sl] = dup
                dup = full
            if (p + 'upconv.conv.weight') in self.sd:
                xu, pad, sd_ = cache[p + 'upconv']
                dxu, dw, db = conv3d_bwd(xu, self.sd[p + 'upconv.conv.weight'], np.ascontiguousarray(dup), pad, True)
                grads[p + 'upconv.conv.weight'], grads[p + 'upconv.conv.bias'] = dw, db
                N_, C_, D_, H_, W_ = dxu.shape     # backward of the nearest up-sampling: sum over each (sd, 2, 2) block
                if self.up_linear:     # transpose of the interpolation
                    Ms = [linear_upsample_matrix(D_ // sd_, sd_), linear_upsample_matrix(H_ // 2, 2), linear_upsample_matrix(W_ // 2, 2)]
                    d = np.einsum('ad,bh,cw,nkabc->nkdhw', *Ms, dxu.astype(np.float64)).astype(np.float32)
                else:
                    d = dxu.astype(np.float64).reshape(N_, C_, D_ // sd_, sd_, H_ // 2, 2, W_ // 2, 2).sum(axis=(3, 5, 7)).astype(np.float32)
            else:
                xin = cache[p + 'upconv']
                d, dw, db = convT_bwd(xin, self.sd[p + 'upconv.weight'], dup)
                grads[p + 'upconv.weight'], grads[p + 'upconv.bias'] = dw, db
        for i in reversed(range(self.n_blocks)):
            p = f'down_convs.{i}.'
            if i < self.n_blocks - 1:
                idx, shp, k = cache[p + 'pool']
                d = maxpool_bwd(d, idx, shp, k)
                d = d + enc_grads[i]
            d = self._norm_act_bwd(p + 'norm1', d, cache, grads)
            d = self._conv_bwd(p + 'conv2', d, cache, grads)
            d = self._norm_act_bwd(p + 'norm0', d, cache, grads)
            d = self._conv_bwd(p + 'conv1', d, cache, grads, need_dx=(i > 0 or need_dx))
        return grads, d


# ----------------------------------------------------------------------------- tiled inference
def tile_plan(out_spatial, tile_shape, overlap_shape):
    """Tile visiting order and coordinates of ``tiled_apply`` (inference.py:153-189): C-order
    ``itertools.product`` over tile indices; input slab = [tile*pos, tile*(pos+1) + 2*overlap) in
    padded coordinates; output slab = [tile*pos, tile*(pos+1))."""
    tile_shape, overlap_shape = np.asarray(tile_shape), np.asarray(overlap_shape)
    tiles = np.ceil(np.asarray(out_spatial) / tile_shape).astype(int)
    plan = []
    for pos in itertools.product(*[range(t) for t in tiles]):
        pos = np.array(pos)
        lo, hi = tile_shape * pos, tile_shape * (pos + 1)
        plan.append((tuple(lo), tuple(hi + 2 * overlap_shape), tuple(lo), tuple(hi)))
    return plan


def tiled_apply(func, inp, tile_shape, overlap_shape, out_shape):
    """inference.py:45-199 for offset=None ('same' networks): zero-pad by overlap, run ``func`` per
    tile, keep the central tile_shape region of each result."""
    inp = _f32(inp)
    tile_shape, overlap_shape = np.asarray(tile_shape), np.asarray(overlap_shape)
    if not np.all(np.mod(out_shape[2:], tile_shape) == 0):
        raise ValueError(f'spatial out shape[2:] {tuple(out_shape[2:])} has to be divisible by tile_shape {tile_shape}.')
    padded = np.zeros(inp.shape[:2] + tuple(np.asarray(inp.shape[2:]) + 2 * overlap_shape), np.float32)
    padded[(slice(None), slice(None)) + tuple(slice(o, o + s) for o, s in zip(overlap_shape, inp.shape[2:]))] = inp
    crop = (slice(None), slice(None)) + tuple(slice(o, o + t) for o, t in zip(overlap_shape, tile_shape))
    out = None
    for ilo, ihi, olo, ohi in tile_plan(out_shape[2:], tile_shape, overlap_shape):
        tile = np.ascontiguousarray(padded[(slice(None), slice(None)) + tuple(slice(l, h) for l, h in zip(ilo, ihi))])
        res = func(tile)[crop]
        if out is None:
            out = np.empty(tuple(out_shape), res.dtype)
        out[(slice(None), slice(None)) + tuple(slice(l, h) for l, h in zip(olo, ohi))] = res
    return out


def predict_tiled(net, inp, tile_shape, overlap_shape, out_shape, apply_softmax=True):
    """``Predictor(model, tile_shape=..., overlap_shape=..., out_shape=..., strict_shapes=False).predict``
    (inference.py:569-687): eval-mode model (+ Softmax(1)), non-divisible out_shape padded up to a tile
    multiple with zero input (inference.py:645-687), result cropped back to ``out_shape``."""
    inp = _f32(inp)
    net.training = False
    tile_shape = np.asarray(tile_shape)
    out_shape = np.asarray(out_shape)  # (C, D, H, W)
    padded_out = out_shape.copy()
    padded_out[1:] = np.ceil(out_shape[1:] / tile_shape) * tile_shape
    pin = np.zeros(inp.shape[:2] + tuple(padded_out[1:]), np.float32)
    pin[(slice(None), slice(None)) + tuple(slice(0, s) for s in inp.shape[2:])] = inp

    def func(tile):
        y = net.forward(tile)
        return softmax_c(y) if apply_softmax else y

    out = tiled_apply(func, pin, tile_shape, overlap_shape, (inp.shape[0],) + tuple(padded_out))
    return out[(slice(None), slice(None)) + tuple(slice(0, s) for s in out_shape[1:])]
