"""Shared helpers for the test-suite (test infrastructure)."""
import os
from collections import OrderedDict

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def sub(npz, prefix):
    """Entries of an npz whose key starts with 'prefix/' as an OrderedDict without the prefix."""
    return OrderedDict((k[len(prefix) + 1:], npz[k]) for k in npz.files if k.startswith(prefix + '/'))


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def unet_cfg(npz):
    cfg = dict(n_blocks=int(npz['cfg.n_blocks']), start_filts=int(npz['cfg.start_filts']),
               planar_blocks=tuple(int(v) for v in npz['cfg.planar_blocks']))
    if 'cfg.dim' in npz.files:
        cfg['dim'] = int(npz['cfg.dim'])
    if 'cfg.normalization' in npz.files:
        cfg['normalization'] = str(npz['cfg.normalization'])
    if 'cfg.conv_mode' in npz.files:
        cfg['conv_mode'] = str(npz['cfg.conv_mode'])
    if 'cfg.up_mode' in npz.files:
        cfg['up_mode'] = str(npz['cfg.up_mode'])
    if 'cfg.activation' in npz.files:
        cfg['activation'] = str(npz['cfg.activation'])
    if 'cfg.merge_mode' in npz.files:
        cfg['merge_mode'] = str(npz['cfg.merge_mode'])
    if 'cfg.resunet' in npz.files:       # elektronn3.models.resunet.UNet: build with elektronn3_amd.resunet.UNet
        cfg['enc_res_blocks'] = int(npz['cfg.enc_res_blocks']); cfg['dec_res_blocks'] = int(npz['cfg.dec_res_blocks'])
    if 'cfg.attention' in npz.files:
        cfg['attention'] = bool(int(npz['cfg.attention']))
    if 'cfg.full_norm' in npz.files:
        cfg['full_norm'] = bool(int(npz['cfg.full_norm']))
    return cfg


def is_prebn_bias(k, names=None, paramless_norms=()):
    """Bias of a (transposed) conv that feeds a train-mode BatchNorm: analytically zero gradient.  ``names`` (all parameter
    names) tells whether the norm after that conv exists at all (normalization='none' / full_norm=False make it nn.Identity)."""
    if k.endswith('.attention.w.0.bias'):       # GridAttention's output transform: 1x1x1 conv -> nn.BatchNorm, always (unet.py:488-491)
        return True
    if '.convs.' in k:       # ResUNet ConvBlock (resunet.py:212-262): conv1 -> norm1, conv2 (+ proj) -> norm2
        if not k.endswith(('.conv1.bias', '.conv2.bias', '.proj.bias')):
            return False
        block = k.rsplit('.', 2)[0]
        norm = 'norm1' if k.endswith('.conv1.bias') else 'norm2'
        return names is None or f'{block}.{norm}.weight' in names or f'{block}.{norm}' in paramless_norms
    if not (k.endswith('.bias') and ('conv1' in k or 'conv2' in k or 'upconv' in k) and not k.startswith('conv_final')):
        return False
    if names is None:
        return True
    stem = k[:-len('.bias')]
    if stem.endswith('.upconv.conv'):           # ResizeConv (up_mode='resizeconv_*'): the conv lives one level deeper
        stem = stem[:-len('.conv')]
    block, conv = stem.rsplit('.', 1)
    if block.startswith('down_convs'):
        norm = {'conv1': 'norm0', 'conv2': 'norm1'}[conv]
    else:
        norm = {'upconv': 'norm0', 'conv1': 'norm1', 'conv2': 'norm2'}[conv]
    return f'{block}.{norm}.weight' in names or f'{block}.{norm}' in paramless_norms      # (nn.InstanceNorm has no parameters)


def embed_2d(sd):
    """A dim=2 state_dict as the equivalent dim=3 one: (Cout, Cin, kh, kw) conv / transposed-conv weights get a depth-1 kernel
    axis.  A 2D U-Net IS the 3D one with every block planar on a depth-1 volume (unet.py:47-74,114-128)."""
    return OrderedDict((k, v[:, :, None] if np.asarray(v).ndim == 4 else v) for k, v in sd.items())


def instance_norm_names(cfg):
    """Norm layers of a normalization='instance' model (no state_dict entries): every norm slot that full_norm leaves in place."""
    if cfg.get('normalization') != 'instance':
        return ()
    nb, full = cfg['n_blocks'], cfg.get('full_norm', True)
    names = []
    for i in range(nb):
        names += ([f'down_convs.{i}.norm0'] if full else []) + [f'down_convs.{i}.norm1']
    for i in range(nb - 1):
        names += ([f'up_convs.{i}.norm0', f'up_convs.{i}.norm1'] if full else []) + [f'up_convs.{i}.norm2']
    return tuple(names)


CLASS_WEIGHTS = (0.2653, 0.7347)


def combined_loss_np(logits, target, cw=CLASS_WEIGHTS):
    """0.5*CrossEntropy(weight=cw) + 0.5*Dice(softmax, weight=cw) in float64 numpy and its gradient
    w.r.t. logits.  Restates the example's criterion (examples/train_unet_neurodata.py:294-296;
    modules/loss.py:19-49,165-189) for the tests only -- the loss itself stays PyTorch in the product."""
    z = np.asarray(logits, np.float64)
    N, C = z.shape[:2]
    t = np.asarray(target)
    cw = np.asarray(cw, np.float64)
    zmax = z.max(axis=1, keepdims=True)
    e = np.exp(z - zmax)
    p = e / e.sum(axis=1, keepdims=True)
    onehot = np.zeros_like(p)
    np.put_along_axis(onehot, t[:, None], 1.0, axis=1)
    wmap = cw[t]  # (N, ...)
    logp = np.log(p)
    ce = -(wmap * np.take_along_axis(logp, t[:, None], axis=1)[:, 0]).sum() / wmap.sum()
    dce = (wmap[:, None] * (p - onehot)) / wmap.sum()
    axes = (0,) + tuple(range(2, z.ndim))
    num = 2 * (p * onehot).sum(axis=axes)
    den = (p + onehot).sum(axis=axes) + 1e-4
    dice = (cw * (1 - num / den)).mean()
    shp = (1, C) + (1,) * (z.ndim - 2)
    dL_dp = (cw / C).reshape(shp) * (-(2 * onehot) / den.reshape(shp) + (num / den ** 2).reshape(shp))
    ddice = p * (dL_dp - (dL_dp * p).sum(axis=1, keepdims=True))
    return 0.5 * ce + 0.5 * dice, 0.5 * dce + 0.5 * ddice


def load_bf16_fixture(path):
    """npz -> {key: fp32 tensor}; keys ending in ':bf16' hold bf16 values as their 16-bit patterns (tests/golden/make_golden.py)."""
    import numpy as np
    import torch
    g = np.load(path)
    out = {}
    for k in g.files:
        a = np.array(g[k])
        if k.endswith(':bf16'):
            out[k[:-5]] = torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).float()
        elif k.endswith(':f16'):            # IEEE half fixtures (tests/golden/unet_nb2_sf32_f16.npz)
            out[k[:-4]] = torch.from_numpy(a.view(np.int16).copy()).view(torch.float16).float()
        else:
            out[k] = torch.from_numpy(a) if a.ndim else torch.tensor(a.item())
    return out


# ---------------------------------------------------------------------------------------------- resident "foreign" kernel (test infrastructure)
_SPIN = None


def spin_lib():
    """ctypes handle of tests/native/spin_kernel.hip, built in-tree with hipcc on first use (None when hipcc is missing)."""
    global _SPIN
    if _SPIN is None:
        import ctypes
        import shutil
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        src, out = os.path.join(here, 'native', 'spin_kernel.hip'), os.path.join(here, 'native', 'libspin.so')
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
            if not os.path.exists(hipcc):
                return None
            tmp = out + f'.tmp{os.getpid()}'
            subprocess.run([hipcc, '--offload-arch=gfx950', '-O2', '-shared', '-fPIC', '-o', tmp, src], check=True, capture_output=True)
            os.replace(tmp, out)
        _SPIN = ctypes.CDLL(out)
        _SPIN.spin_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
        _SPIN.spin_launch.restype = ctypes.c_int
    return _SPIN


def spin(stream, blocks, microseconds, lds_bytes=16384, sink=None):
    """`blocks` workgroups of 256 threads resident on `stream` (a torch.cuda.Stream) for `microseconds`."""
    import torch
    lib = spin_lib()
    assert lib is not None, 'hipcc not available: cannot build the spin kernel'
    if sink is None:
        sink = torch.zeros(4, dtype=torch.int32, device='cuda')
    rc = lib.spin_launch(stream.cuda_stream, int(blocks), int(lds_bytes), int(microseconds * 100), sink.data_ptr())
    assert rc == 0, rc
    return sink


# ---------------------------------------------------------------------------------------------- full-size digest fixtures
# A train step at BASELINE.json configs[1]'s own size (UNet(1, 2, n_blocks=4, start_filts=32), batch 2 of 64 x 128 x 128) has 22 M parameters and
# 2 M voxels: neither the state_dict nor the gradients fit a "small fixture".  Both sides therefore REGENERATE parameters, input and target from
# one seed with numpy's PCG64 (stable across platforms), and the fixture keeps a DIGEST of what the reference computed from them: a strided sample
# of the logits, and per gradient tensor its norm, a strided sample and four projections on seeded Gaussian vectors (for a Gaussian r,
# E <e, r>^2 = |e|^2: the projections of a difference estimate its norm), from the reference's fp32 and fp64 runs.
def digest_state_dict(shapes, seed):
    """shapes: OrderedDict name -> shape in the reference's state_dict order (float entries only).  Conv / transposed-conv weights ~ N(0, 2 / fan_in),
    norm weights 1 + 0.2 N, biases 0.1 N, running_mean 0, running_var 1."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for k, shp in shapes.items():
        shp = tuple(int(v) for v in shp)
        if k.endswith('num_batches_tracked'):
            sd[k] = np.zeros(shp, np.int64)
        elif k.endswith('running_mean'):
            sd[k] = np.zeros(shp, np.float32)
        elif k.endswith('running_var'):
            sd[k] = np.ones(shp, np.float32)
        elif len(shp) >= 3:
            fan_in = int(np.prod(shp[1:]))
            sd[k] = (rng.standard_normal(shp, dtype=np.float32) * np.float32(np.sqrt(2.0 / max(fan_in, 1))))
        elif 'norm' in k and k.endswith('weight'):
            sd[k] = (1.0 + 0.2 * rng.standard_normal(shp, dtype=np.float32)).astype(np.float32)
        else:
            sd[k] = (0.1 * rng.standard_normal(shp, dtype=np.float32)).astype(np.float32)
    return sd


def digest_inputs(batch, shape, seed):
    rng = np.random.default_rng(seed + 1)
    x = rng.standard_normal((batch, 1, *shape), dtype=np.float32)
    t = rng.integers(0, 2, (batch, *shape), dtype=np.int64)
    return x, t


def digest_vectors(name, numel, seed, n=4):
    import zlib
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    return [rng.standard_normal(numel, dtype=np.float32) for _ in range(n)]


def digest_of(name, g, seed):
    """(norm, strided sample of <= 256 elements, 4 projections) of a gradient tensor, in float64."""
    g = np.asarray(g, np.float64).reshape(-1)
    step = max(1, g.size // 256)
    return float(np.linalg.norm(g)), g[::step][:256].copy(), np.array([float(g @ r.astype(np.float64)) for r in digest_vectors(name, g.size, seed)])
