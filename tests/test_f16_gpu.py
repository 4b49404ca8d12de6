"""The 16-bit path compiled for IEEE half (float16) -- the reference's own reduced-precision mode: ``torch.cuda.amp.autocast`` defaults to
float16 (``Trainer(mixed_precision=True)``, trainer.py:367,519) and ``Predictor(float16=True)`` calls ``model.half()`` (inference.py:445-446).

Same kernels as the bfloat16 path (tests/test_bf16_gpu.py), different element type, conversions and matrix instruction.  Per-op: against the
op in fp64 on the SAME float16-valued inputs (what is left is the fp32 accumulation order and the final rounding, half an ulp = 2^-11 relative).
Whole network: a ``model.half()`` module against the fp32 HIP path on the same float16-valued parameters; float16 autocast with GradScaler.
"""
import os

import pytest
import torch
import torch.nn.functional as F

from helpers import GOLDEN, load_bf16_fixture

pytestmark = pytest.mark.gpu
DEV = 'cuda'
HF = torch.float16


def _ndhwc(t):
    return t.permute(0, 2, 3, 4, 1).contiguous()


def _ncdhw(t):
    return t.permute(0, 4, 1, 2, 3).contiguous()


def _hvals(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(HF)


def _close_f16(got, ref64, what, ulps=1.1, atol=1e-6):
    """|got - ref| <= ulps * 2^-11 * |ref| + fp32 accumulation noise (relative to the size of the terms: 1e-5 of the largest output)."""
    got = got.double().cpu(); ref64 = ref64.double().cpu()
    err = (got - ref64).abs()
    bound = ulps * 2.0 ** -11 * ref64.abs() + atol + 1e-5 * float(ref64.abs().max())
    bad = err > bound
    assert not bool(bad.any()), f'{what}: {int(bad.sum())} of {bad.numel()} beyond float16 rounding; worst {float((err / bound).max()):.2f}x the bound'


@pytest.mark.parametrize('N,D,H,W,Cin,Cout', [(1, 1, 8, 16, 32, 32), (2, 3, 19, 37, 64, 32), (1, 1, 70, 100, 32, 64), (2, 16, 128, 64, 32, 32), (1, 2, 13, 21, 128, 64)])
def test_conv3d_f16_planar_forward_dgrad_wgrad_vs_fp64(N, D, H, W, Cin, Cout):
    """1x3x3 convs of planar blocks / dim = 2 networks in float16 (one-slice bricks, bf16_conv.hip Geo::FLAT)."""
    test_conv3d_f16_forward_dgrad_wgrad_vs_fp64(N, D, H, W, Cin, Cout, planar=True)


@pytest.mark.parametrize('N,D,H,W,Cin,Cout', [(1, 4, 8, 16, 32, 32), (2, 5, 11, 21, 64, 32), (1, 3, 9, 17, 128, 64), (2, 32, 64, 64, 32, 32),
                                            (1, 9, 13, 100, 64, 64)])
def test_conv3d_f16_forward_dgrad_wgrad_vs_fp64(N, D, H, W, Cin, Cout, planar=False):
    from elektronn3_amd import ops
    pad = (0, 1, 1) if planar else 1
    x = _hvals(N, Cin, D, H, W, seed=1)
    w = _hvals(Cout, Cin, 1 if planar else 3, 3, 3, seed=2, scale=((9 if planar else 27) * Cin) ** -0.5)
    b = torch.randn(Cout, generator=torch.Generator().manual_seed(3))
    dy = _hvals(N, Cout, D, H, W, seed=4)
    xd, wd, dyd = x.to(DEV), w.to(DEV), dy.to(DEV)
    y, stats = ops.conv3d_bf16(_ndhwc(xd), wd.float(), b.to(DEV), planar=planar, want_stats=True)
    assert y.dtype == HF
    ref = F.conv3d(x.double(), w.double(), b.double(), padding=pad)
    _close_f16(_ncdhw(y), ref, 'conv forward')
    # statistics records of the stored (rounded) values merge to the tensor's mean / variance
    yv = _ncdhw(y).double().cpu()
    n, mean, m2 = stats[:, :, 0].double().cpu(), stats[:, :, 1].double().cpu(), stats[:, :, 2].double().cpu()
    tot = n.sum(0); gm = (n * mean).sum(0) / tot
    var = ((m2 + n * (mean - gm) ** 2).sum(0)) / tot
    assert torch.allclose(gm, yv.mean((0, 2, 3, 4)), atol=1e-4) and torch.allclose(var, yv.var((0, 2, 3, 4), unbiased=False), rtol=1e-3, atol=1e-4)
    dx = ops.conv3d_dgrad_bf16(_ndhwc(dyd), wd.float(), planar=planar)
    refdx = F.conv_transpose3d(dy.double(), w.double(), padding=pad)
    _close_f16(_ncdhw(dx), refdx, 'conv dgrad')
    dw = ops.conv3d_wgrad_bf16(_ndhwc(xd), _ndhwc(dyd), planar=planar)
    xr = x.double().requires_grad_(False); wr = w.double().requires_grad_(True)
    F.conv3d(xr, wr, None, padding=pad).backward(dy.double())
    assert float((dw.double().cpu() - wr.grad).norm() / wr.grad.norm()) < 1e-5        # fp32 accumulation, fp32 result


@pytest.mark.parametrize('N,D,H,W,Cin,Cout,odd', [(1, 4, 8, 8, 64, 32, False), (2, 3, 5, 9, 128, 64, True)])
def test_convT_f16_forward_dgrad_wgrad_vs_fp64(N, D, H, W, Cin, Cout, odd):
    from elektronn3_amd import ops
    x = _hvals(N, Cin, D, H, W, seed=5)
    w = _hvals(Cin, Cout, 2, 2, 2, seed=6, scale=Cin ** -0.5)
    b = torch.randn(Cout, generator=torch.Generator().manual_seed(7))
    Do, Ho, Wo = (2 * D - 1, 2 * H, 2 * W - 1) if odd else (2 * D, 2 * H, 2 * W)
    y = ops.convT_bf16(_ndhwc(x.to(DEV)), w.to(DEV).float(), b.to(DEV), out_dims=(Do, Ho, Wo))
    xr = x.double().requires_grad_(True); wr = w.double().requires_grad_(True)
    ref = F.conv_transpose3d(xr, wr, b.double(), stride=2)[:, :, :Do, :Ho, :Wo]
    _close_f16(_ncdhw(y), ref.detach(), 'convT forward')
    dy = _hvals(N, Cout, Do, Ho, Wo, seed=8)
    ref.backward(dy.double())
    dx = ops.convT_dgrad_bf16(_ndhwc(dy.to(DEV)), w.to(DEV).float(), (D, H, W))
    _close_f16(_ncdhw(dx), xr.grad, 'convT dgrad')
    dw = ops.convT_wgrad_bf16(_ndhwc(x.to(DEV)), _ndhwc(dy.to(DEV)))
    assert float((dw.double().cpu() - wr.grad).norm() / wr.grad.norm()) < 1e-5


def _models(nb, sf, seed=0):
    from elektronn3_amd.unet import UNet
    torch.manual_seed(seed)
    m32 = UNet(1, 2, n_blocks=nb, start_filts=sf)
    with torch.no_grad():        # float16-valued parameters in both modules
        for p in m32.parameters():
            p.copy_(p.to(HF).float())
    m16 = UNet(1, 2, n_blocks=nb, start_filts=sf)
    m16.load_state_dict(m32.state_dict())
    return m32.to(DEV), m16.to(DEV).half()


def _train_step(m, x, dlogits):
    m.train()
    m.zero_grad(set_to_none=True)
    y = m(x)
    assert y.dtype == x.dtype
    y.backward(dlogits.to(y.dtype))
    torch.cuda.synchronize()
    assert all(p.grad.dtype == p.dtype for p in m.parameters())
    return y.detach().float().cpu(), {k: p.grad.detach().float().cpu() for k, p in m.named_parameters()}


@pytest.mark.parametrize('nb,sf,shape', [(2, 32, (2, 1, 8, 16, 16)), (3, 32, (1, 1, 9, 17, 21)), (4, 32, (2, 1, 32, 64, 64))])
def test_unet_f16_train_step_tracks_the_fp32_path(nb, sf, shape):
    """model.half() vs the fp32 HIP path on the same float16-valued parameters and input.  The incoming gradient is of O(1) (a scaled loss,
    as GradScaler provides: unscaled segmentation gradients of ~1e-7 per voxel would underflow float16 in ANY implementation)."""
    from elektronn3_amd import _lib
    m32, m16 = _models(nb, sf, seed=nb)
    assert _lib.load().e3_unet_f16_supported(m16._plan().handle) == 1
    x = _hvals(*shape, seed=21)
    dl = _hvals(shape[0], 2, *shape[2:], seed=22, scale=1.0)
    y32, g32 = _train_step(m32, x.float().to(DEV), dl.float().to(DEV))
    y16, g16 = _train_step(m16, x.to(DEV), dl.to(DEV))
    scale = float(y32.abs().max())
    err = (y16 - y32).abs()
    assert float(err.max()) < 2e-2 * scale, f'logits: max {float(err.max())} vs scale {scale}'
    gscale = max(float(g.norm()) for g in g32.values())
    worst = 0.0
    for k, g in g32.items():
        assert torch.isfinite(g16[k]).all(), k
        if k.endswith('.bias') and 'norm' not in k and not k.startswith('conv_final'):
            assert float(g16[k].norm()) < 2e-2 * gscale, k         # analytically zero (bias in front of a train-mode BN)
            continue
        rel = float((g16[k] - g).norm() / max(float(g.norm()), 1e-3 * gscale))
        worst = max(worst, rel)
        assert rel < 0.35, f'gradient {k}: rel-L2 {rel}'      # (measured: 0.07 / 0.11 / 0.23 for 2 / 3 / 4 blocks; bfloat16: up to 0.6)
    for (k, a), (_, b) in zip(m32.named_buffers(), m16.named_buffers()):
        if 'running' in k:
            torch.testing.assert_close(b.float(), a, rtol=3e-3, atol=3e-3 * float(a.abs().max()), msg=lambda s: f'{k}: {s}')
    print(f'nb={nb}: logits max err {float(err.max()):.3e} (scale {scale:.2f}); worst gradient rel-L2 {worst:.3e}')


def test_unet_f16_eval_softmax_and_determinism():
    m32, m16 = _models(3, 32, seed=5)
    x = _hvals(1, 1, 12, 24, 24, seed=31)
    for _ in range(2):
        m32.train()(x.float().to(DEV)); m16.train()(x.to(DEV))
    m32.eval(); m16.eval()
    with torch.no_grad():
        y32 = m32(x.float().to(DEV)).cpu()
        ya = m16(x.to(DEV)).float().cpu()
        yb = m16(x.to(DEV)).float().cpu()
        s16 = m16.forward_softmax(x.to(DEV)).float().cpu()
    assert torch.equal(ya, yb)
    assert float((ya - y32).abs().max()) < 1e-2 * float(y32.abs().max())
    torch.testing.assert_close(s16, torch.softmax(ya, 1), rtol=0, atol=2e-3)


def test_unet_f16_autocast_with_gradscaler():
    """The reference's Trainer(mixed_precision=True) protocol (trainer.py:367-368,519,539-542): torch.autocast (float16) around forward + loss,
    GradScaler around backward / step.  float16 compute and logits, fp32 master parameters and gradients; the unscaled gradients track the
    fp32 run, and an optimizer step goes through."""
    m32, _ = _models(2, 32, seed=9)
    x = torch.randn(2, 1, 8, 16, 16, device=DEV)
    t = torch.randint(0, 2, (2, 8, 16, 16), device=DEV)
    crit = torch.nn.CrossEntropyLoss()
    m32.train()
    m32.zero_grad(set_to_none=True)
    crit(m32(x), t).backward()
    g_ref = {k: p.grad.detach().clone() for k, p in m32.named_parameters()}
    sd = {k: v.clone() for k, v in m32.state_dict().items()}
    m32.load_state_dict(sd)
    opt = torch.optim.SGD(m32.parameters(), lr=1e-3)
    scaler = torch.amp.GradScaler('cuda')
    m32.zero_grad(set_to_none=True)
    with torch.autocast('cuda', dtype=HF):
        y = m32(x)
        loss = crit(y, t)
    assert y.dtype == HF
    scaler.scale(loss).backward()
    scaler.unscale_(opt)
    assert all(p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all() for p in m32.parameters())
    for k, p in m32.named_parameters():
        if k.endswith('.bias') and 'norm' not in k and not k.startswith('conv_final'):
            continue
        rel = float((p.grad - g_ref[k]).norm() / g_ref[k].norm())
        assert rel < 0.1, (k, rel)
    scaler.step(opt); scaler.update()


def test_predictor_float16_switch_runs_the_native_f16_kernels():
    """Predictor(float16=True) (inference.py:445-446: model.half(), float16 tiles): native float16 kernels, float16 result, close to fp32."""
    from elektronn3_amd.inference import Predictor
    from elektronn3_amd.unet import UNet
    torch.manual_seed(3)
    m = UNet(1, 2, n_blocks=2, start_filts=32).to(DEV)
    m.train()
    with torch.no_grad():
        for _ in range(2):
            m(torch.randn(2, 1, 16, 32, 32, device=DEV))
    m.eval()
    vol = torch.randn(1, 1, 24, 48, 64)
    kw = dict(device=DEV, tile_shape=(12, 24, 32), overlap_shape=(4, 8, 8), offset=None, out_shape=(2, 24, 48, 64), apply_softmax=True)
    y32 = Predictor(m, **kw).predict(vol)
    y16 = Predictor(m, float16=True, **kw).predict(vol)
    assert y16.dtype == HF and y32.dtype == torch.float32
    assert next(m.parameters()).dtype == torch.float32            # the caller's module is left in fp32 (inference.py:402-407)
    assert float((y16.float() - y32).abs().max()) < 5e-3


def test_unet_f16_against_reference_fixture():
    """The imported reference in float16 (model.half() on CPU, tests/golden/make_golden.py f16) on a committed fixture: one train step with an
    O(1) logits gradient; beside it the reference's fp32 run of the same float16-valued weights and input.  The native path must be as close
    to the fp32 run as the reference's own float16 run is (x3), logits and every gradient tensor."""
    from elektronn3_amd.unet import UNet
    g = load_bf16_fixture(os.path.join(GOLDEN, 'unet_nb2_sf32_f16.npz'))
    sd = {k[4:]: v for k, v in g.items() if k.startswith('sd0/')}
    m = UNet(1, 2, n_blocks=int(g['cfg.n_blocks']), start_filts=int(g['cfg.start_filts']))
    m.load_state_dict(sd)
    m = m.to(DEV).half().train()
    y = m(g['x'].half().to(DEV))
    y.backward(g['dlogits'].half().to(DEV))
    torch.cuda.synchronize()
    ref16, ref32 = g['logits_f16'], g['logits_fp32']
    scale = float(ref32.abs().max())
    err_ref = float((ref16 - ref32).abs().max())
    err = float((y.float().cpu() - ref32).abs().max())
    assert err < max(3e-3 * scale, 2 * err_ref), f'logits: {err} (reference float16 vs fp32: {err_ref}, scale {scale})'
    for k, p in m.named_parameters():
        if k.endswith('.bias') and 'norm' not in k and not k.startswith('conv_final'):
            continue
        r32, r16 = g['grad32/' + k], g['grad16/' + k]
        e_ref = float((r16 - r32).norm() / r32.norm())
        e = float((p.grad.float().cpu() - r32).norm() / r32.norm())
        assert e < max(3 * e_ref, 1e-2), f'{k}: rel-L2 {e} (reference float16: {e_ref})'
    for k, v in m.state_dict().items():
        if 'running' in k:
            torch.testing.assert_close(v.float().cpu(), g['sd1_f16/' + k], rtol=4e-3, atol=4e-3 * float(g['sd1_f16/' + k].abs().max()), msg=lambda s: f'{k}: {s}')


@pytest.mark.parametrize('lowp', [torch.float16, torch.bfloat16], ids=['f16', 'bf16'])
@pytest.mark.parametrize('kw,shape', [(dict(n_blocks=3, planar_blocks=(0,)), (2, 1, 6, 24, 40)),
                                      (dict(n_blocks=3, planar_blocks=(0, 1)), (1, 1, 5, 33, 47)),
                                      (dict(n_blocks=4, planar_blocks=(0,)), (2, 1, 16, 64, 64)),        # the example script's network (train_unet_neurodata.py:96-106)
                                      (dict(n_blocks=3, planar_blocks=(1,)), (1, 1, 8, 20, 28)),
                                      (dict(n_blocks=3, dim=2), (2, 1, 44, 60))],
                         ids=['planar0', 'planar01_odd', 'example_net', 'planar1', 'dim2'])
def test_16bit_path_with_planar_blocks_tracks_the_fp32_path(lowp, kw, shape):
    """Planar blocks (1x3x3 convs, (1,2,2) pooling and up-convolution, unet.py:114-128) and dim=2 on the native 16-bit path -- the reference's
    example network has planar_blocks=(0,), so Trainer(mixed_precision=True) on it is this path.  Train step against the fp32 HIP path on the
    same 16-bit-valued parameters and input, O(1) incoming gradient."""
    from elektronn3_amd import _lib
    from elektronn3_amd.unet import UNet
    torch.manual_seed(17)
    m32 = UNet(1, 2, start_filts=32, **kw)
    with torch.no_grad():
        for p in m32.parameters():
            p.copy_(p.to(lowp).float())
    m16 = UNet(1, 2, start_filts=32, **kw)
    m16.load_state_dict(m32.state_dict())
    m32, m16 = m32.to(DEV), m16.to(DEV).to(lowp)
    lib = _lib.load()
    assert (lib.e3_unet_f16_supported if lowp == torch.float16 else lib.e3_unet_bf16_supported)(m16._plan().handle) == 1
    g = torch.Generator().manual_seed(23)
    x = torch.randn(*shape, generator=g).to(lowp)
    dl = torch.randn(shape[0], 2, *shape[2:], generator=g).to(lowp)
    y32, g32 = _train_step(m32, x.float().to(DEV), dl.float().to(DEV))
    y16, g16 = _train_step(m16, x.to(DEV), dl.to(DEV))
    tol, gtol = (2e-2, 0.35) if lowp == torch.float16 else (1e-1, 0.7)
    scale = float(y32.abs().max())
    assert float((y16 - y32).abs().max()) < tol * scale, (float((y16 - y32).abs().max()), scale)
    gscale = max(float(v.norm()) for v in g32.values())
    for k, v in g32.items():
        assert torch.isfinite(g16[k]).all(), k
        if k.endswith('.bias') and 'norm' not in k and not k.startswith('conv_final'):
            continue
        rel = float((g16[k] - v).norm() / max(float(v.norm()), 1e-3 * gscale))
        assert rel < gtol, f'gradient {k}: rel-L2 {rel}'
    m32.eval(); m16.eval()
    with torch.no_grad():
        ye32, ye16 = m32(x.float().to(DEV)).cpu(), m16(x.to(DEV)).float().cpu()
    assert float((ye16 - ye32).abs().max()) < tol * float(ye32.abs().max())
