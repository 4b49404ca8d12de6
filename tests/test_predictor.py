"""Predictor / tiled_apply host logic.

CPU (not gpu): the tiling, padding, cropping, batching and option handling of elektronn3_amd.inference.Predictor,
driven with a plain torch module built from the oracle's functional network (tests may use the oracle), must
reproduce the reference's Predictor outputs stored in tests/golden/predictor.npz.

GPU: the same fixtures through the native HIP UNet (eval-mode BN folded, softmax fused into the last kernel).
"""
import numpy as np
import pytest
import torch
from torch import nn

from helpers import load_npz, sub


class RefModule(nn.Module):
    """nn.Module around oracle/torch_ref.unet_forward (generic-module path of the Predictor)."""

    def __init__(self, sd, n_blocks):
        super().__init__()
        self.sd = nn.ParameterDict()
        self._names = {}
        for k, v in sd.items():
            key = k.replace('.', '__')
            self._names[k] = key
            t = torch.from_numpy(np.array(v))
            if t.is_floating_point():
                self.sd[key] = nn.Parameter(t, requires_grad=False)
        self.n_blocks = n_blocks

    def forward(self, x):
        from oracle.torch_ref import unet_forward
        sd = {k: self.sd[key] for k, key in self._names.items() if key in self.sd}
        return unet_forward(sd, x, self.n_blocks, (), training=self.training)


@pytest.fixture(scope='module')
def gold():
    return load_npz('predictor.npz')


def test_tile_plan_matches_reference_order(gold):
    from elektronn3_amd.inference import tile_plan
    plan = tile_plan((24, 48, 48), gold['tile'], gold['overlap'])
    assert len(plan) == 27
    assert plan[0] == ((0, 0, 0), (16, 32, 32), (0, 0, 0), (8, 16, 16))
    assert plan[1][0] == (0, 0, 16) and plan[3][0] == (0, 16, 0) and plan[9][0] == (8, 0, 0)   # C-order, W fastest


def test_predictor_cpu_generic_module_matches_reference(gold):
    from elektronn3_amd.inference import Predictor
    model = RefModule(sub(gold, 'sd'), 2)
    pred = Predictor(model, device='cpu', tile_shape=tuple(gold['tile']), overlap_shape=tuple(gold['overlap']), offset=(0, 0, 0),
                     out_shape=tuple(gold['out_shape']), apply_softmax=True, strict_shapes=False)
    y = pred.predict(gold['vol'])
    assert not y.is_cuda and tuple(y.shape) == gold['out_tiled'].shape
    np.testing.assert_allclose(y.numpy(), gold['out_tiled'], rtol=1e-4, atol=1e-5)
    assert not model.training            # side effect of the reference kept: the caller's module is put in eval mode
    # untiled
    y2 = Predictor(model, device='cpu', apply_softmax=True).predict(torch.from_numpy(gold['vol']))
    np.testing.assert_allclose(y2.numpy(), gold['out_untiled'], rtol=1e-4, atol=1e-5)
    # argmax + batch splitting
    pred3 = Predictor(model, device='cpu', tile_shape=tuple(gold['tile']), overlap_shape=tuple(gold['overlap']), offset=(0, 0, 0),
                      out_shape=(2, 16, 32, 32), apply_softmax=True, apply_argmax=True, batch_size=1)
    y3 = pred3.predict(gold['vol3'])
    assert y3.dtype == torch.uint8 and tuple(y3.shape) == gold['out3_argmax'].shape
    assert (y3.numpy() != gold['out3_argmax']).mean() < 1e-4


def test_predictor_argument_validation(gold):
    from elektronn3_amd.inference import Predictor, tiled_apply
    model = RefModule(sub(gold, 'sd'), 2)
    with pytest.raises(ValueError):
        Predictor(model, device='cpu', tile_shape=(8, 16, 16), overlap_shape=(4, 8, 8), offset=(1, 1, 1), out_shape=(2, 16, 32, 32))
    with pytest.raises(ValueError):
        Predictor(model, device='cpu', apply_softmax=False, augmentations=3)
    with pytest.raises(ValueError):   # strict shapes: non-divisible out_shape
        Predictor(model, device='cpu', tile_shape=(8, 16, 16), overlap_shape=(4, 8, 8), offset=(0, 0, 0), out_shape=(2, 20, 40, 36),
                  strict_shapes=True).predict(gold['vol'])
    with pytest.raises(ValueError):
        tiled_apply(lambda t, c: t, torch.zeros(1, 1, 8, 8, 8), (3, 3, 3), (1, 1, 1), None, (1, 1, 8, 8, 8))
    with pytest.raises(ValueError):
        tiled_apply(lambda t, c: t, torch.zeros(1, 1, 8, 8, 8), (4, 4), (1, 1), None, (1, 1, 8, 8, 8))


def test_predictor_tta_cpu(gold):
    """Flip test-time augmentation (inference.py:507-517): mean over identity + flips equals a manual computation."""
    from elektronn3_amd.inference import Predictor
    model = RefModule(sub(gold, 'sd'), 2).eval()
    x = torch.from_numpy(gold['vol3'][:1, :, :8, :16, :16].copy())
    y = Predictor(model, device='cpu', apply_softmax=True, augmentations=2).predict(x)
    with torch.no_grad():
        f = lambda t: torch.softmax(model(t), 1)
        manual = (f(x) + torch.flip(f(torch.flip(x, (2,))), (2,)) + torch.flip(f(torch.flip(x, (3,))), (3,))) / 3
    np.testing.assert_allclose(y.numpy(), manual.numpy(), rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------- GPU
def native_model(gold):
    from elektronn3_amd.unet import UNet
    m = UNet(1, 2, n_blocks=2, start_filts=8)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sub(gold, 'sd').items()})
    return m.cuda()


@pytest.mark.gpu
def test_predictor_native_matches_reference(gold):
    from elektronn3_amd.inference import Predictor
    m = native_model(gold)
    pred = Predictor(m, device='cuda', tile_shape=tuple(gold['tile']), overlap_shape=tuple(gold['overlap']), offset=None,
                     out_shape=tuple(gold['out_shape']), apply_softmax=True, strict_shapes=False)
    y = pred.predict(gold['vol'])
    assert not y.is_cuda and tuple(y.shape) == gold['out_tiled'].shape
    np.testing.assert_allclose(y.numpy(), gold['out_tiled'], rtol=1e-4, atol=1e-5)
    y2 = Predictor(m, device='cuda', apply_softmax=True).predict(gold['vol'])
    np.testing.assert_allclose(y2.numpy(), gold['out_untiled'], rtol=1e-4, atol=1e-5)
    pred3 = Predictor(m, device='cuda', tile_shape=tuple(gold['tile']), overlap_shape=tuple(gold['overlap']), offset=(0, 0, 0),
                      out_shape=(2, 16, 32, 32), apply_softmax=True, apply_argmax=True)
    y3 = pred3.predict(gold['vol3'])
    assert y3.dtype == torch.uint8
    # integer output: bit-exact wherever the decision is not a tie -- a voxel may only differ from the reference's argmax if the
    # two class probabilities are within the fp32 forward tolerance of each other there
    y3p = Predictor(m, device='cuda', tile_shape=tuple(gold['tile']), overlap_shape=tuple(gold['overlap']), offset=(0, 0, 0),
                    out_shape=(2, 16, 32, 32), apply_softmax=True).predict(gold['vol3']).numpy()
    diff = y3.numpy() != gold['out3_argmax']
    margin = np.broadcast_to(np.abs(y3p[:, 0] - y3p[:, 1])[:, None], diff.shape)     # (the reference's argmax output keeps out_shape's channel axis)
    assert not diff[margin > 1e-5].any(), f'{int(diff[margin > 1e-5].sum())} argmax voxels differ away from ties'
    assert diff.mean() < 1e-4
    # logits path (no softmax) + float16 option (computed in fp32, cast on output)
    y4 = Predictor(m, device='cuda', apply_softmax=False).predict(gold['vol'])
    y5 = Predictor(m, device='cuda', apply_softmax=True, float16=True).predict(gold['vol'])
    assert y5.dtype == torch.float16
    np.testing.assert_allclose(torch.softmax(y4, 1).numpy(), gold['out_untiled'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(y5.float().numpy(), gold['out_untiled'], rtol=2e-2, atol=2e-3)
    assert next(m.parameters()).dtype == torch.float32    # float16=True deep-copies, the caller's model is untouched


@pytest.mark.gpu
def test_predictor_native_tta_and_cfg5_shaped_tiling(gold):
    """cfg-5-shaped run at reduced size: tile 96x192x192 / overlap 16 geometry scaled down (non-divisible volume, halo
    tiles), plus size-independent properties: the tiled result equals running each tile by hand, and the prediction
    of an all-zero padded region is finite."""
    from elektronn3_amd.inference import Predictor, tile_plan
    m = native_model(gold).eval()
    vol = torch.randn(1, 1, 40, 72, 88)
    tile, ov = (24, 32, 32), (8, 8, 8)
    pred = Predictor(m, device='cuda', tile_shape=tile, overlap_shape=ov, offset=None, out_shape=(2, 40, 72, 88), apply_softmax=True)
    y = pred.predict(vol)
    assert tuple(y.shape) == (1, 2, 40, 72, 88) and torch.isfinite(y).all()
    # hand-rolled tiles
    padded_out = (48, 96, 96)
    padded = torch.zeros(1, 1, *(p + 2 * o for p, o in zip(padded_out, ov)))
    padded[:, :, 8:48, 8:80, 8:96] = vol
    full = torch.zeros(1, 2, *padded_out)
    with torch.no_grad():
        for ilo, ihi, olo, ohi in tile_plan(padded_out, tile, ov):
            t = padded[:, :, ilo[0]:ihi[0], ilo[1]:ihi[1], ilo[2]:ihi[2]].cuda()
            o = m.forward_softmax(t)[:, :, 8:32, 8:40, 8:40].cpu()
            full[:, :, olo[0]:ohi[0], olo[1]:ohi[1], olo[2]:ohi[2]] = o
    assert torch.equal(y, full[:, :, :40, :72, :88])
    # the streaming host<->device pipeline (z slabs up, finished tile rows down on side streams) changes nothing
    import os
    os.environ['E3_PREDICTOR_NO_PIPELINE'] = '1'
    try:
        y_plain = pred.predict(vol)
    finally:
        del os.environ['E3_PREDICTOR_NO_PIPELINE']
    assert torch.equal(y, y_plain)
    yb = Predictor(m, device='cuda', tile_shape=tile, overlap_shape=ov, offset=None, out_shape=(2, 40, 72, 88), apply_softmax=True,
                   apply_argmax=True).predict(torch.cat([vol, vol.flip(2)]))       # batch of 2, uint8 output through the pipeline
    assert yb.dtype == torch.uint8 and tuple(yb.shape) == (2, 1, 40, 72, 88) and torch.equal(yb[0, 0], y[0].argmax(0).to(torch.uint8))
    ya = Predictor(m, device='cuda', apply_softmax=True, augmentations=3).predict(vol[:, :, :16, :32, :32])
    assert torch.allclose(ya.sum(1), torch.ones_like(ya[:, 0]), atol=1e-5)


@pytest.mark.gpu
def test_predictor_dim2_native_tiled():
    """dim=2 model through the same Predictor (tiled_apply is rank-generic, inference.py:115-125): non-divisible image, halo
    tiles; equals the hand-rolled tiles, and one tile equals the ATen op sequence on PyTorch-ROCm."""
    from elektronn3_amd.inference import Predictor, tile_plan
    from elektronn3_amd.unet import UNet
    from oracle.torch_ref import unet_forward
    torch.manual_seed(8)
    m = UNet(1, 2, n_blocks=3, start_filts=16, dim=2).cuda()
    m.train()
    with torch.no_grad():
        for _ in range(2):
            m(torch.randn(2, 1, 64, 64, device='cuda'))      # non-trivial running statistics
    m.eval()
    img = torch.randn(1, 1, 150, 200)
    tile, ov = (64, 96), (16, 16)
    y = Predictor(m, device='cuda', tile_shape=tile, overlap_shape=ov, offset=None, out_shape=(2, 150, 200), apply_softmax=True).predict(img)
    assert tuple(y.shape) == (1, 2, 150, 200) and torch.isfinite(y).all()
    padded_out = (192, 288)
    padded = torch.zeros(1, 1, *(p + 2 * o for p, o in zip(padded_out, ov)))
    padded[:, :, 16:166, 16:216] = img
    full = torch.zeros(1, 2, *padded_out)
    sd = {k: v.double() if v.is_floating_point() else v for k, v in m.state_dict().items()}
    with torch.no_grad():
        for i, (ilo, ihi, olo, ohi) in enumerate(tile_plan(padded_out, tile, ov)):
            t = padded[:, :, ilo[0]:ihi[0], ilo[1]:ihi[1]].cuda()
            o = m.forward_softmax(t)
            if i == 0:
                ref = torch.softmax(unet_forward(sd, t.double(), 3, (), training=False), 1)
                assert torch.allclose(o.double(), ref, rtol=1e-4, atol=1e-5)
            full[:, :, olo[0]:ohi[0], olo[1]:ohi[1]] = o[:, :, 16:80, 16:112].cpu()
    assert torch.equal(y, full[:, :, :150, :200])


@pytest.mark.gpu
def test_predictor_resunet_with_attention_native_tiled():
    """The Predictor is model-agnostic in the reference; here a ``resunet.UNet`` with residual blocks and attention gates goes through the
    native path (fused softmax, pipelined tiles): equals the hand-rolled tiles, and one tile equals the fp64 op sequence."""
    from elektronn3_amd.inference import Predictor, tile_plan
    from elektronn3_amd.resunet import UNet
    from oracle.torch_ref import resunet_forward
    torch.manual_seed(18)
    m = UNet(1, 2, n_blocks=3, start_filts=16, enc_res_blocks=2, dec_res_blocks=1, attention=True).cuda()
    m.train()
    with torch.no_grad():
        for _ in range(2):
            m(torch.randn(2, 1, 16, 32, 32, device='cuda'))      # non-trivial running statistics
    m.eval()
    vol = torch.randn(1, 1, 20, 70, 90)
    tile, ov = (16, 32, 48), (4, 8, 8)
    y = Predictor(m, device='cuda', tile_shape=tile, overlap_shape=ov, offset=None, out_shape=(2, 20, 70, 90), apply_softmax=True).predict(vol)
    assert tuple(y.shape) == (1, 2, 20, 70, 90) and torch.isfinite(y).all()
    padded_out = (32, 96, 96)
    padded = torch.zeros(1, 1, *(p + 2 * o for p, o in zip(padded_out, ov)))
    padded[:, :, 4:24, 8:78, 8:98] = vol
    full = torch.zeros(1, 2, *padded_out)
    sd = {k: v.double() if v.is_floating_point() else v for k, v in m.state_dict().items()}
    with torch.no_grad():
        for i, (ilo, ihi, olo, ohi) in enumerate(tile_plan(padded_out, tile, ov)):
            t = padded[:, :, ilo[0]:ihi[0], ilo[1]:ihi[1], ilo[2]:ihi[2]].cuda()
            o = m.forward_softmax(t)
            if i == 0:
                ref = torch.softmax(resunet_forward(sd, t.double(), 3, (), False, 2, 1), 1)
                assert torch.allclose(o.double(), ref, rtol=1e-4, atol=1e-5)
            full[:, :, olo[0]:ohi[0], olo[1]:ohi[1], olo[2]:ohi[2]] = o[:, :, 4:20, 8:40, 8:56].cpu()
    assert torch.equal(y, full[:, :, :20, :70, :90])


@pytest.mark.gpu
@pytest.mark.parametrize('world', [3, 7])
def test_predictor_tile_parallel_ranks_share_one_output(tmp_path, world):
    """the (z, y) rows of tiles are split over the ranks, every rank writes its rows into ONE output buffer in
    shared host memory, no data-path collective.  `world` gloo ranks share the one GPU of the test box (the sharding logic does
    not care which device a rank drives); 7 ranks > 6 rows leaves a rank without work.  Result == single-process result, bit
    for bit, on every rank; uint8 argmax output too; nothing is left behind in /dev/shm."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'pred_ranks.py'
    script.write_text('''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from elektronn3_amd.unet import UNet
from elektronn3_amd.inference import Predictor
dist.init_process_group('gloo')
torch.cuda.set_device(0)
torch.manual_seed(0)
m = UNet(1, 2, n_blocks=2, start_filts=8).cuda()
m.train()
with torch.no_grad():
    m(torch.randn(2, 1, 16, 32, 32, device='cuda'))
vol = torch.randn(1, 1, 40, 72, 88, generator=torch.Generator().manual_seed(1))
kw = dict(device='cuda', tile_shape=(24, 32, 32), overlap_shape=(8, 8, 8), offset=None, out_shape=(2, 40, 72, 88), apply_softmax=True)
y = Predictor(m, tile_parallel=True, **kw).predict(vol)
ref = Predictor(m, tile_parallel=False, **kw).predict(vol)
ya = Predictor(m, tile_parallel=True, apply_argmax=True, **kw).predict(vol)
ok = torch.equal(y, ref) and ya.dtype == torch.uint8 and torch.equal(ya[0, 0], ref[0].argmax(0).to(torch.uint8))
left = [f for f in os.listdir('/dev/shm') if f.startswith('e3pred_')]
dist.barrier()
print('RANK_OK' if ok and not left else 'RANK_BAD', dist.get_rank(), left, flush=True)
''' % root)
    port = 29600 + world
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count('RANK_OK') == world and 'RANK_BAD' not in r.stdout, r.stdout[-2000:]


def test_shared_host_tensor_roundtrip():
    """The tile-parallel Predictor's shared output buffer (host logic, no GPU): a second mapping of the same /dev/shm file sees the
    writes of the first, the name can be unlinked while mappings live, dtype/shape views are exact."""
    import os
    from elektronn3_amd.inference import _SharedHostTensor
    a = _SharedHostTensor.create((2, 3, 4, 5, 6), torch.float32)
    b = _SharedHostTensor.open(a.name, (2, 3, 4, 5, 6), torch.float32)
    a.tensor.zero_()
    b.tensor[1, 2, 3] = 7.5
    assert float(a.tensor[1, 2, 3, 4, 5]) == 7.5 and float(a.tensor.sum()) == 7.5 * 30
    a.unlink_if_owner(); b.unlink_if_owner()
    assert not os.path.exists(a.name)
    a.tensor[0, 0, 0, 0, 0] = 1.0
    assert float(b.tensor[0, 0, 0, 0, 0]) == 1.0          # mappings outlive the name
    u = _SharedHostTensor.create((1, 1, 7, 9, 11), torch.uint8)
    assert u.tensor.dtype == torch.uint8 and tuple(u.tensor.shape) == (1, 1, 7, 9, 11)
    u.unlink_if_owner()


@pytest.mark.gpu
def test_predictor_valid_conv_model_offset_path():
    """conv_mode='valid' model through the Predictor's offset logic (inference.py:476-494, tiled_apply without padding): the output is
    smaller than the input by 2*offset, tiles of the OUTPUT grid, each fed its input tile incl. the offset border; equals the
    hand-rolled tiles and the untiled forward of the whole volume (valid convs see no padding, so tiling is exact here)."""
    from elektronn3_amd.inference import Predictor, calculate_offset
    from elektronn3_amd.unet import UNet
    torch.manual_seed(21)
    m = UNet(1, 2, n_blocks=2, start_filts=8, conv_mode='valid').cuda()
    m.train()
    with torch.no_grad():
        m(torch.randn(2, 1, 40, 40, 40, device='cuda'))
    m.eval()
    off = calculate_offset(m, tile_shape=(48, 48, 48))
    assert (off > 0).all()
    tile = np.array([8, 12, 16])
    S = 2 * tile + 2 * off                                    # 2 x 2 x 2 output tiles
    vol = torch.randn(1, 1, *S)
    pred = Predictor(m, device='cuda', tile_shape=tuple(tile), offset=tuple(off), out_shape=(2, *S), apply_softmax=True)
    y = pred.predict(vol)
    assert tuple(y.shape) == (1, 2, *(2 * tile))
    with torch.no_grad():
        whole = torch.softmax(m(vol.cuda()), 1).cpu()
        assert tuple(whole.shape) == tuple(y.shape)
        full = torch.zeros_like(y)
        for iz in range(2):
            for iy in range(2):
                for ix in range(2):
                    lo = tile * np.array([iz, iy, ix]); hi = lo + tile + 2 * off
                    t = vol[:, :, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].cuda()
                    o = torch.softmax(m(t), 1).cpu()
                    assert tuple(o.shape[2:]) == tuple(tile)
                    full[:, :, lo[0]:lo[0] + tile[0], lo[1]:lo[1] + tile[1], lo[2]:lo[2] + tile[2]] = o
    assert torch.allclose(y, full, rtol=1e-5, atol=1e-6)
    assert torch.allclose(y, whole, rtol=1e-4, atol=1e-5)
    # offset=None: estimated by a probe forward, like the reference does for unknown models
    y2 = Predictor(m, device='cuda', tile_shape=tuple(tile), offset=None, out_shape=(2, *S), apply_softmax=True).predict(vol)
    assert torch.equal(y2, y)


@pytest.mark.gpu
def test_predictor_bf16_module_runs_the_native_bf16_kernels_and_tracks_fp32():
    """A module that lives in bfloat16 (model.to(torch.bfloat16): the bf16 counterpart of Predictor(float16=True), inference.py:445-446) gets
    bfloat16 tiles, i.e. the native bf16 eval path (BN folded into the conv epilogues, softmax fused into the head); the tiled softmax volume
    stays within bf16 accumulation noise of the fp32 Predictor's and the default result dtype is the compute dtype."""
    import copy
    from elektronn3_amd.inference import Predictor
    from elektronn3_amd.unet import UNet
    torch.manual_seed(3)
    m = UNet(1, 2, n_blocks=3, start_filts=32).to('cuda:0')
    m.train()
    with torch.no_grad():
        for _ in range(3):
            m(torch.randn(2, 1, 16, 32, 32, device='cuda:0'))
    m.eval()
    vol = torch.randn(1, 1, 40, 72, 88)
    kw = dict(device='cuda:0', tile_shape=(16, 32, 32), overlap_shape=(8, 8, 8), out_shape=(2, 40, 72, 88), apply_softmax=True, strict_shapes=False)
    y32 = Predictor(m, **kw).predict(vol)
    mb = copy.deepcopy(m).to(torch.bfloat16)
    pb = Predictor(mb, **kw)
    assert pb.dtype == torch.bfloat16
    yb = pb.predict(vol)
    assert yb.dtype == torch.bfloat16 and tuple(yb.shape) == tuple(y32.shape)
    err = (yb.float() - y32.float()).abs()
    assert float(err.max()) < 4e-2 and float(err.mean()) < 4e-3, (float(err.max()), float(err.mean()))
    yf = Predictor(mb, out_dtype=torch.float32, **kw).predict(vol)
    assert yf.dtype == torch.float32 and float((yf - yb.float()).abs().max()) < 8e-3
    # the fp32 module is untouched and a float16 request still takes the reference's half-precision path
    assert next(m.parameters()).dtype == torch.float32


@pytest.mark.gpu
@pytest.mark.parametrize('cfg', [
    dict(n_blocks=3, start_filts=32, shape=(48, 96, 96), roi=((8, 40), (16, 80), (16, 80))),       # persistent Winograd kernel at the top level
    dict(n_blocks=4, start_filts=32, shape=(64, 64, 96), roi=((16, 48), (16, 48), (16, 80))),      # the Predictor's crop: overlap 16 of every side
    dict(n_blocks=2, start_filts=32, shape=(23, 45, 53), roi=((3, 17), (5, 41), (17, 30))),        # odd sizes, box not on brick edges
    dict(n_blocks=3, start_filts=16, shape=(32, 64, 64), roi=((0, 32), (0, 64), (20, 44))),        # region that touches the tensor's faces
])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_forward_roi_equals_the_whole_forward_inside_the_region(cfg, dtype):
    """UNet.forward_roi / e3_unet_forward_roi (fp32), _roi_bf16, _roi_f16: bit-identical to the whole forward inside the region (the same
    bricks compute the same voxels; the others are skipped), with and without the fused softmax."""
    from elektronn3_amd.unet import UNet
    if dtype != torch.float32 and cfg['start_filts'] % 32:
        pytest.skip('the native 16-bit path needs start_filts % 32 == 0')
    torch.manual_seed(11)
    m = UNet(in_channels=1, out_channels=2, n_blocks=cfg['n_blocks'], start_filts=cfg['start_filts'], normalization='batch').cuda()
    for mod in m.modules():          # eval-mode BatchNorm with non-trivial running statistics
        if isinstance(mod, nn.BatchNorm3d):
            mod.running_mean.normal_(0, 0.2); mod.running_var.uniform_(0.5, 1.5)
    m.eval()
    m = m.to(dtype)
    x = torch.randn(2, 1, *cfg['shape'], device='cuda').to(dtype)
    sl = (slice(None), slice(None)) + tuple(slice(a, b) for a, b in cfg['roi'])
    with torch.no_grad():
        for sm in (False, True):
            whole = m.forward_softmax(x) if sm else m(x)
            # poison the scratch arena between the calls (every byte 0xFF = NaN in fp32, bf16 and float16): a needed voxel whose producer
            # was skipped would otherwise read the whole forward's stale-but-correct value of the same buffer and pass
            from elektronn3_amd import unet as unet_mod
            torch.cuda.synchronize()
            assert unet_mod._scratch, 'the forward keeps its scratch arena cached per (device, stream)'
            for buf in unet_mod._scratch.values():
                buf.fill_(0xFF)
            part = m.forward_roi(x, cfg['roi'], softmax=sm)
            assert part.shape == whole.shape
            assert torch.equal(part[sl], whole[sl])
            assert torch.isfinite(part[sl].float()).all()


@pytest.mark.gpu
def test_predictor_needed_region_switch_gives_the_same_volume():
    """Predictor.predict with the needed-region forward (default) and with whole tiles (E3_PREDICTOR_NO_ROI): identical volumes, in the
    pipelined host<->device path and in the device-resident one."""
    from elektronn3_amd import inference
    from elektronn3_amd.unet import UNet
    torch.manual_seed(5)
    m = UNet(in_channels=1, out_channels=2, n_blocks=3, start_filts=32, normalization='batch').cuda().eval()
    vol = torch.randn(1, 1, 64, 128, 160)
    outs = {}
    for roi in (True, False):
        inference._ROI = roi
        try:
            p = inference.Predictor(m, device='cuda', tile_shape=(32, 64, 80), overlap_shape=(16, 16, 16), out_shape=(2, 64, 128, 160),
                                    apply_softmax=True)
            outs[roi, 'pipe'] = p.predict(vol).clone()
            outs[roi, 'dev'] = p.predict(vol.cuda()).cpu()
        finally:
            inference._ROI = True
    assert torch.equal(outs[True, 'pipe'], outs[False, 'pipe'])
    assert torch.equal(outs[True, 'dev'], outs[False, 'dev'])
    assert torch.equal(outs[True, 'pipe'], outs[True, 'dev'])


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_predictor_pinned_staging_switch_gives_the_same_volume(monkeypatch, dtype):
    """The host <-> device pipeline through the page-locked staging rings (default; slabs larger than a 256 MB slot are split, a fp32 volume for
    a 16-bit model is converted on the device) and through the runtime's pageable copies (E3_PREDICTOR_NO_PINNED=1): identical volumes."""
    from elektronn3_amd import inference
    from elektronn3_amd.unet import UNet
    torch.manual_seed(6)
    m = UNet(in_channels=1, out_channels=2, n_blocks=2, start_filts=32, normalization='batch').cuda().eval().to(dtype)
    vol = torch.randn(1, 1, 40, 96, 130)                      # fp32 host volume, odd width (cropped rows on the way back)
    outs = []
    for pinned in ('rings with device twins', 'rings alone', False):        # (device twins of the slots: the default since round 6; without: torch's own temporaries)
        if pinned:
            monkeypatch.delenv('E3_PREDICTOR_NO_PINNED', raising=False)
            dev = torch.device('cuda:0') if pinned == 'rings with device twins' else None
            monkeypatch.setattr(inference, '_RINGS', {0: (inference._PinnedRing(3 << 20, dev), inference._PinnedRing(1 << 20, dev))})   # small slots: several slabs per row
        else:
            monkeypatch.setenv('E3_PREDICTOR_NO_PINNED', '1')
        p = inference.Predictor(m, device='cuda', tile_shape=(20, 48, 65), overlap_shape=(8, 8, 8), out_shape=(2, 40, 96, 130), apply_softmax=True)
        assert p.dtype == dtype
        outs.append(p.predict(vol).clone())
    assert outs[0].dtype == outs[1].dtype == outs[2].dtype and torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[2])


@pytest.mark.gpu
def test_forward_roi_resunet_and_wider_channel_counts():
    """ResUNet (residual units stop the region's way back through the decoder) and a model with several input / output channels."""
    from elektronn3_amd.unet import UNet
    from elektronn3_amd import resunet
    torch.manual_seed(4)
    for m, cin in ((resunet.UNet(in_channels=1, out_channels=2, n_blocks=3, start_filts=16, enc_res_blocks=1, dec_res_blocks=1), 1),
                   (UNet(in_channels=2, out_channels=5, n_blocks=3, start_filts=32), 2)):
        m = m.cuda().eval()
        x = torch.randn(1, cin, 40, 64, 96, device='cuda')
        roi = ((8, 32), (16, 48), (16, 80))
        with torch.no_grad():
            whole, part = m(x), m.forward_roi(x, roi)
        sl = (slice(None), slice(None)) + tuple(slice(a, b) for a, b in roi)
        assert torch.equal(part[sl], whole[sl])


@pytest.mark.gpu
@pytest.mark.parametrize('kw', [
    dict(planar_blocks=(0,)),                       # planar level 0: the box propagates through 1x3x3 convs / (1,2,2) transposed convs, the planar kernels ignore it
    dict(attention=True),                           # no needed region with attention gates
    dict(conv_mode='valid'),                        # ... nor with valid convs
    dict(up_mode='resizeconv_nearest'),             # ResizeConv: everything in front of it in full
    dict(activation='leaky'),                       # two-pass units (conv, then the apply pass over the whole tensor)
    dict(merge_mode='add'),
])
def test_forward_roi_on_configurations_without_the_facility(kw):
    """Configurations whose kernels do not take a needed region compute (at least) what the region needs: same values inside it."""
    from elektronn3_amd.unet import UNet
    torch.manual_seed(3)
    m = UNet(in_channels=1, out_channels=2, n_blocks=3, start_filts=16, normalization='batch', **kw).cuda().eval()
    x = torch.randn(1, 1, 44, 60, 76, device='cuda')
    with torch.no_grad():
        whole = m(x)
        D, H, W = whole.shape[2:]
        roi = ((1, D - 1), (2, H - 1), (3, W - 2))
        part = m.forward_roi(x, roi)
    sl = (slice(None), slice(None)) + tuple(slice(a, b) for a, b in roi)
    assert part.shape == whole.shape and torch.equal(part[sl], whole[sl])


def test_threaded_host_copy_is_a_copy():
    """inference._host_copy (slabs above 64 MB are split along z over four threads) against Tensor.copy_, contiguous and strided views."""
    from elektronn3_amd import inference
    src = torch.arange(1 * 2 * 20 * 512 * 1024, dtype=torch.float32).view(1, 2, 20, 512, 1024)          # 80 MB
    dst = torch.empty_like(src)
    inference._host_copy(dst, src)
    assert torch.equal(dst, src)
    big = torch.zeros(1, 2, 24, 600, 1100)
    inference._host_copy(big[:, :, 2:22, 40:552, 30:1054], src)                                          # strided destination
    assert torch.equal(big[:, :, 2:22, 40:552, 30:1054], src)
    big[:, :, 2:22, 40:552, 30:1054] = 0
    assert not big.any()                    # nothing outside the view was touched
    small = torch.randn(1, 1, 3, 8, 8); out = torch.empty_like(small)
    inference._host_copy(out, small)
    assert torch.equal(out, small)


@pytest.mark.gpu
@pytest.mark.parametrize('vshape,tile', [((2, 1, 50, 100, 150), (32, 64, 80)), ((1, 1, 44, 150, 170), (40, 128, 128))],
                         ids=['small_tiles', 'tiles_of_1400_first_conv_bricks'])       # (the second case reaches conv_first_mfma_kernel through the strides of the padded volume)
def test_predictor_in_place_tiles_switch_gives_the_same_volume(monkeypatch, vshape, tile):
    """Tiles read in place from the padded volume and written in place into the output volume (UNet.forward_tile / e3_unet_forward_tile, the default
    for the native fp32 model) against the copied-tile path (E3_PREDICTOR_NO_INPLACE=1): identical volumes, also with a ragged last tile."""
    from elektronn3_amd import inference
    from elektronn3_amd.unet import UNet
    torch.manual_seed(8)
    m = UNet(in_channels=1, out_channels=3, n_blocks=3, start_filts=32, normalization='batch').cuda().eval()
    vol = torch.randn(*vshape)                                # not a multiple of the tile shape: padded, cropped on the way back
    outs = []
    calls = []
    real = UNet.forward_tile
    monkeypatch.setattr(UNet, 'forward_tile', lambda self, *a, **k: (calls.append(1), real(self, *a, **k))[1])
    for in_place in (True, False):
        if in_place:
            monkeypatch.delenv('E3_PREDICTOR_NO_INPLACE', raising=False)
        else:
            monkeypatch.setenv('E3_PREDICTOR_NO_INPLACE', '1')
        p = inference.Predictor(m, device='cuda', tile_shape=tile, overlap_shape=(8, 16, 16), out_shape=(3, *vshape[2:]), apply_softmax=True,
                                strict_shapes=False)
        n0 = len(calls)
        outs.append(p.predict(vol).clone())
        assert (len(calls) > n0) == in_place
    assert torch.equal(outs[0], outs[1])
    assert torch.allclose(outs[0].sum(1), torch.ones_like(outs[0][:, 0]), atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16], ids=['fp32_in_place', 'bf16_copied_tiles'])
def test_predictor_edge_tiles_compute_only_what_lies_inside_the_volume(monkeypatch, dt):
    """The last tile of an axis hangs over the end of the volume (the padded volume is a multiple of the tile shape, inference.py:153-199); its kept crop is
    clipped to the real volume before it becomes the needed region (E3_PREDICTOR_NO_CLIP=1 / inference._NO_CLIP: the whole crop).  What predict() returns is
    the real volume only: both forms must give the identical tensor."""
    from elektronn3_amd import inference
    from elektronn3_amd.unet import UNet
    torch.manual_seed(9)
    m = UNet(in_channels=1, out_channels=2, n_blocks=3, start_filts=32, normalization='batch').cuda().eval()
    if dt != torch.float32:
        m = m.to(dt)
    vol = torch.randn(1, 1, 41, 150, 171)                       # 41 = 1.3 x 32, 150 = 2.3 x 64, 171 = 2.1 x 80: every axis ends inside a tile
    outs = []
    for no_clip in (False, True):
        monkeypatch.setattr(inference, '_NO_CLIP', no_clip)
        p = inference.Predictor(m, device='cuda', tile_shape=(32, 64, 80), overlap_shape=(8, 16, 16), out_shape=(2, 41, 150, 171), apply_softmax=True,
                                strict_shapes=False, float16=False)
        outs.append(p.predict(vol.to(dt) if dt != torch.float32 else vol).clone())
    assert outs[0].shape[-3:] == (41, 150, 171)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.gpu
def test_frozen_weights_scope_packs_once_and_never_serves_stale_weights():
    """UNet.frozen_weights() (the Predictor's tile loop runs inside it): inference forwards of one shape reuse the packed weights of the first one
    (E3_FWD_REUSE_PACKED) -- bit-identical outputs --, and every way out of the promise packs again: another shape, a training call in between, leaving the
    scope, a parameter changed OUTSIDE a scope (also through ``.data``, which bumps no version counter: training/swa.py:182-202 swaps weights that way)."""
    from elektronn3_amd.unet import UNet
    from elektronn3_amd import unet as unet_mod
    torch.manual_seed(3)
    m = UNet(1, 2, n_blocks=2, start_filts=32).cuda()
    with torch.no_grad():
        m.train(); m(torch.randn(2, 1, 8, 16, 32, device='cuda')); m.eval()
        x = torch.randn(1, 1, 16, 32, 48, device='cuda'); x2 = torch.randn(1, 1, 8, 16, 32, device='cuda')
        ref, ref2 = m(x).clone(), m(x2).clone()
        key = next(iter(unet_mod._scratch))
        with m.frozen_weights():
            a = m(x).clone()
            assert unet_mod._packed.get(key) is not None
            b = m(x).clone()                 # reuses
            c = m(x2).clone()                # another shape: packs again
            d = m(x).clone()                 # ... and again
            r = m.forward_roi(x, ((2, 14), (4, 28), (8, 40))).clone()
        for t_ in (a, b, d):
            assert torch.equal(t_, ref)
        assert torch.equal(c, ref2)
        assert torch.equal(r[:, :, 2:14, 4:28, 8:40], ref[:, :, 2:14, 4:28, 8:40])
        assert m.__dict__.get('_frozen_scope') is None
        m.down_convs[0].conv2.weight.data.mul_(1.5)          # outside a scope: the next forward must see it
        changed = m(x)
        assert not torch.equal(changed, ref)
        with m.frozen_weights():
            e = m(x).clone()
            m.train(); m(x2); m.eval()       # a training call clears the buffer's token (and moves the running statistics)
            f = m(x).clone()
        assert torch.equal(e, changed) and not torch.equal(f, e)


@pytest.mark.gpu
def test_bench_volume_tiles_against_fp64():
    """BASELINE.json configs[4] at ITS size: bench.py's Predictor leg -- the full 512x2048x2048 synthetic volume, tile 96x192x192, overlap 16, 726 tiles
    through the pipelined native path -- and then four of its tiles against the fp64 op sequence + softmax on the same (zero-padded) input tiles:
    the near corner, the far corner (ragged along every axis: the volume is not a multiple of the tile), a face tile and an interior tile
    (the full-size run was only checked for finiteness).  Tolerance 1e-4 on probabilities (fp32 against fp64)."""
    import bench
    from elektronn3_amd.inference import Predictor
    from elektronn3_amd.unet import UNet
    from oracle.torch_ref import unet_forward
    try:
        import psutil
        if psutil.virtual_memory().available < 40 * 2 ** 30:
            pytest.skip('needs ~30 GB of host memory for the volume and its two-class result')
    except ImportError:
        pass
    shape, tile, ov = (512, 2048, 2048), (96, 192, 192), (16, 16, 16)
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = UNet(1, 2, n_blocks=4, start_filts=32).to(dev).train()
    with torch.no_grad():
        for _ in range(10):
            model(torch.randn(2, 1, 32, 64, 64, device=dev))
    model.eval()
    vol = torch.empty(1, 1, *shape)
    gen = torch.Generator().manual_seed(0)
    for z in range(0, shape[0], 32):
        vol[0, 0, z:z + 32].normal_(generator=gen)
    pred = Predictor(model, device=dev, tile_shape=tile, overlap_shape=ov, offset=None, out_shape=(2, *shape), apply_softmax=True, strict_shapes=False)
    out = pred.predict(vol)
    assert tuple(out.shape) == (1, 2, *shape)
    ntile = [-(-n // t) for n, t in zip(shape, tile)]
    assert ntile == [6, 11, 11]
    sd = {k: v.double() if v.is_floating_point() else v for k, v in model.state_dict().items()}
    worst = 0.0
    for idx in [(0, 0, 0), (5, 10, 10), (0, 5, 5), (2, 5, 5)]:
        lo = [i * t - o for i, t, o in zip(idx, tile, ov)]                    # input tile in volume coordinates (may hang over every face)
        tin = torch.zeros(1, 1, *[t + 2 * o for t, o in zip(tile, ov)])
        src = [slice(max(l, 0), min(l + t + 2 * o, n)) for l, t, o, n in zip(lo, tile, ov, shape)]
        dst = [slice(s.start - l, s.stop - l) for s, l in zip(src, lo)]
        tin[(0, 0, *dst)] = vol[(0, 0, *src)]
        with torch.no_grad():
            ref = torch.softmax(unet_forward(sd, tin.to(dev).double(), 4, (), training=False), 1)[:, :, ov[0]:ov[0] + tile[0], ov[1]:ov[1] + tile[1], ov[2]:ov[2] + tile[2]].cpu()
        olo = [i * t for i, t in zip(idx, tile)]
        ohi = [min(l + t, n) for l, t, n in zip(olo, tile, shape)]
        got = out[:, :, olo[0]:ohi[0], olo[1]:ohi[1], olo[2]:ohi[2]].double()
        want = ref[:, :, :ohi[0] - olo[0], :ohi[1] - olo[1], :ohi[2] - olo[2]]
        err = float((got - want).abs().max())
        worst = max(worst, err)
        assert err < 1e-4, (idx, err)
        del ref, tin
    print(f'bench volume, 4 of 726 tiles against fp64: worst |p - p64| = {worst:.2e}')
    assert bench.CROP == (64, 128, 128)
