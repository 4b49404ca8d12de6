"""Pins the CPU oracle (oracle/) against golden vectors produced by the imported reference
(tests/golden/make_golden.py).  CPU only.

Tolerances: the reference's own fp32-vs-fp64 noise floor is ~5e-6..4e-5 max-abs on
logits and 2e-4..4e-3 rel-L2 on weight gradients.  The oracle accumulates in double, so it is
compared (a) against the reference's fp32 results within that floor and (b) against the
reference's fp64 results, where it must be at least as close as the fp32 reference is.
"""
import numpy as np
import pytest

from helpers import combined_loss_np, embed_2d, instance_norm_names, is_prebn_bias, load_npz, rel_l2, sub, unet_cfg
from oracle import unet_oracle as orc


@pytest.fixture(scope='module')
def ops():
    return load_npz('ops.npz')


@pytest.mark.parametrize('tag,pad', [('conv3', (1, 1, 1)), ('conv3p', (0, 1, 1)), ('conv3c1', (1, 1, 1)), ('conv1', (0, 0, 0))])
def test_conv(ops, tag, pad):
    x, w, b = ops[f'{tag}.x'], ops[f'{tag}.w'], ops[f'{tag}.b']
    y = orc.conv3d_fwd(x, w, b, pad)
    np.testing.assert_allclose(y, ops[f'{tag}.y'], rtol=1e-5, atol=2e-5)
    dx, dw, db = orc.conv3d_bwd(x, w, ops[f'{tag}.dy'], pad)
    np.testing.assert_allclose(dx, ops[f'{tag}.dx'], rtol=1e-5, atol=2e-5)
    assert rel_l2(dw, ops[f'{tag}.dw']) < 1e-5
    assert rel_l2(db, ops[f'{tag}.db']) < 1e-5


@pytest.mark.parametrize('tag', ['convT', 'convTp'])
def test_convT(ops, tag):
    x, w, b = ops[f'{tag}.x'], ops[f'{tag}.w'], ops[f'{tag}.b']
    np.testing.assert_allclose(orc.convT_fwd(x, w, b), ops[f'{tag}.y'], rtol=1e-5, atol=2e-5)
    dx, dw, db = orc.convT_bwd(x, w, ops[f'{tag}.dy'])
    np.testing.assert_allclose(dx, ops[f'{tag}.dx'], rtol=1e-5, atol=2e-5)
    assert rel_l2(dw, ops[f'{tag}.dw']) < 1e-5
    assert rel_l2(db, ops[f'{tag}.db']) < 1e-5


def test_batchnorm_relu(ops):
    rm, rv = ops['bn.rm0'].copy(), ops['bn.rv0'].copy()
    z, mean, invstd = orc.bn_train_fwd(ops['bn.x'], ops['bn.gamma'], ops['bn.beta'], rm, rv)
    np.testing.assert_allclose(z, ops['bn.z'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rm, ops['bn.rm1'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(rv, ops['bn.rv1'], rtol=1e-6, atol=1e-6)
    a = orc.relu_fwd(z)
    np.testing.assert_allclose(a, ops['bn.a'], rtol=1e-5, atol=1e-5)
    dz = orc.relu_bwd(ops['bn.da'], ops['bn.a'])
    dx, dg, db = orc.bn_train_bwd(dz, ops['bn.x'], ops['bn.gamma'], mean, invstd)
    np.testing.assert_allclose(dx, ops['bn.dx'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dg, ops['bn.dgamma'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(db, ops['bn.dbeta'], rtol=1e-4, atol=1e-4)
    ze = orc.bn_eval_fwd(ops['bn.x'], ops['bn.gamma'], ops['bn.beta'], ops['bn.rm1'], ops['bn.rv1'])
    np.testing.assert_allclose(ze, ops['bn.z_eval'], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('tag,k', [('pool', (2, 2, 2)), ('poolp', (1, 2, 2))])
def test_maxpool_ceil(ops, tag, k):
    x = ops[f'{tag}.x']
    y, idx = orc.maxpool_fwd(x, k)
    np.testing.assert_array_equal(y, ops[f'{tag}.y'])  # bit-exact: pure selection
    dx = orc.maxpool_bwd(ops[f'{tag}.dy'], idx, x.shape, k)
    np.testing.assert_array_equal(dx, ops[f'{tag}.dx'])


def test_softmax(ops):
    np.testing.assert_allclose(orc.softmax_c(ops['softmax.x']), ops['softmax.y'], rtol=1e-6, atol=1e-7)


CASES = ['unet_nb2_sf8.npz', 'unet_nb3_sf8_planar0_odd.npz', 'unet_nb4_sf8_planar01.npz', 'unet2d_nb3_sf8_odd.npz',
         'unet_nb2_sf8_nonorm.npz', 'unet_nb3_sf8_planar0_sparsenorm.npz', 'unet_nb3_sf8_add_odd.npz', 'unet_nb3_sf8_instance.npz',
         'unet_nb3_sf8_group4_odd.npz', 'unet_nb3_sf8_leaky_odd.npz', 'unet_nb2_sf8_lin_nonorm.npz',
         'unet_nb3_sf8_silu_odd.npz', 'unet_nb3_sf8_resizeconv_odd.npz',
         'unet_nb3_sf8_resizelinear_odd.npz', 'unet_nb3_sf8_resizenearest1_odd.npz',
         'unet_nb3_sf8_prelu_odd.npz', 'unet_nb3_sf8_valid.npz']


@pytest.mark.parametrize('case', CASES)
def test_unet_train_step(case):
    g = load_npz(case)
    cfg = unet_cfg(g)
    if cfg.get('dim', 3) == 2:      # dim=2 fixture on the 3D oracle: depth-1 volume, every block planar
        net = orc.OracleUNet(embed_2d(sub(g, 'sd0')), cfg['n_blocks'], tuple(range(cfg['n_blocks'])))
        logits = net.forward(g['x'][:, :, None])[:, :, 0]
    else:
        net = orc.OracleUNet(sub(g, 'sd0'), cfg['n_blocks'], cfg['planar_blocks'], normalization=cfg.get('normalization', 'batch'))
        net.instance_norms = instance_norm_names(cfg)
        net.valid = cfg.get('conv_mode') == 'valid'
        net.up_linear = str(cfg.get('up_mode')).startswith('resizeconv_linear')
        net.act_slope = {'relu': 0.0, 'leaky': 0.1, 'lin': 1.0, 'silu': 2.0, 'prelu': 3.0}[cfg.get('activation', 'relu')]
        logits = net.forward(g['x'])
    # forward: fp32 reference noise floor is <= 4e-5 max-abs
    np.testing.assert_allclose(logits, g['logits'], rtol=1e-4, atol=1e-4)
    assert np.abs(logits - g['logits64']).max() <= max(2 * np.abs(g['logits'] - g['logits64']).max(), 2e-6)
    # BN running statistics after the step
    for k, v in sub(g, 'sd1').items():
        if k.endswith('num_batches_tracked'):
            assert int(net.sd[k]) == int(v)
        else:
            np.testing.assert_allclose(net.sd[k], v, rtol=1e-5, atol=1e-6, err_msg=k)
    # loss + dlogits restated in numpy (tests/helpers.py) against the reference criterion
    loss, dlogits = combined_loss_np(logits, g['target'])
    assert abs(loss - float(g['loss'])) < 1e-5
    np.testing.assert_allclose(dlogits, g['dlogits'], rtol=1e-3, atol=1e-9)
    if cfg.get('dim', 3) == 2:
        grads, _ = net.backward(g['dlogits'][:, :, None])
        grads = {k: v.reshape(g['grad/' + k].shape) for k, v in grads.items()}
    else:
        grads, _ = net.backward(g['dlogits'])
    ref32, ref64 = sub(g, 'grad'), sub(g, 'grad64')
    assert set(grads) == set(ref32)
    gnorm = np.sqrt(sum(float(np.sum(v.astype(np.float64) ** 2)) for v in ref64.values()))
    for k in ref32:
        # (a conv bias in front of a GroupNorm has a REAL gradient: only the group mean is removed, not the channel's own shift)
        if is_prebn_bias(k, set() if str(cfg.get('normalization')).startswith('group') else set(ref32), instance_norm_names(cfg)):  # analytically zero gradient (bias feeding a train-mode BN): absolute tolerance only
            assert np.abs(grads[k]).max() <= 1e-5 * gnorm, k
            continue
        err_o = rel_l2(grads[k], ref64[k])
        err_r = rel_l2(ref32[k], ref64[k])
        assert err_o <= max(3 * err_r, 1e-4), (k, err_o, err_r)


@pytest.mark.parametrize('case', ['unet_nb3_sf8_attention_odd.npz', 'unet2d_nb3_sf8_attention.npz', 'unet_nb3_sf8_attention_valid_planar0.npz',
                                  'unet_nb3_sf8_attention_add.npz'])
def test_torch_restatement_of_grid_attention_train_step(case):
    """attention=True is outside the C oracle; its checker is oracle/torch_ref.py's ATen restatement of GridAttention.forward
    (unet.py:509-530).  Pinned here: logits, loss, every gradient and the running statistics of the reference's own train step."""
    import torch
    from oracle.torch_ref import combined_loss, unet_forward
    g = load_npz(case)
    cfg = unet_cfg(g)
    sd = {k: torch.from_numpy(np.array(v)).clone() for k, v in sub(g, 'sd0').items()}
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k:
            v.requires_grad_(True)
    if cfg.get('conv_mode') == 'valid':
        sd['__valid__'] = True
    if cfg.get('activation') == 'leaky':
        sd['__act_slope__'] = 0.1
    out = unet_forward(sd, torch.from_numpy(g['x']), cfg['n_blocks'], cfg['planar_blocks'], training=True)
    np.testing.assert_allclose(out.detach().numpy(), g['logits'], rtol=1e-5, atol=1e-6)
    loss = combined_loss(out, torch.from_numpy(g['target']))
    assert abs(float(loss) - float(g['loss'])) < 1e-6
    loss.backward()
    ref = sub(g, 'grad')
    gnorm = np.sqrt(sum(float(np.sum(v.astype(np.float64) ** 2)) for v in ref.values()))
    for k, v in ref.items():
        assert np.abs(sd[k].grad.numpy() - v).max() <= 1e-5 * gnorm, k
    for k, v in sub(g, 'sd1').items():
        if 'running' in k:
            np.testing.assert_allclose(sd[k].numpy(), v, rtol=1e-5, atol=1e-7, err_msg=k)


@pytest.mark.parametrize('case', ['resunet_nb3_sf8_res00.npz', 'resunet_nb3_sf8_res21_odd.npz', 'resunet_nb3_sf8_res12_add_attention.npz',
                                  'resunet_nb2_sf8_res11_nonorm.npz'])
def test_torch_restatement_of_resunet_train_step(case):
    """elektronn3.models.resunet.UNet is outside the C oracle; its checker is oracle/torch_ref.py::resunet_forward (resunet.py:254-262,944-967
    restated with ATen ops).  Pinned here by the reference's own train step: logits, loss, every gradient, the running statistics."""
    import torch
    from oracle.torch_ref import combined_loss, resunet_forward
    g = load_npz(case)
    cfg = unet_cfg(g)
    sd = {k: torch.from_numpy(np.array(v)).clone() for k, v in sub(g, 'sd0').items()}
    for k, v in sd.items():
        if v.is_floating_point() and 'running' not in k:
            v.requires_grad_(True)
    if cfg.get('activation') == 'leaky':
        sd['__act_slope__'] = 0.1
    out = resunet_forward(sd, torch.from_numpy(g['x']), cfg['n_blocks'], cfg['planar_blocks'], True, cfg['enc_res_blocks'], cfg['dec_res_blocks'])
    np.testing.assert_allclose(out.detach().numpy(), g['logits'], rtol=1e-5, atol=1e-6)
    loss = combined_loss(out, torch.from_numpy(g['target']))
    assert abs(float(loss) - float(g['loss'])) < 1e-6
    loss.backward()
    ref = sub(g, 'grad')
    gnorm = np.sqrt(sum(float(np.sum(v.astype(np.float64) ** 2)) for v in ref.values()))
    for k, v in ref.items():
        assert np.abs(sd[k].grad.numpy() - v).max() <= 1e-5 * gnorm, k
    for k, v in sub(g, 'sd1').items():
        if 'running' in k:
            np.testing.assert_allclose(sd[k].numpy(), v, rtol=1e-5, atol=1e-7, err_msg=k)


def test_resunet_parameter_table_matches_the_reference_state_dict():
    """Host logic: elektronn3_amd.resunet.UNet has the reference's state_dict keys in the reference's order, and the native plan's table
    names every one of them (residual blocks with identity and projected shortcuts, attention)."""
    import torch
    from elektronn3_amd.resunet import UNet
    for case in ['resunet_nb3_sf8_res21_odd.npz', 'resunet_nb3_sf8_res12_add_attention.npz']:
        g = load_npz(case)
        sd = {k: torch.from_numpy(np.array(v)) for k, v in sub(g, 'sd0').items()}
        m = UNet(1, 2, **unet_cfg(g))
        m.load_state_dict(sd)
        assert list(m.state_dict().keys()) == list(sd.keys())
        plan = m._plan()
        assert set(plan.names) == {k for k in sd if not k.endswith('num_batches_tracked')}
        assert any(n.endswith('.proj.weight') for n in plan.names)


def test_unet_eval_forward():
    g = load_npz('unet_nb2_sf8.npz')
    cfg = unet_cfg(g)
    sd = sub(g, 'sd0')
    sd.update(sub(g, 'sd1'))  # eval logits were produced after the train step updated the running stats
    net = orc.OracleUNet(sd, cfg['n_blocks'], cfg['planar_blocks'])
    net.training = False
    np.testing.assert_allclose(net.forward(g['x']), g['logits_eval'], rtol=1e-4, atol=1e-5)


def test_unet_eval_forward_rrelu():
    """activation='rrelu' in eval mode: nn.RReLU's fixed slope (lower + upper) / 2 = (1/8 + 1/3) / 2 (get_activation, unet.py:183-199)."""
    g = load_npz('unet_nb3_sf8_rrelu_eval.npz')
    cfg = unet_cfg(g)
    net = orc.OracleUNet(sub(g, 'sd0'), cfg['n_blocks'], cfg['planar_blocks'])
    net.act_slope = (1.0 / 8 + 1.0 / 3) / 2
    net.training = False
    np.testing.assert_allclose(net.forward(g['x']), g['logits_eval'], rtol=1e-4, atol=1e-5)


def test_predictor_tiled():
    g = load_npz('predictor.npz')
    net = orc.OracleUNet(sub(g, 'sd'), 2)
    out = orc.predict_tiled(net, g['vol'], g['tile'], g['overlap'], g['out_shape'])
    assert out.shape == g['out_tiled'].shape
    np.testing.assert_allclose(out, g['out_tiled'], rtol=1e-4, atol=1e-5)
    # tiled != untiled by construction (overlap < receptive field), SURVEY.md 8c
    assert np.abs(g['out_tiled'] - g['out_untiled']).max() > 1e-4
    plan = orc.tile_plan(np.ceil(g['out_shape'][1:] / g['tile']) * g['tile'], g['tile'], g['overlap'])
    assert len(plan) == 3 * 3 * 3 and plan[0][0] == (0, 0, 0) and plan[1][0] == (0, 0, 16)
    out3 = orc.predict_tiled(net, g['vol3'], g['tile'], g['overlap'], (2, 16, 32, 32))
    am = out3.argmax(axis=1)[:, None].astype(np.uint8)
    assert (am != g['out3_argmax']).mean() < 1e-4  # argmax may flip only where p0 ~ p1 within fp32 noise


def test_adamw_trajectory():
    """orc_adamw_step against torch.optim.AdamW's own 5-step trajectory with a changing lr (tests/golden/adamw.npz)."""
    g = load_npz('adamw.npz')
    for i in range(int(g['n'])):
        p = g[f'p0/{i}'].copy()
        m = np.zeros_like(p); v = np.zeros_like(p)
        for t, lr in enumerate(g['lrs']):
            orc.adamw_step(p, g[f'g{t}/{i}'], m, v, t + 1, lr=float(lr), weight_decay=0.5e-4)
            # m = m + 0.1 (g - m) cancels where the result crosses zero: absolute slack of a few ulp of the tensor's scale
            np.testing.assert_allclose(m, g[f'm{t + 1}/{i}'], rtol=2e-6, atol=3e-7 * np.abs(g[f'm{t + 1}/{i}']).max())
            np.testing.assert_allclose(v, g[f'v{t + 1}/{i}'], rtol=2e-6, atol=0)
            np.testing.assert_allclose(p, g[f'p{t + 1}/{i}'], rtol=2e-6, atol=1e-7)


def test_swa_running_average():
    """oracle.swa_update against the reference SWA wrapper's buffers (tests/golden/swa.npz: automatic mode, swa_start=2, swa_freq=2)."""
    g = load_npz('swa.npz')
    n, steps, start, freq = int(g['n']), int(g['steps']), int(g['swa_start']), int(g['swa_freq'])
    bufs = [np.zeros_like(g[f'p0/{i}']) for i in range(n)]
    n_avg = 0
    for t in range(1, steps + 1):
        if t > start and t % freq == 0:
            bufs = [orc.swa_update(bufs[i], g[f'p{t}/{i}'], n_avg) for i in range(n)]
            n_avg += 1
        assert n_avg == int(g[f'n_avg{t}'])
        for i in range(n):
            if f'b{t}/{i}' in g.files:
                np.testing.assert_array_equal(bufs[i], g[f'b{t}/{i}'])
    for i in range(n):      # swap_swa_sgd
        np.testing.assert_array_equal(g[f'p_swapped/{i}'], bufs[i])
        np.testing.assert_array_equal(g[f'b_swapped/{i}'], g[f'p{steps}/{i}'])


def test_torch_restatement_at_the_headline_size_against_the_reference_digest():
    """oracle/torch_ref.py (the checker of every full-size GPU parity test and bench.py's cpu_baseline) against the REFERENCE's own forward at
    BASELINE.json configs[1]'s size: tests/golden/cfg2_digest.npz holds a strided sample of the reference's train-mode logits and its loss for
    parameters / input / target that are regenerated here from the digest's seed (helpers.digest_*).  Forward + criterion only (the backward at this
    size is covered on the GPU side by test_full_size_against_the_reference_digest); ~10 s of CPU."""
    from collections import OrderedDict
    import torch
    from helpers import digest_inputs, digest_state_dict
    from oracle.torch_ref import combined_loss, unet_forward
    g = load_npz('cfg2_digest.npz')
    seed = int(g['seed'])
    shapes = OrderedDict((str(k), tuple(int(i) for i in str(sh).split(',')) if str(sh) else ()) for k, sh in zip(g['names'], g['shapes']))
    sd = {k: torch.from_numpy(v) for k, v in digest_state_dict(shapes, seed).items()}
    x_np, t_np = digest_inputs(int(g['batch']), tuple(int(v) for v in g['shape']), seed)
    with torch.no_grad():
        out = unet_forward(sd, torch.from_numpy(x_np), int(g['cfg.n_blocks']), (), training=True)
        loss = float(combined_loss(out, torch.from_numpy(t_np)))
    samp = out[:, :, ::8, ::8, ::8].numpy()
    err_ref = float(g['logits_err_ref'])
    assert float(np.abs(samp - g['logits64']).max()) <= max(3 * err_ref, 2e-5)
    np.testing.assert_allclose(samp, g['logits32'], rtol=1e-4, atol=1e-4)
    assert abs(loss - float(g['loss64'])) < 2e-5
