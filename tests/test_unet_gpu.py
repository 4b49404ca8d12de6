"""GPU parity of the whole-network path (UNet.forward / backward through the C ABI) against

  (a) the golden train steps the reference produced (tests/golden/unet_*.npz): logits, BN running statistics,
      every parameter gradient (judged against the reference's own fp64 run), eval-mode logits;
  (b) the CPU oracle on a mid-size seeded case;
  (c) at BASELINE.json's full size (cfg 2: UNet(1,2,n_blocks=4,start_filts=32), 2x1x64x128x128) the same ATen op
      sequence executed by PyTorch-ROCm on the GPU (oracle/torch_ref.py), plus size-independent properties
      (determinism, batch independence in eval mode, zero gradient of pre-BN biases).

Stated tolerances (fp32): logits atol=rtol=1e-4 vs fp32 references; gradients rel-L2 per tensor
<= max(3 x the reference's own fp32-vs-fp64 error, 1e-4) on the golden cases, <= 2e-2 on random-init full-size nets
against a second fp32 implementation (gradient parity is ill-conditioned there: SURVEY.md 7 "hard parts").
"""
import os
import numpy as np
import pytest
import torch

from helpers import instance_norm_names, is_prebn_bias, load_npz, rel_l2, sub, unet_cfg

pytestmark = pytest.mark.gpu

CASES = ['unet_nb2_sf8.npz', 'unet_nb3_sf8_planar0_odd.npz', 'unet_nb4_sf8_planar01.npz', 'unet2d_nb3_sf8_odd.npz',
         'unet_nb2_sf8_nonorm.npz', 'unet_nb3_sf8_planar0_sparsenorm.npz', 'unet_nb3_sf8_add_odd.npz', 'unet_nb3_sf8_instance.npz',
         'unet_nb3_sf8_group4_odd.npz', 'unet_nb3_sf8_leaky_odd.npz', 'unet_nb2_sf8_lin_nonorm.npz',
         'unet_nb3_sf8_silu_odd.npz', 'unet_nb3_sf8_resizeconv_odd.npz',
         'unet_nb3_sf8_resizelinear_odd.npz', 'unet_nb3_sf8_resizenearest1_odd.npz',
         'unet_nb3_sf8_prelu_odd.npz', 'unet_nb3_sf8_valid.npz',
         # attention=True (GridAttention): odd sizes (both resizes are real interpolations), dim=2, conv_mode='valid' + planar block, merge 'add'
         'unet_nb3_sf8_attention_odd.npz', 'unet2d_nb3_sf8_attention.npz', 'unet_nb3_sf8_attention_valid_planar0.npz',
         'unet_nb3_sf8_attention_add.npz',
         # elektronn3.models.resunet.UNet: plain ConvBlocks; residual ones (2 per encoder block + planar + odd; with attention, 'add', leaky; no norm)
         'resunet_nb3_sf8_res00.npz', 'resunet_nb3_sf8_res21_odd.npz', 'resunet_nb3_sf8_res12_add_attention.npz', 'resunet_nb2_sf8_res11_nonorm.npz',
         # start_filts=32 at a size where the fp32 Winograd kernels run (persistent kernel at level 0, plain kernel at level 1, Winograd wgrad), all-odd extents:
         # closes the chain reference -> fixture -> HIP path for the kernels that carry 85 % of the headline step
         'unet_nb2_sf32_wino_odd.npz']


def build(cfg, sd_np):
    from elektronn3_amd.unet import UNet
    if 'enc_res_blocks' in cfg:         # fixture of the reference's models/resunet.py
        from elektronn3_amd.resunet import UNet
    m = UNet(in_channels=1, out_channels=2, **cfg)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()}
    m.load_state_dict(sd)           # reference key names and shapes must match exactly
    return m.cuda()


@pytest.mark.parametrize('case', CASES)
def test_train_step_matches_reference(case):
    from oracle.torch_ref import combined_loss
    g = load_npz(case)
    cfg = unet_cfg(g)
    m = build(cfg, sub(g, 'sd0'))
    m.train()
    x = torch.from_numpy(g['x']).cuda()
    t = torch.from_numpy(g['target']).cuda()
    out = m(x)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g['logits'], rtol=1e-4, atol=1e-4)
    err_build = np.abs(out.detach().cpu().numpy() - g['logits64']).max()
    err_ref = np.abs(g['logits'] - g['logits64']).max()
    assert err_build <= max(3 * err_ref, 2e-5), (err_build, err_ref)
    loss = combined_loss(out, t)
    assert abs(float(loss.detach()) - float(g['loss'])) < 2e-5
    loss.backward()
    sd = m.state_dict()
    for k, v in sub(g, 'sd1').items():
        if k.endswith('num_batches_tracked'):
            assert int(sd[k]) == int(v)
        else:
            np.testing.assert_allclose(sd[k].cpu().numpy(), v, rtol=1e-5, atol=1e-6, err_msg=k)
    ref32, ref64 = sub(g, 'grad'), sub(g, 'grad64')
    gnorm = np.sqrt(sum(float(np.sum(v.astype(np.float64) ** 2)) for v in ref64.values()))

    def grad_errors(model):
        gr = {k: p.grad.detach().cpu().numpy() for k, p in model.named_parameters()}
        assert set(gr) == set(ref32)
        errs = {}
        for k in ref32:
            assert gr[k].shape == ref32[k].shape, k
            # analytically zero (bias feeding a train-mode BN / InstanceNorm; NOT for GroupNorm): absolute tolerance only
            # (conv_mode='valid' + attention: the gate's BatchNorm shift is a per-channel constant that an un-padded conv1 hands to a
            # train-mode norm1, which removes it -- also analytically zero)
            att_beta_zero = (cfg.get('conv_mode') == 'valid' and k.endswith('.attention.w.1.bias')
                             and k.replace('attention.w.1.bias', 'norm1.weight') in ref32)
            if att_beta_zero or is_prebn_bias(k, set() if str(cfg.get('normalization')).startswith('group') else set(ref32), instance_norm_names(cfg)):
                assert np.abs(gr[k]).max() <= 1e-5 * gnorm, (k, np.abs(gr[k]).max())
                continue
            errs[k] = (rel_l2(gr[k], ref64[k]), rel_l2(ref32[k], ref64[k]))
        return errs

    # Gradients are piecewise smooth in the activations: one ReLU / arg-max decision on an element whose pre-activation is ~1e-7 flips under ANY fp32-level
    # perturbation, and a flipped decision moves whole gradient tensors by ~4e-3 rel-L2.  The fixtures are therefore GENERATED tie-free (make_golden.py moves a
    # fixture's seed until the reference's own fp64 run has no pre-activation within 2e-6 of zero and no pooling window whose two largest inputs are closer than
    # that; tests/golden/tie_counts.json records the census): on them the tight bound max(3 x the reference's fp32-vs-fp64 error, 1e-4) must hold on the
    # FIRST run, no dither, no retry.  The one exception is the Winograd-size fixture -- 60 M activations have ~70 pre-activations inside that band whatever
    # the seed -- where SURVEY 8c's bound (rel-L2 <= 1e-2 per tensor) must hold on every run and the tight bound on at least one of a few runs with inputs
    # dithered far below fp32 resolution of the problem (3e-7 relative).
    import json
    import os
    census = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'tie_counts.json')))['cases']
    assert {k for k, v in census.items() if v['act_ties'] or v['pool_ties']} <= {'unet_nb2_sf32_wino_odd.npz'}, 'a small fixture was generated with a near-tie decision'
    ties = census[case]
    runs = [grad_errors(m)]
    if ties['act_ties'] == 0 and ties['pool_ties'] == 0:
        bad = {k: v for k, v in runs[0].items() if v[0] > max(3 * v[1], 1e-4)}
        assert not bad, ('no near-tie decision in this fixture: the tight bound holds without retries', bad)
        return
    torch.manual_seed(123)
    for _ in range(5):
        if all(eb <= max(3 * er, 1e-4) for eb, er in runs[-1].values()):
            break
        m2 = build(cfg, sub(g, 'sd0')).train()
        xd = x * (1 + 3e-7 * torch.randn_like(x))
        combined_loss(m2(xd), t).backward()
        runs.append(grad_errors(m2))
    for errs in runs:
        for k, (eb, er) in errs.items():
            assert eb <= 1e-2, (k, eb, er)
    assert any(all(eb <= max(3 * er, 1e-4) for eb, er in errs.values()) for errs in runs), \
        [max(errs.items(), key=lambda kv: kv[1][0]) for errs in runs]


@pytest.mark.parametrize('case,key', [('unet_nb2_sf32_bf16.npz', 'bf16'), ('unet_nb2_sf32_f16.npz', 'f16')])
def test_fp32_path_on_the_fp32_leg_of_the_16bit_fixtures(case, key):
    """The 16-bit fixtures also hold the REFERENCE's fp32 run (logits_fp32, grad32/*) on the same 16-bit-valued weights and input,
    start_filts=32: consumed here by the fp32 HIP path (logits 1e-4; gradients within SURVEY 8c's hard bound of 1e-2 rel-L2 per tensor --
    the fixture has no fp64 twin to take the reference's own noise from)."""
    import os
    from helpers import load_bf16_fixture
    from elektronn3_amd.unet import UNet
    g = load_bf16_fixture(os.path.join(os.path.dirname(__file__), 'golden', case))
    sd = {k[4:]: v.float() for k, v in g.items() if k.startswith('sd0/')}
    m = UNet(1, 2, n_blocks=int(g['cfg.n_blocks']), start_filts=int(g['cfg.start_filts']))
    m.load_state_dict(sd)
    m = m.cuda().train()
    x = g['x'].float().cuda()
    y = m(x)
    ref = g['logits_fp32']
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-4 * float(ref.abs().max()))
    y.backward(g['dlogits'].float().cuda())
    for k, p in m.named_parameters():
        if k.endswith('.bias') and 'norm' not in k and not k.startswith('conv_final'):
            continue          # analytically zero (bias in front of a train-mode BatchNorm)
        r = g['grad32/' + k].float()
        e = float((p.grad.cpu() - r).norm() / r.norm())
        assert e <= 1e-2, (k, e)


def test_eval_forward_matches_reference():
    g = load_npz('unet_nb2_sf8.npz')
    sd = sub(g, 'sd0'); sd.update(sub(g, 'sd1'))
    m = build(unet_cfg(g), sd).eval()
    with torch.no_grad():
        y = m(torch.from_numpy(g['x']).cuda())
    np.testing.assert_allclose(y.cpu().numpy(), g['logits_eval'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('case', [c for c in CASES if 'attention' in c])
def test_attention_eval_forward_matches_reference(case):
    """attention=True in eval mode (the gate's BatchNorm folds into its 1x1x1 conv): the reference's own eval output."""
    g = load_npz(case)
    sd = sub(g, 'sd0'); sd.update(sub(g, 'sd1'))          # (the fixture's eval output follows its train step)
    m = build(unet_cfg(g), sd).eval()
    with torch.no_grad():
        out = m(torch.from_numpy(g['x']).cuda())
    np.testing.assert_allclose(out.cpu().numpy(), g['logits_eval'], rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize('kw', [dict(), dict(dim=2), dict(conv_mode='valid', planar_blocks=(0,))], ids=['3d_odd', '2d', 'valid_planar'])
def test_attention_maps_are_kept_on_the_blocks(kw):
    """UpConvBlock.att (unet.py:382,394-395; the Trainer plots it, trainer.py:611-617): after a forward every decoder block holds its attention
    map (N, 1, *spatial of the skip), in train and in eval mode; values against the fp64 op sequence.  None without attention."""
    from elektronn3_amd.unet import UNet
    from oracle.torch_ref import unet_forward
    torch.manual_seed(5)
    m = UNet(1, 2, n_blocks=3, start_filts=16, attention=True, **kw).cuda()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith('bias'):
                p.copy_(0.1 * torch.randn_like(p))
    shape = (37, 50) if kw.get('dim') == 2 else ((30, 53, 55) if kw.get('conv_mode') == 'valid' else (11, 22, 27))
    x = torch.randn(2, 1, *shape, device='cuda')
    for train in (True, False):
        m.train(train)
        sd = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in m.state_dict().items()}
        sd['__valid__'] = kw.get('conv_mode') == 'valid'
        with torch.no_grad():
            y = m(x)
        atts = []
        ref = unet_forward(sd, x.double(), 3, tuple(kw.get('planar_blocks', ())), training=train, atts=atts)
        assert torch.allclose(y.double(), ref, rtol=1e-4, atol=1e-4)
        assert len(atts) == len(m.up_convs) == 2
        for blk, a in zip(m.up_convs, atts):
            assert blk.att.shape == a.shape and blk.att.dtype == torch.float32
            assert float((blk.att.double() - a).abs().max()) < 1e-5
            assert 0.0 <= float(blk.att.min()) and float(blk.att.max()) <= 1.0
    assert all(b.att is None for b in UNet(1, 2, n_blocks=2, start_filts=8).up_convs)


@pytest.mark.gpu
def test_rrelu_eval_forward_matches_reference_and_train_mode_runs():
    g = load_npz('unet_nb3_sf8_rrelu_eval.npz')
    m = build(unet_cfg(g), sub(g, 'sd0')).eval()
    x = torch.from_numpy(g['x']).cuda()
    with torch.no_grad():
        y = m(x)
    np.testing.assert_allclose(y.cpu().numpy(), g['logits_eval'], rtol=1e-4, atol=1e-5)
    # train mode: random slopes per element (checked in detail by test_rrelu_train_mode_draws_slopes_and_backward_recomputes_them)
    yt = m.train()(x)
    yt.sum().backward()
    assert bool(torch.isfinite(yt).all()) and all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())


def test_three_adamw_steps_match_reference_trajectory():
    """Trainer._train_step (trainer.py:509-543) with the example's optimizer/criterion: 3 steps, loss trajectory and
    final weights against the reference run (tests/golden/trainsteps.npz)."""
    from oracle.torch_ref import combined_loss
    g = load_npz('trainsteps.npz')
    m = build(dict(n_blocks=2, start_filts=8, planar_blocks=()), sub(g, 'sd0'))
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.5e-4)
    m.train()
    for i in range(3):
        out = m(torch.from_numpy(g['xs'][i]).cuda())
        loss = combined_loss(out, torch.from_numpy(g['ts'][i]).cuda())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        assert abs(float(loss.detach()) - g['losses'][i]) < 5e-5, (i, float(loss.detach()), g['losses'][i])
    sd = m.state_dict()
    for k, v in sub(g, 'sd3').items():
        if k.endswith('num_batches_tracked'):
            assert int(sd[k]) == 3
        else:
            # Adam's first steps are sign-like (lr*g/|g|): wherever |g| is below the fp32 gradient noise floor
            # (rel-L2 2e-4..4e-3 per tensor for the REFERENCE itself) an element moves by up to 2*lr per
            # step in either implementation.  So: every element within the hard bound 3 steps * 2*lr, at most 10 % of the
            # elements off by more than lr/2, and the tensor within 1e-2 rel-L2.
            got = sd[k].cpu().numpy()
            np.testing.assert_allclose(got, v, rtol=0, atol=6e-3, err_msg=k)
            if is_prebn_bias(k):
                continue   # zero-gradient parameters: Adam turns pure round-off noise into +-lr steps in BOTH implementations
            if got.size >= 256:
                assert (np.abs(got - v) > 5e-4).mean() < 0.10, k
            assert rel_l2(got, v) < 1e-2 or np.abs(v).max() < 1e-2, k


def test_midsize_vs_cpu_oracle():
    from oracle import unet_oracle as orc
    from helpers import combined_loss_np
    from elektronn3_amd.unet import UNet
    torch.manual_seed(3)
    m = UNet(1, 2, n_blocks=3, start_filts=16, planar_blocks=(1,)).cuda().train()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if 'norm' in n and n.endswith('weight'):
                p.copy_(1 + 0.2 * torch.randn_like(p))
            elif n.endswith('bias'):
                p.copy_(0.1 * torch.randn_like(p))
    sd0 = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    x = torch.randn(2, 1, 12, 40, 36)
    t = torch.randint(0, 2, (2, 12, 40, 36))
    net = orc.OracleUNet(sd0, 3, (1,))
    logits_ref = net.forward(x.numpy())
    out = m(x.cuda())
    np.testing.assert_allclose(out.detach().cpu().numpy(), logits_ref, rtol=1e-4, atol=1e-4)
    _, dlogits = combined_loss_np(logits_ref, t.numpy())
    grads_ref, _ = net.backward(dlogits.astype(np.float32))
    out.backward(torch.from_numpy(dlogits.astype(np.float32)).cuda())
    gnorm = np.sqrt(sum(float(np.sum(v.astype(np.float64) ** 2)) for v in grads_ref.values()))
    for k, p in m.named_parameters():
        gb = p.grad.cpu().numpy()
        if is_prebn_bias(k):
            assert np.abs(gb).max() <= 1e-5 * gnorm, k
        else:
            assert rel_l2(gb, grads_ref[k]) < 5e-3, (k, rel_l2(gb, grads_ref[k]))
    for k in sd0:
        if 'running' in k:
            np.testing.assert_allclose(m.state_dict()[k].cpu().numpy(), net.sd[k], rtol=1e-5, atol=1e-6, err_msg=k)


@pytest.fixture(scope='module')
def cfg2():
    from elektronn3_amd.unet import UNet
    torch.manual_seed(0)
    m = UNet(in_channels=1, out_channels=2, n_blocks=4, start_filts=32, normalization='batch').cuda()
    x = torch.randn(2, 1, 64, 128, 128, device='cuda')
    t = torch.randint(0, 2, (2, 64, 128, 128), device='cuda')
    return m, x, t


def _train_step_vs_pytorch_rocm(m, x, t, dt=torch.float32, atol=2e-4, nb=4, planar=(), budget=False):
    """One train step of the HIP path against the reference's ATen op sequence (oracle/torch_ref.py) run by PyTorch-ROCm on this GPU in
    `dt`.  budget=True: `dt` must be fp64, and the SAME op sequence is run a second time in fp32 (MIOpen): every gradient tensor of
    the HIP path must then be within max(3 x the fp32 reference's own distance from fp64, 1e-2) rel-L2 of the fp64 result -- SURVEY.md
    8c's stated budget (a single flipped ReLU / arg-max decision moves a gradient tensor by up to 4e-3, DESIGN.md section 5, hence the floor)."""
    from oracle.torch_ref import combined_loss, unet_forward
    m.train()
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    out = m(x)
    loss = combined_loss(out, t)
    m.zero_grad(set_to_none=True)
    loss.backward()

    def run_ref(dtype):
        sd_ref = {k: (v.to(dtype) if v.is_floating_point() else v.clone()).requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd0.items()}
        ref = unet_forward(sd_ref, x.to(dtype), nb, planar, training=True)
        lref = combined_loss(ref, t)
        lref.backward()
        return sd_ref, ref.detach(), float(lref.detach())

    sd_ref, ref, lref = run_ref(dt)
    assert torch.allclose(out.double(), ref.double(), rtol=1e-3, atol=atol), float((out.double() - ref.double()).abs().max())
    assert abs(float(loss.detach()) - lref) < 1e-5
    for k in sd0:
        if 'running' in k:
            assert torch.allclose(m.state_dict()[k], sd_ref[k].float(), rtol=1e-4, atol=1e-6), k
    err_ref = {}
    if budget:
        assert dt == torch.float64
        grads64 = {k: sd_ref[k].grad.clone() for k, _ in m.named_parameters()}
        sd32, ref32, _ = run_ref(torch.float32)
        err_ref = {k: float((sd32[k].grad.double() - grads64[k]).norm() / grads64[k].norm().clamp_min(1e-30)) for k in grads64}
        e_out_ref = float((ref32.double() - ref.double()).abs().max())
        assert float((out.double() - ref.double()).abs().max()) <= max(3 * e_out_ref, 1e-4)
        del sd32, ref32
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters()))
    worst = (0.0, 0.0, '')
    for k, p in m.named_parameters():
        gr = sd_ref[k].grad
        if is_prebn_bias(k):
            assert float(p.grad.abs().max()) <= 1e-5 * float(gn), k
            continue
        err = float((p.grad.to(gr.dtype) - gr).norm() / gr.norm().clamp_min(1e-30))
        bound = max(3 * err_ref[k], 1e-2) if budget else 2e-2
        assert err < bound, (k, err, err_ref.get(k))
        if err > worst[0]:
            worst = (err, err_ref.get(k, float('nan')), k)
    m.load_state_dict(sd0)
    return worst


@pytest.mark.parametrize('case', ['cfg2_digest.npz', 'cfg4_digest.npz'])
def test_full_size_against_the_reference_digest(case):
    """The REFERENCE's own train step at BASELINE.json configs[1]'s size (8192 Winograd bricks at level 0), closing reference -> fixture -> HIP at the size
    the headline is measured on: tests/golden/cfg2_digest.npz (make_golden.py cfg2) holds, from the reference's fp32 and fp64 CPU runs on parameters /
    input / target that both sides regenerate from one seed, a strided sample of the logits, the loss, the running statistics, and per gradient tensor
    its norm, a strided sample and four Gaussian projections.  Bounds as in test_train_step_matches_reference: logits within max(3 x the reference's own
    fp32-vs-fp64 error, 2e-5) of the fp64 sample; per gradient tensor the error norm estimated from the projections (E <e, r>^2 = |e|^2) and the sampled
    rel-L2 within SURVEY 8c's 1e-2 always, and within 2 x max(3 x the reference's own error, 1e-4) -- the factor 2 is the spread of a four-sample norm
    estimate -- unless this fixture's near-tie count (act_ties + pool_ties of the reference's fp64 run, stored in the
    digest) says a flipped ReLU / arg-max decision may move a tensor.  cfg4_digest.npz is batch 1 (the fp64 reference run of batch 2 does not fit the authoring
    container)."""
    from collections import OrderedDict
    from helpers import digest_state_dict, digest_inputs, digest_of
    from oracle.torch_ref import combined_loss
    g = load_npz(case)      # (cfg4_digest.npz: the same for BASELINE configs[3], UNet(planar_blocks=(0, 1), start_filts=64) on 2 x 32x256x256 -- the planar Winograd kernels at their own size)
    seed = int(g['seed'])
    shapes = OrderedDict((str(k), tuple(int(i) for i in str(sh).split(',')) if str(sh) else ()) for k, sh in zip(g['names'], g['shapes']))
    sd0 = digest_state_dict(shapes, seed)
    m = build(unet_cfg(g), sd0).train()
    x_np, t_np = digest_inputs(int(g['batch']), tuple(int(v) for v in g['shape']), seed)
    x, t = torch.from_numpy(x_np).cuda(), torch.from_numpy(t_np).cuda()
    out = m(x)
    samp = out.detach()[:, :, ::8, ::8, ::8].cpu().numpy()
    err_ref = float(g['logits_err_ref'])
    near_ties = int(g['act_ties']) + int(g['pool_ties'])
    np.testing.assert_allclose(samp, g['logits32'], rtol=1e-4, atol=1e-4)
    assert float(np.abs(samp - g['logits64']).max()) <= max(3 * err_ref, 2e-5), (float(np.abs(samp - g['logits64']).max()), err_ref)
    loss = combined_loss(out, t)
    assert abs(float(loss.detach()) - float(g['loss64'])) < 2e-5, (float(loss.detach()), float(g['loss64']), float(g['loss32']))
    loss.backward()
    sd = m.state_dict()
    for k, v in sub(g, 'sd1').items():
        np.testing.assert_allclose(sd[k].cpu().numpy(), v, rtol=1e-5, atol=1e-6, err_msg=k)
    names = {k for k, _ in m.named_parameters()}
    gnorm = np.sqrt(sum(float(g['g/' + k][0]) ** 2 for k in names))
    worst = (0.0, 0.0, '')
    margins = []            # per tensor: what the HIP path scores against the fp64 digest, beside the reference's own fp32 error and the bounds
    for k, p in m.named_parameters():
        rec = g['g/' + k]
        n64, err_own = float(rec[0]), float(rec[2])
        p64 = rec[3:7]
        ns = int(rec[11]); s64 = rec[12:12 + ns]
        gr = p.grad.detach().cpu().numpy()
        if is_prebn_bias(k, names):
            assert np.abs(gr).max() <= 1e-5 * gnorm, (k, float(np.abs(gr).max()))
            continue
        _, s_h, p_h = digest_of(k, gr, seed)
        est = float(np.sqrt(np.mean((p_h - p64) ** 2))) / max(n64, 1e-30)          # estimated rel-L2 of (HIP - fp64 reference) over the whole tensor
        smp = float(np.linalg.norm(s_h - s64) / max(np.linalg.norm(s64), 1e-30))   # the same on the sampled elements
        margins.append({'tensor': k, 'est': est, 'sampled': smp, 'err_own': err_own, 'tight_bound': 2 * max(3 * err_own, 1e-4), 'hard_cap': 1e-2})
        assert est <= 1e-2 and (smp <= 1e-2 or np.linalg.norm(s64) < 1e-3 * n64), (k, est, smp, err_own)
        if est / max(err_own, 1e-30) > worst[0] / max(worst[1], 1e-30) or worst[2] == '':
            worst = (est, err_own, k)
        bound = 2 * max(3 * err_own, 1e-4)
        # (4e-3: the size of one flipped near-tie decision, see test_train_step_matches_reference -- allowed only where the fixture's own census of the
        # reference's fp64 run says there ARE decisions within 2e-6 of a tie; at these sizes there are hundreds)
        assert est <= bound or (near_ties > 0 and est <= 4e-3), (k, est, err_own, near_ties)
    print(f'{case}: worst projected gradient error {worst[0]:.2e} (reference fp32 itself: {worst[1]:.2e}) at {worst[2]}; logits err_ref {err_ref:.2e}')
    # the per-tensor table goes to gpurun_out/ (merged back from the GPU box); tools/parity_margins.py renders profiles/r06_parity_margins.md from it
    try:
        import json
        outdir = os.environ.get('E3_MARGINS_DIR') or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        os.makedirs(outdir, exist_ok=True)
        with open(os.path.join(outdir, 'parity_margins_' + case.replace('.npz', '.json')), 'w') as f:
            json.dump({'case': case, 'near_ties': near_ties, 'logits_err': float(np.abs(samp - g['logits64']).max()), 'logits_err_ref': err_ref,
                       'logits_bound': max(3 * err_ref, 2e-5), 'loss_err': abs(float(loss.detach()) - float(g['loss64'])), 'tensors': margins}, f, indent=1)
    except OSError:
        pass


def test_full_size_cfg2_against_pytorch_rocm(cfg2):
    """BASELINE.json configs[1] at full size: same weights, same input, the reference's ATen op sequence executed by
    PyTorch-ROCm (MIOpen) on this GPU vs the HIP path."""
    _train_step_vs_pytorch_rocm(*cfg2)


def test_full_size_cfg2_gradient_budget_against_fp64(cfg2):
    """BASELINE.json configs[1] at full size with SURVEY.md 8c's stated budget: logits and every gradient tensor against the fp64 run
    of the reference's op sequence, each allowed max(3 x the fp32 reference's own error, 1e-2)."""
    worst = _train_step_vs_pytorch_rocm(*cfg2, dt=torch.float64, atol=1e-4, budget=True)
    print(f'cfg 2 full size: worst gradient rel-L2 vs fp64 {worst[0]:.2e} (fp32 reference: {worst[1]:.2e}) at {worst[2]}')


def test_full_size_ragged_crop_against_pytorch_rocm(cfg2):
    """The cfg-2 network on a full-size crop whose extents are all odd (61 x 131 x 125: ragged bricks at every edge of the persistent
    Winograd kernels, ceil-mode pooling and the one-voxel autocrop of every up-conv, unet.py:289-299) vs the reference's op sequence run by PyTorch-ROCm in FP64:
    MIOpen's fp32 kernels for this shape are themselves 6e-4 (output) / 2.7e-2 (gradients) away from fp64 (tools/ragged_diag.py; the HIP path:
    3e-5 / 6e-3), so they cannot serve as the reference here."""
    m = cfg2[0]
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(2, 1, 61, 131, 125, device='cuda', generator=g)
    t = (torch.rand(2, 61, 131, 125, device='cuda', generator=g) < 0.3).long()
    _train_step_vs_pytorch_rocm(m, x, t, dt=torch.float64, atol=1e-4)


def test_split_k_bottom_level_against_fp64():
    """Few bricks, many channels: UNet(n_blocks=2, start_filts=64) on 2 x 16x32x64 has a bottom level of 8x16x32 voxels (16 bricks per
    sample) with 64->128 and 128->128 convs -- the Winograd kernels split the input channels over two workgroups per brick and a
    reduction pass adds the partial sums, the bias and takes the statistics (conv_wino_splitk; forward AND data gradient).  Same
    weights, same input, the reference's op sequence in fp64."""
    from elektronn3_amd.unet import UNet
    torch.manual_seed(4)
    m = UNet(in_channels=1, out_channels=2, n_blocks=2, start_filts=64).cuda()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if 'norm' in n and n.endswith('weight'):
                p.copy_(1 + 0.2 * torch.randn_like(p))
            elif n.endswith('bias'):
                p.copy_(0.1 * torch.randn_like(p))
    x = torch.randn(2, 1, 16, 32, 64, device='cuda')
    t = torch.randint(0, 2, (2, 16, 32, 64), device='cuda')
    _train_step_vs_pytorch_rocm(m, x, t, dt=torch.float64, atol=1e-4, nb=2)


def test_many_classes_against_fp64():
    """out_channels = 12 (the head, its fused backward and the criterion are compiled for 1..16 classes): train step vs the fp64 op sequence."""
    from elektronn3_amd.unet import UNet
    from oracle.torch_ref import combined_loss, unet_forward
    torch.manual_seed(6)
    m = UNet(in_channels=1, out_channels=12, n_blocks=2, start_filts=16).cuda().train()
    x = torch.randn(2, 1, 12, 24, 40, device='cuda')
    t = torch.randint(0, 12, (2, 12, 24, 40), device='cuda')
    cw = tuple(float(v) for v in (torch.rand(12) + 0.2))
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    out = m(x); loss = combined_loss(out, t, cw); loss.backward()
    from elektronn3_amd.loss import CombinedCEDiceLoss
    assert abs(float(CombinedCEDiceLoss(weight=cw).cuda()(out.detach(), t)) - float(loss.detach())) < 2e-6
    sd = {k: (v.double() if v.is_floating_point() else v.clone()).requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd0.items()}
    ref = unet_forward(sd, x.double(), 2, (), training=True)
    lref = combined_loss(ref, t, cw); lref.backward()
    assert torch.allclose(out.double(), ref, rtol=1e-4, atol=1e-4) and abs(float(loss.detach()) - float(lref.detach())) < 1e-5
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters()))
    for k, p in m.named_parameters():
        if is_prebn_bias(k):
            assert float(p.grad.abs().max()) <= 1e-5 * float(gn), k
            continue
        err = float((p.grad.double() - sd[k].grad).norm() / sd[k].grad.norm().clamp_min(1e-30))
        assert err < 1e-2, (k, err)
    m.eval()
    with torch.no_grad():
        ye = m(x)
        sde = {k: v.double() if v.is_floating_point() else v for k, v in m.state_dict().items()}
        assert torch.allclose(ye.double(), unet_forward(sde, x.double(), 2, (), training=False), rtol=1e-4, atol=1e-4)


def test_full_size_cfg4_anisotropic_against_pytorch_rocm():
    """BASELINE.json configs[3] (anisotropic UNet, planar_blocks=(0,1), start_filts=64) on a 16x128x128 crop (a quarter of
    the 32x256x256 of the config in every direction: MIOpen needs minutes to pick kernels for the full size) -- mixed
    1x3x3 / 3x3x3 convs, (1,2,2) pooling and transposed convs -- vs PyTorch-ROCm on the same weights."""
    from elektronn3_amd.unet import UNet
    from oracle.torch_ref import combined_loss, unet_forward
    torch.manual_seed(3)
    m = UNet(in_channels=1, out_channels=2, n_blocks=4, start_filts=64, planar_blocks=(0, 1), normalization='batch').cuda().train()
    x = torch.randn(1, 1, 16, 128, 128, device='cuda')
    t = torch.randint(0, 2, (1, 16, 128, 128), device='cuda')
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    out = m(x)
    loss = combined_loss(out, t)
    m.zero_grad(set_to_none=True)
    loss.backward()
    sd_ref = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd0.items()}
    ref = unet_forward(sd_ref, x, 4, (0, 1), training=True)
    lref = combined_loss(ref, t)
    lref.backward()
    assert torch.allclose(out, ref, rtol=1e-3, atol=2e-4), float((out - ref).abs().max())
    assert abs(float(loss.detach()) - float(lref.detach())) < 1e-5
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters()))
    for k, p in m.named_parameters():
        gr = sd_ref[k].grad
        if is_prebn_bias(k):
            assert float(p.grad.abs().max()) <= 1e-5 * float(gn), k
            continue
        err = float((p.grad - gr).norm() / gr.norm().clamp_min(1e-30))
        assert err < 2e-2, (k, err)


def test_full_size_cfg4_at_its_own_size_against_fp64():
    """BASELINE.json configs[3] at ITS size: anisotropic UNet(planar_blocks=(0,1), start_filts=64), one 32x256x256 crop (the per-GPU
    shard of the 4-GPU config), against the fp64 op sequence (torch's fp64 convolutions, no MIOpen kernel search) with the 8c budget."""
    from elektronn3_amd.unet import UNet
    torch.manual_seed(3)
    m = UNet(in_channels=1, out_channels=2, n_blocks=4, start_filts=64, planar_blocks=(0, 1), normalization='batch').cuda()
    x = torch.randn(1, 1, 32, 256, 256, device='cuda')
    t = torch.randint(0, 2, (1, 32, 256, 256), device='cuda')
    worst = _train_step_vs_pytorch_rocm(m, x, t, dt=torch.float64, atol=1e-4, nb=4, planar=(0, 1), budget=True)
    print(f'cfg 4 full size: worst gradient rel-L2 vs fp64 {worst[0]:.2e} (fp32 reference: {worst[1]:.2e}) at {worst[2]}')
    del m
    torch.cuda.empty_cache()


def test_cfg5_tile_eval_softmax_against_fp64():
    """BASELINE.json configs[4]'s model and tile: eval-mode UNet(n_blocks=4, start_filts=32) (running statistics from a few train
    batches, BN folded into the conv epilogues, softmax fused into the head) on ONE 128x224x224 input tile (tile 96x192x192 + overlap 16)
    against the fp64 op sequence + softmax."""
    from elektronn3_amd.unet import UNet
    from oracle.torch_ref import unet_forward
    torch.manual_seed(0)
    m = UNet(1, 2, n_blocks=4, start_filts=32).cuda().train()
    with torch.no_grad():
        for _ in range(4):
            m(torch.randn(2, 1, 32, 64, 64, device='cuda'))
    m.eval()
    x = torch.randn(1, 1, 128, 224, 224, device='cuda', generator=torch.Generator(device='cuda').manual_seed(7))
    with torch.no_grad():
        y = m.forward_softmax(x)
        sd = {k: v.double() if v.is_floating_point() else v for k, v in m.state_dict().items()}
        ref = torch.softmax(unet_forward(sd, x.double(), 4, (), training=False), 1)
    err = float((y.double() - ref).abs().max())
    assert err < 1e-4, err
    assert float((y.sum(1) - 1).abs().max()) < 1e-5


def test_eval_forward_with_the_pool_in_the_conv_epilogue_at_odd_sizes_against_fp64():
    """Eval mode at start_filts=32: the ceil-mode max-pool behind an encoder block is taken in the epilogue of the persistent Winograd kernel
    (conv3_wino_pkernel<true, true, true>, ConvArgs::pool_out; models/unet.py:244-253 pooling = MaxPool3d(2, ceil_mode=True)).  Odd extents on
    every axis and at every level: windows that hang over the tensor's edge take part with the voxels inside only."""
    from elektronn3_amd.unet import UNet
    from oracle.torch_ref import unet_forward
    torch.manual_seed(3)
    m = UNet(1, 2, n_blocks=3, start_filts=32).cuda().train()
    with torch.no_grad():
        for _ in range(3):
            m(torch.randn(2, 1, 16, 32, 32, device='cuda'))
    m.eval()
    for shape in [(2, 1, 37, 75, 85), (1, 1, 64, 96, 130)]:
        x = torch.randn(*shape, device='cuda', generator=torch.Generator(device='cuda').manual_seed(11))
        with torch.no_grad():
            y = m(x)
            sd = {k: v.double() if v.is_floating_point() else v for k, v in m.state_dict().items()}
            ref = unet_forward(sd, x.double(), 3, (), training=False)
        err = float((y.double() - ref).abs().max())
        assert err < 1e-4, (shape, err)


@pytest.mark.parametrize('kw', [dict(normalization='none'), dict(normalization='batch', full_norm=False, planar_blocks=(0,)),
                                dict(merge_mode='add'), dict(merge_mode='add', normalization='none', planar_blocks=(0,)),
                                dict(normalization='instance', planar_blocks=(0,)), dict(normalization='group', planar_blocks=(0,)),
                                dict(normalization='group16', full_norm=False, merge_mode='add'),
                                dict(activation='leaky'), dict(activation='leaky', normalization='none', planar_blocks=(0,)),
                                dict(activation='lin', normalization='group', full_norm=False),
                                dict(activation='silu', planar_blocks=(0,)), dict(activation='silu', normalization='none'),
                                dict(activation='prelu', planar_blocks=(0,)), dict(activation='prelu', normalization='none'),
                                dict(up_mode='resizeconv_nearest'), dict(up_mode='resizeconv_nearest', planar_blocks=(0,), normalization='group', full_norm=False),
                                dict(up_mode='resizeconv_linear', planar_blocks=(0,)), dict(up_mode='resizeconv_linear1', planar_blocks=(0,)),
                                dict(conv_mode='valid'), dict(conv_mode='valid', planar_blocks=(0,), merge_mode='add', normalization='group'),
                                dict(conv_mode='valid', up_mode='resizeconv_nearest', activation='leaky', full_norm=False),
                                dict(attention=True), dict(attention=True, planar_blocks=(0,), merge_mode='add'),
                                dict(attention=True, conv_mode='valid', activation='leaky', full_norm=False),
                                dict(attention=True, normalization='none', up_mode='resizeconv_nearest')],
                         ids=['nonorm', 'sparsenorm', 'add', 'add_nonorm_planar', 'instance', 'group8', 'group16_sparse_add',
                              'leaky', 'leaky_nonorm_planar', 'lin_group', 'silu_planar', 'silu_nonorm', 'prelu_planar', 'prelu_nonorm', 'resizeconv', 'resizeconv_planar_group_sparse', 'resizelinear_planar', 'resizelinear1_planar', 'valid', 'valid_planar_add_group', 'valid_resizeconv_leaky_sparse',
                              'attention', 'attention_planar_add', 'attention_valid_leaky_sparse', 'attention_nonorm_resizeconv'])
def test_option_variants_against_pytorch_rocm(kw):
    """normalization='none' and full_norm=False (norm layers = nn.Identity, unet.py:77-80,238-242,369-375) at a size that runs the
    Winograd kernels (conv -> bias -> ReLU fused in their epilogue, also in training), and merge_mode='add' (unet.py:398-401: the skip
    connection is summed, conv1 has C input channels): forward, loss, all gradients (conv biases before an Identity have REAL
    gradients), running statistics, vs the fp64 ATen op sequence on PyTorch-ROCm."""
    from elektronn3_amd.unet import UNet
    from oracle.torch_ref import combined_loss, instance_norm_names as inorms, unet_forward
    torch.manual_seed(9)
    paramless = inorms(3, kw.get('full_norm', True)) if kw.get('normalization') == 'instance' else ()
    m = UNet(in_channels=1, out_channels=2, n_blocks=3, start_filts=32, **kw).cuda().train()
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith('.bias'):
                p.copy_(0.1 * torch.randn_like(p))
            elif '.act' in k:
                p.copy_(0.25 + 0.3 * torch.randn_like(p))
    valid = kw.get('conv_mode') == 'valid'
    x = torch.randn(2, 1, *((44, 76, 76) if valid else (32, 64, 64)), device='cuda')      # ('valid' shrinks every conv by 2)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    out = m(x)
    assert valid == (out.shape[2:] != x.shape[2:])
    t = torch.randint(0, 2, (2, *out.shape[2:]), device='cuda')
    loss = combined_loss(out, t)
    m.zero_grad(set_to_none=True)
    loss.backward()
    sd_ref = {k: (v.double() if v.is_floating_point() else v.clone()).requires_grad_(v.is_floating_point() and 'running' not in k)
              for k, v in sd0.items()}
    pl = tuple(kw.get('planar_blocks', ()))
    group = str(kw.get('normalization', '')).startswith('group')
    sd_ref['__valid__'] = valid
    sd_ref['__up_linear__'] = str(kw.get('up_mode')).startswith('resizeconv_linear')
    sd_ref['__act_slope__'] = {'relu': 0.0, 'leaky': 0.1, 'lin': 1.0, 'silu': 2.0, 'prelu': 3.0}[kw.get('activation', 'relu')]
    sd_ref['__instance_norms__'] = paramless
    sd_ref['__num_groups__'] = 8 if kw.get('normalization') == 'group' else (int(kw['normalization'][5:]) if group else 0)
    ref = unet_forward(sd_ref, x.double(), 3, pl, training=True)
    lref = combined_loss(ref, t)
    lref.backward()
    assert torch.allclose(out.double(), ref, rtol=1e-4, atol=1e-4), float((out - ref).abs().max())
    assert abs(float(loss.detach()) - float(lref.detach())) < 1e-5
    for k, v in m.state_dict().items():
        if 'running' in k:
            torch.testing.assert_close(v.double(), sd_ref[k], rtol=1e-5, atol=1e-6, msg=k)
    names = {k for k, _ in m.named_parameters()}
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters()))
    n_real_bias = 0
    ref32 = None
    act_gmax = max([float(sd_ref[k].grad.norm()) for k, _ in m.named_parameters() if '.act' in k] or [0.0])
    for k, p in m.named_parameters():
        gr = sd_ref[k].grad
        if is_prebn_bias(k, set() if group else names, paramless):
            assert float(p.grad.abs().max()) <= 1e-5 * float(gn), k
            continue
        n_real_bias += k.endswith('.bias') and 'conv' in k and not k.startswith('conv_final')
        err = float((p.grad.double() - gr).norm() / gr.norm().clamp_min(1e-30))
        bound = 1e-2
        if '.act' in k:      # (scalars: judged against at least 1e-2 of the largest slope gradient of the network)
            err = float((p.grad.double() - gr).norm() / gr.norm().clamp_min(1e-2 * act_gmax))
        if '.act' in k and err >= bound:
            # a PReLU slope gradient is ONE scalar, a sum over the whole tensor with heavy cancellation (|g| ~ 1e-4 of the layer's other
            # gradients): SURVEY 8c's budget of 3 x the error of the reference's own fp32 run applies (tools/prelu_diag.py: MIOpen fp32 is
            # 2.9e-2 off fp64 on down_convs.2.act2 of this very case, the HIP path 0.7e-2 .. 1.7e-2 depending on the conv algorithm)
            if ref32 is None:
                sd32 = {k2: (v.float().detach().clone().requires_grad_(v.requires_grad) if torch.is_tensor(v) and v.is_floating_point() else v) for k2, v in sd_ref.items()}
                combined_loss(unet_forward(sd32, x, 3, pl, training=True), t).backward()
                ref32 = sd32
            bound = max(bound, 3 * float((ref32[k].grad.double() - gr).norm() / gr.norm().clamp_min(1e-30)))
        assert err < bound, (k, err, bound)
    assert n_real_bias >= 3 or kw.get('normalization', 'batch') in ('batch', 'instance') and kw.get('full_norm', True)
    m.eval()
    with torch.no_grad():
        ye = m(x)
    sd_e = {k: v.double() if v.is_floating_point() else v for k, v in m.state_dict().items()}
    sd_e['__instance_norms__'] = paramless
    sd_e['__num_groups__'] = sd_ref['__num_groups__']
    sd_e['__act_slope__'] = sd_ref['__act_slope__']
    sd_e['__up_linear__'] = sd_ref['__up_linear__']
    sd_e['__valid__'] = valid
    assert torch.allclose(ye.double(), unet_forward(sd_e, x.double(), 3, pl, training=False), rtol=1e-4, atol=1e-4)
    if paramless or group:       # instance / group statistics in eval mode too: eval output == train output, and a batch equals its samples one by one
        assert torch.equal(ye, out.detach())
        assert torch.equal(m(x[1:2]), ye[1:2])


@pytest.mark.parametrize('kw', [dict(enc_res_blocks=1, dec_res_blocks=1), dict(enc_res_blocks=2, dec_res_blocks=2, planar_blocks=(0,)),
                                dict(enc_res_blocks=0, dec_res_blocks=0, activation='leaky'),
                                dict(enc_res_blocks=1, dec_res_blocks=0, attention=True, merge_mode='add'),
                                dict(enc_res_blocks=2, dec_res_blocks=1, normalization='none', up_mode='resizeconv_nearest'),
                                dict(enc_res_blocks=1, dec_res_blocks=1, normalization='group', activation='silu')],
                         ids=['res11', 'res22_planar', 'res00_leaky', 'res10_attention_add', 'res21_nonorm_resizeconv', 'res11_group_silu'])
def test_resunet_variants_against_pytorch_rocm(kw):
    """elektronn3.models.resunet.UNet (resunet.py:598-934) at a size that runs the Winograd kernels (2 x 32 x 64 x 64, start_filts 32): residual
    ConvBlocks with identity and projected shortcuts, several per block; forward, loss, every gradient (projection weights and biases included),
    running statistics, the eval-mode forward, and the backward through the eval-mode forward -- against the fp64 op sequence on PyTorch-ROCm."""
    from elektronn3_amd.resunet import UNet
    from oracle.torch_ref import combined_loss, resunet_forward
    torch.manual_seed(13)
    m = UNet(in_channels=1, out_channels=2, n_blocks=3, start_filts=32, **kw).cuda().train()
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith('.bias'):
                p.copy_(0.1 * torch.randn_like(p))
    group = str(kw.get('normalization', '')).startswith('group')
    pl = tuple(kw.get('planar_blocks', ()))
    e, d = kw['enc_res_blocks'], kw['dec_res_blocks']
    flags = {'__act_slope__': {'relu': 0.0, 'leaky': 0.1, 'silu': 2.0}[kw.get('activation', 'relu')], '__num_groups__': 8 if group else 0}
    x = torch.randn(2, 1, 32, 64, 64, device='cuda')
    t = torch.randint(0, 2, (2, 32, 64, 64), device='cuda')

    def reference(sd0, training):
        sd = {k: (v.double() if v.is_floating_point() else v.clone()).requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd0.items()}
        sd.update(flags)
        out = resunet_forward(sd, x.double(), 3, pl, training, e, d)
        loss = combined_loss(out, t)
        loss.backward()
        return sd, out, loss

    def check_grads(sd, zero_bias_ok, bound):
        names = {k for k, _ in m.named_parameters()}
        gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters()))
        for k, p in m.named_parameters():
            gr = sd[k].grad
            if zero_bias_ok and is_prebn_bias(k, set() if group else names):
                assert float(p.grad.abs().max()) <= 1e-5 * float(gn), k
                continue
            err = float((p.grad.double() - gr).norm() / gr.norm().clamp_min(1e-30))
            assert err < bound, (k, err)

    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    out = m(x)
    loss = combined_loss(out, t)
    m.zero_grad(set_to_none=True)
    loss.backward()
    sd, ref, lref = reference(sd0, True)
    assert torch.allclose(out.double(), ref, rtol=1e-4, atol=1e-4), float((out - ref).abs().max())
    assert abs(float(loss.detach()) - float(lref.detach())) < 1e-5
    for k, v in m.state_dict().items():
        if 'running' in k:
            torch.testing.assert_close(v.double(), sd[k], rtol=1e-5, atol=1e-6, msg=k)
    assert any(k.endswith('.proj.weight') for k, _ in m.named_parameters()) == (e >= 1 or d >= 1)
    check_grads(sd, True, 1e-2)
    # eval mode: inference path, then a backward through it (frozen statistics; GroupNorm has none: same function as above)
    m.eval()
    sd1 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ye = m(x)
    m.zero_grad(set_to_none=True)
    oe = m(x)
    combined_loss(oe, t).backward()
    sde, refe, _ = reference(sd1, False)
    assert torch.allclose(ye.double(), refe, rtol=1e-4, atol=1e-4)
    assert torch.allclose(oe.double(), refe, rtol=1e-4, atol=1e-4)
    check_grads(sde, group, 1e-2)


def test_dim2_unet_against_pytorch_rocm():
    """dim=2 (unet.py:47-74: Conv2d / ConvTranspose2d / MaxPool2d / BatchNorm2d, 4D input) at 2x1x384x512 with the headline
    widths (n_blocks=4, start_filts=32): runs on the planar kernels incl. the planar Winograd ones; vs the same op sequence
    on PyTorch-ROCm (fp64, so that its own BN-statistics error does not enter) -- forward, loss, running stats, gradients."""
    from elektronn3_amd.unet import UNet
    from oracle.torch_ref import combined_loss, unet_forward
    torch.manual_seed(5)
    m = UNet(in_channels=1, out_channels=2, n_blocks=4, start_filts=32, normalization='batch', dim=2).cuda().train()
    assert isinstance(m.down_convs[0].conv1, torch.nn.Conv2d) and isinstance(m.up_convs[0].norm0, torch.nn.BatchNorm2d)
    x = torch.randn(2, 1, 384, 512, device='cuda')
    t = torch.randint(0, 2, (2, 384, 512), device='cuda')
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    out = m(x)
    assert out.shape == (2, 2, 384, 512)
    loss = combined_loss(out, t)
    m.zero_grad(set_to_none=True)
    loss.backward()
    sd_ref = {k: (v.double() if v.is_floating_point() else v.clone()).requires_grad_(v.is_floating_point() and 'running' not in k)
              for k, v in sd0.items()}
    ref = unet_forward(sd_ref, x.double(), 4, (), training=True)
    lref = combined_loss(ref, t)
    lref.backward()
    assert torch.allclose(out.double(), ref, rtol=1e-4, atol=1e-4), float((out - ref).abs().max())
    assert abs(float(loss.detach()) - float(lref.detach())) < 1e-5
    sd1 = m.state_dict()
    for k in sd0:
        if 'running' in k:
            torch.testing.assert_close(sd1[k].double(), sd_ref[k], rtol=1e-5, atol=1e-6, msg=k)
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters()))
    for k, p in m.named_parameters():
        gr = sd_ref[k].grad
        assert p.grad.shape == p.shape
        if is_prebn_bias(k):
            assert float(p.grad.abs().max()) <= 1e-5 * float(gn), k
            continue
        err = float((p.grad.double() - gr).norm() / gr.norm().clamp_min(1e-30))
        assert err < 1e-2, (k, err)
    # eval mode + the wrong rank fail loudly
    m.eval()
    with torch.no_grad():
        ye = m(x)
    sd_e = {k: v.double() if v.is_floating_point() else v for k, v in m.state_dict().items()}
    assert torch.allclose(ye.double(), unet_forward(sd_e, x.double(), 4, (), training=False), rtol=1e-4, atol=1e-4)
    with pytest.raises(ValueError):
        m(x.unsqueeze(2))


@pytest.mark.parametrize('variant', ['attention', 'resunet'])
def test_attention_and_residual_steps_are_deterministic_and_batch_independent_in_eval(variant):
    """The gates' and the shortcuts' GEMMs reduce over voxels in a fixed split order (no atomics): two training steps from the same state are
    bit-identical, odd sizes included; eval mode treats the samples of a batch independently."""
    from elektronn3_amd import resunet, unet
    torch.manual_seed(21)
    if variant == 'attention':
        m = unet.UNet(1, 2, n_blocks=3, start_filts=16, attention=True).cuda()
    else:
        m = resunet.UNet(1, 2, n_blocks=3, start_filts=16, enc_res_blocks=2, dec_res_blocks=1, attention=True).cuda()
    x = torch.randn(2, 1, 21, 38, 45, device='cuda')
    m.train()
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    outs, grads, atts = [], [], []
    for _ in range(2):
        m.load_state_dict(sd0)
        m.zero_grad(set_to_none=True)
        o = m(x)
        o.backward(torch.ones_like(o) * 1e-3 + 1e-4 * torch.sign(o.detach()))
        outs.append(o.detach().clone())
        grads.append([p.grad.clone() for p in m.parameters()])
        atts.append([b.att.clone() for b in m.up_convs])
    assert torch.equal(outs[0], outs[1])
    assert all(torch.equal(a, b) for a, b in zip(*grads))
    assert all(torch.equal(a, b) for a, b in zip(*atts))
    m.eval()
    with torch.no_grad():
        y2 = m(x)
        assert torch.equal(y2[:1], m(x[:1])) and torch.equal(y2[1:], m(x[1:]))


def _minimal_case(n_blocks, planar, dim, resunet_blocks=None):
    import itertools  # noqa: F401
    from elektronn3_amd import resunet, unet
    from oracle.torch_ref import resunet_forward, unet_forward
    torch.manual_seed(100 * n_blocks + len(planar) + (7 if dim == 2 else 0))
    if resunet_blocks is None:
        m = unet.UNet(in_channels=1, out_channels=2, n_blocks=n_blocks, planar_blocks=planar, dim=dim).cuda().train()
    else:
        m = resunet.UNet(in_channels=1, out_channels=2, n_blocks=n_blocks, planar_blocks=planar, dim=dim, enc_res_blocks=resunet_blocks[0],
                         dec_res_blocks=resunet_blocks[1]).cuda().train()
    side = 2 ** n_blocks
    shape = (side, side) if dim == 2 else (side // (2 ** len(planar)), side, side)
    x = torch.randn(1, 1, *shape, device='cuda')
    sd = {k: (v.detach().double() if v.is_floating_point() else v.clone()).requires_grad_(v.is_floating_point() and 'running' not in k)
          for k, v in m.state_dict().items()}
    out = m(x)
    assert tuple(out.shape) == (1, 2, *shape)
    out.sum().backward()
    if resunet_blocks is None:
        ref = unet_forward(sd, x.double(), n_blocks, planar, training=True)
    else:
        ref = resunet_forward(sd, x.double(), n_blocks, planar, True, *resunet_blocks)
    ref.sum().backward()
    assert torch.allclose(out.double(), ref, rtol=1e-3, atol=1e-3), float((out.double() - ref).abs().max())
    gn = float(torch.sqrt(sum((v.grad ** 2).sum() for v in sd.values() if v.grad is not None)))
    for k, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        # (a handful of voxels per BatchNorm channel at the bottom levels: statistics of 4..8 values amplify fp32 rounding; judged against the
        # gradient's global scale)
        assert float((p.grad.double() - sd[k].grad).abs().max()) <= 2e-2 * gn + 1e-6, (k, float((p.grad.double() - sd[k].grad).abs().max()), gn)


def test_reference_self_test_sweep_2d():
    """The reference's own inline self-test (`python unet.py`: test_2d_config, unet.py:997-1000): n_blocks 1..4 on the minimal 2**n_blocks input,
    out.sum().backward(), output shape -- here with values and gradients against the fp64 op sequence as well."""
    for n_blocks in range(1, 5):
        _minimal_case(n_blocks, (), 2)


def test_reference_self_test_sweep_planar_configs():
    """test_planar_configs (unet.py:1003-1013): n_blocks 1..4 x EVERY subset of planar blocks on the minimal input
    (depth 2**n_blocks // 2**len(planar_blocks))."""
    import itertools
    for n_blocks in range(1, 5):
        for r in range(n_blocks + 1):
            for planar in itertools.combinations(range(n_blocks), r):
                _minimal_case(n_blocks, planar, 3)


def test_resunet_self_test_sweep():
    """The same sweep for the ResUNet (resunet.py:990-1079), plain and residual ConvBlocks."""
    import itertools
    for n_blocks in range(1, 4):
        for r in range(n_blocks + 1):
            for planar in itertools.combinations(range(n_blocks), r):
                _minimal_case(n_blocks, planar, 3, (0, 0) if (n_blocks + r) % 2 else (1, 1))
    _minimal_case(4, (0, 1), 3, (2, 1))


def test_full_size_properties(cfg2):
    m, x, t = cfg2
    # determinism: two training forwards+backwards from the same state are bit-identical (no atomics anywhere)
    m.train()
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    outs, grads = [], []
    for _ in range(2):
        m.load_state_dict(sd0)
        m.zero_grad(set_to_none=True)
        o = m(x)
        o.backward(torch.ones_like(o) * 1e-3 + 1e-4 * torch.sign(o.detach()))
        outs.append(o.detach().clone())
        grads.append([p.grad.clone() for p in m.parameters()])
    assert torch.equal(outs[0], outs[1])
    assert all(torch.equal(a, b) for a, b in zip(*grads))
    # eval mode: samples are independent (BN uses running stats) -> batch of 2 == two batches of 1, bit for bit
    m.eval()
    with torch.no_grad():
        y2 = m(x)
        y0, y1 = m(x[:1]), m(x[1:])
        assert torch.equal(y2[:1], y0) and torch.equal(y2[1:], y1)
        # softmax head fused into the last kernel == torch.softmax of the logits
        ps = m.forward_softmax(x[:1])
        assert torch.allclose(ps, torch.softmax(y0, 1), rtol=1e-5, atol=1e-6)
        assert torch.allclose(ps.sum(1), torch.ones_like(ps[:, 0]), atol=1e-6)


def test_state_dict_roundtrip_pickle_and_reference_keys(tmp_path):
    import copy
    from elektronn3_amd.unet import UNet
    g = load_npz('unet_nb4_sf8_planar01.npz')
    m = build(unet_cfg(g), sub(g, 'sd0'))
    assert list(m.state_dict().keys()) == list(sub(g, 'sd0').keys())   # same keys, same ORDER as the reference
    torch.save(m, tmp_path / 'model.pt')                  # Trainer._save_model pickles the module (trainer.py:874)
    m2 = torch.load(tmp_path / 'model.pt', weights_only=False)
    m3 = copy.deepcopy(m)                                 # Predictor(float16=True) deep-copies (inference.py:402-407)
    x = torch.from_numpy(g['x']).cuda()
    m.eval(); m2.eval(); m3.eval()
    with torch.no_grad():
        assert torch.equal(m(x), m2(x)) and torch.equal(m(x), m3(x))
    import torch.nn as nn
    assert sum(isinstance(mod, nn.modules.batchnorm._BatchNorm) for mod in m.modules()) == 17   # SWA.bn_update walks these


def test_trainer_protocol_autocast_gradscaler_dataparallel():
    """What Trainer._train_step wraps around the model (training/trainer.py:517-543): ``torch.autocast`` (fp16 by default),
    ``GradScaler.scale(loss).backward()`` + ``scaler.step(optimizer)``, and ``nn.DataParallel`` (benchmark/train_benchmark.py:109-110,
    one visible GPU here).  A configuration outside the native 16-bit path computes in fp32 under autocast: the plain call's logits rounded once."""
    from elektronn3_amd.loss import CombinedCEDiceLoss
    from elektronn3_amd.optim import AdamW
    from elektronn3_amd.unet import UNet
    torch.manual_seed(12)
    m = UNet(n_blocks=3, start_filts=16, planar_blocks=(0,)).cuda().train()
    crit = CombinedCEDiceLoss(weight=[0.2653, 0.7347]).cuda()
    x = torch.randn(2, 1, 12, 40, 48, device='cuda'); t = torch.randint(0, 2, (2, 12, 40, 48), device='cuda')
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    out0 = m(x); crit(out0, t).backward()
    g0 = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.load_state_dict(sd0); m.zero_grad(set_to_none=True)
    dp = torch.nn.DataParallel(m)
    opt = AdamW(m.parameters(), lr=1e-3, weight_decay=0.5e-4)
    scaler = torch.amp.GradScaler('cuda', init_scale=256.0)
    with torch.autocast('cuda', dtype=torch.float16):
        out1 = dp(x)
        loss = crit(out1, t)
    # start_filts=16 is not on the native 16-bit path: fp32 compute, ONE rounding to the autocast dtype at the end -- the output has the autocast
    # dtype whichever kernels ran (as the reference's does), so its incoming gradient is a float16 tensor too
    assert out1.dtype == torch.float16 and torch.equal(out1, out0.half())
    scaler.scale(loss).backward()
    for k, p in m.named_parameters():
        if is_prebn_bias(k, set(g0), set()):
            continue
        e = float((p.grad / 256.0 - g0[k]).norm() / g0[k].norm())
        assert e < 3e-3, (k, e)          # the logits and their (scaled) gradient were rounded to float16 once each
    before = [p.detach().clone() for p in m.parameters()]
    scaler.step(opt); scaler.update()
    assert all(torch.isfinite(p).all() for p in m.parameters())
    assert any(not torch.equal(p.detach(), b) for p, b in zip(m.parameters(), before))
    assert float(opt.state[next(m.parameters())]['step']) == 1


def test_cpu_input_fails_loudly():
    from elektronn3_amd.unet import UNet
    m = UNet(n_blocks=2, start_filts=8)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 8, 8, 8))


def test_gradsync_rccl_single_rank_path(tmp_path):
    """The data-parallel path end to end on ONE GPU: torch.distributed.run with one rank, RCCL process group, the HIP
    event recorded by libe3unet mid-backward, the side-stream all-reduce (AVG over 1 rank = identity).  The gradients
    must equal the plain single-GPU gradients bit for bit and bench.py must print its JSON line."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (E3_WGRAD_NO_DEFER: with a bucket event the executor keeps one weight-gradient launch per layer -- the early bucket wants final gradients --, without it the
    # small layers take the cross-layer stream-K launch, whose sums have another order: bit-identity between the modes is a statement about ONE of the two forms)
    env = dict(os.environ, E3_FORCE_GRADSYNC='1', HSA_ENABLE_IPC_MODE_LEGACY='0', E3_WGRAD_NO_DEFER='1')
    script = tmp_path / 'dp_check.py'
    script.write_text('''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from elektronn3_amd.unet import UNet
from elektronn3_amd.dataparallel import GradSync
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
torch.manual_seed(0)
m = UNet(1, 2, n_blocks=3, start_filts=8).cuda().train()
x = torch.randn(2, 1, 16, 32, 32, device='cuda')
out = m(x); out.backward(torch.ones_like(out) * 1e-3)
ref = [p.grad.clone() for p in m.parameters()]
bad = []
for overlap in (False, True):          # GradSync's default (one all-reduce behind the backward) and the overlapped bucket (no CU reserve: bit-identical weight gradients)
    sync = GradSync(m, bucket_after_down_block=2, overlap=overlap, cu_reserve=0)
    for p in m.parameters(): p.grad = None
    m.load_state_dict(m.state_dict())
    out = m(x); out.backward(torch.ones_like(out) * 1e-3)
    torch.cuda.synchronize()
    assert sync._split > 0 and (sync._event is not None) == overlap, (overlap, sync._event)
    bad += [(overlap, n) for (n, p), r in zip(m.named_parameters(), ref) if not torch.equal(p.grad, r)]
print('BAD', bad)
dist.destroy_process_group()
''' % root)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29541', str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'BAD []' in r.stdout, r.stdout[-2000:]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29542', os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--no-cpu-baseline']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    res = json.loads(line)
    assert res['n_gpus'] == 1 and res['value'] > 1e6 and res['roofline']['achieved'] > 10
    assert res['dp_mode'].startswith('serial'), res['dp_mode']         # bench.py says which data-parallel mode ran


# ------------------------------------------------------------------------------------------------ device criterion
@pytest.mark.gpu
@pytest.mark.parametrize('C,shape,weighted', [(2, (2, 9, 17, 21), True), (4, (1, 8, 16, 16), True), (3, (3, 5, 6, 7), False), (11, (2, 6, 9, 10), True), (16, (1, 4, 8, 12), False)])
def test_ce_dice_loss_matches_reference_criterion(C, shape, weighted):
    """CombinedCEDiceLoss == 0.5*CrossEntropyLoss(w) + 0.5*DiceLoss(softmax, w) of the reference (numpy fp64 restatement
    tests/helpers.combined_loss_np for the value; PyTorch autograd of the same formula in fp64 for the gradient)."""
    import torch
    from elektronn3_amd.loss import CombinedCEDiceLoss
    from helpers import combined_loss_np
    rng = np.random.default_rng(7 + C)
    n = shape[0]
    z = (rng.standard_normal((n, C) + shape[1:]) * 2).astype(np.float32)
    t = rng.integers(0, C, (n,) + shape[1:]).astype(np.int64)
    w = (rng.random(C) + 0.2).astype(np.float32) if weighted else None
    crit = CombinedCEDiceLoss(weight=w).cuda()
    zt = torch.from_numpy(z).cuda().requires_grad_(True)
    loss = crit(zt, torch.from_numpy(t).cuda())
    (loss * 1.7).backward()
    # fp64 reference with autograd
    zr = torch.from_numpy(z).double().requires_grad_(True)
    tr = torch.from_numpy(t)
    wr = torch.ones(C, dtype=torch.float64) if w is None else torch.from_numpy(w).double()
    ce = torch.nn.functional.cross_entropy(zr, tr, weight=wr)
    p = torch.softmax(zr, 1)
    oh = torch.zeros_like(p).scatter_(1, tr.unsqueeze(1), 1)
    dims = (0,) + tuple(range(2, zr.dim()))
    dice = (wr * (1 - 2 * (p * oh).sum(dims) / ((p + oh).sum(dims) + 1e-4))).mean()
    lref = 0.5 * ce + 0.5 * dice
    (lref * 1.7).backward()
    assert abs(float(loss) - float(lref)) < 2e-6 * max(1.0, abs(float(lref)))
    g, gr = zt.grad.cpu().numpy(), zr.grad.numpy()
    assert np.abs(g - gr).max() < 2e-6 * np.abs(gr).max() + 1e-10
    if w is not None:      # the numpy restatement used by the whole-net tests agrees as well
        lnp, gnp = combined_loss_np(z, t, w)
        assert abs(float(loss) - lnp) < 2e-6 * max(1.0, abs(lnp))
        assert np.abs(g - 1.7 * gnp).max() < 2e-6 * np.abs(gnp).max() * 1.7 + 1e-10
    # deterministic (fixed reduction order)
    loss2 = crit(zt.detach(), torch.from_numpy(t).cuda())
    assert float(loss2) == float(loss)


@pytest.mark.gpu
def test_ce_dice_loss_over_sharded_minibatch_equals_gathered_batch():
    """CombinedCEDiceLoss(global_batch=True): two 'ranks' each hold half of the minibatch; with their 2+3C sums added
    (the all-reduce, emulated in-process) both return the loss of the GATHERED batch -- what the reference computes on GPU 0 behind
    nn.DataParallel (training/trainer.py:520-524) -- and their logit gradients, scaled back by the world size the mode pre-multiplies
    for gradient AVERAGING, concatenate to the gradient of that loss.  Checked against the unsharded HIP criterion, the fp64 oracle sums
    (oracle/unet_oracle.ce_dice_sums) and PyTorch autograd in fp64."""
    import oracle.unet_oracle as O
    from oracle.torch_ref import combined_loss
    from elektronn3_amd.loss import CombinedCEDiceLoss
    from elektronn3_amd import _lib
    from elektronn3_amd._lib import c_size_t, check, ptr, stream_ptr
    rng = np.random.default_rng(11)
    C, shape, world = 3, (4, 5, 9, 11), 2
    z = (rng.standard_normal((shape[0], C) + shape[1:]) * 2).astype(np.float32)
    t = (rng.random(shape) < np.array([0.1, 0.3, 0.6, 0.9]).reshape(4, 1, 1, 1)).astype(np.int64) * 2      # unbalanced shards
    z[:2, 2] += 3 * (t[:2] == 2)
    w = np.array([0.2, 0.5, 0.3], np.float32)
    zs = [torch.from_numpy(z[r * 2:(r + 1) * 2]).cuda().requires_grad_(True) for r in range(world)]
    ts = [torch.from_numpy(t[r * 2:(r + 1) * 2]).cuda() for r in range(world)]
    # each rank's sums through the C ABI; their total = the all-reduce
    L = _lib.load()
    nbytes = L.e3_ce_dice_workspace_bytes(C)
    wt = torch.from_numpy(w).cuda()
    local = []
    for zr, tr in zip(zs, ts):
        ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda'); sm = torch.empty(2 + 3 * C, dtype=torch.float64, device='cuda')
        check(L.e3_ce_dice_sums(stream_ptr(zr.device), ptr(zr.detach()), ptr(tr), ptr(wt), C, 2, *shape[1:], ptr(ws), c_size_t(nbytes), ptr(sm)))
        want = O.ce_dice_sums(zr.detach().cpu().numpy(), tr.cpu().numpy(), w)
        assert np.abs(sm.cpu().numpy() - want).max() < 3e-6 * np.abs(want).max()
        local.append(sm)
    total = local[0] + local[1]

    class Sharded(CombinedCEDiceLoss):
        def _world(self): return world
        def _reduce_sums(self, sums): return total.clone()

    crit = Sharded(weight=w, global_batch=True).cuda()
    losses = [crit(zr, tr) for zr, tr in zip(zs, ts)]
    for l in losses: (l * 1.3).backward()
    # unsharded HIP criterion and fp64 autograd on the gathered batch
    zf = torch.from_numpy(z).cuda().requires_grad_(True)
    lf = CombinedCEDiceLoss(weight=w).cuda()(zf, torch.from_numpy(t).cuda()); (lf * 1.3).backward()
    zd = torch.from_numpy(z).double().requires_grad_(True)
    ld = combined_loss(zd, torch.from_numpy(t), tuple(float(v) for v in w)); (ld * 1.3).backward()
    assert float(losses[0]) == float(losses[1])
    assert abs(float(losses[0]) - float(ld)) < 2e-6 and abs(float(lf) - float(ld)) < 2e-6
    g = torch.cat([zr.grad for zr in zs]).cpu().numpy() / world
    gd = zd.grad.numpy()
    assert np.abs(g - gd).max() < 2e-6 * np.abs(gd).max() + 1e-10
    assert np.abs(g - zf.grad.cpu().numpy()).max() < 1e-6 * np.abs(gd).max() + 1e-10
    # the mean of per-shard losses is something else
    per = [float(CombinedCEDiceLoss(weight=w).cuda()(zr.detach(), tr)) for zr, tr in zip(zs, ts)]
    assert abs(sum(per) / world - float(ld)) > 1e-4


def test_forward_with_loss_on_a_sharded_minibatch_equals_the_two_calls():
    """UNet.forward_with_loss with CombinedCEDiceLoss(global_batch=True) (what bench.py --gpus N runs per rank): the head emits the criterion's
    2 + 3C sums of the shard, they are summed over the 'ranks' (emulated in-process), the finaliser writes the batch-wide loss and the
    coefficients, and e3_unet_backward_loss seeds the backward from them.  Loss and parameter gradients must equal the two separate calls
    out = model(x); loss = criterion(out, target) on the same shard with the same exchanged sums, and a scaled loss must scale the gradients."""
    from elektronn3_amd.loss import CombinedCEDiceLoss
    from elektronn3_amd.unet import UNet
    torch.manual_seed(3)
    world = 2
    xs = [torch.randn(1, 1, 12, 20, 24, device='cuda') for _ in range(world)]
    ts = [torch.randint(0, 2, (1, 12, 20, 24), device='cuda') for _ in range(world)]
    ref = UNet(1, 2, n_blocks=2, start_filts=8).cuda().train()
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    # the 'all-reduce': shard sums of the unfused criterion on the logits of every shard
    from elektronn3_amd import _lib
    from elektronn3_amd._lib import c_size_t, check, ptr, stream_ptr
    L = _lib.load()
    nbytes = L.e3_ce_dice_workspace_bytes(2)
    wt = torch.tensor([0.2653, 0.7347], device='cuda')
    total = torch.zeros(8, dtype=torch.float64, device='cuda')
    for xr, tr in zip(xs, ts):
        m = UNet(1, 2, n_blocks=2, start_filts=8).cuda().train(); m.load_state_dict(sd)
        z = m(xr).detach().contiguous()
        ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda'); sm = torch.empty(8, dtype=torch.float64, device='cuda')
        check(L.e3_ce_dice_sums(stream_ptr(z.device), ptr(z), ptr(tr), ptr(wt), 2, 1, 12, 20, 24, ptr(ws), c_size_t(nbytes), ptr(sm)))
        total += sm

    class Sharded(CombinedCEDiceLoss):
        def _world(self): return world
        def _reduce_sums(self, sums): return total.clone()

    crit = Sharded(weight=[0.2653, 0.7347], global_batch=True).cuda()
    for xr, tr in zip(xs, ts):
        ma = UNet(1, 2, n_blocks=2, start_filts=8).cuda().train(); ma.load_state_dict(sd)
        mb = UNet(1, 2, n_blocks=2, start_filts=8).cuda().train(); mb.load_state_dict(sd)
        out_a, loss_a = ma.forward_with_loss(xr, tr, crit)
        (loss_a * 1.7).backward()
        out_b = mb(xr); loss_b = crit(out_b, tr)
        (loss_b * 1.7).backward()
        assert torch.equal(out_a.detach(), out_b.detach())
        assert abs(float(loss_a.detach()) - float(loss_b.detach())) <= 2e-6 * max(1.0, abs(float(loss_b.detach())))
        for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
            ga, gb = pa.grad.cpu().numpy(), pb.grad.cpu().numpy()
            if k.endswith('.bias') and not k.startswith('conv_final') and 'norm' not in k:
                continue       # analytically-zero gradients (bias feeding a train-mode BN)
            assert rel_l2(ga, gb) <= 2e-5, (k, rel_l2(ga, gb))


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
def test_low_precision_module_trains_with_fp32_compute(dt):
    """model.bfloat16() / model.half() (BASELINE configs[2] stores the model in bf16; Predictor(float16=True)): parameters, input and
    output keep the low-precision dtype, the kernels compute in fp32 on up-cast copies.  One training step must equal the fp32 model
    loaded with the SAME (rounded) parameters, up to the rounding of the results to the storage dtype."""
    from elektronn3_amd.unet import UNet
    from oracle.torch_ref import combined_loss
    torch.manual_seed(31)
    m = UNet(1, 2, n_blocks=3, start_filts=16, planar_blocks=(0,)).cuda().to(dt).train()
    ref = UNet(1, 2, n_blocks=3, start_filts=16, planar_blocks=(0,)).cuda().train()
    ref.load_state_dict({k: v.float() for k, v in m.state_dict().items()})
    x = torch.randn(2, 1, 12, 40, 48, device='cuda').to(dt)
    t = torch.randint(0, 2, (2, 12, 40, 48), device='cuda')
    out = m(x)
    assert out.dtype == dt
    out_ref = ref(x.float())
    eps = torch.finfo(dt).eps
    torch.testing.assert_close(out.float(), out_ref, rtol=2 * eps, atol=2 * eps)
    combined_loss(out.float(), t).backward()
    combined_loss(out_ref, t).backward()
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        assert p.grad.dtype == dt
        # (the low-precision run back-propagates through the rounded logits: allow the storage rounding on top)
        assert float((p.grad.float() - q.grad).norm()) <= 3e-2 * float(q.grad.norm()) + 1e-6, k
    for (k, b), (_, c) in zip(m.named_buffers(), ref.named_buffers()):
        if 'running' in k:
            torch.testing.assert_close(b.float(), c, rtol=2 * eps, atol=2 * eps, msg=k)
            assert b.dtype == dt
    assert not torch.equal(m.down_convs[0].norm0.running_mean.float(), torch.zeros(16, device='cuda'))


@pytest.mark.parametrize('kw', [dict(), dict(planar_blocks=(0,), full_norm=False), dict(normalization='none'), dict(attention=True),
                                dict(attention=True, normalization='none')])      # (the gates' BatchNorm layers are frozen also when the blocks have no norm)
def test_backward_through_eval_mode_forward_against_fp64(kw):
    """Autograd through a module in eval mode (frozen-BatchNorm fine-tuning; the reference's autograd supports it, e.g. around
    training/recalibration.py:53-73): BatchNorm uses the running statistics as constants, which stay untouched.  Against the fp64 op
    sequence in eval mode; the split-K bottom level and odd sizes included."""
    from elektronn3_amd.unet import UNet
    from oracle.torch_ref import combined_loss, unet_forward
    torch.manual_seed(11)
    planar = kw.get('planar_blocks', ())
    m = UNet(1, 2, n_blocks=3, start_filts=16, **kw).cuda()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if 'norm' in n and n.endswith('weight'):
                p.copy_(1 + 0.2 * torch.randn_like(p))
            elif n.endswith('bias'):
                p.copy_(0.1 * torch.randn_like(p))
    x = torch.randn(2, 1, 13, 30, 36, device='cuda')
    t = torch.randint(0, 2, (2, 13, 30, 36), device='cuda')
    m.train()
    with torch.no_grad():
        for _ in range(3):
            m(torch.randn(2, 1, 13, 30, 36, device='cuda'))      # non-trivial running statistics
    m.eval()
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.zero_grad(set_to_none=True)
    out = m(x)
    loss = combined_loss(out, t)
    loss.backward()
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd0[k]), k                            # eval mode: no buffer moves
    sd = {k: (v.double() if v.is_floating_point() else v.clone()).requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd0.items()}
    ref = unet_forward(sd, x.double(), 3, planar, training=False)
    lref = combined_loss(ref, t)
    lref.backward()
    assert torch.allclose(out.double(), ref, rtol=1e-4, atol=1e-4), float((out.double() - ref).abs().max())
    with torch.no_grad():
        assert torch.allclose(m(x), out, rtol=1e-5, atol=1e-5)       # the inference path (folded BN) computes the same function
    for k, p in m.named_parameters():
        g = sd[k].grad
        err = float((p.grad.double() - g).norm() / g.norm().clamp_min(1e-30))
        assert err < 5e-3, (k, err)


@pytest.mark.gpu
def test_rrelu_train_mode_draws_slopes_and_backward_recomputes_them():
    """activation='rrelu' in TRAIN mode (get_activation, unet.py:183-199 -> nn.RReLU(): slope of a negative input ~ U(1/8, 1/3) per element).
    The kernels draw the slopes from a hash of (per-call seed, unit, element index); nothing can be compared element-wise with torch's generator,
    so the network is wired as a probe: conv1 = -x on every channel, conv2 / head = identity, no normalisation, positive input.  Then
    out[c] = -a1[c] * a2[c] * x with two independent draws per element: the ratios must lie in [1/64, 1/9] with mean ((1/8 + 1/3)/2)^2, runs with the
    same torch seed must agree bit for bit, and the gradients -- which the backward computes from RE-DRAWN slopes -- must equal the ones implied by
    the forward's outputs.  Eval mode keeps nn.RReLU's fixed slope."""
    import torch
    from elektronn3_amd.unet import UNet
    dev = torch.device('cuda:0')
    m = UNet(1, 2, n_blocks=1, start_filts=8, normalization='none', activation='rrelu').to(dev)
    with torch.no_grad():
        for p in m.parameters():
            p.zero_()
        m.down_convs[0].conv1.weight[:, 0, 1, 1, 1] = -1.0
        for c in range(8):
            m.down_convs[0].conv2.weight[c, c, 1, 1, 1] = 1.0
        m.conv_final.weight[0, 0] = 1.0
        m.conv_final.weight[1, 1] = 1.0
    x = (torch.rand(2, 1, 12, 20, 36, device=dev) + 0.5).requires_grad_(True)
    gout = torch.randn(2, 2, 12, 20, 36, device=dev)
    m.train()
    torch.manual_seed(5)
    y = m(x)
    y.backward(gout)
    rho = (-y / x.detach()).flatten()                               # a1 * a2 per element and channel
    assert float(rho.min()) >= 1.0 / 64 - 1e-6 and float(rho.max()) <= 1.0 / 9 + 1e-6
    assert abs(float(rho.mean()) - ((1.0 / 8 + 1.0 / 3) / 2) ** 2) < 1e-3
    assert float(rho.std()) > 0.01                                   # (a fixed slope would give 0)
    # the two channels, and neighbouring voxels, draw independently
    r0, r1 = (-y[:, 0] / x.detach()[:, 0]).flatten(), (-y[:, 1] / x.detach()[:, 0]).flatten()
    assert abs(float(torch.corrcoef(torch.stack([r0, r1]))[0, 1])) < 0.02
    assert abs(float(torch.corrcoef(torch.stack([r0[:-1], r0[1:]]))[0, 1])) < 0.02
    # backward: dL/dx = sum_c gout[c] * d out[c] / dx = sum_c gout[c] * out[c] / x   (the same slopes as the forward drew)
    torch.testing.assert_close(x.grad, (gout * y.detach() / x.detach()).sum(1, keepdim=True), rtol=2e-5, atol=1e-6)
    gw = m.conv_final.weight.grad
    torch.testing.assert_close(gw[0, 0].reshape(()), (gout[:, 0] * y.detach()[:, 0]).sum(), rtol=1e-4, atol=1e-5)
    # repeatable under torch.manual_seed, different for another seed
    torch.manual_seed(5)
    assert torch.equal(m(x.detach()), y.detach())
    torch.manual_seed(6)
    assert not torch.equal(m(x.detach()), y.detach())
    # eval mode: the fixed slope (lower + upper) / 2 on both layers
    m.eval()
    with torch.no_grad():
        ye = m(x.detach())
    torch.testing.assert_close(ye[:, 0:1], -(((1.0 / 8 + 1.0 / 3) / 2) ** 2) * x.detach(), rtol=1e-5, atol=1e-7)


def test_two_threads_share_one_plan_without_sharing_rrelu_state():
    """Plans are cached per configuration and shared by every module (and thread) that has it; the RReLU seed of a call is per calling
    thread (e3_unet_set_rrelu), so a train-mode RReLU module on one thread and an eval-mode module of the SAME configuration on another
    give exactly the results of the serial runs (nn.DataParallel-style replicas, train_benchmark.py:109-110)."""
    import threading
    from elektronn3_amd.unet import UNet
    torch.manual_seed(0)
    kw = dict(n_blocks=2, start_filts=8, activation='rrelu')
    ma, mb = UNet(1, 2, **kw).cuda(), UNet(1, 2, **kw).cuda()
    mb.load_state_dict(ma.state_dict())
    assert ma._plan() is mb._plan()
    ma.train(); mb.eval()
    x = torch.randn(2, 1, 12, 16, 20, device='cuda')
    dy = torch.randn(2, 2, 12, 16, 20, device='cuda')
    sd0 = {k: v.clone() for k, v in ma.state_dict().items()}

    def train_steps(n):
        ma.load_state_dict(sd0)
        torch.manual_seed(7)
        outs = []
        for _ in range(n):
            ma.zero_grad(set_to_none=True)
            y = ma(x)
            y.backward(dy)
            outs.append((y.detach().clone(), ma.down_convs[0].conv1.weight.grad.clone()))
        torch.cuda.synchronize()
        return outs

    def eval_steps(n, res):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(n):
                res.append(mb(x).clone())
        s.synchronize()

    serial_train = train_steps(6)
    serial_eval = []
    eval_steps(3, serial_eval)
    assert not torch.equal(serial_train[0][0], serial_eval[0])          # (train mode really draws slopes)
    par_eval = []
    th = threading.Thread(target=eval_steps, args=(40, par_eval))
    th.start()
    par_train = train_steps(6)
    th.join()
    for (ys, gs), (yp, gp) in zip(serial_train, par_train):
        assert torch.equal(ys, yp) and torch.equal(gs, gp)
    assert all(torch.equal(e, serial_eval[0]) for e in par_eval)


@pytest.mark.parametrize('name,kw,shape,dtype', [
    ('cfg2', dict(n_blocks=4, start_filts=32), (2, 1, 64, 128, 128), torch.float32),
    ('cfg4', dict(n_blocks=4, start_filts=64, planar_blocks=(0, 1)), (1, 1, 32, 256, 256), torch.float32),
    ('odd', dict(n_blocks=3, start_filts=16), (2, 1, 21, 45, 47), torch.float32),
    ('odd_valid', dict(n_blocks=3, start_filts=8, conv_mode='valid', planar_blocks=(0,)), (1, 1, 21, 45, 47), torch.float32),
    ('cfg3_bf16', dict(n_blocks=4, start_filts=32), (2, 1, 64, 128, 128), torch.bfloat16),
])
def test_arenas_are_not_overrun(name, kw, shape, dtype, monkeypatch):
    """`saved` and `scratch` are sized by e3_unet_sizes and owned by the caller; a kernel that writes past its share of an arena would
    corrupt a neighbour's memory unnoticed.  The library gets EXACTLY the requested bytes framed by 1 MiB of canaries on both sides
    (forward + backward of cfg 2, cfg 4 at its own size, all-odd crops incl. conv_mode='valid', the native bf16 path); the canaries
    must be intact afterwards and the arenas must have been written at all."""
    from elektronn3_amd import unet as U
    PAD = 1 << 20
    frames = []

    def framed(device, nbytes):
        nbytes = int(nbytes)
        big = torch.full((nbytes + 2 * PAD,), 0xA5, dtype=torch.uint8, device=device)
        frames.append((big, nbytes))
        return big[PAD:PAD + nbytes]

    U.release_scratch()
    monkeypatch.setattr(U, '_alloc_saved', framed)
    monkeypatch.setattr(U, '_get_scratch', framed)
    torch.manual_seed(0)
    m = U.UNet(1, 2, **kw).cuda().train()
    if dtype != torch.float32:
        m = m.to(dtype)
    x = torch.randn(*shape, device='cuda').to(dtype)
    y = m(x)
    y.float().square().mean().backward()
    torch.cuda.synchronize()
    assert len(frames) >= 3          # saved + scratch of the forward, scratch of the backward
    for big, n in frames:
        assert bool((big[:PAD] == 0xA5).all()) and bool((big[PAD + n:] == 0xA5).all()), (name, n)
        assert not bool((big[PAD:PAD + n] == 0xA5).all())
    assert all(torch.isfinite(p.grad.float()).all() for p in m.parameters())
    monkeypatch.undo()
    U.release_scratch()


@pytest.mark.parametrize('name,kw,shape,dtype', [
    ('cfg2', dict(n_blocks=4, start_filts=32), (2, 1, 64, 128, 128), torch.float32),
    ('odd', dict(n_blocks=3, start_filts=16), (2, 1, 21, 45, 47), torch.float32),
    ('odd_valid_planar', dict(n_blocks=3, start_filts=8, conv_mode='valid', planar_blocks=(0,)), (1, 1, 21, 45, 47), torch.float32),
    ('cfg3_bf16', dict(n_blocks=4, start_filts=32), (2, 1, 64, 128, 128), torch.bfloat16),
])
def test_results_do_not_depend_on_what_the_arenas_and_their_surroundings_held(name, kw, shape, dtype, monkeypatch):
    """READ-side counterpart of test_arenas_are_not_overrun (a mis-sized buffer descriptor would pass every write canary).  The
    same training step runs twice: once with `saved`, `scratch` and 1 MiB on both sides of them -- and of the input tensor -- filled with zero
    bytes, once with 0xFF bytes (NaN in fp32, bf16 and fp16).  A kernel that reads anything it (or an earlier kernel of the call) did not
    write -- stale scratch, a halo read that leaves its view inside an arena, a read past the input -- would turn results into NaN or make the
    two runs differ; logits, loss gradients and the updated running statistics must be finite and bit-identical.  (Reads that leave a buffer
    descriptor's range return 0 by hardware: that IS the zero padding of the convolutions.)"""
    from elektronn3_amd import unet as U
    PAD = 1 << 20
    fill = [0]

    def framed(device, nbytes):
        nbytes = int(nbytes)
        big = torch.full((nbytes + 2 * PAD,), fill[0], dtype=torch.uint8, device=device)
        return big[PAD:PAD + nbytes]

    def run(byte):
        fill[0] = byte
        U.release_scratch()
        torch.manual_seed(0)
        m = U.UNet(1, 2, **kw).cuda().train()
        if dtype != torch.float32:
            m = m.to(dtype)
        torch.manual_seed(1)
        x0 = torch.randn(*shape, device='cuda').to(dtype)
        esz = x0.element_size()
        xb = torch.full((x0.numel() * esz + 2 * PAD,), byte, dtype=torch.uint8, device='cuda')
        x = xb[PAD:PAD + x0.numel() * esz].view(dtype).view(shape)
        x.copy_(x0)
        y = m(x)
        y.float().square().mean().backward()
        torch.cuda.synchronize()
        return y.detach().clone(), [p.grad.detach().clone() for p in m.parameters()], [b.detach().clone() for b in m.buffers()]

    monkeypatch.setattr(U, '_alloc_saved', framed)
    monkeypatch.setattr(U, '_get_scratch', framed)
    ya, ga, ba = run(0x00)
    yb, gb, bb = run(0xFF)
    monkeypatch.undo()
    U.release_scratch()
    assert bool(torch.isfinite(ya.float()).all()) and bool(torch.isfinite(yb.float()).all()), name
    assert torch.equal(ya, yb), name
    for a, b in zip(ga, gb):
        assert bool(torch.isfinite(b.float()).all()) and torch.equal(a, b), name
    for a, b in zip(ba, bb):
        assert torch.equal(a, b), name


@pytest.mark.parametrize('case', ['unet_nb2_sf8.npz', 'unet_nb3_sf8_planar0_odd.npz', 'unet_nb2_sf32_wino_odd.npz'])
def test_forward_with_loss_matches_the_reference_step(case):
    """UNet.forward_with_loss (the criterion evaluated inside the 1x1x1 head) against the reference's golden train step:
    logits, loss value ('loss'), the seed d loss / d logits ('dlogits', through the parameter gradients) -- and bit-identical logits / equal
    gradients to the two-call form criterion(model(x), target)."""
    from elektronn3_amd.loss import CombinedCEDiceLoss
    g = load_npz(case)
    cfg = unet_cfg(g)
    x = torch.from_numpy(g['x']).cuda()
    t = torch.from_numpy(g['target']).cuda()
    crit = CombinedCEDiceLoss(weight=torch.tensor([0.2653, 0.7347])).cuda()
    ma = build(cfg, sub(g, 'sd0')).train()
    out, loss = ma.forward_with_loss(x, t, crit)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g['logits'], rtol=1e-4, atol=1e-4)
    assert abs(float(loss.detach()) - float(g['loss'])) < 2e-5
    loss.backward()
    mb = build(cfg, sub(g, 'sd0')).train()
    out_b = mb(x)
    loss_b = crit(out_b, t)
    loss_b.backward()
    assert torch.equal(out.detach(), out_b.detach())
    assert abs(float(loss.detach()) - float(loss_b.detach())) <= 1e-6 * max(1.0, abs(float(loss_b.detach())))
    ref64 = sub(g, 'grad64')
    for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        ga, gb = pa.grad.cpu().numpy(), pb.grad.cpu().numpy()
        if is_prebn_bias(k, set(ref64), instance_norm_names(cfg)):
            continue
        assert rel_l2(ga, gb) <= 1e-5, (k, rel_l2(ga, gb))            # (different summation order of the criterion's partial sums only)
        assert rel_l2(ga, ref64[k]) <= 1e-2, (k, rel_l2(ga, ref64[k]))
    # the logits can be used next to the loss: their gradient adds to the criterion's
    mc = build(cfg, sub(g, 'sd0')).train()
    out_c, loss_c = mc.forward_with_loss(x, t, crit)
    (loss_c + 1e-3 * out_c.square().mean()).backward()
    md = build(cfg, sub(g, 'sd0')).train()
    out_d = md(x)
    (crit(out_d, t) + 1e-3 * out_d.square().mean()).backward()
    k0 = 'down_convs.0.conv1.weight'
    assert rel_l2(dict(mc.named_parameters())[k0].grad.cpu().numpy(), dict(md.named_parameters())[k0].grad.cpu().numpy()) <= 1e-5


def test_head_in_the_last_conv_epilogue_is_bit_identical_to_the_separate_head(tmp_path):
    """Inference at start_filts=32: the 1x1x1 head (+ softmax) is evaluated in the epilogue of the last 3x3x3 conv (conv3_wino_pkernel<true, true, false, true>,
    ConvArgs::head_*; unet.py:881,912) with conv_final_fwd_kernel's arithmetic and summation order.  A child process with E3_WINO_NO_HEAD=1 (the switch is read
    once per process) computes the same forwards through the separate head kernel: logits, softmax output and a needed-region forward must agree bit for bit."""
    import subprocess, sys, os
    script = r"""
import sys, torch
sys.path.insert(0, %r)
from elektronn3_amd.unet import UNet
torch.manual_seed(4)
m = UNet(1, 3, n_blocks=2, start_filts=32).cuda().train()
with torch.no_grad():
    for _ in range(2):
        m(torch.randn(2, 1, 16, 32, 32, device='cuda'))
m.eval()
x = torch.randn(2, 1, 37, 70, 83, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5))
with torch.no_grad():
    out = {'logits': m(x).cpu(), 'softmax': m.forward_softmax(x).cpu(), 'roi': m.forward_roi(x[:1], ((4, 30), (8, 61), (16, 70)), softmax=True)[:, :, 4:30, 8:61, 16:70].cpu()}
torch.save(out, sys.argv[1])
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for tag, extra in (('fused', {}), ('separate', {'E3_WINO_NO_HEAD': '1'})):
        f = str(tmp_path / f'{tag}.pt')
        r = subprocess.run([sys.executable, '-c', script, f], env={**os.environ, **extra}, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(f))
    for k in outs[0]:
        assert torch.isfinite(outs[0][k]).all() and torch.equal(outs[0][k], outs[1][k]), k


def test_channel_chunked_conv1_to_conv2_chain_is_bit_identical_to_rows(tmp_path):
    """Inference: where conv1 and conv2 of a block both run on the F(2x2x4) Winograd kernel and conv2 is the tensor's only reader, conv1 writes its activation
    channel-chunked ([C / 8][voxel][8], ConvArgs::y_chunk) and conv2 stages it that way (x_chunk) -- a change of layout only (unet.py:131-149, 244-253: the same
    conv / BatchNorm / ReLU sequence).  A child process with E3_NO_CHUNKED_FWD=1 (read once per process) computes the same forwards through [voxel][C] rows:
    logits, softmax output and a needed-region forward must agree bit for bit, on ragged grids and with the kernel forced onto every grid (E3_WINO4_MIN=1)."""
    import subprocess, sys, os
    script = r"""
import sys, torch
sys.path.insert(0, %r)
from elektronn3_amd.unet import UNet
out = {}
for nb, sf, shape in ((3, 16, (1, 1, 37, 70, 83)), (2, 32, (2, 1, 24, 40, 48))):
    torch.manual_seed(4 + nb)
    m = UNet(1, 3, n_blocks=nb, start_filts=sf).cuda().train()
    with torch.no_grad():
        for _ in range(2):
            m(torch.randn(2, 1, 16, 32, 32, device='cuda'))
    m.eval()
    x = torch.randn(*shape, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5))
    with torch.no_grad():
        out[f'logits{nb}'] = m(x).cpu(); out[f'softmax{nb}'] = m.forward_softmax(x).cpu()
        out[f'roi{nb}'] = m.forward_roi(x[:1], ((4, 20), (8, 33), (16, 40)), softmax=True)[:, :, 4:20, 8:33, 16:40].cpu()
torch.save(out, sys.argv[1])
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for tag, extra in (('chunked', {'E3_WINO4_MIN': '1'}), ('rows', {'E3_WINO4_MIN': '1', 'E3_NO_CHUNKED_FWD': '1'})):
        f = str(tmp_path / f'{tag}.pt')
        r = subprocess.run([sys.executable, '-c', script, f], env={**os.environ, **extra}, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(f))
    for k in outs[0]:
        assert torch.isfinite(outs[0][k]).all() and torch.equal(outs[0][k], outs[1][k]), k


def test_inference_tensor_layouts_fuzz():
    """tools/fuzz_eval_layouts.py on a handful of random configurations (options, ResUNet blocks, ragged grids, needed regions): the default inference layouts
    (channel-chunked tensors, store boxes, transposed-conv boxes) against plain rows / whole tensors in two child processes, bit for bit."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'fuzz_eval_layouts.py'), '14', '3'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and ' 0 mismatches' in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
