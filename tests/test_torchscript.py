"""TorchScript boundary: ``torch.jit.script(model)`` is the reference training script's default (examples/train_unet_neurodata.py:110-113,
Trainer._save_model training/trainer.py:876-881).  The scripted forward gathers the module's tensors and calls the registered operator
e3unet::unet_fwd; these tests pin the gathering order (CPU) and script -> save -> load -> identical results (GPU)."""
import io

import pytest
import torch

CONFIGS = [dict(), dict(normalization='none'), dict(normalization='group'), dict(activation='prelu', full_norm=False),
           dict(up_mode='resizeconv_nearest', planar_blocks=(0,)), dict(merge_mode='add', activation='leaky')]


@pytest.mark.parametrize('kw', CONFIGS)
def test_scripted_forward_gathers_the_parameter_table_in_order(kw, monkeypatch):
    """CPU: the eager run of the scripted forward's body hands the operator exactly the plan's table (same tensors, same order), the BN
    momenta in table order, and the scripted module compiles."""
    from elektronn3_amd.unet import UNet
    m = UNet(1, 2, n_blocks=3, start_filts=8, **kw)
    plan = m._plan()
    want = [m.get_parameter(n) if k == 0 else m.get_buffer(n) for n, k in zip(plan.names, plan.kinds)]
    seen = {}

    def fake_op(x, tensors, key, momenta, training, softmax):
        seen.update(tensors=tensors, key=key, momenta=momenta, training=training)
        bufs = [t.clone() + 1 for t, k in zip(tensors, plan.kinds) if k != 0]
        return [x.new_zeros(1), x.new_zeros(0)] + (bufs if training else [])

    monkeypatch.setattr(torch.ops.e3unet, 'unet_fwd', fake_op, raising=False)
    m.train()
    rm_before = [b.clone() for n, b in m.named_buffers() if 'running' in n]
    m._scripted_forward(torch.zeros(1, 1, 8, 8, 8))
    assert len(seen['tensors']) == len(want) and all(a is b for a, b in zip(seen['tensors'], want))
    assert tuple(seen['key']) == tuple(float(v) for v in m._plan_key())
    assert seen['momenta'] == m._momenta(plan)[:len(seen['momenta'])] and len(seen['momenta']) == sum(1 for n in plan.names if n.endswith('running_mean'))
    # the returned statistics were copied back and the batch counters advanced
    for before, (n, b) in zip(rm_before, [(n, b) for n, b in m.named_buffers() if 'running' in n]):
        assert torch.equal(b, before + 1), n
    assert all(int(b) == 1 for n, b in m.named_buffers() if n.endswith('num_batches_tracked'))
    monkeypatch.undo()
    torch.jit.script(m)                      # compiles (TorchScript resolves e3unet::unet_fwd from the operator registry)


@pytest.mark.parametrize('kw', [dict(), dict(enc_res_blocks=1, dec_res_blocks=1), dict(enc_res_blocks=2, dec_res_blocks=1, normalization='none', activation='prelu', planar_blocks=(0,))])
def test_resunet_scripts_and_gathers_its_parameter_table_in_order(kw, monkeypatch):
    """elektronn3.models.resunet.UNet is scriptable and Trainer._save_model scripts it with save_jit='script' (trainer.py:871-887): the mirror
    must compile too (the inherited forward walked blk.conv1 / norm0, which the ResUNet blocks do not have) and hand the operator
    the plan's table (ConvBlocks: conv1, norm1, act1, conv2, norm2, act2, proj)."""
    from elektronn3_amd.resunet import UNet
    m = UNet(1, 2, n_blocks=3, start_filts=8, **kw)
    plan = m._plan()
    want = [m.get_parameter(n) if k == 0 else m.get_buffer(n) for n, k in zip(plan.names, plan.kinds)]
    seen = {}

    def fake_op(x, tensors, key, momenta, training, softmax):
        seen.update(tensors=tensors, key=key, momenta=momenta)
        bufs = [t.clone() for t, k in zip(tensors, plan.kinds) if k != 0]
        return [x.new_zeros(1), x.new_zeros(0)] + (bufs if training else [])

    monkeypatch.setattr(torch.ops.e3unet, 'unet_fwd', fake_op, raising=False)
    m.train()
    m._scripted_forward(torch.zeros(1, 1, 8, 8, 8))
    assert len(seen['tensors']) == len(want) and all(a is b for a, b in zip(seen['tensors'], want))
    assert tuple(seen['key']) == tuple(float(v) for v in m._plan_key())
    monkeypatch.undo()
    torch.jit.script(m)


@pytest.mark.gpu
@pytest.mark.parametrize('kw', [dict(), dict(normalization='group', planar_blocks=(0,)), dict(resunet=True, enc_res_blocks=1, dec_res_blocks=1)])
def test_script_save_load_gives_identical_results(kw):
    from elektronn3_amd.unet import UNet
    kw = dict(kw)
    if kw.pop('resunet', False):
        from elektronn3_amd.resunet import UNet
    torch.manual_seed(0)
    m = UNet(1, 2, n_blocks=3, start_filts=8, **kw).cuda()
    x = torch.randn(2, 1, 12, 20, 24, device='cuda')
    dy = torch.randn(2, 2, 12, 20, 24, device='cuda')
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    # eager train step
    m.train(); m.zero_grad(set_to_none=True)
    y = m(x); y.backward(dy)
    g_eager = {k: p.grad.clone() for k, p in m.named_parameters()}
    sd1 = {k: v.clone() for k, v in m.state_dict().items()}
    m.eval()
    with torch.no_grad():
        y_eval = m(x)
    # scripted module from the same initial state: train step, then save -> load -> eval
    m.load_state_dict(sd0)
    sm = torch.jit.script(m)
    sm.train(); sm.zero_grad(set_to_none=True)
    ys = sm(x); ys.backward(dy)
    assert torch.equal(ys, y)
    for k, p in sm.named_parameters():
        assert torch.equal(p.grad, g_eager[k]), k
    for k, v in sm.state_dict().items():
        assert torch.equal(v, sd1[k]), k
    buf = io.BytesIO()
    torch.jit.save(sm, buf); buf.seek(0)
    lm = torch.jit.load(buf, map_location='cuda').eval()
    with torch.no_grad():
        assert torch.equal(lm(x), y_eval)
    # the Predictor accepts the loaded ScriptModule like any other module
    from elektronn3_amd.inference import Predictor
    out = Predictor(lm, device='cuda', apply_softmax=True).predict(x.cpu())
    torch.testing.assert_close(out, torch.softmax(y_eval, 1).cpu(), rtol=1e-5, atol=1e-6)
