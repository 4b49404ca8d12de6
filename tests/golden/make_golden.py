#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the authoring container (needs /root/reference).  The reference's Python is imported
by file path (``import elektronn3`` fails here: no colorlog / generated _version.py),
executed on CPU with torch, and only DATA (inputs, weights, outputs, gradients) is written out as
.npz.  No reference source text is stored.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz

Fixtures
    ops.npz        per-op vectors made with the reference's own layer factories
                   (unet.py: conv3 :131, upconv2 :152, conv1 :178, get_normalization :77, get_maxpool :67)
    unet_*.npz     whole-network train step: input, target, state_dict, logits, loss, all parameter
                   gradients, BN running stats after the step (UNet.forward unet.py:894, Trainer._train_step
                   trainer.py:509-543 with the example's criterion train_unet_neurodata.py:294-296)
    predictor.npz  Predictor.predict tiled + padded-shape case (inference.py:569-687)
    trainsteps.npz 3 AdamW steps: loss trajectory + final weights
    adamw.npz      torch.optim.AdamW alone: 5 steps on random tensors with a changing lr (parameters + both moments per step)
    unet_nb2_sf32_wino_odd.npz  a train step at a size where the fp32 Winograd kernels (persistent and plain) run (`python make_golden.py wino`)
    unet_nb2_sf32_bf16.npz  the reference module cast to bf16 (BASELINE configs[2]): train step in bf16 and in fp32 on the same numbers
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference/elektronn3'
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    """Import unet.py / inference.py / loss.py by path with minimal package stubs."""
    for pkg in ('elektronn3', 'elektronn3.data', 'elektronn3.modules', 'elektronn3.models'):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []
            sys.modules[pkg] = m
    unet = _load('elektronn3.models.unet', f'{REF}/models/unet.py')
    # elektronn3.data.utils needs h5py etc.; inference.py only uses utils.calculate_offset, which
    # needs nothing but torch/numpy -> execute just that file's function in a stub module.
    utils = types.ModuleType('elektronn3.data.utils')
    src = open(f'{REF}/data/utils.py').read()
    start = src.index('def calculate_offset')
    nxt = src.find('\ndef ', start + 1)
    src = src[start:nxt if nxt > 0 else None]
    ns = {'torch': torch, 'np': np, 'logger': __import__('logging').getLogger('golden')}
    exec(compile(src, f'{REF}/data/utils.py', 'exec'), ns)  # runs reference code in place; nothing is copied
    utils.calculate_offset = ns['calculate_offset']
    sys.modules['elektronn3.data.utils'] = utils
    sys.modules['elektronn3.data'].utils = utils
    inference = _load('elektronn3.inference.inference', f'{REF}/inference/inference.py')
    lov = types.ModuleType('elektronn3.modules.lovasz_losses')
    lov.lovasz_softmax = None
    sys.modules['elektronn3.modules.lovasz_losses'] = lov
    loss = _load('elektronn3.modules.loss', f'{REF}/modules/loss.py')
    return unet, inference, loss


def npy(t):
    return t.detach().cpu().numpy().copy()  # copy: .numpy() aliases tensors that are later updated in place


def make_ops(unet, out):
    g = torch.Generator().manual_seed(1234)
    r = lambda *s: torch.randn(*s, generator=g)
    d = {}
    # conv 3x3x3, odd sizes, batch 2
    for tag, cin, cout, shape, planar in (('conv3', 8, 16, (5, 7, 9), False), ('conv3p', 8, 8, (3, 6, 10), True),
                                          ('conv3c1', 1, 8, (6, 5, 7), False)):
        m = unet.conv3(cin, cout, planar=planar)
        with torch.no_grad():
            m.weight.copy_(r(*m.weight.shape) * 0.2)
            m.bias.copy_(r(*m.bias.shape))
        x = r(2, cin, *shape).requires_grad_()
        y = m(x)
        dy = r(*y.shape)
        y.backward(dy)
        d.update({f'{tag}.x': npy(x), f'{tag}.w': npy(m.weight), f'{tag}.b': npy(m.bias), f'{tag}.y': npy(y),
                  f'{tag}.dy': npy(dy), f'{tag}.dx': npy(x.grad), f'{tag}.dw': npy(m.weight.grad),
                  f'{tag}.db': npy(m.bias.grad)})
    # transposed conv k=s=2 and planar (1,2,2)
    for tag, cin, cout, shape, planar in (('convT', 16, 8, (3, 4, 5), False), ('convTp', 8, 8, (3, 4, 5), True)):
        m = unet.upconv2(cin, cout, mode='transpose', planar=planar)
        with torch.no_grad():
            m.weight.copy_(r(*m.weight.shape) * 0.2)
            m.bias.copy_(r(*m.bias.shape))
        x = r(2, cin, *shape).requires_grad_()
        y = m(x)
        dy = r(*y.shape)
        y.backward(dy)
        d.update({f'{tag}.x': npy(x), f'{tag}.w': npy(m.weight), f'{tag}.b': npy(m.bias), f'{tag}.y': npy(y),
                  f'{tag}.dy': npy(dy), f'{tag}.dx': npy(x.grad), f'{tag}.dw': npy(m.weight.grad),
                  f'{tag}.db': npy(m.bias.grad)})
    # conv 1x1x1
    m = unet.conv1(8, 2)
    with torch.no_grad():
        m.weight.copy_(r(*m.weight.shape))
        m.bias.copy_(r(*m.bias.shape))
    x = r(2, 8, 4, 5, 6).requires_grad_()
    y = m(x)
    dy = r(*y.shape)
    y.backward(dy)
    d.update({'conv1.x': npy(x), 'conv1.w': npy(m.weight), 'conv1.b': npy(m.bias), 'conv1.y': npy(y),
              'conv1.dy': npy(dy), 'conv1.dx': npy(x.grad), 'conv1.dw': npy(m.weight.grad), 'conv1.db': npy(m.bias.grad)})
    # batch norm (train incl. running stats, backward; eval) followed by the reference's ReLU
    bn = unet.get_normalization('batch', 8)
    act = unet.get_activation('relu')
    with torch.no_grad():
        bn.weight.copy_(r(8) * 0.5 + 1.0)
        bn.bias.copy_(r(8) * 0.3)
        bn.running_mean.copy_(r(8) * 0.1)
        bn.running_var.copy_(torch.rand(8, generator=g) + 0.5)
    d.update({'bn.gamma': npy(bn.weight), 'bn.beta': npy(bn.bias), 'bn.rm0': npy(bn.running_mean), 'bn.rv0': npy(bn.running_var)})
    x = (r(2, 8, 5, 6, 7) * 1.7 + 0.4).requires_grad_()
    bn.train()
    z = bn(x)
    a = act(z)
    da = r(*a.shape)
    a.backward(da)
    d.update({'bn.x': npy(x), 'bn.z': npy(z), 'bn.a': npy(a), 'bn.da': npy(da), 'bn.dx': npy(x.grad),
              'bn.dgamma': npy(bn.weight.grad), 'bn.dbeta': npy(bn.bias.grad), 'bn.rm1': npy(bn.running_mean),
              'bn.rv1': npy(bn.running_var), 'bn.nbt': npy(bn.num_batches_tracked)})
    bn.eval()
    d['bn.z_eval'] = npy(bn(x))
    # max pool, ceil mode, odd sizes; planar
    for tag, planar, shape in (('pool', False, (5, 7, 9)), ('poolp', True, (3, 7, 8))):
        ks = unet.planar_kernel(2) if planar else 2
        pool = unet.get_maxpool(3)(kernel_size=ks, ceil_mode=True)
        x = r(2, 8, *shape).requires_grad_()
        y = pool(x)
        dy = r(*y.shape)
        y.backward(dy)
        d.update({f'{tag}.x': npy(x), f'{tag}.y': npy(y), f'{tag}.dy': npy(dy), f'{tag}.dx': npy(x.grad)})
    # softmax over C (Predictor wraps model in Sequential(model, Softmax(1)), inference.py:443-444)
    x = r(2, 2, 3, 4, 5) * 3
    d.update({'softmax.x': npy(x), 'softmax.y': npy(torch.nn.Softmax(1)(x))})
    np.savez_compressed(out, **d)
    print('wrote', out, f'{os.path.getsize(out) / 1e6:.2f} MB')


def criterion(loss_mod):
    """train_unet_neurodata.py:290-296 -- CombinedLoss([CE(w), DiceLoss(softmax, w)], [0.5, 0.5])."""
    cw = torch.tensor([0.2653, 0.7347])
    return loss_mod.CombinedLoss([torch.nn.CrossEntropyLoss(weight=cw),
                                  loss_mod.DiceLoss(apply_softmax=True, weight=cw)], weight=[0.5, 0.5])


def make_rrelu_eval(unet, out, seed=17):
    """activation='rrelu' in EVAL mode (nn.RReLU(): the fixed slope (1/8 + 1/3)/2; train mode draws random slopes and cannot be pinned):
    eval-mode logits of the reference with non-trivial running statistics."""
    torch.manual_seed(seed)
    model = unet.UNet(in_channels=1, out_channels=2, n_blocks=3, start_filts=8, planar_blocks=(0,), activation='rrelu')
    with torch.no_grad():
        for name, p in model.named_parameters():
            if 'norm' in name and name.endswith('weight'):
                p.copy_(1.0 + 0.2 * torch.randn_like(p))
            elif name.endswith('bias'):
                p.copy_(0.1 * torch.randn_like(p))
        for name, b in model.named_buffers():
            if name.endswith('running_mean'):
                b.copy_(0.3 * torch.randn_like(b))
            elif name.endswith('running_var'):
                b.copy_(0.5 + torch.rand_like(b))
    x = torch.randn(2, 1, 9, 14, 19)
    model.eval()
    with torch.no_grad():
        y = model(x)
    d = {'cfg.n_blocks': 3, 'cfg.start_filts': 8, 'cfg.planar_blocks': np.array((0,), dtype=np.int64), 'cfg.activation': np.array('rrelu'),
         'x': npy(x), 'logits_eval': npy(y)}
    for k, v in model.state_dict().items():
        d['sd0/' + k] = npy(v).copy()
    np.savez_compressed(out, **d)


def make_unet_case(unet, loss_mod, out, seed, n_blocks, start_filts, planar_blocks, shape, batch, dim=3, normalization='batch', full_norm=True, merge_mode='concat', activation='relu', up_mode='transpose', conv_mode='same', attention=False, res_blocks=None, tie_free=False):
    # res_blocks = (enc_res_blocks, dec_res_blocks): `unet` is then the reference's models/resunet.py module
    extra = {} if res_blocks is None else dict(enc_res_blocks=res_blocks[0], dec_res_blocks=res_blocks[1])
    torch.manual_seed(seed)
    model = unet.UNet(in_channels=1, out_channels=2, n_blocks=n_blocks, start_filts=start_filts,
                      planar_blocks=planar_blocks, activation=activation, normalization=normalization, dim=dim, full_norm=full_norm, merge_mode=merge_mode, up_mode=up_mode, conv_mode=conv_mode, attention=attention, **extra)
    # make BN affine + conv bias non-trivial so that the fixtures exercise them
    with torch.no_grad():
        for name, p in model.named_parameters():
            if 'norm' in name and name.endswith('weight'):
                p.copy_(1.0 + 0.2 * torch.randn_like(p))
            elif name.endswith('bias'):
                p.copy_(0.1 * torch.randn_like(p))
            elif '.act' in name:                     # nn.PReLU slopes: distinct values, also negative ones
                p.copy_(0.25 + 0.3 * torch.randn_like(p))
    sd0 = {k: npy(v).copy() for k, v in model.state_dict().items()}
    x = torch.randn(batch, 1, *shape)
    if tie_free:
        # the fixture must not contain a ReLU / arg-max decision within TIE_TOL of a tie in the reference's fp64 run (a flipped decision moves whole gradient
        # tensors by ~4e-3: no implementation can be held to a tight bound on such a fixture).  A seed that has one is replaced by seed + 1000 (recursively);
        # the seed that generated the file is stored in it.
        rng_state = torch.get_rng_state()       # (building the checker consumes random numbers: the fixture itself is generated as without the check)
        m64t = unet.UNet(in_channels=1, out_channels=2, n_blocks=n_blocks, start_filts=start_filts,
                         planar_blocks=planar_blocks, activation=activation, normalization=normalization, dim=dim, full_norm=full_norm, merge_mode=merge_mode, up_mode=up_mode, conv_mode=conv_mode, attention=attention, **extra).double()
        m64t.load_state_dict({k: torch.as_tensor(v).double() if v.dtype != np.int64 else torch.as_tensor(v) for k, v in sd0.items()})
        stat, _ = count_ties(m64t, x.double())
        torch.set_rng_state(rng_state)
        if stat['act_ties'] or stat['pool_ties']:
            print(f'{os.path.basename(out)}: seed {seed} has {stat["act_ties"]} + {stat["pool_ties"]} near-ties, trying {seed + 1000}')
            return make_unet_case(unet, loss_mod, out, seed + 1000, n_blocks, start_filts, planar_blocks, shape, batch, dim=dim, normalization=normalization, full_norm=full_norm,
                                  merge_mode=merge_mode, activation=activation, up_mode=up_mode, conv_mode=conv_mode, attention=attention, res_blocks=res_blocks, tie_free=True)
    model.train()
    crit = criterion(loss_mod)
    out_t = model(x)
    target = torch.randint(0, 2, (batch, *out_t.shape[2:]))      # (conv_mode='valid': the output is smaller than the input)
    loss = crit(out_t, target)
    dout, = torch.autograd.grad(loss, out_t, retain_graph=True)
    loss.backward()
    d = {'cfg.n_blocks': n_blocks, 'cfg.start_filts': start_filts, 'cfg.planar_blocks': np.array(planar_blocks, dtype=np.int64), 'seed': np.array(seed),
         'x': npy(x), 'target': npy(target), 'logits': npy(out_t), 'loss': npy(loss), 'dlogits': npy(dout)}
    if dim != 3:
        d['cfg.dim'] = dim
    if normalization != 'batch':
        d['cfg.normalization'] = np.array(normalization)
    if not full_norm:
        d['cfg.full_norm'] = np.array(0)
    if merge_mode != 'concat':
        d['cfg.merge_mode'] = np.array(merge_mode)
    if activation != 'relu':
        d['cfg.activation'] = np.array(activation)
    if up_mode != 'transpose':
        d['cfg.up_mode'] = np.array(up_mode)
    if conv_mode != 'same':
        d['cfg.conv_mode'] = np.array(conv_mode)
    if attention:
        d['cfg.attention'] = np.array(1)
    if res_blocks is not None:
        d['cfg.resunet'] = np.array(1)
        d['cfg.enc_res_blocks'] = np.array(res_blocks[0]); d['cfg.dec_res_blocks'] = np.array(res_blocks[1])
    for k, v in sd0.items():
        d['sd0/' + k] = v
    for k, v in model.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            d['sd1/' + k] = npy(v)
    for k, p in model.named_parameters():
        d['grad/' + k] = npy(p.grad)
    model.eval()
    with torch.no_grad():
        d['logits_eval'] = npy(model(x))
    # fp64 reference of the same step (tolerances are stated against it)
    m64 = unet.UNet(in_channels=1, out_channels=2, n_blocks=n_blocks, start_filts=start_filts,
                    planar_blocks=planar_blocks, activation=activation, normalization=normalization, dim=dim, full_norm=full_norm, merge_mode=merge_mode, up_mode=up_mode, conv_mode=conv_mode, attention=attention, **extra).double()
    m64.load_state_dict({k: torch.as_tensor(v).double() if v.dtype != np.int64 else torch.as_tensor(v) for k, v in sd0.items()})
    m64.train()
    o64 = m64(x.double())
    crit64 = criterion(loss_mod).double()
    l64 = crit64(o64, target)
    l64.backward()
    d['logits64'] = npy(o64).astype(np.float32)  # fp64 result rounded once to fp32 (keeps the fixture small)
    for k, p in m64.named_parameters():
        d['grad64/' + k] = npy(p.grad).astype(np.float32)
    np.savez_compressed(out, **d)
    print('wrote', out, f'{os.path.getsize(out) / 1e6:.2f} MB', 'loss', float(loss))


def make_predictor(unet, inference, out):
    torch.manual_seed(7)
    model = unet.UNet(in_channels=1, out_channels=2, n_blocks=2, start_filts=8, normalization='batch')
    model.train()
    with torch.no_grad():
        for _ in range(3):  # move the running stats off their init values
            model(torch.randn(2, 1, 8, 16, 16) * 1.5 + 0.3)
    sd = {k: npy(v).copy() for k, v in model.state_dict().items()}
    vol = torch.randn(1, 1, 20, 40, 36)
    tile, overlap, out_shape = (8, 16, 16), (4, 8, 8), (2, 20, 40, 36)
    pred = inference.Predictor(model, device='cpu', tile_shape=tile, overlap_shape=overlap, offset=(0, 0, 0),
                               out_shape=out_shape, apply_softmax=True, strict_shapes=False)
    y = pred.predict(vol)
    # untiled whole-volume prediction through the same class
    pred2 = inference.Predictor(model.eval(), device='cpu', apply_softmax=True)
    y2 = pred2.predict(vol)
    # divisible case + argmax output
    vol3 = torch.randn(2, 1, 16, 32, 32)
    pred3 = inference.Predictor(model, device='cpu', tile_shape=tile, overlap_shape=overlap, offset=(0, 0, 0),
                                out_shape=(2, 16, 32, 32), apply_softmax=True, apply_argmax=True)
    y3 = pred3.predict(vol3)
    d = {'vol': npy(vol), 'tile': np.array(tile), 'overlap': np.array(overlap), 'out_shape': np.array(out_shape),
         'out_tiled': npy(y), 'out_untiled': npy(y2), 'vol3': npy(vol3), 'out3_argmax': npy(y3)}
    for k, v in sd.items():
        d['sd/' + k] = v
    np.savez_compressed(out, **d)
    print('wrote', out, f'{os.path.getsize(out) / 1e6:.2f} MB')


def make_trainsteps(unet, loss_mod, out):
    torch.manual_seed(11)
    model = unet.UNet(in_channels=1, out_channels=2, n_blocks=2, start_filts=8, normalization='batch')
    sd0 = {k: npy(v).copy() for k, v in model.state_dict().items()}
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.5e-4)  # train_unet_neurodata.py:257-262 (w/o SWA wrapper)
    crit = criterion(loss_mod)
    xs = torch.randn(3, 2, 1, 8, 16, 16)
    ts = torch.randint(0, 2, (3, 2, 8, 16, 16))
    losses = []
    model.train()
    for i in range(3):
        out_t = model(xs[i])
        loss = crit(out_t, ts[i])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    d = {'xs': npy(xs), 'ts': npy(ts), 'losses': np.array(losses, dtype=np.float64)}
    for k, v in sd0.items():
        d['sd0/' + k] = v
    for k, v in model.state_dict().items():
        d['sd3/' + k] = npy(v)
    np.savez_compressed(out, **d)
    print('wrote', out, f'{os.path.getsize(out) / 1e6:.2f} MB', losses)


def make_adamw(out):
    """torch.optim.AdamW as the example configures it (train_unet_neurodata.py:257-262), lr varied per step like CyclicLR does."""
    torch.manual_seed(5)
    shapes = [(7,), (3, 5), (2, 3, 3, 3, 3), (1030,)]
    ps = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    opt = torch.optim.AdamW(ps, lr=1e-3, weight_decay=0.5e-4, foreach=False)
    lrs = [1e-3, 2e-3, 5e-4, 1e-6, 3e-3]
    d = {'lrs': np.array(lrs), 'n': np.array(len(shapes))}
    for i, p in enumerate(ps):
        d[f'p0/{i}'] = npy(p).copy()
    for t, lr in enumerate(lrs):
        for grp in opt.param_groups:
            grp['lr'] = lr
        for i, p in enumerate(ps):
            p.grad = torch.randn(p.shape) * (10.0 ** (i - 2))      # gradient scales 1e-2 .. 10
            d[f'g{t}/{i}'] = npy(p.grad).copy()
        opt.step()
        for i, p in enumerate(ps):
            d[f'p{t + 1}/{i}'] = npy(p).copy()
            d[f'm{t + 1}/{i}'] = npy(opt.state[p]['exp_avg']).copy()
            d[f'v{t + 1}/{i}'] = npy(opt.state[p]['exp_avg_sq']).copy()
    np.savez_compressed(out, **d)
    print('wrote', out, f'{os.path.getsize(out) / 1e6:.2f} MB')


def make_swa(out):
    """The reference's SWA wrapper (training/swa.py, imported by path: it needs torch only) around torch.optim.SGD on CPU: automatic mode
    (swa_start=2, swa_freq=2) over 8 steps, then swap_swa_sgd.  Pins the running-average arithmetic (two rounded fp32 operations per
    update), the update schedule and the swap."""
    swa = _load('e3ref_swa', f'{REF}/training/swa.py')
    torch.manual_seed(6)
    shapes = [(5,), (3, 4), (2, 3, 3, 3, 3), (1027,)]
    ps = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    opt = swa.SWA(torch.optim.SGD(ps, lr=0.1), swa_start=2, swa_freq=2)
    d = {'n': np.array(len(shapes)), 'steps': np.array(8), 'swa_start': np.array(2), 'swa_freq': np.array(2), 'lr': np.array(0.1)}
    for i, p in enumerate(ps):
        d[f'p0/{i}'] = npy(p).copy()
    for t in range(8):
        for i, p in enumerate(ps):
            p.grad = torch.randn(p.shape)
            d[f'g{t}/{i}'] = npy(p.grad).copy()
        opt.step()
        for i, p in enumerate(ps):
            d[f'p{t + 1}/{i}'] = npy(p).copy()
            if 'swa_buffer' in opt.state[p]:
                d[f'b{t + 1}/{i}'] = npy(opt.state[p]['swa_buffer']).copy()
        d[f'n_avg{t + 1}'] = np.array(opt.param_groups[0]['n_avg'])
    opt.swap_swa_sgd()
    for i, p in enumerate(ps):
        d[f'p_swapped/{i}'] = npy(p).copy()
        d[f'b_swapped/{i}'] = npy(opt.state[p]['swa_buffer']).copy()
    np.savez_compressed(out, **d)
    print('wrote', out, f'{os.path.getsize(out) / 1e6:.2f} MB')


def make_unet_bf16(unet, out, seed=21, n_blocks=2, start_filts=32, shape=(9, 18, 20), batch=2, lowp=torch.bfloat16, dl_scale=1e-3):
    """BASELINE configs[2] ("same UNet bf16"): the reference module cast with model.to(lowp) and fed a bf16 input (the style of
    benchmark/pred_benchmark.py:55,71 and Predictor(float16=True), inference.py:445-446 -- the reference has no bf16 switch of its own,
    SURVEY 0.6), one train step with a given logits gradient; beside it the fp32 run of the SAME bf16-valued weights and input, so that
    a test can state its error relative to the reference's own bf16-vs-fp32 distance.  lowp=torch.float16: the same in IEEE half (model.half())."""
    NAME = 'f16' if lowp == torch.float16 else 'bf16'
    TAG = ':' + NAME      # 16-bit values are stored as their bit patterns under keys with this suffix
    torch.manual_seed(seed)
    model = unet.UNet(in_channels=1, out_channels=2, n_blocks=n_blocks, start_filts=start_filts)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if 'norm' in name and name.endswith('weight'):
                p.copy_(1.0 + 0.2 * torch.randn_like(p))
            elif name.endswith('bias'):
                p.copy_(0.1 * torch.randn_like(p))
        for p in model.parameters():
            p.copy_(p.to(lowp).float())             # bf16-valued parameters: both runs start from identical numbers
    bits = lambda t: npy(t.detach().to(lowp).view(torch.int16)).view(np.uint16)     # bf16 values as their 16-bit patterns (half the file)
    sd0 = {k: npy(v).copy() for k, v in model.state_dict().items()}
    x = torch.randn(batch, 1, *shape).to(lowp)
    dl = (torch.randn(batch, 2, *shape) * dl_scale).to(lowp)
    d = {'cfg.n_blocks': n_blocks, 'cfg.start_filts': start_filts, 'x' + TAG: bits(x), 'dlogits' + TAG: bits(dl)}
    for k, v in model.state_dict().items():
        if v.is_floating_point() and float((v - v.to(lowp).float()).abs().max()) == 0.0:
            d['sd0/' + k + TAG] = bits(v)
        else:
            d['sd0/' + k] = npy(v)
    model.train()
    y32 = model(x.float())
    y32.backward(dl.float())
    d['logits_fp32'] = npy(y32)
    for k, p in model.named_parameters():
        d['grad32/' + k] = npy(p.grad)
    for k, v in model.state_dict().items():
        if 'running' in k:
            d['sd1_fp32/' + k] = npy(v)
    m16 = unet.UNet(in_channels=1, out_channels=2, n_blocks=n_blocks, start_filts=start_filts)
    m16.load_state_dict({k: torch.as_tensor(v) for k, v in sd0.items()})
    m16 = m16.to(lowp).train()
    y16 = m16(x)
    assert y16.dtype == lowp
    y16.backward(dl)
    d['logits_' + NAME + TAG] = bits(y16)
    for k, p in m16.named_parameters():
        d['grad16/' + k + TAG] = bits(p.grad)
    for k, v in m16.state_dict().items():
        if 'running' in k:
            d['sd1_' + NAME + '/' + k + TAG] = bits(v)
    np.savez_compressed(out, **d)
    e = float((y16.float() - y32).abs().max()), float(y32.abs().max())
    print('wrote', out, f'{os.path.getsize(out) / 1e6:.2f} MB', 'reference bf16 vs fp32 logits: max err %.3e at scale %.2f' % e)


TIE_TOL = 2e-6      # |pre-activation| (resp. gap between the two largest values of a pooling window) below which an fp32 run may decide differently


def make_cfg2_digest(unet, loss_mod, out, seed=2024, n_blocks=4, start_filts=32, shape=(64, 128, 128), batch=2, planar_blocks=()):
    """A train step of the reference at BASELINE.json configs[1]'s OWN size -- UNet(1, 2, n_blocks=4, start_filts=32), batch 2 of 64 x 128 x 128: 8192
    Winograd bricks at level 0 -- in fp32 and fp64, kept as a digest (tests/helpers.py: parameters, input and target are regenerated from the seed on
    both sides; the fixture holds samples, norms and projections of what the reference computed).  `python make_golden.py cfg2`, a few minutes of CPU."""
    from collections import OrderedDict
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import digest_state_dict, digest_inputs, digest_of, rel_l2
    kw = dict(in_channels=1, out_channels=2, n_blocks=n_blocks, start_filts=start_filts, planar_blocks=planar_blocks)
    model = unet.UNet(**kw)
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in model.state_dict().items())
    sd0 = digest_state_dict(shapes, seed)
    x_np, t_np = digest_inputs(batch, shape, seed)
    x, target = torch.from_numpy(x_np), torch.from_numpy(t_np)
    d = {'seed': np.array(seed), 'cfg.n_blocks': n_blocks, 'cfg.start_filts': start_filts, 'cfg.planar_blocks': np.array(planar_blocks, dtype=np.int64),
         'batch': np.array(batch), 'shape': np.array(shape), 'names': np.array(list(shapes)), 'shapes': np.array([','.join(str(i) for i in s) for s in shapes.values()])}
    runs = {}
    for tag, dt in (('32', torch.float32), ('64', torch.float64)):
        m = unet.UNet(**kw).to(dt)
        m.load_state_dict({k: torch.as_tensor(v).to(dt) if v.dtype != np.int64 else torch.as_tensor(v) for k, v in sd0.items()})
        m.train()
        stat, handles = tie_hooks(m) if tag == '64' else (None, [])      # near-tie ReLU / arg-max decisions of the fp64 run (stored: the test's allowance for a flipped decision hangs on it)
        o = m(x.to(dt))
        for h in handles:
            h.remove()
        if stat is not None:
            d['act_ties'], d['pool_ties'], d['tie_tol'] = np.array(stat['act_ties']), np.array(stat['pool_ties']), np.array(TIE_TOL)
            print('near-ties of the fp64 run:', stat['act_ties'], '+', stat['pool_ties'], 'of', stat['act_elements'], '+', stat['pool_windows'], flush=True)
        crit = criterion(loss_mod).to(dt)
        l = crit(o, target)
        l.backward()
        runs[tag] = (npy(o), float(l), {k: npy(p.grad) for k, p in m.named_parameters()}, {k: npy(v) for k, v in m.state_dict().items() if 'running' in k})
        print(tag, 'loss', float(l), flush=True)
        del m, o, l
    o32, l32, g32, r32 = runs['32']
    o64, l64, g64, r64 = runs['64']
    d['loss32'], d['loss64'] = np.array(l32), np.array(l64)
    d['logits32'] = o32[:, :, ::8, ::8, ::8].astype(np.float32); d['logits64'] = o64[:, :, ::8, ::8, ::8].astype(np.float64)
    d['logits_err_ref'] = np.array(np.abs(o32 - o64).max())
    for k, v in r32.items():
        d['sd1/' + k] = v
    for k in g64:
        n64, s64, p64 = digest_of(k, g64[k], seed)
        n32, s32, p32 = digest_of(k, g32[k], seed)
        d['g/' + k] = np.concatenate([[n64, n32, rel_l2(g32[k], g64[k])], p64, p32, [len(s64)], s64, s32])
    np.savez_compressed(out, **d)
    print('wrote', out, f'{os.path.getsize(out) / 1e6:.2f} MB')


def tie_hooks(m):
    """Forward hooks on every piecewise-linear activation (count |input| < TIE_TOL) and every max-pool (count windows whose two largest inputs are closer
    than TIE_TOL) of the reference model `m`: returns (statistics dict, filled by the next forwards; hook handles)."""
    import torch.nn as nn
    import torch.nn.functional as F
    stat = {'act_ties': 0, 'pool_ties': 0, 'min_abs_preact': float('inf'), 'min_pool_gap': float('inf'), 'act_elements': 0, 'pool_windows': 0}

    def act_hook(_m, inp, _out):
        z = inp[0].detach().abs()
        stat['act_ties'] += int((z < TIE_TOL).sum())
        stat['min_abs_preact'] = min(stat['min_abs_preact'], float(z.min()))
        stat['act_elements'] += z.numel()

    def pool_hook(pm, inp, _out):
        x = inp[0].detach()
        nd = x.dim() - 2
        ks = pm.kernel_size if isinstance(pm.kernel_size, (tuple, list)) else (pm.kernel_size,) * nd
        pad = []
        for n, k in zip(reversed(x.shape[2:]), reversed(ks)):       # ceil_mode=True: windows may overhang; pad with -inf
            pad += [0, (-n) % k]
        xp = F.pad(x, pad, value=float('-inf'))
        v = xp
        for ax, k in enumerate(ks):                                 # -> (..., n_ax / k, k) per axis, window elements gathered last
            v = v.unflatten(2 + 2 * ax, (xp.shape[2 + ax] // k, k))
        perm = list(range(2)) + [2 + 2 * a for a in range(nd)] + [3 + 2 * a for a in range(nd)]
        w = v.permute(perm).flatten(2 + nd)
        top = w.topk(2, dim=-1).values
        gap = top[..., 0] - top[..., 1]
        # (two exact zeros -- ReLU outputs -- are an exact tie in every implementation: the first one wins, and its gradient dies in the ReLU)
        live = torch.isfinite(gap) & ~((top[..., 0] == 0) & (top[..., 1] == 0))
        gap = gap[live]
        stat['pool_ties'] += int((gap < TIE_TOL).sum())
        stat['min_pool_gap'] = min(stat['min_pool_gap'], float(gap.min())) if gap.numel() else stat['min_pool_gap']
        stat['pool_windows'] += gap.numel()

    handles = []
    for sub_m in m.modules():
        if isinstance(sub_m, (nn.ReLU, nn.LeakyReLU, nn.PReLU, nn.RReLU)):
            handles.append(sub_m.register_forward_hook(act_hook))
        elif isinstance(sub_m, (nn.MaxPool3d, nn.MaxPool2d)):
            handles.append(sub_m.register_forward_hook(pool_hook))
    return stat, handles


def count_ties(m, x64):
    """Near-tie decisions of one fp64 train-mode forward of the reference model `m` on `x64`.  Returns (statistics, output)."""
    stat, handles = tie_hooks(m)
    was_training = m.training
    m.train()
    with torch.no_grad():
        o = m(x64)
    for h in handles:
        h.remove()
    m.train(was_training)
    return {k: (v if np.isfinite(v) else None) for k, v in stat.items()}, o


def make_tie_counts(unet, out):
    """Which train-step fixtures contain a ReLU / max-pool decision that is a near-tie in the reference's fp64 run?

    The gradients of such a fixture are not a smooth function of fp32-level perturbations (a flipped decision moves whole gradient tensors
    by ~4e-3 rel-L2), so tests/test_unet_gpu.py::test_train_step_matches_reference may only demand its tight bound unconditionally where there
    is none.  Per fixture: the reference model is rebuilt in fp64 from the fixture's own cfg / sd0 / x and run in train mode with count_ties().
    The small fixtures are GENERATED tie-free (make_unet_case(tie_free=True) moves the seed until the count is zero); only the Winograd-size
    fixture -- 60 M activations: ~70 within TIE_TOL of zero whatever the seed -- keeps ties.  Written to tie_counts.json."""
    import glob
    import json
    sys.path.insert(0, os.path.dirname(HERE))
    from helpers import unet_cfg
    res = {}
    for path in sorted(glob.glob(f'{HERE}/*.npz')):
        g = np.load(path)
        if 'grad64/conv_final.weight' not in g.files or 'cfg.n_blocks' not in g.files:
            continue
        cfg = unet_cfg(g)
        mod = unet
        if 'enc_res_blocks' in cfg:
            mod = _load('elektronn3.models.resunet', f'{REF}/models/resunet.py')
        m = mod.UNet(in_channels=1, out_channels=2, **cfg).double()
        sd0 = {k[4:]: g[k] for k in g.files if k.startswith('sd0/')}
        m.load_state_dict({k: torch.as_tensor(v).double() if v.dtype != np.int64 else torch.as_tensor(v) for k, v in sd0.items()})
        stat, o = count_ties(m, torch.as_tensor(g['x']).double())
        assert np.abs(npy(o).astype(np.float32) - g['logits64']).max() < 1e-6, path      # the rebuilt fp64 run IS the fixture's
        res[os.path.basename(path)] = stat
        print(os.path.basename(path), stat)
    json.dump({'tol': TIE_TOL, 'cases': res}, open(out, 'w'), indent=1, sort_keys=True)
    print('wrote', out)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'ties':      # near-tie ReLU / max-pool decisions of every train-step fixture (tie_counts.json)
        unet, inference, loss_mod = load_reference()
        make_tie_counts(unet, f'{HERE}/tie_counts.json')
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'cfg2':      # digest of a train step at BASELINE configs[1]'s own size (cfg2_digest.npz)
        torch.set_num_threads(8)
        unet, inference, loss_mod = load_reference()
        make_cfg2_digest(unet, loss_mod, f'{HERE}/cfg2_digest.npz')
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'cfg4':      # the same for BASELINE configs[3]: anisotropic UNet(planar_blocks=(0, 1), start_filts=64), batch 2 of 32 x 256 x 256 (cfg4_digest.npz; ~20 GB, tens of minutes)
        torch.set_num_threads(8)
        unet, inference, loss_mod = load_reference()
        make_cfg2_digest(unet, loss_mod, f'{HERE}/cfg4_digest.npz', seed=2025, n_blocks=4, start_filts=64, shape=(32, 256, 256), batch=1, planar_blocks=(0, 1))      # (batch 1: the fp64 run of batch 2 does not fit this container's 62 GB)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'f16':      # the reference in float16 (model.half(), inference.py:445-446): O(1) incoming gradient, as GradScaler provides
        torch.set_num_threads(8)
        unet, _, _ = load_reference()
        make_unet_bf16(unet, f'{HERE}/unet_nb2_sf32_f16.npz', lowp=torch.float16, dl_scale=1.0)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'bf16':
        torch.set_num_threads(8)
        unet, _, _ = load_reference()
        make_unet_bf16(unet, f'{HERE}/unet_nb2_sf32_bf16.npz')
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'swa':
        make_swa(f'{HERE}/swa.npz')
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'adamw':      # only the optimizer fixture (needs torch, not the reference)
        make_adamw(f'{HERE}/adamw.npz')
        sys.exit(0)
    torch.set_num_threads(8)
    unet, inference, loss_mod = load_reference()
    # ---- the train-step fixtures: (file, command-line group, reference module, arguments).  `python make_golden.py <group>` writes one group, no argument all
    # of them + the other fixtures.  All but the Winograd-size one are generated tie-free (make_unet_case: the seed moves by 1000 until the fp64 run has no
    # ReLU / arg-max decision within TIE_TOL of a tie; the seed used is stored in the file).
    U, R = 'unet', 'resunet'
    UNET_CASES = [
        # cfg 1 of BASELINE.json at a reduced crop: UNet(1,2,n_blocks=2,start_filts=8)
        ('unet_nb2_sf8', 'base', U, dict(seed=0, n_blocks=2, start_filts=8, planar_blocks=(), shape=(16, 24, 24), batch=1)),
        # odd sizes (ceil-mode pooling + autocrop of the up-convolved tensor), batch 2, planar first block
        ('unet_nb3_sf8_planar0_odd', 'base', U, dict(seed=1, n_blocks=3, start_filts=8, planar_blocks=(0,), shape=(9, 17, 21), batch=2)),
        # the headline depth (n_blocks=4) at start_filts=8, cfg-4 style mixed 3D/2D (planar_blocks=(0,1))
        ('unet_nb4_sf8_planar01', 'base', U, dict(seed=2, n_blocks=4, start_filts=8, planar_blocks=(0, 1), shape=(8, 32, 32), batch=2)),
        # dim=2 (Conv2d/BatchNorm2d/MaxPool2d/ConvTranspose2d), odd sizes
        ('unet2d_nb3_sf8_odd', 'unet2d', U, dict(seed=3, n_blocks=3, start_filts=8, planar_blocks=(), shape=(37, 46), batch=2, dim=2)),
        # nn.Identity norms: normalization='none'; full_norm=False ("sparse" normalization scheme of the examples' comments)
        ('unet_nb2_sf8_nonorm', 'norms', U, dict(seed=4, n_blocks=2, start_filts=8, planar_blocks=(), shape=(10, 13, 18), batch=2, normalization='none')),
        ('unet_nb3_sf8_planar0_sparsenorm', 'norms', U, dict(seed=5, n_blocks=3, start_filts=8, planar_blocks=(0,), shape=(8, 18, 21), batch=2, full_norm=False)),
        # merge_mode='add' (skip connection summed instead of concatenated), odd sizes, planar middle block
        ('unet_nb3_sf8_add_odd', 'add', U, dict(seed=6, n_blocks=3, start_filts=8, planar_blocks=(1,), shape=(9, 14, 19), batch=2, merge_mode='add')),
        # nn.InstanceNorm3d norms (no parameters, per-sample statistics in training and eval mode)
        ('unet_nb3_sf8_instance', 'instance', U, dict(seed=7, n_blocks=3, start_filts=8, planar_blocks=(0,), shape=(8, 17, 20), batch=2, normalization='instance')),
        # nn.GroupNorm(4, C) norms (affine, per-sample group statistics, no running stats), odd sizes
        ('unet_nb3_sf8_group4_odd', 'group', U, dict(seed=8, n_blocks=3, start_filts=8, planar_blocks=(0,), shape=(9, 14, 21), batch=2, normalization='group4')),
        # other activations: LeakyReLU(0.1) with BatchNorm, identity ('lin') without a norm, SiLU
        ('unet_nb3_sf8_leaky_odd', 'act', U, dict(seed=9, n_blocks=3, start_filts=8, planar_blocks=(1,), shape=(10, 13, 19), batch=2, activation='leaky')),
        ('unet_nb2_sf8_lin_nonorm', 'act', U, dict(seed=10, n_blocks=2, start_filts=8, planar_blocks=(), shape=(6, 10, 12), batch=2, activation='lin', normalization='none')),
        ('unet_nb3_sf8_silu_odd', 'silu', U, dict(seed=11, n_blocks=3, start_filts=8, planar_blocks=(0,), shape=(9, 15, 18), batch=2, activation='silu')),
        # up_mode='resizeconv_*' (up-sampling + conv instead of the transposed conv), odd sizes, planar blocks
        ('unet_nb3_sf8_resizeconv_odd', 'resize', U, dict(seed=12, n_blocks=3, start_filts=8, planar_blocks=(0,), shape=(9, 14, 19), batch=2, up_mode='resizeconv_nearest')),
        ('unet_nb3_sf8_resizelinear_odd', 'resizelin', U, dict(seed=13, n_blocks=3, start_filts=8, planar_blocks=(0,), shape=(7, 15, 18), batch=2, up_mode='resizeconv_linear')),
        ('unet_nb3_sf8_resizenearest1_odd', 'resize1', U, dict(seed=14, n_blocks=3, start_filts=8, planar_blocks=(1,), shape=(9, 13, 18), batch=2, up_mode='resizeconv_nearest1')),
        # nn.PReLU(1) activations (learnable slopes) with the sparse norm scheme: slopes behind a norm and behind nn.Identity
        ('unet_nb3_sf8_prelu_odd', 'prelu', U, dict(seed=15, n_blocks=3, start_filts=8, planar_blocks=(0,), shape=(8, 15, 19), batch=2, activation='prelu', full_norm=False)),
        # conv_mode='valid' (padding 0: shrinking grids, centre-cropped skips), planar first block, odd sizes
        ('unet_nb3_sf8_valid', 'valid', U, dict(seed=16, n_blocks=3, start_filts=8, planar_blocks=(0,), shape=(21, 45, 47), batch=2, conv_mode='valid')),
        # attention=True (GridAttention, unet.py:452-541): odd sizes (phi(g) and the gate are resized), dim=2, conv_mode='valid' + a planar block (theta halves
        # the depth the pooling kept), merge_mode='add'
        ('unet_nb3_sf8_attention_odd', 'attention', U, dict(seed=17, n_blocks=3, start_filts=8, planar_blocks=(), shape=(9, 14, 19), batch=2, attention=True)),
        ('unet2d_nb3_sf8_attention', 'attention', U, dict(seed=18, n_blocks=3, start_filts=8, planar_blocks=(), shape=(37, 46), batch=2, dim=2, attention=True)),
        ('unet_nb3_sf8_attention_valid_planar0', 'attention', U, dict(seed=19, n_blocks=3, start_filts=8, planar_blocks=(0,), shape=(22, 45, 47), batch=2, conv_mode='valid', attention=True)),
        ('unet_nb3_sf8_attention_add', 'attention', U, dict(seed=20, n_blocks=3, start_filts=8, planar_blocks=(), shape=(8, 12, 16), batch=2, merge_mode='add', activation='leaky', attention=True)),
        # elektronn3.models.resunet.UNet (resunet.py:598-934): plain ConvBlocks, residual ones (identity and projected shortcuts, several per block), with planar
        # blocks / attention / merge 'add' / no norm
        ('resunet_nb3_sf8_res00', 'resunet', R, dict(seed=21, n_blocks=3, start_filts=8, planar_blocks=(), shape=(8, 12, 16), batch=2, res_blocks=(0, 0))),
        ('resunet_nb3_sf8_res21_odd', 'resunet', R, dict(seed=22, n_blocks=3, start_filts=8, planar_blocks=(0,), shape=(9, 14, 19), batch=2, res_blocks=(2, 1))),
        ('resunet_nb3_sf8_res12_add_attention', 'resunet', R, dict(seed=23, n_blocks=3, start_filts=8, planar_blocks=(), shape=(9, 13, 18), batch=2, merge_mode='add', activation='leaky', attention=True, res_blocks=(1, 2))),
        ('resunet_nb2_sf8_res11_nonorm', 'resunet', R, dict(seed=24, n_blocks=2, start_filts=8, planar_blocks=(), shape=(10, 12, 14), batch=2, normalization='none', res_blocks=(1, 1))),
    ]
    resunet = None

    def run_cases(group):
        global_resunet = None
        for name, grp, which, kw in UNET_CASES:
            if group is not None and grp != group:
                continue
            mod = unet
            if which == R:
                if global_resunet is None:
                    global_resunet = _load('elektronn3.models.resunet', f'{REF}/models/resunet.py')
                mod = global_resunet
            make_unet_case(mod, loss_mod, f'{HERE}/{name}.npz', tie_free=True, **kw)

    if len(sys.argv) > 1 and sys.argv[1] == 'wino':      # a size at which the fp32 Winograd kernels run: start_filts=32, batch 2 of 31 x 61 x 67 -- 1280 bricks at level 0
        # (the persistent kernel), 384 at level 1 (one brick per workgroup), all-odd extents (partial bricks and tiles in every dimension, autocrop).  NOT tie-free:
        # 60 M activations have ~70 pre-activations within TIE_TOL of zero whatever the seed (tie_counts.json)
        make_unet_case(unet, loss_mod, f'{HERE}/unet_nb2_sf32_wino_odd.npz', seed=32, n_blocks=2, start_filts=32, planar_blocks=(), shape=(31, 61, 67), batch=2)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'rrelu':
        make_rrelu_eval(unet, f'{HERE}/unet_nb3_sf8_rrelu_eval.npz')
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'steps':     # every train-step fixture of the table
        run_cases(None)
        sys.exit(0)
    if len(sys.argv) > 1:
        assert sys.argv[1] in {g for _, g, _, _ in UNET_CASES}, f'unknown fixture group {sys.argv[1]}'
        run_cases(sys.argv[1])
        sys.exit(0)
    make_ops(unet, f'{HERE}/ops.npz')
    run_cases(None)
    make_rrelu_eval(unet, f'{HERE}/unet_nb3_sf8_rrelu_eval.npz')
    make_predictor(unet, inference, f'{HERE}/predictor.npz')
    make_adamw(f'{HERE}/adamw.npz')
    make_swa(f'{HERE}/swa.npz')
    make_trainsteps(unet, loss_mod, f'{HERE}/trainsteps.npz')
