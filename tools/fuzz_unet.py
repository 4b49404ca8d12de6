"""Randomised whole-network fuzzing: UNet / ResUNet configs (attention gates, residual blocks included) / crop sizes (odd extents, planar blocks, batch sizes) against the reference's
ATen op sequence run by PyTorch-ROCm IN FP64 (oracle/torch_ref.py) -- train-mode forward, loss, all gradients, running
statistics.  fp64 because PyTorch-ROCm's own fp32 batch-norm statistics are only good to ~4e-6 for channel counts such as 24/48/96,
which train-mode normalisation then amplifies to 1e-3 in the output (ours agree with fp64 to 4e-11).
Gradients are only comparable when no ReLU input sits within fp32 rounding of zero: one voxel whose mask flips changes a BN bias
gradient of these tiny crops by ~1e-2 (verified: the error equals that voxel's incoming gradient exactly).  The fp64 run records the
smallest |pre-ReLU value| and the smallest gap between the two largest values of a max-pool window; cases with a margin < 3e-5 get
the loose gradient bound -- after subtracting 3 x the movement the fp64 reference ITSELF shows when its input is dithered at fp32-noise
level (which settles whether a deviation is a decision flip or a defect) -- all others the tight one (threshold 3e-5: BN-scaled fp32 rounding of pre-activations of O(1..10)).
Usage: python tools/fuzz_unet.py [n_cases] [seed]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd.unet import UNet
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from helpers import is_prebn_bias
import oracle.torch_ref as R
from oracle.torch_ref import combined_loss, unet_forward
_relu = torch.nn.functional.relu
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
bad = 0
FLIP = 3e-5      # smallest |pre-activation| / arg-max gap below which fp32 rounding of the two implementations may decide differently
for case in range(n_cases):
    nb = ri(2, 4); sf = 8 * ri(1, 4); inc = ri(1, 2); outc = ri(2, 3)
    planar = tuple(sorted(set(ri(0, nb - 1) for _ in range(ri(0, 2))))) if ri(0, 1) else ()
    planar = tuple(range(len(planar)))   # the reference examples only make the first blocks planar
    mult = 2 ** (nb - 1)
    D = ri(1, 5) * mult + (ri(0, 3) if ri(0, 1) else 0); H = ri(2, 9) * mult + ri(0, 5); W = ri(2, 10) * mult + ri(0, 5)
    N = ri(1, 3)
    kw = {}
    v = ri(0, 7)                                      # option variants: dim=2, normalization='none'/'instance', full_norm=False, merge_mode='add'
    if v == 0: kw, planar, D = dict(dim=2), (), None
    elif v == 1: kw = dict(normalization='none')
    elif v == 2: kw = dict(full_norm=False)
    elif v == 3: kw = dict(normalization='instance', full_norm=bool(ri(0, 1)))
    elif v == 4: kw = dict(normalization=('group', 'group4', 'group2')[ri(0, 2)], full_norm=bool(ri(0, 1)))
    if ri(0, 3) == 0: kw['merge_mode'] = 'add'
    if ri(0, 3) == 0 and 'merge_mode' not in kw: kw['up_mode'] = ('resizeconv_nearest', 'resizeconv_linear', 'resizeconv_nearest1', 'resizeconv_linear1')[ri(0, 3)]
    if ri(0, 3) == 0:
        kw['conv_mode'] = 'valid'          # every conv shrinks the grid by 2: needs a larger input
    if ri(0, 3) == 0: kw['activation'] = ('leaky', 'lin', 'silu', 'prelu')[ri(0, 3)]
    per_sample = kw.get('normalization', 'batch') == 'instance' or str(kw.get('normalization', '')).startswith('group')
    if ri(0, 3) == 0 and not per_sample: kw['attention'] = True          # GridAttention gates (their BatchNorm needs the whole batch)
    res = None
    if ri(0, 3) == 0 and D is not None and kw.get('normalization') != 'instance':      # elektronn3.models.resunet.UNet
        res = (ri(0, 2), ri(0, 2))
        kw.pop('full_norm', None)
        if res[0] or res[1]: kw.pop('conv_mode', None)
    shape = (H, W) if D is None else (D, H, W)
    if kw.get('conv_mode') == 'valid':
        shape = tuple(s_ + 4 * (2 ** nb) for s_ in shape)
    torch.manual_seed(case)
    print(f'case {case}: nb={nb} sf={sf} in={inc} out={outc} planar={planar} {kw} res={res} N={N} shape={shape}', flush=True)
    try:
        if res is not None:
            from elektronn3_amd.resunet import UNet as ResUNet
            m = ResUNet(in_channels=inc, out_channels=outc, n_blocks=nb, start_filts=sf, planar_blocks=planar, enc_res_blocks=res[0], dec_res_blocks=res[1], **kw).cuda().train()
        else:
            m = UNet(in_channels=inc, out_channels=outc, n_blocks=nb, start_filts=sf, planar_blocks=planar, **kw).cuda().train()
    except Exception as e:
        print('skip (ctor):', nb, sf, planar, e); continue
    x = torch.randn(N, inc, *shape, device='cuda')
    try:
        with torch.no_grad(): oshape = tuple(m.eval()(x[:1]).shape[2:]); m.train()
    except ValueError as e:
        print('skip (too small for valid):', shape, e); continue
    t = torch.randint(0, outc, (N, *oshape), device='cuda')
    cw = tuple(float(v) for v in (torch.rand(outc, generator=g) + 0.2))
    with torch.no_grad():
        for k, p in m.named_parameters():
            if '.act' in k: p.copy_(0.25 + 0.3 * torch.randn_like(p))      # PReLU slopes: distinct, some negative
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    out = m(x); loss = combined_loss(out, t, cw); m.zero_grad(set_to_none=True); loss.backward()
    sd_ref = {k: (v.double() if v.is_floating_point() else v.clone()).requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd0.items()}
    paramless = R.instance_norm_names(nb, kw.get('full_norm', True)) if kw.get('normalization') == 'instance' else ()
    sd_ref['__instance_norms__'] = paramless
    sd_ref['__valid__'] = kw.get('conv_mode') == 'valid'
    sd_ref['__up_linear__'] = str(kw.get('up_mode')).startswith('resizeconv_linear')
    sd_ref['__act_slope__'] = {'relu': 0.0, 'leaky': 0.1, 'lin': 1.0, 'silu': 2.0, 'prelu': 3.0}[kw.get('activation', 'relu')]
    group = str(kw.get('normalization', '')).startswith('group')
    sd_ref['__num_groups__'] = (8 if kw['normalization'] == 'group' else int(kw['normalization'][5:])) if group else 0
    margin = [float('inf')]
    def rec_relu(z, *a, **k):
        margin[0] = min(margin[0], float(z.detach().abs().min())); return _relu(z, *a, **k)
    _leaky = torch.nn.functional.leaky_relu
    def rec_leaky(z, *a, **k):                      # (a LeakyReLU mask flip changes the local derivative by 0.9 instead of 1)
        margin[0] = min(margin[0], float(z.detach().abs().min())); return _leaky(z, *a, **k)
    _mp3, _mp2 = torch.nn.functional.max_pool3d, torch.nn.functional.max_pool2d
    def rec_pool(fn):                               # arg-max decisions: smallest non-zero gap between a window's two largest values
        def f(z, **k):                              # (exact ties, e.g. ReLU zeros, resolve identically in both implementations)
            m1, idx = fn(z, return_indices=True, **k)
            z2 = z.detach().clone().flatten(2); z2.scatter_(2, idx.flatten(2), float('-inf'))
            m2 = fn(z2.view_as(z), **k)
            gap = (m1.detach() - m2)[torch.isfinite(m2)]
            gap = gap[gap > 0]
            if gap.numel(): margin[0] = min(margin[0], float(gap.min()))
            return m1
        return f
    _prelu = torch.nn.functional.prelu
    def rec_prelu(z, *a, **k):
        margin[0] = min(margin[0], float(z.detach().abs().min())); return _prelu(z, *a, **k)
    R.F.prelu = rec_prelu
    R.F.relu = rec_relu; R.F.max_pool3d = rec_pool(_mp3); R.F.max_pool2d = rec_pool(_mp2)
    if sd_ref['__act_slope__'] == 0.1: R.F.leaky_relu = rec_leaky
    fwd = (lambda sd_, x_: R.resunet_forward(sd_, x_, nb, planar, True, res[0], res[1])) if res is not None else (lambda sd_, x_: unet_forward(sd_, x_, nb, planar, training=True))
    try: ref = fwd(sd_ref, x.double())
    finally: R.F.relu = _relu; R.F.prelu = _prelu; R.F.leaky_relu = _leaky; R.F.max_pool3d = _mp3; R.F.max_pool2d = _mp2; lref = combined_loss(ref, t, cw); lref.backward()
    e_out = float((out - ref).detach().abs().max()) / max(1.0, float(ref.abs().max()))
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())))
    worst, wk = 0.0, ''
    detail = []
    names = {k for k, _ in m.named_parameters()}
    # In the flip regime the fp64 reference itself tells how far a decision flip moves each gradient: re-run it on inputs dithered by
    # 3e-6 relative (the size of the fp32 rounding noise that reaches a BN-scaled pre-activation) and allow every tensor 3 x the largest movement seen.
    sens = {}
    if margin[0] < FLIP:
        for trial in range(4):
            sd2 = {k: (v.detach().clone().requires_grad_(v.requires_grad) if torch.is_tensor(v) else v) for k, v in sd_ref.items()}
            xd = x.double() * (1 + 3e-6 * torch.randn(x.shape, device=x.device, dtype=torch.float64, generator=torch.Generator(device=x.device).manual_seed(77 + trial)))
            combined_loss(fwd(sd2, xd), t, cw).backward()
            for k in names:
                sens[k] = max(sens.get(k, 0.0), float((sd2[k].grad - sd_ref[k].grad).norm()))
    act_gmax = max([float(sd_ref[k].grad.norm()) for k in names if '.act' in k] or [0.0])
    for k, p in m.named_parameters():
        gr = sd_ref[k].grad
        prebn = is_prebn_bias(k, set() if group else names, paramless)
        if group and p.numel() == sd_ref['__num_groups__'] and is_prebn_bias(k, names, paramless):
            prebn = True        # groups of ONE channel: GroupNorm removes the channel's own mean, the bias gradient is analytically zero
        if float(gr.norm()) < 1e-10 * gn:
            prebn = True        # analytically zero in fp64 (e.g. a norm bias whose shift the next norm removes through an identity activation): absolute tolerance
        floor = 1e-4 * gn
        if '.act' in k:     # a PReLU slope gradient is ONE scalar (a sum over the tensor with heavy cancellation): judged against at least
            floor = max(floor, (1e-1 if margin[0] < FLIP else 1e-3) * act_gmax)     # 1e-3 of the largest slope gradient of the network (as tests/test_unet_gpu.py does);
                                                                                  # a flipped mask moves it by as much as any other tensor, in ABSOLUTE terms
        err = float(p.grad.abs().max()) / gn if prebn else float(((p.grad - gr).norm() - 3 * sens.get(k, 0.0)).clamp_min(0) / gr.norm().clamp_min(floor))     # (gradients that are analytically ~0, e.g. a norm bias whose
                                                                                       # shift the next norm removes entirely, are judged against the global scale)
        if '.act' in k and margin[0] >= FLIP: err /= 3.0      # (one scalar summed over a whole tensor in fp32: 3e-4 of the floor above is summation noise, seen at margin 4e-5)
        if (not prebn and err > worst): worst, wk = err, k
        if not prebn: detail.append((err, k, float(gr.norm()), float(p.grad.norm())))
        if prebn and err > 1e-5: worst, wk = 1.0, k + ' (pre-BN bias not ~0)'
    e_rs = max([float((m.state_dict()[k] - sd_ref[k]).abs().max()) for k in sd0 if 'running' in k] or [0.0])
    # one flipped ReLU mask moves a per-channel sum over n voxels by ~1/sqrt(n): the loose bound follows the smallest level
    n_bottom = N
    for i, v in enumerate(shape):
        n_bottom *= -(-v // (1 if (len(shape) == 3 and i == 0 and all(b in planar for b in range(nb - 1))) else mult))
    loose = max(3e-2, 1.0 / n_bottom ** 0.5)
    ok = e_out < 5e-5 and worst < (loose if margin[0] < FLIP else 1e-4) and e_rs < 1e-5
    bad += not ok
    print(f'{"ok  " if ok else "BAD "} nb={nb} sf={sf} in={inc} out={outc} planar={planar} {kw} res={res} N={N} {"x".join(map(str, shape))}: out {e_out:.1e} worst grad {worst:.1e} ({wk}) running {e_rs:.1e} relu margin {margin[0]:.0e}', flush=True)
    if not ok and os.environ.get('FUZZ_VERBOSE'):
        for e_, k_, a_, b_ in sorted(detail, reverse=True)[:8]: print(f'      {k_:34s} err {e_:.2e} |ref| {a_:.3e} |ours| {b_:.3e}  (global {gn:.3e}, act max {act_gmax:.3e})')
print('BAD CASES:', bad)
sys.exit(1 if bad else 0)
