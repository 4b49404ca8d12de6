"""Randomised whole-network check of the native bf16 path: random depth / widths / input channels / batch / odd crop extents, one train step of a
bfloat16 module against the fp32 HIP path on the same bf16-valued parameters and input (the bound of tests/test_bf16_gpu.py: logits within bf16
accumulation noise, per-tensor gradients at bf16 resolution), plus an eval-mode forward.  Finds indexing defects of the brick / tile / split-K /
two-tensor variants on shapes the fixed tests do not hit.    python tools/fuzz_bf16.py [n_cases] [seed]
The gradient bound against fp32 is loose by nature (0.2-0.8 rel-L2 for deep random nets), so the same cases are ALSO run in a second process with
the alternative kernels of every op (2-deep bricks, 2x16 tiles, no split-K, generic transposed convs, first conv on the VALU) and compared with the
default ones and the deviation printed.  Measured: the two kernel sets differ from each other by almost as much as either differs from fp32 (logits
1-2 % of scale, gradients 0.1-0.4 rel-L2 on deep nets): another summation order flips bf16 roundings, those flip ReLU / arg-max decisions, and the
gradients of a random net move.  Whole-net gradient distances therefore only catch gross defects (bound 0.9); the sharp statements are per op
(tools/fuzz_ops_bf16.py, tests/test_bf16_gpu.py: every output within bf16 rounding of fp64)."""
import os, sys, copy, subprocess, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd.unet import UNet

ALT_ENV = dict(E3_B16_BD='2', E3_B16_TW='16', E3_B16_NO_SPLITK='1', E3_B16_UP_GENERIC='1', E3_B16_FIRST_VALU='1')
dump = os.environ.get('E3_FUZZ_DUMP')            # (set in the child process: save the bf16 results, skip the checks)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
dev = torch.device('cuda:0')
bf = lambda t: t.to(torch.bfloat16)
bad = 0
saved = []
for case in range(n_cases):
    nb = ri(2, 4); sf = 32 * ri(1, 2); inc = ri(1, 3); outc = ri(2, 4)
    mult = 2 ** (nb - 1)
    D = ri(1, 4) * mult + (ri(0, 3) if ri(0, 1) else 0); H = ri(1, 6) * mult + ri(0, 5); W = ri(1, 12) * mult + ri(0, 7); N = ri(1, 3)
    dim = 2 if ri(0, 4) == 0 else 3
    planar = () if dim == 2 or ri(0, 1) else tuple(i for i in range(nb) if ri(0, 1))      # (planar blocks: 1x3x3 convs, (1,2,2) pooling / up-convolution)
    if dim == 2: H, W = H * ri(1, 6), W * ri(1, 3)
    torch.manual_seed(case)
    m32 = UNet(inc, outc, n_blocks=nb, start_filts=sf, planar_blocks=planar, dim=dim).to(dev)
    with torch.no_grad():
        for p in m32.parameters():
            p.copy_(bf(p).float())
    m16 = copy.deepcopy(m32).to(torch.bfloat16)
    sp = (H, W) if dim == 2 else (D, H, W)
    x = bf(torch.randn(N, inc, *sp, generator=g)); dl = bf(torch.randn(N, outc, *sp, generator=g) * 1e-3)
    res = {}
    for tag, m, xx, dd in (('f32', m32, x.float(), dl.float()), ('bf16', m16, x, dl)):
        m.train(); m.zero_grad(set_to_none=True)
        y = m(xx.to(dev)); y.backward(dd.to(dev)); torch.cuda.synchronize()
        m.eval()
        with torch.no_grad():
            ye = m(xx.to(dev))
        res[tag] = (y.detach().float().cpu(), {k: p.grad.float().cpu() for k, p in m.named_parameters()}, ye.float().cpu())
    (y32, g32, e32), (y16, g16, e16) = res['f32'], res['bf16']
    saved.append((y16, g16, e16))
    if dump:
        continue
    scale = float(y32.abs().max()); err = (y16 - y32).abs().flatten()
    p999 = float(err.kthvalue(max(1, int(0.999 * err.numel()))).values)
    escale = float(e32.abs().max()); eerr = float((e16 - e32).abs().max())
    gscale = max(float(v.norm()) for v in g32.values())
    worst, wk = 0.0, ''
    for k, v in g32.items():
        if k.endswith('.bias') and 'norm' not in k and not k.startswith('conv_final'):
            continue
        rel = float((g16[k] - v).norm() / max(float(v.norm()), 1e-3 * gscale))
        if rel > worst: worst, wk = rel, k
    ok = p999 < 4e-2 * scale and float(err.max()) < 1e-1 * scale and worst < 0.9 and eerr < 1e-1 * escale and bool(torch.isfinite(y16).all())
    bad += not ok
    print(f'{"ok " if ok else "BAD"} nb={nb} sf={sf} in={inc} out={outc} planar={planar} x={(N,) + sp}: logits p99.9 {p999 / scale:.2e} max {float(err.max()) / scale:.2e} of scale; '
          f'eval max {eerr / escale:.2e}; worst gradient {worst:.3f} ({wk})', flush=True)
if dump:
    torch.save(saved, dump)
    sys.exit(0)
with tempfile.TemporaryDirectory() as td:
    f = os.path.join(td, 'alt.pt')
    subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:3], env={**os.environ, **ALT_ENV, 'E3_FUZZ_DUMP': f}, check=True)
    alt = torch.load(f)
vbad = 0
for case, ((y, gr, e), (ya, gra, ea)) in enumerate(zip(saved, alt)):
    sc = float(y.abs().max())
    dy = float((y - ya).abs().max()) / sc; de = float((e - ea).abs().max()) / max(float(e.abs().max()), 1e-30)
    gscale = max(float(v.norm()) for v in gr.values())
    worst, wk = 0.0, ''
    for k, v in gr.items():
        if k.endswith('.bias') and 'norm' not in k and not k.startswith('conv_final'):
            continue
        rel = float((gra[k] - v).norm() / max(float(v.norm()), 1e-3 * gscale))
        if rel > worst: worst, wk = rel, k
    ok = dy < 1e-1 and de < 5e-2 and worst < 0.9
    vbad += not ok
    print(f'{"ok " if ok else "BAD"} case {case}: default vs alternative kernels: logits {dy:.2e}, eval {de:.2e} of scale; worst gradient rel-L2 {worst:.3f} ({wk})')
print(f'{bad} beyond the fp32 bound, {vbad} kernel-variant disagreements, of {n_cases}')
sys.exit(1 if vbad else 0)
