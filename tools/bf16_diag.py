"""Per-tensor errors of the native bf16 path on the reference fixture: ours vs the reference's fp32 run, beside the reference's own
bf16-vs-fp32 distance.  python tools/bf16_diag.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import load_bf16_fixture  # noqa: E402
from elektronn3_amd.unet import UNet  # noqa: E402

g = load_bf16_fixture(os.path.join(ROOT, 'tests', 'golden', 'unet_nb2_sf32_bf16.npz'))
sd = {k[4:]: v for k, v in g.items() if k.startswith('sd0/')}
BF = torch.bfloat16
for mode in ('bf16', 'fp32'):
    m = UNet(1, 2, n_blocks=int(g['cfg.n_blocks']), start_filts=int(g['cfg.start_filts']))
    m.load_state_dict(sd)
    m = m.to('cuda').train()
    x, dl = g['x'].cuda(), g['dlogits'].cuda()
    if mode == 'bf16':
        m = m.to(BF); x = x.to(BF); dl = dl.to(BF)
    y = m(x); y.backward(dl.to(y.dtype)); torch.cuda.synchronize()
    r32, r16 = g['logits_fp32'], g['logits_bf16']
    print(f'== {mode}: logits err vs ref fp32 {float((y.float().cpu() - r32).abs().max()):.3e}   (ref bf16 vs ref fp32 {float((r16 - r32).abs().max()):.3e}, scale {float(r32.abs().max()):.2f})')
    for k, p in m.named_parameters():
        a32, a16 = g['grad32/' + k], g['grad16/' + k]
        n = float(a32.norm())
        print(f'{k:34s} |g| {n:9.3e}  ours {float((p.grad.float().cpu() - a32).norm()) / max(n, 1e-30):9.3e}   ref-bf16 {float((a16 - a32).norm()) / max(n, 1e-30):9.3e}')
