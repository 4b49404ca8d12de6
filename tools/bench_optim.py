"""Optimizer step on cfg 2's parameter set (70 tensors, 5.6 M fp32 elements): elektronn3_amd.optim.AdamW (one launch) against
torch.optim.AdamW (foreach, the default on GPU) and torch's fused=True variant.  Wall time per step incl. Python, and GPU time."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd.unet import UNet
from elektronn3_amd.optim import AdamW

m = UNet(in_channels=1, out_channels=2, n_blocks=4, start_filts=32, normalization='batch').cuda()
ps = list(m.parameters())
for p in ps: p.grad = torch.randn_like(p)
n = sum(p.numel() for p in ps)
for name, mk in (('elektronn3_amd.optim.AdamW', lambda: AdamW(ps, lr=1e-3, weight_decay=0.5e-4)),
                 ('torch.optim.AdamW (foreach)', lambda: torch.optim.AdamW(ps, lr=1e-3, weight_decay=0.5e-4)),
                 ('torch.optim.AdamW (fused=True)', lambda: torch.optim.AdamW(ps, lr=1e-3, weight_decay=0.5e-4, fused=True))):
    opt = mk()
    for _ in range(5): opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(50): opt.step()
    e1.record(); torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 50 * 1e6
    gpu = e0.elapsed_time(e1) / 50 * 1e3
    print(f'{name:34s} {len(ps)} tensors {n} elements: {wall:7.1f} us wall / step, {gpu:7.1f} us on the stream  '
          f'({28.0 * n / (gpu * 1e-6) / 1e12:.2f} TB/s of the 28 B/element)', flush=True)
