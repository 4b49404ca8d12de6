#!/usr/bin/env python3
"""Micro-benchmark of the bf16 transposed-conv kernels on the cfg-2 shapes.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from elektronn3_amd import ops

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

for name, (cin, cout, dims) in {'L1->L0 64->32': (64, 32, (32, 64, 64)), 'L2->L1 128->64': (128, 64, (16, 32, 32)), 'L3->L2 256->128': (256, 128, (8, 16, 16))}.items():
    D, H, W = dims
    x = torch.randn(2, D, H, W, cin, device='cuda').bfloat16()
    w = torch.randn(cin, cout, 2, 2, 2, device='cuda') * 0.1
    b = torch.zeros(cout, device='cuda')
    cat = torch.empty(2, 2 * D, 2 * H, 2 * W, 2 * cout, device='cuda', dtype=torch.bfloat16)     # written into the first half of a concat buffer
    dy = torch.randn(2, 2 * D, 2 * H, 2 * W, cout, device='cuda').bfloat16()
    f = timeit(lambda: ops.convT_bf16(x, w, b, want_stats=True, out=cat[..., :cout]))
    d = timeit(lambda: ops.convT_dgrad_bf16(dy, w, (D, H, W)))
    g = timeit(lambda: ops.convT_wgrad_bf16(x, dy))
    mb = (x.numel() + dy.numel()) * 2 / 1e6
    print(f'{name:18s} {mb:7.1f} MB  fwd {f:7.1f} us ({mb / f * 1e-3:5.2f} TB/s)  dgrad {d:7.1f} us ({mb / d * 1e-3:5.2f} TB/s)  wgrad {g:7.1f} us', flush=True)
