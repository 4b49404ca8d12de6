#!/usr/bin/env python3
"""Micro-benchmark of the MFMA conv kernels on the cfg-2 layer shapes (per-op C ABI).  GPU only.

    python tools/bench_conv.py [--layers L0_32_32,L1_64_64,...] [--iters 10] [--what fwd,dgrad,wgrad]
"""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from elektronn3_amd import ops

LAYERS = {  # name: (Cin, Cout, (D,H,W), N)
    'L0_32_32': (32, 32, (64, 128, 128), 2), 'L0_64_32': (64, 32, (64, 128, 128), 2),
    'L1_32_64': (32, 64, (32, 64, 64), 2), 'L1_64_64': (64, 64, (32, 64, 64), 2), 'L1_128_64': (128, 64, (32, 64, 64), 2),
    'L2_64_128': (64, 128, (16, 32, 32), 2), 'L2_128_128': (128, 128, (16, 32, 32), 2), 'L2_256_128': (256, 128, (16, 32, 32), 2),
    'L3_128_256': (128, 256, (8, 16, 16), 2), 'L3_256_256': (256, 256, (8, 16, 16), 2),
}


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', default=','.join(LAYERS))
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--what', default='fwd,dgrad,wgrad')
    ap.add_argument('--dtype', default='f32', choices=('f32', 'bf16'))
    ap.add_argument('--no-stats', action='store_true', help='forward without the BatchNorm statistics epilogue')
    a = ap.parse_args()
    what = a.what.split(',')
    tot = {w: [0.0, 0.0] for w in what}
    for name in a.layers.split(','):
        cin, cout, (D, H, W), N = LAYERS[name]
        x = torch.randn(N, D, H, W, cin, device='cuda')
        dy = torch.randn(N, D, H, W, cout, device='cuda')
        w = torch.randn(cout, cin, 3, 3, 3, device='cuda') * 0.05
        b = torch.zeros(cout, device='cuda')
        if a.dtype == 'bf16':
            x, dy = x.bfloat16(), dy.bfloat16()
            conv, dgrad, wgrad = ops.conv3d_bf16, ops.conv3d_dgrad_bf16, ops.conv3d_wgrad_bf16
        else:
            conv, dgrad, wgrad = ops.conv3d, ops.conv3d_dgrad, ops.conv3d_wgrad
        fl = 2.0 * cin * cout * 27 * N * D * H * W
        line = f'{name:12s} {fl / 1e9:7.1f} GF '
        for wh in what:
            if wh == 'fwd':
                ms = timeit(lambda: conv(x, w, b, want_stats=not a.no_stats), a.iters)
            elif wh == 'dgrad':
                ms = timeit(lambda: dgrad(dy, w), a.iters)
            else:
                ms = timeit(lambda: wgrad(x, dy), a.iters)
            tot[wh][0] += fl; tot[wh][1] += ms
            line += f'| {wh} {ms * 1e3:8.1f} us {fl / ms / 1e9:6.1f} TF '
        print(line, flush=True)
    for wh in what:
        print(f'total {wh}: {tot[wh][1]:.3f} ms, {tot[wh][0] / tot[wh][1] / 1e9:.1f} TF')


if __name__ == '__main__':
    main()
