#!/usr/bin/env python3
"""Step time (forward + loss + backward, fp32) of the headline configuration -- UNet(1, 2, n_blocks=4, start_filts=32), 2 x 64 x 128 x 128 -- and of its
attention / ResUNet variants, to put a number on what the general-purpose kernels of csrc/attention.hip and the residual plumbing cost.

    python tools/bench_variants.py [--steps 10]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--only', type=int, default=-1, help='index of the single variant to run (for rocprofv3)')
    args = ap.parse_args()
    from elektronn3_amd import resunet, unet
    from elektronn3_amd.loss import CombinedCEDiceLoss
    variants = [('unet', unet.UNet, {}), ('unet attention=True', unet.UNet, dict(attention=True)),
                ('resunet res_blocks=(0,0)', resunet.UNet, {}), ('resunet res_blocks=(1,1)', resunet.UNet, dict(enc_res_blocks=1, dec_res_blocks=1)),
                ('resunet res_blocks=(2,2)', resunet.UNet, dict(enc_res_blocks=2, dec_res_blocks=2)),
                ('resunet res_blocks=(1,1) attention=True', resunet.UNet, dict(enc_res_blocks=1, dec_res_blocks=1, attention=True))]
    x = torch.randn(2, 1, 64, 128, 128, device='cuda')
    t = torch.randint(0, 2, (2, 64, 128, 128), device='cuda')
    crit = CombinedCEDiceLoss(weight=torch.tensor([0.2653, 0.7347])).cuda()
    print('| model | ms / step | vs unet |\n|---|---|---|')
    base = None
    for vi, (name, cls, kw) in enumerate(variants):
        if args.only >= 0 and vi != args.only:
            continue
        m = cls(1, 2, n_blocks=4, start_filts=32, **kw).cuda().train()

        def step():
            m.zero_grad(set_to_none=True)
            out = m(x)
            loss = crit(out, t)
            loss.backward()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        base = base or ms
        print(f'| {name} | {ms:.2f} | {ms / base:.2f}x |', flush=True)


if __name__ == '__main__':
    main()
