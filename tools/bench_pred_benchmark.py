"""The reference's own inference benchmark protocol (benchmark/pred_benchmark.py:48-104): UNet(dim, out_channels=2, n_blocks=4, start_filts=32,
'relu', 'batch').to(device, dtype) -- float32 and float16 -- on inputs (8,1,640,640) and (8,1,80,80,80); 1 warm-up, then the mean of n = 10 runs of
`model(x.to(device)).cpu(); synchronize()` (model left in train mode, grad enabled, host copies included); MVox/s = prod(inp_shape) / mean time.
Usage: python tools/bench_pred_benchmark.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd.unet import UNet

device = torch.device('cuda')
n = 10
print('| dim | input | dtype | mean time s | MVox/s |\n|---|---|---|---|---|')
for inp_shape in [(8, 1, 640, 640), (8, 1, 80, 80, 80)]:
    for float16 in (False, True):
        dim = len(inp_shape) - 2
        dtype = torch.float16 if float16 else torch.float32
        model = UNet(dim=dim, out_channels=2, n_blocks=4, start_filts=32, activation='relu', normalization='batch').to(device, dtype)
        r = model(torch.randn(*inp_shape, dtype=dtype).to(device)).cpu()
        torch.cuda.synchronize()
        del r
        xm = [torch.randn(*inp_shape, dtype=dtype) for _ in range(n)]
        t0 = time.time()
        for i in range(n):
            model(xm[i].to(device)).cpu()
            torch.cuda.synchronize()
        dt = (time.time() - t0) / n
        print(f'| {dim} | {inp_shape} | {str(dtype).replace("torch.", "")} | {dt:.4f} | {np.prod(inp_shape) / dt / 1e6:.1f} |', flush=True)
