"""Which engine moves a device -> pinned-host copy?  (profiles/r06_predictor_copies.md)
    rocprofv3 --kernel-trace --memory-copy-trace -- python tools/d2h_probe.py <case>
cases: fresh   D2H copies on a stream that never ran a kernel
       after   D2H copies on a stream, each behind a kernel of that stream
       split   kernel on stream A, event, D2H on stream B (which never runs kernels)
       h2d     H2D copies behind kernels of the same stream (for comparison)"""
import sys
import torch

case = sys.argv[1]
dev = torch.device('cuda:0')
n = 64 << 20
d = torch.zeros(n, dtype=torch.float32, device=dev)
h = torch.empty(n, dtype=torch.float32).pin_memory()
torch.cuda.synchronize()
a, b = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for _ in range(5):
    if case == 'fresh':
        with torch.cuda.stream(b):
            h.copy_(d, non_blocking=True)
    elif case == 'after':
        with torch.cuda.stream(a):
            d.add_(1.0)
            h.copy_(d, non_blocking=True)
    elif case == 'split':
        with torch.cuda.stream(a):
            d.add_(1.0)
            ev = torch.cuda.Event(); ev.record(a)
        with torch.cuda.stream(b):
            b.wait_event(ev)
            h.copy_(d, non_blocking=True)
    elif case == 'h2d':
        with torch.cuda.stream(a):
            d.add_(1.0)
            d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
print('done', case)
