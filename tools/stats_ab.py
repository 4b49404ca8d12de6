import sys, os, torch
sys.path.insert(0, '/root/repo')
from elektronn3_amd import ops
from tools.bench_conv import timeit
x = torch.randn(2, 64, 128, 128, 32, device='cuda'); w = torch.randn(32, 32, 3, 3, 3, device='cuda') * 0.05; b = torch.zeros(32, device='cuda')
for r in range(3):
    t0 = timeit(lambda: ops.conv3d(x, w, b, want_stats=False), 20)
    t1 = timeit(lambda: ops.conv3d(x, w, b, want_stats=True), 20)
    t2 = timeit(lambda: ops.conv3d(x, w, None, want_stats=False), 20)
    print(f'no-stats {t0*1e3:.1f} us   stats {t1*1e3:.1f} us   no-bias {t2*1e3:.1f} us')
