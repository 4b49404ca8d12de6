"""Elementwise (HBM-bound) kernels at the full-resolution level of cfg 2: effective GB/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd import ops

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for C, shp in ((32, (2, 64, 128, 128)), (64, (2, 32, 64, 64))):
    x = torch.randn(*shp, C, device='cuda'); g1 = torch.randn(*shp, C, device='cuda')
    gamma = torch.rand(C, device='cuda') + 0.5; beta = torch.randn(C, device='cuda') * 0.1
    mean = x.mean(dim=(0, 1, 2, 3)); var = x.var(dim=(0, 1, 2, 3), unbiased=False); invstd = 1 / torch.sqrt(var + 1e-5)
    scale = gamma * invstd; shift = beta - mean * scale
    mb = x.numel() * 4 / 1e6
    out = torch.empty_like(x)
    t = timeit(lambda: ops.bn_relu_apply(x, scale, shift, out=out))
    print(f'C={C} {shp}: tensor {mb:.0f} MB | bn_relu_apply {t:6.1f} us {2*mb/t*1e-3*1e3:6.0f} GB/s', end='')
    t = timeit(lambda: ops.bn_relu_apply(x, scale, shift, pool_kd=2))
    print(f' | +pool {t:6.1f} us {2.125*mb/t*1e3*1e-3:6.0f} GB/s', end='')
    t = timeit(lambda: ops.bn_relu_bwd(x, mean, invstd, gamma, scale, shift, g1=g1))
    print(f' | bn_relu_bwd (2 passes + finalisers) {t:6.1f} us {5*mb/t*1e3*1e-3:6.0f} GB/s')
