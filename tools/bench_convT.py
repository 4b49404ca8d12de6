"""Transposed-conv (k = s = 2) micro-benchmark: forward (+BN statistics), dgrad, wgrad for the three up-convs of cfg 2."""
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

tot = [0, 0, 0]
for name, cin, cout, shp in (('L3->L2 256->128', 256, 128, (2, 8, 16, 16)), ('L2->L1 128->64', 128, 64, (2, 16, 32, 32)), ('L1->L0 64->32', 64, 32, (2, 32, 64, 64))):
    x = torch.randn(*shp, cin, device='cuda'); w = torch.randn(cin, cout, 2, 2, 2, device='cuda') * 0.05; b = torch.zeros(cout, device='cuda')
    dy = torch.randn(shp[0], 2 * shp[1], 2 * shp[2], 2 * shp[3], cout, device='cuda')
    vox = shp[0] * shp[1] * shp[2] * shp[3]
    fl = 2.0 * vox * cin * cout * 8
    mb_out = vox * 8 * cout * 4 / 1e6; mb_in = vox * cin * 4 / 1e6
    t0 = timeit(lambda: ops.convT(x, w, b, want_stats=True))
    t1 = timeit(lambda: ops.convT_dgrad(dy, w, shp[1:]))
    t2 = timeit(lambda: ops.convT_wgrad(x, dy, 2))
    tot = [tot[0] + t0, tot[1] + t1, tot[2] + t2]
    print(f'{name:18s} {fl/1e9:5.1f} GF  in {mb_in:5.0f} MB out {mb_out:5.0f} MB | fwd {t0:7.1f} us {fl/t0/1e6:6.1f} TF {(mb_in+mb_out)/t0*1e-3*1e3:6.0f} GB/s | dgrad {t1:7.1f} us {fl/t1/1e6:6.1f} TF | wgrad {t2:7.1f} us {fl/t2/1e6:6.1f} TF')
print(f'total fwd {tot[0]:.0f} us, dgrad {tot[1]:.0f} us, wgrad {tot[2]:.0f} us')
