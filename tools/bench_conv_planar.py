"""Planar 1x3x3 conv micro-benchmark on the layer shapes of BASELINE.json configs[3] (planar blocks 0 and 1, start_filts=64)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd import ops

def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

tot = [0.0, 0.0, 0.0]; totf = 0.0
for name, cin, cout, shp in (('L0_64_64', 64, 64, (2, 32, 256, 256)), ('L0_128_64', 128, 64, (2, 32, 256, 256)), ('L1_64_128', 64, 128, (2, 32, 128, 128)),
                             ('L1_128_128', 128, 128, (2, 32, 128, 128)), ('L1_256_128', 256, 128, (2, 32, 128, 128))):
    x = torch.randn(*shp, cin, device='cuda'); w = torch.randn(cout, cin, 1, 3, 3, device='cuda') * 0.05; b = torch.zeros(cout, device='cuda')
    dy = torch.randn(*shp, cout, device='cuda')
    fl = 2.0 * cin * cout * 9 * shp[0] * shp[1] * shp[2] * shp[3]
    t0 = timeit(lambda: ops.conv3d(x, w, b, planar=True, want_stats=True))
    t1 = timeit(lambda: ops.conv3d_dgrad(dy, w, planar=True))
    t2 = timeit(lambda: ops.conv3d_wgrad(x, dy, planar=True))
    tot = [tot[0] + t0, tot[1] + t1, tot[2] + t2]; totf += fl
    print(f'{name:12s} {fl/1e9:6.1f} GF | fwd {t0:7.1f} us {fl/t0/1e6:6.1f} TF | dgrad {t1:7.1f} us {fl/t1/1e6:6.1f} TF | wgrad {t2:7.1f} us {fl/t2/1e6:6.1f} TF')
for k, nme in enumerate(('fwd', 'dgrad', 'wgrad')):
    print(f'total {nme}: {tot[k]/1e3:.3f} ms, {totf/tot[k]/1e6:.1f} TF')
