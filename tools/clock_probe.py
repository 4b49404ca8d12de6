"""Sustained shader clock inside the conv kernels: s_memtime (shader clock) over s_memrealtime (100 MHz) per workgroup.
Needs the debug stamps of conv_v3.hip: a developer build of the library (E3_HIPCC_EXTRA=-DE3_TIMING python -m elektronn3_amd.build --force;
the release library compiles the stamps and the E3_CONV_ABLATE switch out) with E3_CONV_ABLATE=1024.  Usage: python tools/clock_probe.py"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('E3_CONV_ABLATE', '1024')
from elektronn3_amd import ops
for cin, cout, shp in ((32, 32, (2, 64, 128, 128)), (64, 32, (2, 64, 128, 128)), (64, 64, (2, 32, 64, 64)), (128, 64, (2, 32, 64, 64)), (128, 128, (2, 16, 32, 32))):
    x = torch.randn(*shp, cin, device='cuda'); w = torch.randn(cout, cin, 3, 3, 3, device='cuda') * 0.05; b = torch.zeros(cout, device='cuda')
    for _ in range(5):
        y, st = ops.conv3d(x, w, b, want_stats=True)
    torch.cuda.synchronize()
    nblk = st.numel() // (cout * 3) * max(1, (cout + 31) // 32) // max(1, (cout + 31) // 32)
    raw = st.view(-1).view(torch.int64).cpu().numpy()
    nb = min(len(raw) // 16, 100000)
    t = raw[: nb * 16].reshape(nb, 16)
    t = t[(t[:, 12] > 0) & (t[:, 13] > t[:, 12]) & (t[:, 15] > t[:, 14])]
    mt = (t[:, 13] - t[:, 12]).astype(np.float64); rt = (t[:, 15] - t[:, 14]).astype(np.float64)
    ck = mt / rt * 100.0
    wall = (t[:, 15].max() - t[:, 14].min()) / 100.0
    fl = 2.0 * cin * cout * 27 * np.prod(shp)
    print(f'{cin}->{cout} {shp}: blocks {len(t)} median block {np.median(rt)/100:.1f} us, clock MHz median {np.median(ck):.0f} (p10 {np.percentile(ck,10):.0f}, p90 {np.percentile(ck,90):.0f}); kernel wall {wall:.0f} us = {fl/wall/1e6:.1f} TF (with stamps)')
