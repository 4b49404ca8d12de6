"""Phase timing of conv3_wino_kernel (s_memtime stamps, E3_CONV_ABLATE=1024).  Needs a developer build of the library:
E3_HIPCC_EXTRA=-DE3_TIMING python -m elektronn3_amd.build --force (the release library compiles the stamps and the switch out)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('E3_CONV_ABLATE', '1024')
from elektronn3_amd import ops
names = ['prologue+first stage', 'chunk 0', 'c1 read+transform', 'c1 mfma', 'c1 lds write', 'c1 barrier', 'rest of loop', 'hw-transform+exchange', 'd-sum+store']
for cin, cout, shp in ((32, 32, (2, 64, 128, 128)), (64, 32, (2, 64, 128, 128)), (128, 64, (2, 32, 64, 64))):
    x = torch.randn(*shp, cin, device='cuda'); w = torch.randn(cout, cin, 3, 3, 3, device='cuda') * 0.05; b = torch.zeros(cout, device='cuda')
    for _ in range(3):
        y, st = ops.conv3d(x, w, b, want_stats=True)
    torch.cuda.synchronize()
    raw = st.view(-1).view(torch.int64).cpu().numpy()
    nb = len(raw) // 16
    t = raw[: nb * 16].reshape(nb, 16)
    t = t[(t[:, 0] > 0) & (t[:, 9] > t[:, 0])]
    d = np.diff(t[:, :10], axis=1).astype(np.float64)
    med = np.median(d, axis=0)
    rt = (t[:, 15] - t[:, 14]).astype(np.float64)
    print(f'{cin}->{cout} {shp}: blocks {len(t)} total {np.median(t[:,9]-t[:,0]):.0f} ticks, {np.median(rt)/100:.1f} us, clock {np.median((t[:,9]-t[:,0])/rt*100):.0f} MHz')
    for nme, v in zip(names, med):
        print(f'    {nme:24s} {v:8.0f}')
