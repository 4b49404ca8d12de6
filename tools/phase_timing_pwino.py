"""Phase timing of conv3_wino_pkernel: s_memtime stamps of every workgroup's second brick (developer build -DE3_WINO_TIMING,
tools/build_timing_lib.sh -> tools/_bin/libe3unet_timing.so, selected with E3_LIB_PATH)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd import ops
names = ['chunk 0 (+ chunks before #1)', 'c1 VALU phase (reads, transforms)', 'c1 MFMA phase', 'c1 barrier', 'remaining chunks', 'out transform (ph,pw) + exchange writes',
         'next-brick requests + barrier', 'pd sum + stores', 'statistics']
for cin, cout, shp in ((32, 32, (2, 64, 128, 128)), (64, 32, (2, 64, 128, 128)), (128, 64, (2, 32, 64, 64))):
    x = torch.randn(*shp, cin, device='cuda'); w = torch.randn(cout, cin, 3, 3, 3, device='cuda') * 0.05; b = torch.zeros(cout, device='cuda')
    for _ in range(3):
        y, st = ops.conv3d(x, w, b, want_stats=True)
    torch.cuda.synchronize()
    raw = st.view(-1).view(torch.int64).cpu().numpy()
    t = raw[: 256 * 32].reshape(256, 32)
    t = t[(t[:, 0] > 0) & (t[:, 9] > t[:, 0])]
    d = np.diff(t[:, :10], axis=1).astype(np.float64)
    med = np.median(d, axis=0)
    print(f'{cin}->{cout} {shp}: workgroups {len(t)}, brick total {np.median(t[:,9]-t[:,0]):.0f} ticks')
    for nme, v in zip(names, med):
        print(f'    {nme:44s} {v:8.0f}')
    tt = raw[: 256 * 32].reshape(256, 32); tt = tt[(tt[:, 0] > 0) & (tt[:, 9] > tt[:, 0])]
    for lab, o in (('second-to-last chunk', 24), ('last chunk', 16)):
        # stamps 1 (start), 2 (after VALU phase), 4 (barrier in front of the last 16 MFMAs), 3 (end)
        print(f'    {lab}: VALU phase {np.median(tt[:, o + 2] - tt[:, o + 1]):.0f}, to barrier {np.median(tt[:, o + 4] - tt[:, o + 2]):.0f}, barrier to end {np.median(tt[:, o + 3] - tt[:, o + 4]):.0f}, total {np.median(tt[:, o + 3] - tt[:, o + 1]):.0f}')
