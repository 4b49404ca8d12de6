"""Phase timing of conv3_wino16_kernel: s_memtime stamps of every workgroup's third brick (developer build -DE3_W16_TIMING:
E3_W16_EXTRA=-DE3_W16_TIMING bash tools/build_w16_variants.sh 0 -> tools/_bin/libe3unet_w16abl0.so, selected with E3_LIB_PATH)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd import ops
for cin, cout, shp in ((32, 32, (2, 64, 128, 128)), (64, 32, (2, 64, 128, 128)), (64, 64, (2, 32, 64, 64))):
    x = torch.randn(*shp, cin, device='cuda'); w = torch.randn(cout, cin, 3, 3, 3, device='cuda') * 0.05; b = torch.zeros(cout, device='cuda')
    for _ in range(3):
        y, st = ops.conv3d(x, w, b, want_stats=True)
    torch.cuda.synchronize()
    raw = st.view(-1).view(torch.int64).cpu().numpy()
    n = min(512, raw.size // 40)
    t = raw[: n * 40].reshape(n, 40).astype(np.float64)
    t = t[(t[:, 0] > 0) & (t[:, 36] > t[:, 0])]
    nch = cin // 8
    med = lambda a: float(np.median(a))
    print(f'{cin}->{cout} {shp}: workgroups {len(t)}, brick total {med(t[:, 36] - t[:, 0]):.0f} ticks')
    prev = t[:, 0]
    for c in range(min(nch, 8)):
        s = t[:, 1 + 4 * c: 5 + 4 * c]
        print(f'    chunk {c}: reads+transform {med(s[:, 0] - prev):6.0f}  dma+mfma {med(s[:, 1] - s[:, 0]):6.0f}  vmcnt wait {med(s[:, 2] - s[:, 1]):6.0f}  barrier {med(s[:, 3] - s[:, 2]):6.0f}  total {med(s[:, 3] - prev):6.0f}')
        prev = s[:, 3]
    print(f'    epilogue: transform+ex writes {med(t[:, 33] - prev):6.0f}  barrier {med(t[:, 34] - t[:, 33]):6.0f}  pd sum + stores {med(t[:, 35] - t[:, 34]):6.0f}  tail {med(t[:, 36] - t[:, 35]):6.0f}')
