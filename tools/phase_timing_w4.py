"""Phase timing of conv3_wino4_kernel: s_memtime stamps of every workgroup's third brick (developer build -DE3_W4_TIMING:
E3_W4_EXTRA=-DE3_W4_TIMING bash tools/build_w4_variants.sh 0 -> tools/_bin/libe3unet_w4abl0.so, selected with E3_LIB_PATH; E3_WINO4=2)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd import ops
for cin, cout, shp in ((32, 32, (2, 64, 128, 128)), (64, 32, (2, 64, 128, 128))):
    x = torch.randn(*shp, cin, device='cuda'); w = torch.randn(cout, cin, 3, 3, 3, device='cuda') * 0.05; b = torch.zeros(cout, device='cuda')
    for _ in range(3):
        y, st = ops.conv3d(x, w, b, want_stats=True)
    torch.cuda.synchronize()
    raw = st.view(-1).view(torch.int64).cpu().numpy()
    n = min(256, raw.size // 48)
    t = raw[: n * 48].reshape(n, 48).astype(np.float64)
    t = t[(t[:, 0] > 0) & (t[:, 44] > t[:, 0])]
    nch = cin // 8
    med = lambda a: float(np.median(a))
    print(f'{cin}->{cout} {shp}: workgroups {len(t)}, brick total {med(t[:, 44] - t[:, 0]):.0f} ticks (wave 0 of every workgroup)')
    prev = t[:, 0]
    for c in range(min(nch, 8)):
        s = t[:, 1 + 5 * c: 6 + 5 * c]
        print(f'    chunk {c}: transform {med(s[:, 0] - prev):6.0f}  vmcnt wait {med(s[:, 1] - s[:, 0]):6.0f}  barrier {med(s[:, 2] - s[:, 1]):6.0f}  6 pairs + DMA {med(s[:, 3] - s[:, 2]):6.0f}  6 pairs {med(s[:, 4] - s[:, 3]):6.0f}  total {med(s[:, 4] - prev):6.0f}')
        prev = s[:, 4]
    print(f'    epilogue: transform+ex writes {med(t[:, 41] - prev):6.0f}  barrier {med(t[:, 42] - t[:, 41]):6.0f}  pd sum + stores {med(t[:, 43] - t[:, 42]):6.0f}  tail {med(t[:, 44] - t[:, 43]):6.0f}')
