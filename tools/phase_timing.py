"""Phase timing of the direct conv kernel (conv_v3.hip).  Needs a developer build of the library: E3_HIPCC_EXTRA=-DE3_TIMING python -m elektronn3_amd.build --force
(the release library compiles the stamps and the E3_CONV_ABLATE switch out)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('E3_CONV_ABLATE', '1024')
from elektronn3_amd import ops
for cin, cout in ((32, 32),):
    x = torch.randn(2, 64, 128, 128, cin, device='cuda'); w = torch.randn(cout, cin, 3, 3, 3, device='cuda') * 0.05; b = torch.zeros(cout, device='cuda')
    for _ in range(3):
        y, st = ops.conv3d(x, w, b, want_stats=True)
    torch.cuda.synchronize()
    t = st.view(-1).view(torch.int64)[: 8192 * 16].view(8192, 16).cpu().numpy()
    nch = cin // 8
    n = 1 + 2 * nch + 4
    d = np.diff(t[:, :n], axis=1).astype(np.float64)
    med = np.median(d, axis=0)
    print(f'{cin}->{cout}: stamps={n} median cycles per phase: prologue+first loads {med[0]:.0f};',
          'per chunk (wait+LDS write, taps):', [(int(med[1 + 2 * c]), int(med[2 + 2 * c])) for c in range(nch)], f'epi: barrier+setup {med[-4]:.0f} bias load {med[-3]:.0f} tile write+stats {med[-2]:.0f} stores {med[-1]:.0f}; total {np.median(t[:, n-1]-t[:, 0]):.0f}')
    dur = (t[:, n - 1].max() - t[:, 0].min())
    print('  kernel span (memtime ticks)', dur, ' sum of block times / (span*768 slots)=', (t[:, n-1]-t[:, 0]).sum() / (dur * 768.0))
    rt = (t[:, 15] - t[:, 14]).astype(np.float64)
    ck = (t[:, n - 1] - t[:, 0]) / np.maximum(rt, 1) * 100.0
    print(f'  per-block realtime ticks (100 MHz) median {np.median(rt):.0f}; memtime ticks per us -> clock MHz: median {np.median(ck):.0f} p10 {np.percentile(ck,10):.0f} p90 {np.percentile(ck,90):.0f}')
    print('  kernel wall (realtime, max-min)', (t[:, 15].max() - t[:, 14].min()) / 100.0, 'us')
