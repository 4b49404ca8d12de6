"""Phase timing inside the persistent bf16 conv kernel (debug build only):

    E3_HIPCC_EXTRA=-DE3_CONV_TIMING python -c "from elektronn3_amd.build import build; build(force=True)"
    python tools/conv_phases.py [Cin Cout]          # on the GPU box

Thread 0 of every workgroup stamps s_memrealtime (100 MHz) at: start, DMA issued, DMA landed, barrier passed, taps done, barrier passed,
stores done, statistics done.  Prints the mean duration of every phase."""
import ctypes
import sys

import numpy as np
import torch

from elektronn3_amd import _lib, ops

cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 32)
if len(sys.argv) > 3:            # another build of the library (experiments)
    _lib._LIB_PATH = sys.argv[3]
    print('library', sys.argv[3])
L = _lib.load(build_if_missing=False)
dev = torch.device('cuda:0')
buf = torch.zeros(8192 * 16, dtype=torch.int64, device=dev)
fn = L.e3_debug_conv_timing
fn.argtypes = [ctypes.c_void_p]
assert fn(buf.data_ptr()) == 0
x = torch.randn(2, 64, 128, 128, cin, device=dev).bfloat16()
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
b = torch.randn(cout, device=dev)
for mode in ('stats',):
    for _ in range(3):
        buf.zero_()
        ops.conv3d_bf16(x, w, b, want_stats=(mode == 'stats'))
        torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(8192, 16).astype(np.int64)
    used = t[:, 0] > 0
    d = np.diff(t[:, :8], axis=1) * 10.0 / 1e3      # us
    names = ['decode + weight prefetch + DMA issue', 'DMA wait', 'barrier', 'taps (last chunk)', 'barrier', 'stores', 'statistics']
    print(f'{cin}->{cout} {mode}: {int(used.sum())} workgroups; per brick (us): ' + ', '.join(f'{n} {d[:, i][used].mean():.2f}' for i, n in enumerate(names))
          + f'; total {(t[:, 7] - t[:, 0])[used].mean() * 10.0 / 1e3:.2f}; kernel span {(t[:, 7].max() - t[:, 0][used].min()) * 10.0 / 1e3:.1f} us')

    if (t[:, 9] > 0).any():          # first chunk of a brick with several chunks: slots 9..13 = the same stamps
        u0 = used & (t[:, 9] > 0)
        c0 = np.diff(t[:, 9:14], axis=1) * 10.0 / 1e3
        print(f'    prologue: start -> decoded {((t[:, 14] - t[:, 0])[u0]).mean() * 10.0 / 1e3:.2f}, -> staging plan {((t[:, 15] - t[:, 14])[u0]).mean() * 10.0 / 1e3:.2f}, -> first DMA issued {((t[:, 9] - t[:, 15])[u0]).mean() * 10.0 / 1e3:.2f}')
        print(f'    first chunk: start -> DMA issued {((t[:, 9] - t[:, 0])[u0]).mean() * 10.0 / 1e3:.2f}, DMA wait {c0[:, 0][u0].mean():.2f}, barrier {c0[:, 1][u0].mean():.2f}, '
              f'taps {c0[:, 2][u0].mean():.2f}, barrier {c0[:, 3][u0].mean():.2f}')

# ---- weight gradient (same debug build): python tools/conv_phases.py 32 32  prints this after the forward's phases
fnw = getattr(L, 'e3_debug_wgrad_timing', None)
if fnw is not None:
    fnw.argtypes = [ctypes.c_void_p]
    wbuf = torch.zeros(1024 * 8 * 4, dtype=torch.int64, device=dev)
    assert fnw(wbuf.data_ptr()) == 0
    dy = torch.randn(2, 64, 128, 128, cout, device=dev).bfloat16()
    for _ in range(3):
        wbuf.zero_()
        ops.conv3d_wgrad_bf16(x, dy)
        torch.cuda.synchronize()
    t = wbuf.cpu().numpy().reshape(1024, 8, 4).astype(np.int64)
    used = t[:, :, 0] > 0
    d = np.diff(t, axis=2) * 10.0 / 1e3
    print(f'wgrad {cin}->{cout}: {int(used.sum())} bricks timed; per brick (us): decode + DMA issue {d[:, :, 0][used].mean():.2f}, DMA wait + barrier {d[:, :, 1][used].mean():.2f}, '
          f'MFMA loop {d[:, :, 2][used].mean():.2f}; brick to brick {np.diff(t[:, :, 0], axis=1)[used[:, 1:] & used[:, :-1]].mean() * 10.0 / 1e3:.2f}')
