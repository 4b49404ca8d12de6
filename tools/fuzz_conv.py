"""Randomised shape fuzzing of the conv kernels (3D and planar, fwd / dgrad / wgrad, + transposed conv) against PyTorch-ROCm.
Usage: python tools/fuzz_conv.py [n_cases] [seed]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd import ops
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
bad = 0
for case in range(n_cases):
    planar = ri(0, 2) == 0
    cin = 8 * ri(1, 12); cout = 8 * ri(1, 12)
    if ri(0, 3) == 0: cin, cout = 32 * ri(1, 4), 32 * ri(1, 4)
    N = ri(1, 3); D = ri(1, 20); H = ri(1, 70); W = ri(1, 90)
    if D * H * W * N * max(cin, cout) > 6e7: D = max(1, D // 4)
    x = torch.randn(N, D, H, W, cin, device='cuda'); kd = 1 if planar else 3
    w = torch.randn(cout, cin, kd, 3, 3, device='cuda') * 0.1; b = torch.randn(cout, device='cuda')
    pad = (0, 1, 1) if planar else (1, 1, 1)
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    ref = torch.nn.functional.conv3d(xr.permute(0, 4, 1, 2, 3), wr, b, padding=pad).permute(0, 2, 3, 4, 1)
    dy = torch.randn_like(ref).contiguous(); ref.backward(dy)
    y, st = ops.conv3d(x, w, b, planar=planar, want_stats=True)
    dx = ops.conv3d_dgrad(dy, w, planar=planar); dw = ops.conv3d_wgrad(x, dy, planar=planar)
    mean, invstd, _, _ = ops.bn_finalize(st, torch.ones(cout, device='cuda'), torch.zeros(cout, device='cuda'))
    e = [float((y - ref).abs().max()), float((dx - xr.grad).abs().max()), float((dw - wr.grad).norm() / wr.grad.norm().clamp_min(1e-20)),
         float((mean - ref.mean(dim=(0, 1, 2, 3))).abs().max())]
    ok = e[0] < 2e-4 and e[1] < 4e-4 and e[2] < 5e-5 and e[3] < 1e-4
    bad += not ok
    print(f'{"ok  " if ok else "BAD "} planar={int(planar)} {cin:3d}->{cout:3d} N={N} {D}x{H}x{W}: y {e[0]:.1e} dx {e[1]:.1e} dw {e[2]:.1e} mean {e[3]:.1e}', flush=True)
for case in range(max(4, n_cases // 5)):
    sd = ri(1, 2); cin = 32 * ri(1, 6) if ri(0, 1) else 8 * ri(1, 12); cout = 32 * ri(1, 3) if ri(0, 1) else 8 * ri(1, 8)
    N = ri(1, 2); D = ri(1, 8); H = ri(1, 30); W = ri(1, 40)
    x = torch.randn(N, D, H, W, cin, device='cuda'); w = torch.randn(cin, cout, sd, 2, 2, device='cuda') * 0.1; b = torch.randn(cout, device='cuda')
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    ref = torch.nn.functional.conv_transpose3d(xr.permute(0, 4, 1, 2, 3), wr, b, stride=(sd, 2, 2)).permute(0, 2, 3, 4, 1)
    dy = torch.randn_like(ref).contiguous(); ref.backward(dy)
    y = ops.convT(x, w, b); dx = ops.convT_dgrad(dy, w, (D, H, W)); dw = ops.convT_wgrad(x, dy, sd)
    e = [float((y - ref).abs().max()), float((dx - xr.grad).abs().max()), float((dw - wr.grad).norm() / wr.grad.norm().clamp_min(1e-20))]
    ok = e[0] < 2e-4 and e[1] < 4e-4 and e[2] < 5e-5
    bad += not ok
    print(f'{"ok  " if ok else "BAD "} convT sd={sd} {cin:3d}->{cout:3d} N={N} {D}x{H}x{W}: y {e[0]:.1e} dx {e[1]:.1e} dw {e[2]:.1e}', flush=True)
print('BAD CASES:', bad)
sys.exit(1 if bad else 0)
