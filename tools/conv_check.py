"""Quick bf16 conv forward check against torch fp32 on a few shapes (GPU box): python tools/conv_check.py"""
import itertools
import torch
import torch.nn.functional as F
from elektronn3_amd import ops

dev = torch.device('cuda:0')
torch.manual_seed(0)
for (N, D, H, W, ci, co) in [(1, 8, 16, 16, 32, 32), (1, 8, 16, 16, 64, 32), (2, 5, 11, 21, 32, 32), (2, 5, 11, 21, 64, 32), (1, 9, 13, 100, 64, 64),
                             (1, 16, 32, 64, 32, 64), (2, 32, 64, 64, 32, 32)]:
    x = torch.randn(N, D, H, W, ci, device=dev).bfloat16()
    w = (torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05).bfloat16().float()
    b = torch.randn(co, device=dev)
    y, st = ops.conv3d_bf16(x, w, b, want_stats=True)
    ref = F.conv3d(x.float().permute(0, 4, 1, 2, 3), w, b, padding=1).permute(0, 2, 3, 4, 1)
    err = (y.float() - ref).abs()
    # statistics: merge the records
    n = st[:, :, 0].double(); m = st[:, :, 1].double(); M2 = st[:, :, 2].double()
    tot = n.sum(0); mean = (n * m).sum(0) / tot
    var = (M2.sum(0) + (n * (m - mean) ** 2).sum(0)) / tot
    yr = y.float().reshape(-1, co).double()
    print(f'{(N, D, H, W, ci, co)}: max err {float(err.max()):.3e} (ref max {float(ref.abs().max()):.2f}), bad rows {int((err.amax(-1) > 0.1).sum())}/{err[..., 0].numel()}; '
          f'stats parts {st.shape[0]} n {float(tot[0])} mean err {float((mean - yr.mean(0)).abs().max()):.2e} var err {float((var - yr.var(0, unbiased=False)).abs().max()):.2e}')
    if float(err.max()) > 0.1:
        bad = (err.amax(-1) > 0.1)[0]
        idx = bad.nonzero()
        print('   first bad voxels (d,h,w):', idx[:6].tolist(), ' last:', idx[-3:].tolist())
