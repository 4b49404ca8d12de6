#!/usr/bin/env python3
"""Pooled BatchNorm backward (REDUCE + APPLY passes through the per-op ABI) at the level-0 shape of cfg 2 with the skip gradient g1 read
(a) from a contiguous [voxel][32] tensor and (b) as the second half of a [voxel][64] concat-gradient buffer (what the plan hands it):
does the half-row stride cost anything?  Also the un-pooled form for reference.    python tools/probe_bnbwd_stride.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd import ops
dev = torch.device('cuda:0'); torch.manual_seed(0)
N, D, H, W, C = 2, 64, 128, 128, 32
x = torch.randn(N, D, H, W, C, device=dev)
stats = torch.stack([torch.full((C,), float(x.numel() // C), device=dev), x.mean((0, 1, 2, 3)), x.var((0, 1, 2, 3), unbiased=False) * (x.numel() // C)], 1)[None].contiguous()
gamma = torch.rand(C, device=dev) + 0.5; beta = torch.randn(C, device=dev) * 0.1
mean, invstd, scale, shift = ops.bn_finalize(stats, gamma, beta)
a, pooled = ops.bn_relu_apply(x, scale, shift, pool_kd=2)
gpool = torch.randn_like(pooled)
g_c = torch.randn(N, D, H, W, C, device=dev)
cat = torch.randn(N, D, H, W, 2 * C, device=dev); cat[..., C:] = g_c
g_s = cat[..., C:]
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for i in range(2):
    tc = t(lambda: ops.bn_relu_bwd(x, mean, invstd, gamma, scale, shift, g1=g_c, gpool=gpool, a=a, pooled=pooled, kd=2))
    ts = t(lambda: ops.bn_relu_bwd(x, mean, invstd, gamma, scale, shift, g1=g_s, gpool=gpool, a=a, pooled=pooled, kd=2))
    pc = t(lambda: ops.bn_relu_bwd(x, mean, invstd, gamma, scale, shift, g1=g_c))
    ps = t(lambda: ops.bn_relu_bwd(x, mean, invstd, gamma, scale, shift, g1=g_s))
    print(f'round {i}: pooled unit (reduce + finalize + apply): g1 contiguous {tc:.1f} us, g1 = half of a 64-channel row {ts:.1f} us;  plain unit: {pc:.1f} / {ps:.1f} us')
r1 = ops.bn_relu_bwd(x, mean, invstd, gamma, scale, shift, g1=g_c, gpool=gpool, a=a, pooled=pooled, kd=2)
r2 = ops.bn_relu_bwd(x, mean, invstd, gamma, scale, shift, g1=g_s, gpool=gpool, a=a, pooled=pooled, kd=2)
print('identical results:', all(torch.equal(p, q) for p, q in zip(r1, r2)))
