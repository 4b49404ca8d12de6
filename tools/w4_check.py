#!/usr/bin/env python3
"""F(2x2x4) Winograd kernel (csrc/conv_wino4.hip) through the per-op C ABI: results against an fp64 convolution, A/B timing against the
F(2x2x2) persistent kernel (child processes with E3_WINO4=0).  GPU only.

    python tools/w4_check.py check            # parity on a set of shapes (E3_WINO4_MIN=1 in the environment: every grid)
    python tools/w4_check.py bench [iters]    # cfg-2 / cfg-5 layer shapes: dgrad, eval forward (folded epilogue), forward with statistics
"""
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F


def merged_stats(st):
    n = st[:, :, 0].double(); m = st[:, :, 1].double(); M2 = st[:, :, 2].double()
    tot = n.sum(0); mean = (n * m).sum(0) / tot
    var = (M2.sum(0) + (n * (m - mean) ** 2).sum(0)) / tot
    return tot, mean, var


def check():
    from elektronn3_amd import ops
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    bad = 0
    shapes = [(1, 8, 16, 32, 32, 32), (2, 5, 11, 21, 32, 32), (1, 9, 13, 100, 64, 64), (2, 16, 32, 64, 64, 32), (1, 4, 4, 16, 8, 32), (1, 7, 9, 35, 16, 48),
              (1, 12, 20, 50, 40, 24), (2, 32, 64, 64, 32, 32), (1, 64, 128, 128, 32, 32)]
    for (N, D, H, W, ci, co) in shapes:
        x = torch.relu(torch.randn(N, D, H, W, ci, device=dev))
        w = torch.randn(co, ci, 3, 3, 3, device=dev) * (2.0 / (27 * ci)) ** 0.5
        b = torch.randn(co, device=dev)
        sc = 1 + 0.2 * torch.randn(co, device=dev); sh = 0.3 * torch.randn(co, device=dev)
        ref = F.conv3d(x.double().permute(0, 4, 1, 2, 3), w.double(), None, padding=1).permute(0, 2, 3, 4, 1)
        s = float(ref.abs().mean())
        # eval forward: relu(acc * scale + shift)
        y = ops.conv3d(x, w, None, epi=(sc, sh))
        r = torch.relu(ref * sc.double() + sh.double())
        e_aff = float((y.double() - r).abs().max()) / s
        # plain forward with bias and statistics
        y2, st = ops.conv3d(x, w, b, want_stats=True)
        r2 = ref + b.double()
        e_st = float((y2.double() - r2).abs().max()) / s
        tot, mean, var = merged_stats(st)
        yr = y2.reshape(-1, co).double()
        e_mean = float((mean - yr.mean(0)).abs().max()); e_var = float((var - yr.var(0, unbiased=False)).abs().max() / yr.var(0, unbiased=False).max())
        n_ok = bool((tot == yr.shape[0]).all())
        # data gradient
        dy = torch.randn(N, D, H, W, co, device=dev)
        dx = ops.conv3d_dgrad(dy, w)
        rd = F.conv_transpose3d(dy.double().permute(0, 4, 1, 2, 3), w.double(), None, padding=1).permute(0, 2, 3, 4, 1)
        e_dg = float((dx.double() - rd).abs().max()) / float(rd.abs().mean())
        ok = e_aff < 2e-5 and e_st < 2e-5 and e_dg < 2e-5 and e_mean < 1e-5 and e_var < 1e-5 and n_ok
        bad += not ok
        print(f'{(N, D, H, W, ci, co)}: max err / mean |y|: eval fwd {e_aff:.2e}, fwd+stats {e_st:.2e} (records {st.shape[0]}, n ok {n_ok}, mean {e_mean:.1e}, var {e_var:.1e}), dgrad {e_dg:.2e}  {"ok" if ok else "BAD"}', flush=True)
    print('bad cases:', bad)
    return bad


LAYERS = {  # name: (Cin, Cout, (D,H,W), N)
    'L0_32_32': (32, 32, (64, 128, 128), 2), 'L0_64_32': (64, 32, (64, 128, 128), 2),
    'L1_32_64': (32, 64, (32, 64, 64), 2), 'L1_64_64': (64, 64, (32, 64, 64), 2), 'L1_128_64': (128, 64, (32, 64, 64), 2),
    'T0_32_32': (32, 32, (128, 224, 224), 1), 'T0_64_32': (64, 32, (128, 224, 224), 1), 'T1_64_64': (64, 64, (64, 112, 112), 1), 'T1_128_64': (128, 64, (64, 112, 112), 1),
}


def bench(iters, layers=None):
    from elektronn3_amd import ops

    def timeit(fn):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / iters * 1e3
    out = {}
    for name, (cin, cout, (D, H, W), N) in LAYERS.items():
        if layers and name not in layers:
            continue
        x = torch.relu(torch.randn(N, D, H, W, cin, device='cuda'))
        dy = torch.randn(N, D, H, W, cout, device='cuda') * (torch.rand(N, D, H, W, cout, device='cuda') < 0.5)      # ReLU-sparse gradient
        w = torch.randn(cout, cin, 3, 3, 3, device='cuda') * 0.05
        b = torch.zeros(cout, device='cuda'); sc = torch.ones(cout, device='cuda')
        yb = torch.empty(N, D, H, W, cout, device='cuda')
        t_dg = timeit(lambda: ops.conv3d_dgrad(dy, w))
        t_ev = timeit(lambda: ops.conv3d(x, w, None, epi=(sc, b), out=yb))
        t_st = timeit(lambda: ops.conv3d(x, w, b, want_stats=True, out=yb)) if name[0] == 'L' else float('nan')
        out[name] = (t_dg, t_ev, t_st)
        print(f'{name:10s} dgrad {t_dg:8.1f} us | eval fwd {t_ev:8.1f} us | fwd+stats {t_st:8.1f} us', flush=True)
    return out


if __name__ == '__main__':
    cmd = sys.argv[1] if len(sys.argv) > 1 else 'check'
    if cmd == 'check':
        sys.exit(1 if check() else 0)
    if cmd == 'bench1':       # the two level-0 shapes only (ablation builds)
        bench(10, ('L0_32_32', 'L0_64_32'))
    if cmd == 'bench':
        iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
        print(f'--- E3_WINO4={os.environ.get("E3_WINO4", "(default)")}')
        bench(iters)
