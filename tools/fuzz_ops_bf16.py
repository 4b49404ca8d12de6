"""Randomised per-op parity of the bf16 kernels: random shapes through the SAME checks as tests/test_bf16_gpu.py (every output element within bf16
rounding of the fp64 result; statistics records; eval epilogue; weight gradients rel-L2 < 2e-6) -- conv forward / dgrad / wgrad, the first conv,
transposed conv.    python tools/fuzz_ops_bf16.py [n_cases] [seed]"""
import os, sys, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_bf16_gpu as T

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
bad = 0
convT = [f for n, f in vars(T).items() if n.startswith('test_convT') and 'vs_fp64' in n]
for case in range(n_cases):
    kind = ri(0, 3)
    N, D, H, W = ri(1, 3), ri(1, 20), ri(1, 40), ri(1, 110)
    try:
        if kind <= 1:
            Cin, Cout = 32 * ri(1, 4), 32 * ri(1, 4)
            if N * D * H * W * max(Cin, Cout) > 3e7: W = max(1, W // 4)
            name = f'conv {(N, D, H, W, Cin, Cout)}'
            T.test_conv3d_bf16_forward_dgrad_wgrad_vs_fp64(N, D, H, W, Cin, Cout)
        elif kind == 2:
            Cin, Cout = ri(1, 3), 32 * ri(1, 2)
            name = f'first conv {(N, D, H, W, Cin, Cout)}'
            T.test_first_conv_bf16_forward_and_wgrad_vs_fp64(N, D, H, W, Cin, Cout)
        else:
            name = 'transposed conv (fixed cases of the test file)'
            import inspect
            for f in convT:
                sig = inspect.signature(f)
                if not sig.parameters:
                    f()
        print('ok ', name, flush=True)
    except AssertionError as e:
        bad += 1
        print('BAD', name, str(e).splitlines()[0][:200], flush=True)
    except Exception:
        bad += 1
        print('ERR', name); traceback.print_exc()
print(f'{bad} bad of {n_cases}')
sys.exit(1 if bad else 0)
