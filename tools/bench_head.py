"""1x1x1 head forward alone (no BatchNorm prologue) on the cfg-2 activation, next to a plain device copy of the same tensor: the practical HBM ceiling
the HBM-bound rows of profiles/r02_per_layer_*.md should be read against.    python tools/bench_head.py"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from elektronn3_amd import ops
dev = torch.device('cuda')
a = torch.randn(2, 64, 128, 128, 32, device=dev)
w = torch.randn(2, 32, device=dev); b = torch.randn(2, device=dev)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print('head fwd (no prologue): %.1f us' % t(lambda: ops.conv1(a, w, b)))
print('head fwd + softmax: %.1f us' % t(lambda: ops.conv1(a, w, b, softmax=True)))
x = torch.empty_like(a)
print('copy 268 MB -> 268 MB: %.1f us' % t(lambda: x.copy_(a)))
print('sum over channels (torch): %.1f us' % t(lambda: a.sum(-1)))
