"""CU-masked streams (hipExtStreamCreateWithCUMask): what a kernel costs on a subset of the CUs.  The mask bits go round-robin over the 8 XCDs
(bit b -> XCD b % 8, CU b / 8 of it; an XCD without any bit keeps ALL its CUs), so `first 32 bits` = 4 CUs per XCD.  DESIGN.md section 5, "Not done"."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd import ops
hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(bits, nwords=8):
    words = (ctypes.c_uint32 * nwords)(*[(bits >> (32 * i)) & 0xffffffff for i in range(nwords)])
    s = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), nwords, words) == 0
    return torch.cuda.ExternalStream(s.value)


def tm(fn, stream, iters=10):
    with torch.cuda.stream(stream):
        fn(); stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters): fn()
        e1.record(stream); stream.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


a = torch.randn(4096, 4096, device='cuda'); b = torch.randn(4096, 4096, device='cuda'); c = torch.empty_like(a)
x = torch.randn(2, 64, 128, 128, 32, device='cuda'); dy = torch.randn(2, 64, 128, 128, 32, device='cuda')
big = torch.randn(64 * 1024 * 1024, device='cuda'); out = torch.empty_like(big)
cases = (('whole chip', None), ('bits 0..127 (16 CUs per XCD)', (1 << 128) - 1), ('bits 0..31 (4 CUs per XCD)', (1 << 32) - 1), ('bits 0..7 (1 CU per XCD)', 0xff))
for label, bits in cases:
    s = torch.cuda.Stream() if bits is None else masked_stream(bits)
    print(f'{label}: fp32 GEMM 4096^3 {tm(lambda: torch.mm(a, b, out=c), s):.0f} us, Winograd wgrad 32->32 {tm(lambda: ops.conv3d_wgrad(x, dy), s):.0f} us, '
          f'268 MB copy {tm(lambda: out.copy_(big), s):.0f} us')
