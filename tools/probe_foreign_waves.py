"""What a collective's resident workgroups do to the training step, and what the CU reserve buys.

One GPU: GradSync runs with its whole event / side-stream machinery (E3_FORCE_GRADSYNC=1) and a stand-in for RCCL's kernel in place of the
all-reduce -- `blocks` workgroups of 256 threads that stay resident on the side stream for `us` microseconds (tests/native/spin_kernel.hip).
  serial            the stand-in starts when the backward has finished (GradSync's default)
  overlap, R = 0    it starts at the bucket event; the kernels after the event use the one-brick-per-workgroup conv kernel (round 3's behaviour)
  overlap, R > 0    the kernels after the event leave R compute units alone (E3_BWD_CU_RESERVE)
Usage: python tools/probe_foreign_waves.py [us=400]
"""
import os
import sys
import time

os.environ['E3_FORCE_GRADSYNC'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from elektronn3_amd.dataparallel import GradSync  # noqa: E402
from elektronn3_amd.loss import CombinedCEDiceLoss  # noqa: E402
from elektronn3_amd.unet import UNet  # noqa: E402
from tests.helpers import spin  # noqa: E402

US = float(sys.argv[1]) if len(sys.argv) > 1 else 400.0
dev = torch.device('cuda')
torch.manual_seed(0)
crit = CombinedCEDiceLoss(weight=[0.2653, 0.7347]).to(dev)
x = torch.randn(2, 1, 64, 128, 128, device=dev)
t = torch.randint(0, 2, (2, 64, 128, 128), device=dev)
sink = torch.zeros(4, dtype=torch.int32, device=dev)


def run(label, overlap=None, reserve=0, blocks=0, iters=8):
    torch.manual_seed(0)
    model = UNet(1, 2, n_blocks=4, start_filts=32).to(dev).train()
    if overlap is not None:
        sync = GradSync(model, overlap=overlap, cu_reserve=reserve)
        calls = [0]

        def fake(tensor):       # runs on GradSync's side stream in place of dist.all_reduce
            calls[0] += 1
            if blocks and tensor.numel() > 1 << 20:      # (the big bucket only: the 0.8 MB remainder is a latency-bound blip)
                spin(torch.cuda.current_stream(), blocks, US, sink=sink)
        sync.collective = fake

    def step():
        out, loss = model.forward_with_loss(x, t, crit)
        for p in model.parameters():
            p.grad = None
        loss.backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    print(f'{label:72s} {ms:7.2f} ms per step', flush=True)
    return ms


base = run('no GradSync')
run('no GradSync (again)')
for blocks in (1, 8, 16, 32):
    print(f'--- stand-in collective: {blocks} resident workgroup(s) for {US:.0f} us')
    run(f'serial, {blocks} wg', overlap=False, blocks=blocks)
    for r in (0, 8, 16, 32):
        run(f'overlap, reserve {r:2d} CUs, {blocks} wg', overlap=True, reserve=r, blocks=blocks)
print(f'(serial costs the collective\'s {US:.0f} us on top of {base:.2f} ms; overlap hides it if the kernels after the event keep their speed)')
