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
        if overlap and os.environ.get("PROBE_VERBOSE"):
            from elektronn3_amd.dataparallel import quiet_side_stream
            sync._flat = x; sync._comm_stream = quiet_side_stream(dev, verbose=True)
        if os.environ.get('PROBE_NO_CALIBRATION') is not None and overlap:       # A/B: the first stream, whatever queue it lands on
            sync._comm_stream = torch.cuda.Stream(device=dev, priority=-1)
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
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(iters):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    per = [evs[i].elapsed_time(evs[i + 1]) for i in range(iters)]
    ms1 = torch.cuda.memory_stats()
    dm = {k: ms1.get(k, 0) - ms0.get(k, 0) for k in ('num_device_alloc', 'num_device_free', 'num_alloc_retries', 'num_sync_all_streams')}
    print(f'{label:72s} {ms:7.2f} ms per step   (per step: {" ".join(f"{v:.1f}" for v in per)})  hipMalloc {dm["num_device_alloc"]} hipFree {dm["num_device_free"]} '
          f'retries {dm["num_alloc_retries"]} reserved {ms1.get("reserved_bytes.all.current", 0) / 2**30:.1f} GiB', flush=True)
    return ms


_keep = [torch.cuda.Stream(priority=-1) for _ in range(int(os.environ.get('PROBE_SKIP_STREAMS', '0')))]     # (which HARDWARE queue a side stream lands on depends on how many streams were USED before it)
for _s in _keep:
    with torch.cuda.stream(_s):
        torch.zeros(16, device=dev)
torch.cuda.synchronize()
if len(sys.argv) > 3 and sys.argv[2] == 'serial':
    run(f'serial, {int(sys.argv[3])} wg', overlap=False, blocks=int(sys.argv[3]), iters=3)
    sys.exit(0)
if len(sys.argv) > 3:        # one configuration (for a rocprofv3 --kernel-trace run): us reserve blocks
    run(f'overlap, reserve {int(sys.argv[2])} CUs, {int(sys.argv[3])} wg', overlap=True, reserve=int(sys.argv[2]), blocks=int(sys.argv[3]), iters=3)
    sys.exit(0)
base = run('no GradSync')
run('no GradSync (again)')
for blocks in (1, 8, 16, 32):
    print(f'--- stand-in collective: {blocks} resident workgroup(s) for {US:.0f} us')
    run(f'serial, {blocks} wg', overlap=False, blocks=blocks)
    for r in (0, 8, 16, 32) if os.environ.get('PROBE_REVERSE') is None else (32, 16, 8, 0):
        run(f'overlap, reserve {r:2d} CUs, {blocks} wg', overlap=True, reserve=r, blocks=blocks)
print(f'(serial costs the collective\'s {US:.0f} us on top of {base:.2f} ms; overlap hides it if the kernels after the event keep their speed)')
