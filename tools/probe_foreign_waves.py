"""What a few foreign waves (a collective's kernels on a side stream) do to the persistent kernels: N one-thread spin kernels on N streams while the training step runs."""
import torch, time, sys
sys.path.insert(0, '.')
from elektronn3_amd.unet import UNet
from elektronn3_amd.loss import CombinedCEDiceLoss
dev = torch.device('cuda')
torch.manual_seed(0)
model = UNet(1, 2, n_blocks=4, start_filts=32).to(dev).train()
crit = CombinedCEDiceLoss(weight=[0.2653, 0.7347]).to(dev)
x = torch.randn(2, 1, 64, 128, 128, device=dev); t = torch.randint(0, 2, (2, 64, 128, 128), device=dev)
def step():
    out, loss = model.forward_with_loss(x, t, crit)
    for p in model.parameters(): p.grad = None
    loss.backward()
for _ in range(3): step()
torch.cuda.synchronize()
def timed(n_sleepers, iters=5):
    streams = [torch.cuda.Stream() for _ in range(n_sleepers)]
    torch.cuda.synchronize()
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(int(2.1e9 * 0.12))          # ~120 ms of spinning
    t0 = time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.current_stream().synchronize()
    dt = (time.perf_counter() - t0) / iters * 1e3
    torch.cuda.synchronize()
    return dt
for n in (0, 0, 1, 2, 3, 0):        # (more side streams than hardware queues would queue the spinners in front of the step itself)
    print(f'{n:3d} spinning one-thread kernels beside the step: {timed(n):.2f} ms per step')
