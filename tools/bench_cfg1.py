"""Training-step time of BASELINE.json configs[0] (the reference's CPU-runnable case): UNet(1,2,n_blocks=2,start_filts=8), batch 1 of 64^3.
A launch-bound case: ~50 kernels of a few microseconds each per step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd.unet import UNet
from elektronn3_amd.loss import CombinedCEDiceLoss
torch.manual_seed(0)
m = UNet(in_channels=1, out_channels=2, n_blocks=2, start_filts=8).cuda().train()
crit = CombinedCEDiceLoss(weight=[0.2653, 0.7347]).cuda()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.randn(N, 1, 64, 64, 64, device='cuda'); t = torch.randint(0, 2, (N, 64, 64, 64), device='cuda')
def step():
    loss = crit(m(x), t)
    for p in m.parameters(): p.grad = None
    loss.backward()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 50
for _ in range(K): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print(f'cfg 1 (n_blocks=2, sf=8, batch {N} x 64^3): {dt*1e3:.3f} ms/step = {x.numel()/dt/1e6:.1f} M voxels/s')
