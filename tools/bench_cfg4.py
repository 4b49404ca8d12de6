"""Training-step time of BASELINE.json configs[3]: anisotropic UNet (planar_blocks=(0,1), start_filts=64), batch 2 of 32x256x256."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd.unet import UNet
from elektronn3_amd.loss import CombinedCEDiceLoss
torch.manual_seed(0)
m = UNet(in_channels=1, out_channels=2, n_blocks=4, start_filts=64, planar_blocks=(0, 1), normalization='batch').cuda().train()
crit = CombinedCEDiceLoss(weight=[0.2653, 0.7347]).cuda()
x = torch.randn(2, 1, 32, 256, 256, device='cuda'); t = torch.randint(0, 2, (2, 32, 256, 256), device='cuda')
def step():
    loss = crit(m(x), t)
    for p in m.parameters(): p.grad = None
    loss.backward()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f'cfg 4 (planar_blocks=(0,1), sf=64, batch 2 x 32x256x256): {dt*1e3:.2f} ms/step = {x.numel()/dt/1e6:.1f} M voxels/s; max memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB')
