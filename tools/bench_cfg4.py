"""Training-step time of BASELINE.json configs[3]: anisotropic UNet (planar_blocks=(0,1), start_filts=64), batch 2 of 32x256x256.
Usage: python tools/bench_cfg4.py [f32|bf16|f16|example_f16]   (example_f16: the example script's network -- n_blocks=4, start_filts=32,
planar_blocks=(0,) -- under float16 autocast, i.e. Trainer(mixed_precision=True), on 2 x 64x128x128)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd.unet import UNet
from elektronn3_amd.loss import CombinedCEDiceLoss
torch.manual_seed(0)
mode = sys.argv[1] if len(sys.argv) > 1 else 'f32'
example = mode.startswith('example')
if example:
    m = UNet(in_channels=1, out_channels=2, n_blocks=4, start_filts=32, planar_blocks=(0,), normalization='batch').cuda().train()
    shape = (64, 128, 128)
else:
    m = UNet(in_channels=1, out_channels=2, n_blocks=4, start_filts=64, planar_blocks=(0, 1), normalization='batch').cuda().train()
    shape = (32, 256, 256)
crit = CombinedCEDiceLoss(weight=[0.2653, 0.7347]).cuda()
x = torch.randn(2, 1, *shape, device='cuda'); t = torch.randint(0, 2, (2, *shape), device='cuda')
lowp = {'bf16': torch.bfloat16, 'f16': torch.float16}.get(mode)
if lowp is not None:
    m = m.to(lowp); x = x.to(lowp)
ac = torch.float16 if mode == 'example_f16' else None
def step():
    if ac is not None:
        with torch.autocast('cuda', dtype=ac):
            loss = crit(m(x), t)
    else:
        loss = crit(m(x), t)
    for p in m.parameters(): p.grad = None
    (loss * 1024.0 if (ac is not None or lowp is torch.float16) else loss).backward()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f'{mode}: ' + ('example net (planar_blocks=(0,), sf=32, 2 x 64x128x128)' if example else 'cfg 4 (planar_blocks=(0,1), sf=64, batch 2 x 32x256x256)') + f': {dt*1e3:.2f} ms/step = {x.numel()/dt/1e6:.1f} M voxels/s; max memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB')
