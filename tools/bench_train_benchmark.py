"""The reference's own train benchmark workload (benchmark/train_benchmark.py via train_benchmark.sh: UNet(n_blocks=4, start_filts=32,
planar_blocks=(0,), bn), batch 8 of 44 x 88 x 88 patches, AdamW + SWA, variants plain / --amp) as a synthetic-data step loop:
forward + CE/Dice loss + backward + optimizer step (+ GradScaler with --amp).  Usage: python tools/bench_train_benchmark.py [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd.unet import UNet
from elektronn3_amd.loss import CombinedCEDiceLoss
from elektronn3_amd.optim import AdamW

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
x = torch.randn(8, 1, 44, 88, 88, device='cuda'); t = torch.randint(0, 2, (8, 44, 88, 88), device='cuda')
for amp in (False, True):
    torch.manual_seed(0)
    m = UNet(in_channels=1, out_channels=2, n_blocks=4, start_filts=32, planar_blocks=(0,), normalization='batch').cuda().train()
    crit = CombinedCEDiceLoss(weight=[0.2653, 0.7347]).cuda()
    opt = AdamW(m.parameters(), lr=1e-3, weight_decay=0.5e-4)
    scaler = torch.amp.GradScaler('cuda', enabled=amp)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.float16, enabled=amp):
            loss = crit(m(x), t)
        scaler.scale(loss).backward()
        scaler.step(opt); scaler.update()
        return loss
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): l = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f'train_benchmark workload, {"--amp (float16 autocast + GradScaler)" if amp else "fp32"}: {dt * 1e3:.2f} ms/step = {x.numel() / dt / 1e6:.1f} MVox/s, loss {float(l):.4f}')
