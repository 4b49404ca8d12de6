"""Per-conv-layer roofline table for cfg 2 (north_star: "achieved fraction of HBM roofline per conv layer").

Every conv / transposed conv / head of UNet(1,2,n_blocks=4,start_filts=32) is timed INSIDE complete training steps with HIP
events around its dominant kernel (e3_unet_profile_select/read), forward, dgrad and wgrad separately.  Algorithmic work
per launch follows SURVEY.md section 8(d): FLOPs = 2*Cin*Cout*taps per output voxel (transposed conv: per input voxel),
bytes = x + y + w (fwd), dy + w + dx (dgrad), x + dy + dw (wgrad), fp32.
Fractions: HBM = bytes / t / 8 TB/s, fp32 = FLOPs / t / 157.3 TFLOP/s (Winograd layers can exceed 1: the F(2x2x2) kernels execute 64/216, the F(2x2x4) data gradients of levels 0 and 1 96/432, the planar ones
16/36 of the algorithmic multiplies).
Usage: python tools/layer_table.py [steps] > profiles/rNN_per_layer.md"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd.unet import UNet
from elektronn3_amd.loss import CombinedCEDiceLoss

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
BF16 = len(sys.argv) > 2 and sys.argv[2] == 'bf16'          # native bf16 path (BASELINE configs[2]): 2-byte tensors, 2.5 PFLOP/s dense bf16 MFMA peak
ESZ, PEAK, PEAKNAME = (2.0, 2500.0, 'bf16') if BF16 else (4.0, 157.3, 'fp32')
N, CROP = 2, (64, 128, 128)
dev = torch.device('cuda')
torch.manual_seed(0)
model = UNet(in_channels=1, out_channels=2, n_blocks=4, start_filts=32, normalization='batch').to(dev).train()
crit = CombinedCEDiceLoss(weight=[0.2653, 0.7347]).to(dev)
x = torch.randn(N, 1, *CROP, device=dev); t = torch.randint(0, 2, (N, *CROP), device=dev)
if BF16:
    model = model.to(torch.bfloat16); x = x.to(torch.bfloat16)


def step():
    loss = crit(model(x), t)
    for p in model.parameters(): p.grad = None
    loss.backward()


for _ in range(3): step()
torch.cuda.synchronize()
rows = []
tot = {0: 0.0, 1: 0.0, 2: 0.0}
for li, (name, cin, cout, taps, level) in enumerate(model.conv_layers()):
    vox_lvl = N * CROP[0] * CROP[1] * CROP[2] // (8 ** level)         # voxels of the layer's resolution level
    up = 'upconv' in name
    vin = vox_lvl // 8 if up else vox_lvl                            # transposed conv reads the coarser level
    vout = vox_lvl
    flops = 2.0 * cin * cout * taps * (vin if up else vout)
    byts = ESZ * (cin * vin + cout * vout + cin * cout * taps)
    ms = []
    for which in (0, 1, 2):
        if which == 1 and li == 0: ms.append(None); continue        # no dx for the network input
        model.profile_select(li, which)
        for _ in range(steps): step()
        torch.cuda.synchronize()
        m, n = model.profile_read()
        ms.append(m if n else None)
        if n: tot[which] += m
    rows.append((name, cin, cout, taps, vout, flops, byts, ms))
model.profile_select(-1, 0)

print(f'# Per-conv-layer roofline (tools/layer_table.py), cfg 2 (UNet n_blocks=4 start_filts=32, batch 2 of 64x128x128, {"bf16 (native path)" if BF16 else "fp32"}), one MI355X')
print()
print(f'HIP events around each layer\'s dominant kernel inside full training steps (`tools/layer_table.py`, mean of {steps} steps per cell). '
      f'HBM = algorithmic bytes / t / 8 TB/s; {PEAKNAME} = algorithmic FLOPs / t / {PEAK} TFLOP/s' + ('' if BF16 else ' (Winograd kernels execute 64/216 of the '
      '3x3x3 multiplies, so their algorithmic fraction can exceed 1)') + '. Bytes/FLOPs per SURVEY.md section 8(d).')
print()
print(f'| layer | Cin→Cout | taps | voxels out | GFLOP | MB | fwd µs | fwd TF/s | fwd {PEAKNAME} frac | fwd HBM frac | dgrad µs | dgrad TF/s | dgrad HBM frac | wgrad µs | wgrad TF/s | wgrad HBM frac |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|')
for name, cin, cout, taps, vout, flops, byts, ms in rows:
    cells = [name, f'{cin}→{cout}', str(taps), str(vout), f'{flops / 1e9:.2f}', f'{byts / 1e6:.1f}']
    for k, m in enumerate(ms):
        if m is None or m <= 0:
            cells += ['—'] * (4 if k == 0 else 3); continue
        tf = flops / (m * 1e-3) / 1e12
        hb = byts / (m * 1e-3) / 8e12
        cells += [f'{m * 1e3:.0f}', f'{tf:.1f}'] + ([f'{tf / PEAK:.3f}'] if k == 0 else []) + [f'{hb:.3f}']
    print('| ' + ' | '.join(cells) + ' |')
print()
print(f'Sum of the timed kernels per step: forward {tot[0]:.2f} ms, dgrad {tot[1]:.2f} ms, wgrad {tot[2]:.2f} ms '
      '(the rest of the step is BN/ReLU/pool passes, statistic finalisers, weight packing, slab reduction and the criterion).')
