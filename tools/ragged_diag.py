import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd.unet import UNet
from oracle.torch_ref import combined_loss, unet_forward
torch.manual_seed(0)
m = UNet(in_channels=1, out_channels=2, n_blocks=4, start_filts=32, normalization='batch').cuda().train()
g = torch.Generator(device='cuda').manual_seed(5)
x = torch.randn(2, 1, 61, 131, 125, device='cuda', generator=g)
t = (torch.rand(2, 61, 131, 125, device='cuda', generator=g) < 0.3).long()
sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
out = m(x); loss = combined_loss(out, t); m.zero_grad(set_to_none=True); loss.backward()
res = {}
for name, dt in (('fp32', torch.float32), ('fp64', torch.float64)):
    sd = {k: (v.detach().to(dt).clone() if v.is_floating_point() else v.clone()).requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd0.items()}
    ref = unet_forward(sd, x.to(dt), 4, (), training=True)
    l = combined_loss(ref, t); l.backward()
    res[name] = (ref.detach(), sd)
r64 = res['fp64'][0]
print('ours  vs fp64: out max abs', float((out - r64).abs().max()))
print('MIOpen fp32 vs fp64: out max abs', float((res['fp32'][0] - r64).abs().max()))
worst_o = worst_r = 0
for k, p in m.named_parameters():
    g64 = res['fp64'][1][k].grad; g32 = res['fp32'][1][k].grad
    if g64.norm() < 1e-6: continue
    eo = float((p.grad - g64).norm() / g64.norm()); er = float((g32 - g64).norm() / g64.norm())
    worst_o = max(worst_o, eo); worst_r = max(worst_r, er)
print('grad rel-L2 worst: ours', worst_o, 'MIOpen fp32', worst_r)
