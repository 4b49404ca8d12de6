"""PReLU slope gradients (scalar sums with heavy cancellation) of the option-variant test configs: HIP path vs fp64, MIOpen fp32 vs fp64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd.unet import UNet
from oracle.torch_ref import combined_loss, unet_forward
for kw in (dict(activation='prelu', planar_blocks=(0,)), dict(activation='prelu', normalization='none')):
    torch.manual_seed(9)
    m = UNet(in_channels=1, out_channels=2, n_blocks=3, start_filts=32, **kw).cuda().train()
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.endswith('.bias'): p.copy_(0.1 * torch.randn_like(p))
            elif '.act' in k: p.copy_(0.25 + 0.3 * torch.randn_like(p))
    x = torch.randn(2, 1, 32, 64, 64, device='cuda')
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    out = m(x); t = torch.randint(0, 2, (2, *out.shape[2:]), device='cuda')
    loss = combined_loss(out, t); m.zero_grad(set_to_none=True); loss.backward()
    res = {}
    for name, dt in (('fp32', torch.float32), ('fp64', torch.float64)):
        sd = {k: (v.detach().to(dt).clone() if v.is_floating_point() else v.clone()).requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in sd0.items()}
        sd['__act_slope__'] = 3.0
        ref = unet_forward(sd, x.to(dt), 3, tuple(kw.get('planar_blocks', ())), training=True)
        combined_loss(ref, t).backward()
        res[name] = sd
    print(kw)
    for k, p in m.named_parameters():
        g64 = res['fp64'][k].grad; g32 = res['fp32'][k].grad
        if g64 is None or float(g64.norm()) < 1e-12: continue
        eo = float((p.grad.double() - g64).norm() / g64.norm()); er = float((g32.double() - g64).norm() / g64.norm())
        if '.act' in k or eo > 3e-3: print(f'  {k:32s} ours {eo:.2e}  MIOpen-fp32 {er:.2e}  |g|={float(g64.norm()):.3e}')
