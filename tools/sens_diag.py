import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import load_npz, rel_l2, sub, unet_cfg
from elektronn3_amd.unet import UNet
g = load_npz('unet_nb3_sf8_planar0_odd.npz'); cfg = unet_cfg(g)
ref64 = sub(g, 'grad64')
torch.manual_seed(0)
for eps in (0.0, 1e-7, 1e-6, 1e-5):
    worsts = []
    for trial in range(4):
        m = UNet(1, 2, **cfg); m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sub(g, 'sd0').items()}); m = m.cuda().train()
        x = torch.from_numpy(g['x']).cuda()
        x = x * (1 + eps * torch.randn_like(x))
        out = m(x)
        out.backward(torch.from_numpy(g['dlogits']).cuda())
        errs = {k: rel_l2(p.grad.cpu().numpy(), ref64[k]) for k, p in m.named_parameters() if not (k.endswith('.bias') and 'norm' not in k and not k.startswith('conv_final'))}
        worsts.append(max(errs.values()))
    print('input perturbation', eps, 'worst grad rel-L2 vs fp64 ref over 4 trials:', ['%.1e' % w for w in worsts])
