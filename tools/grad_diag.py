import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import load_npz, rel_l2, sub, unet_cfg
from oracle.torch_ref import combined_loss
from elektronn3_amd.unet import UNet
case = sys.argv[1] if len(sys.argv) > 1 else 'unet_nb3_sf8_planar0_odd.npz'
g = load_npz(case); cfg = unet_cfg(g)
for rep in range(2):
    m = UNet(1, 2, **cfg); m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sub(g, 'sd0').items()}); m = m.cuda().train()
    out = m(torch.from_numpy(g['x']).cuda())
    loss = combined_loss(out, torch.from_numpy(g['target']).cuda()); loss.backward()
    ref32, ref64 = sub(g, 'grad'), sub(g, 'grad64')
    bad = []
    for k, p in m.named_parameters():
        eb, er = rel_l2(p.grad.cpu().numpy(), ref64[k]), rel_l2(ref32[k], ref64[k])
        if eb > max(3 * er, 1e-4) and not (k.endswith('.bias') and 'norm' not in k and not k.startswith('conv_final')):
            bad.append((k, f'{eb:.2e}', f'{er:.2e}'))
    print('rep', rep, 'logit err', float(np.abs(out.detach().cpu().numpy() - g['logits']).max()), 'bad:', bad)
