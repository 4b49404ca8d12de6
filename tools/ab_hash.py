import os, sys, hashlib, torch
sys.path.insert(0, os.getcwd())
from elektronn3_amd.unet import UNet
from elektronn3_amd.loss import CombinedCEDiceLoss
torch.manual_seed(0)
m = UNet(1, 2, n_blocks=4, start_filts=32).cuda().train()
x = torch.randn(2, 1, 64, 128, 128, device='cuda'); t = torch.randint(0, 2, (2, 64, 128, 128), device='cuda')
crit = CombinedCEDiceLoss(weight=[0.2653, 0.7347]).cuda()
h = hashlib.sha256()
for i in range(2):
    out, loss = m.forward_with_loss(x, t, crit)
    for p in m.parameters(): p.grad = None
    loss.backward()
h.update(out.detach().cpu().numpy().tobytes())
for p in m.parameters(): h.update(p.grad.cpu().numpy().tobytes())
m.eval()
with torch.no_grad():
    y = m(torch.randn(1, 1, 64, 96, 112, device='cuda'))
h.update(y.cpu().numpy().tobytes())
print('HASH', h.hexdigest()[:16], float(loss))
