"""Run-to-run identity of the first conv (conv_first_mfma_kernel) and of a whole eval forward under the library named by E3_LIB_PATH
(tools/repro_first_soffset.sh): 20 calls on the same input, the number of distinct results, and the result against the in-tree library's hash."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from elektronn3_amd import ops  # noqa: E402
from elektronn3_amd.unet import UNet  # noqa: E402

torch.manual_seed(0)
x = ops.to_ndhwc(torch.randn(2, 1, 62, 126, 130, device='cuda'))          # ragged against the 4 x 8 x 32 bricks
w = torch.randn(32, 1, 3, 3, 3, device='cuda') * 0.2
b = torch.randn(32, device='cuda')
hs = set()
for _ in range(20):
    y = ops.conv3d(x, w, b)
    hs.add(hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:12])
print('first conv (rows, with statistics off): distinct results over 20 runs:', len(hs), sorted(hs)[:3])
m = UNet(1, 2, n_blocks=4, start_filts=32).cuda().eval()
xx = torch.randn(1, 1, 64, 96, 112, device='cuda')
hs = set()
with torch.no_grad():
    for _ in range(20):
        hs.add(hashlib.sha256(m(xx).cpu().numpy().tobytes()).hexdigest()[:12])
print('eval forward (chunked first-conv output): distinct results over 20 runs:', len(hs), sorted(hs)[:3])
