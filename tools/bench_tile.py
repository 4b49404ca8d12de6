#!/usr/bin/env python3
"""Per-tile compute time of the cfg-5 Predictor tile (128x224x224 input, kept 96x192x192) WITHOUT any host <-> device traffic beside it:
UNet.forward_tile in a loop over a device-resident padded volume.  Compared with predict()'s compute-stream time per tile it tells what the
concurrent H2D / D2H copies cost the persistent kernels.    python tools/bench_tile.py [n_tiles]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elektronn3_amd.unet import UNet
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
torch.manual_seed(0)
m = UNet(1, 2, n_blocks=4, start_filts=32).cuda().train()
with torch.no_grad():
    for _ in range(4):
        m(torch.randn(2, 1, 32, 64, 64, device='cuda'))
m.eval()
vol = torch.randn(1, 1, 128, 224, 224 * 3, device='cuda')
out = torch.zeros(1, 2, 96, 192, 192 * 3, device='cuda')
roi = [(16, 112), (16, 208), (16, 208)]
with torch.no_grad(), m.frozen_weights():      # (as in Predictor.predict: the weights are packed for the first tile only)
    for i in range(3):
        m.forward_tile(vol, (0, 0, 224 * (i % 3)), (128, 224, 224), out, (0, 0, 192 * (i % 3)), roi, softmax=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        m.forward_tile(vol, (0, 0, 224 * (i % 3)), (128, 224, 224), out, (0, 0, 192 * (i % 3)), roi, softmax=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f'tile forward alone: {dt * 1e3:.3f} ms per tile = {96 * 192 * 192 / dt / 1e6:.1f} MVox/s of kept voxels')
# the same loop beside continuous H2D + D2H traffic on two side streams (page-locked buffers, 256 MB pieces): what copies cost the persistent kernels
if os.environ.get('E3_TILE_WITH_COPIES'):
    import threading
    h_in = torch.empty(64 * 2 ** 20, dtype=torch.float32).pin_memory(); h_out = torch.empty(64 * 2 ** 20, dtype=torch.float32).pin_memory()
    d_in = torch.empty(64 * 2 ** 20, dtype=torch.float32, device='cuda'); d_out = torch.randn(64 * 2 ** 20, device='cuda')
    s_up, s_down = torch.cuda.Stream(), torch.cuda.Stream()
    stop = [False]
    def pump():
        while not stop[0]:
            with torch.cuda.stream(s_up): d_in.copy_(h_in, non_blocking=True)
            with torch.cuda.stream(s_down): h_out.copy_(d_out, non_blocking=True)
            s_up.synchronize(); s_down.synchronize()
    th = threading.Thread(target=pump); th.start()
    time.sleep(0.2)
    with torch.no_grad():
        torch.cuda.current_stream().synchronize(); t0 = time.perf_counter()
        for i in range(n):
            m.forward_tile(vol, (0, 0, 224 * (i % 3)), (128, 224, 224), out, (0, 0, 192 * (i % 3)), roi, softmax=True)
        torch.cuda.current_stream().synchronize(); dt2 = (time.perf_counter() - t0) / n
    stop[0] = True; th.join()
    print(f'tile forward beside continuous H2D + D2H copies: {dt2 * 1e3:.3f} ms per tile')
# several tiles of a row as ONE call: a "batch" whose sample stride is the tile step along w (overlapping views of the same padded volume)
if os.environ.get('E3_TILE_BATCH'):
    nb = int(os.environ['E3_TILE_BATCH'])
    volb = torch.randn(1, 1, 128, 224, 32 + 192 * nb, device='cuda')
    outb = torch.zeros(1, 2, 96, 192, 192 * nb, device='cuda')
    vb = torch.as_strided(volb, (nb, 1, 128, 224, 224), (192, volb.stride(1), volb.stride(2), volb.stride(3), 1))
    ob = torch.as_strided(outb, (nb, 2, 96, 192, 192), (192, outb.stride(1), outb.stride(2), outb.stride(3), 1))
    with torch.no_grad():
        for _ in range(2):
            m.forward_tile(vb, (0, 0, 0), (128, 224, 224), ob, (0, 0, 0), roi, softmax=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(max(1, n // nb)):
            m.forward_tile(vb, (0, 0, 0), (128, 224, 224), ob, (0, 0, 0), roi, softmax=True)
        torch.cuda.synchronize(); dtb = (time.perf_counter() - t0) / max(1, n // nb) / nb
        # the same tiles one by one
        ref = torch.zeros_like(outb)
        for i in range(nb):
            m.forward_tile(volb, (0, 0, 192 * i), (128, 224, 224), ref, (0, 0, 192 * i), roi, softmax=True)
        torch.cuda.synchronize()
    print(f'{nb} tiles of a row per call: {dtb * 1e3:.3f} ms per tile; identical to one-by-one: {bool(torch.equal(ref, outb))}, max diff {float((ref - outb).abs().max()):.2e}')
