#!/usr/bin/env python3
"""Predictor MVox/s on a cfg-5-shaped workload (BASELINE.json configs[4]): tiled sliding-window inference of
UNet(1,2,n_blocks=4,start_filts=32) over a synthetic fp32 volume, tile 96x192x192, overlap 16, softmax output.

    python tools/bench_predictor.py                 # full 512x2048x2048 volume (8 GiB in, 16 GiB out)
    python tools/bench_predictor.py --shape 128 448 448

With torch.distributed initialised (torchrun) the tiles are sharded round-robin over the ranks (tile_parallel).
Reports input-voxel based MVox/s like benchmark/pred_benchmark.py:101 (and the reference Predictor's own
output-element based figure, inference.py:637-640).
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', type=int, nargs=3, default=[512, 2048, 2048])
    ap.add_argument('--tile', type=int, nargs=3, default=[96, 192, 192])
    ap.add_argument('--overlap', type=int, nargs=3, default=[16, 16, 16])
    ap.add_argument('--device-input', action='store_true', help='generate the volume on the GPU (skips the H2D copy)')
    a = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1')); rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if 'RANK' in os.environ:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    from elektronn3_amd.unet import UNet
    from elektronn3_amd.inference import Predictor
    torch.manual_seed(0)
    model = UNet(1, 2, n_blocks=4, start_filts=32).cuda()
    model.train()
    with torch.no_grad():                    # running statistics from 10 warm-up batches (SURVEY 8d cfg 5)
        for _ in range(10):
            model(torch.randn(2, 1, 32, 64, 64, device='cuda'))
    D, H, W = a.shape
    g = torch.Generator(device='cuda' if a.device_input else 'cpu').manual_seed(0)
    t0 = time.time()
    vol = torch.randn(1, 1, D, H, W, generator=g, device='cuda' if a.device_input else 'cpu')
    t_gen = time.time() - t0
    pred = Predictor(model, device=f'cuda:{local}', tile_shape=tuple(a.tile), overlap_shape=tuple(a.overlap), offset=None,
                     out_shape=(2, D, H, W), apply_softmax=True, strict_shapes=False, tile_parallel=world > 1)
    # warm-up on one tile-sized volume (allocator, plan, clocks)
    Predictor(model, device=f'cuda:{local}', apply_softmax=True).predict(torch.randn(1, 1, *[t + 2 * o for t, o in zip(a.tile, a.overlap)]))
    torch.cuda.synchronize()
    t0 = time.time()
    out = pred.predict(vol)
    dt = time.time() - t0
    if rank == 0:
        ntiles = int(np.prod(np.ceil(np.array(a.shape) / np.array(a.tile))))
        res = {'metric': 'Predictor MVox/s (input voxels / predict() wall time incl. H2D and D2H)', 'value': D * H * W / dt / 1e6,
               'unit': 'MVox/s', 'n_gpus': world, 'seconds': dt, 'tiles': ntiles, 'volume': a.shape, 'tile': a.tile, 'overlap': a.overlap,
               'out_elements_MVox_s': out.numel() / dt / 1e6, 'volume_generation_s': t_gen, 'device_input': a.device_input,
               'finite': bool(torch.isfinite(out[..., ::64, ::64]).all())}
        print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
