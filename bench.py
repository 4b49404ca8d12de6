#!/usr/bin/env python3
"""Benchmark of the hot path: 3D U-Net training step (forward + backward) on synthetic 64x128x128 crops.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` (for N > 1 launched by
``python -m torch.distributed.run --nproc-per-node N``), prints ONE JSON line on rank 0.

Workload = BASELINE.json configs[1]: ``UNet(in=1, out=2, n_blocks=4, start_filts=32, normalization='batch')``, fp32,
batch 2 per GPU of 1x64x128x128 crops (N > 1: the same per-GPU batch on every rank = weak scaling, gradients
all-reduced with RCCL on a side stream overlapped with the backward).  A step is forward + loss + backward
(+ gradient all-reduce); the optimizer is excluded.  Inputs are resident in HBM before the timed region.

Extra objects on the JSON line:
  roofline      the conv of the heaviest layer (up_convs.2.conv1, 64->32 at full resolution): ALGORITHMIC flops per launch
                (direct-convolution count 2*Cin*Cout*27 per voxel) / mean launch time measured with HIP
                events on the compute stream INSIDE the timed steps (e3_unet_profile_*), against the 157.3 TFLOP/s fp32
                matrix peak.  The forward/dgrad kernel is Winograd F(2x2x2,3x3x3): it EXECUTES 64/216 of those flops on
                the matrix cores, so `frac` can exceed 1; `mfma_executed` reports the executed matrix flops against
                the same peak.  wgrad (direct implicit GEMM) executes exactly the algorithmic count.
  cpu_baseline  the reference's ATen op sequence (oracle/torch_ref.py) on the host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CROP = (64, 128, 128)
BATCH_PER_GPU = 2
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
FWD_FLOP_PER_VOXEL = 427.2e3    # SURVEY.md 8d (cfg 2 network)
FWDBWD_FLOP_PER_VOXEL = 1279.9e3


def conv_flops(cin, cout, taps, voxels):
    return 2.0 * cin * cout * taps * voxels


def cpu_baseline(iters=2):
    """The reference's CPU PyTorch path (same ATen op sequence, oracle/torch_ref.py) timed on this box's host cores."""
    from oracle.torch_ref import combined_loss, unet_forward
    from elektronn3_amd.unet import UNet
    torch.manual_seed(0)
    m = UNet(in_channels=1, out_channels=2, n_blocks=4, start_filts=32)
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in m.state_dict().items()}
    x = torch.randn(BATCH_PER_GPU, 1, *CROP)
    t = torch.randint(0, 2, (BATCH_PER_GPU, *CROP))
    times = []
    for i in range(iters + 1):
        t0 = time.time()
        out = unet_forward(sd, x, 4, (), training=True)
        loss = combined_loss(out, t)
        loss.backward()
        for v in sd.values():
            v.grad = None
        times.append(time.time() - t0)
    dt = sum(times[1:]) / iters
    return {'value': x.numel() / dt, 'unit': 'voxels/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'full cfg-2 batch ({BATCH_PER_GPU}x1x{"x".join(map(str, CROP))}) fp32 fwd+bwd, 1 warm-up + {iters} timed iterations '
                      f'of the reference\'s ATen op sequence (oracle/torch_ref.py) with torch {torch.__version__} on the host CPU',
            's_per_step': dt}


def predictor_leg(dev, shape=(288, 1152, 1152), tile=(96, 192, 192), overlap=(16, 16, 16), tile_parallel=False):
    """BASELINE.json's second metric ("Predictor MVox/s", configs[4]) on the cfg-5 geometry -- tile 96x192x192, overlap 16, eval-mode
    UNet(n_blocks=4, start_filts=32), softmax output, fp32 volume in HOST memory, result back in host memory -- over a
    288x1152x1152 sub-volume (108 tiles) so that the default bench run stays short; tools/bench_predictor.py runs the full
    512x2048x2048 volume (726 tiles).  Input voxels / predict() wall time incl. H2D and D2H (benchmark/pred_benchmark.py:101)."""
    from elektronn3_amd.inference import Predictor
    from elektronn3_amd.unet import UNet
    torch.manual_seed(0)
    model = UNet(1, 2, n_blocks=4, start_filts=32).to(dev)
    model.train()
    with torch.no_grad():                    # running statistics from 10 warm-up batches (SURVEY 8d cfg 5)
        for _ in range(10):
            model(torch.randn(2, 1, 32, 64, 64, device=dev))
    vol = torch.randn(1, 1, *shape, generator=torch.Generator().manual_seed(0))
    Predictor(model, device=dev, apply_softmax=True).predict(torch.randn(1, 1, *[t + 2 * o for t, o in zip(tile, overlap)]))   # warm-up tile
    pred = Predictor(model, device=dev, tile_shape=tile, overlap_shape=overlap, offset=None, out_shape=(2, *shape), apply_softmax=True,
                     strict_shapes=False, tile_parallel=tile_parallel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pred.predict(vol)
    dt = time.perf_counter() - t0
    ntiles = 1
    for n, t in zip(shape, tile):
        ntiles *= -(-n // t)
    return {'metric': 'Predictor MVox/s', 'value': vol.numel() / dt / 1e6, 'unit': 'MVox/s (input voxels / predict() wall time incl. H2D + D2H)',
            'seconds': dt, 'volume': list(shape), 'tile': list(tile), 'overlap': list(overlap), 'tiles': ntiles, 'dtype': 'f32',
            'finite': bool(torch.isfinite(out[..., ::32, ::32]).all()),
            'note': 'cfg-5 geometry on a sub-volume; the full 512x2048x2048 volume (726 tiles): tools/bench_predictor.py, DESIGN.md section 5'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-predictor', action='store_true', help='skip the Predictor MVox/s leg (N=1 only)')
    ap.add_argument('--profile-layer', default='up_convs.2.conv1')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU (the HIP path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or 'RANK' in os.environ:   # launched by torch.distributed.run (also with a single rank)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # RCCL's kernels must get compute units while the (512-register, one-workgroup-per-CU) conv kernels of the backward own the
        # chip: run the collectives on a high-priority stream so that they are dispatched first whenever a CU frees up
        pg_opts = None
        try:
            pg_opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        except Exception:  # noqa: BLE001
            pg_opts = None
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev, **({'pg_options': pg_opts} if pg_opts is not None else {}))

    from elektronn3_amd.unet import UNet
    from elektronn3_amd.loss import CombinedCEDiceLoss   # the example's criterion (0.5 CE + 0.5 Dice, class weights) on device

    torch.manual_seed(0)                                   # identical replica on every rank
    model = UNet(in_channels=1, out_channels=2, n_blocks=4, start_filts=32, normalization='batch').to(dev).train()
    # N > 1: ONE loss over the global minibatch, as the reference computes on the batch nn.DataParallel gathers (trainer.py:520-524):
    # the ranks exchange the criterion's 2 + 3C sums (one all-reduce of 8 doubles) between forward and backward
    criterion = CombinedCEDiceLoss(weight=[0.2653, 0.7347], global_batch=dist is not None).to(dev)
    sync = None
    if world > 1 or (dist is not None):
        from elektronn3_amd.dataparallel import GradSync
        sync = GradSync(model)
    torch.manual_seed(1000 + rank)                         # different synthetic crops per rank
    x = torch.randn(BATCH_PER_GPU, 1, *CROP, device=dev)
    tgt = torch.randint(0, 2, (BATCH_PER_GPU, *CROP), device=dev)

    layers = model.conv_layers()
    names = [l[0] for l in layers]
    li = names.index(args.profile_layer)
    _, lcin, lcout, ltaps, llevel = layers[li]

    def step():
        out = model(x)
        loss = criterion(out, tgt)
        for p in model.parameters():
            p.grad = None
        loss.backward()
        if sync is not None:
            sync.wait()
        return loss

    def timed(nsteps, which):
        model.profile_select(li, which)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        ms, n = model.profile_read()
        return dt, ms, n

    for _ in range(args.warmup):
        step()
    dt, k_ms, k_n = timed(args.steps, 0)
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms_per_step = dt / args.steps * 1e3
    # secondary kernel timings (outside the reported timed region): dgrad and wgrad of the same layer
    extra = {}
    for which, tag in ((1, 'dgrad'), (2, 'wgrad')):
        _, ms, n = timed(2, which)
        extra[tag] = ms

    multi_pred = None
    if dist is not None and world > 1 and os.environ.get('E3_BENCH_PREDICTOR_MULTI') and not args.no_predictor:
        # opt-in (a collective path that the 1-GPU development box cannot exercise over RCCL): every rank predicts its share of
        # the tile rows of the same sub-volume; all ranks call this
        multi_pred = predictor_leg(dev, tile_parallel=True)
    if rank == 0:
        vox_per_step = world * BATCH_PER_GPU * CROP[0] * CROP[1] * CROP[2]
        lvox = BATCH_PER_GPU * CROP[0] * CROP[1] * CROP[2] // (8 ** llevel)
        lflops = conv_flops(lcin, lcout, ltaps, lvox)
        ach = lflops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        res = {
            'metric': 'voxels/sec (train fwd+bwd) 3D UNet 64x128x128',
            'value': vox_per_step / (ms_per_step * 1e-3),
            'unit': 'voxels/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE.json configs[1]: UNet(in=1,out=2,n_blocks=4,start_filts=32,bn) fp32 train fwd+bwd, '
                                   f'batch {BATCH_PER_GPU}/GPU of 1x64x128x128 random crops, CE+Dice loss, optimizer excluded',
                       'global_batch': world * BATCH_PER_GPU, 'crop': list(CROP),
                       'parallelism': f'dp{world}' if world > 1 else 'single',
                       'step_tflops': FWDBWD_FLOP_PER_VOXEL * vox_per_step / world / (ms_per_step * 1e-3) / 1e12},
            'roofline': {'bound': 'mfma', 'achieved': ach, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': ach / FP32_MFMA_PEAK_TFLOPS,
                         # HBM bytes per launch from rocprofv3 PMC passes of this same command (FETCH_SIZE x2 per the gfx950
                         # correction + WRITE_SIZE, profiles/r01_pmc_wino.md); only known for the default layer/config
                         'traffic': 1.30e9 if (args.profile_layer == 'up_convs.2.conv1' and ltaps == 27) else None,
                         'kernel': f'conv3_wino_pkernel (persistent Winograd F(2x2x2,3x3x3), fp32 MFMA) fwd of {args.profile_layer} '
                                   f'({lcin}->{lcout}, {ltaps} taps, {lvox} voxels)' if ltaps == 27 else
                                   f'conv3_v3_kernel fwd of {args.profile_layer} ({lcin}->{lcout}, {ltaps} taps, {lvox} voxels)',
                         'mfma_executed': {'flops_per_launch': lflops * (64.0 / 216.0 if ltaps == 27 else 1.0),
                                           'tflops': ach * (64.0 / 216.0 if ltaps == 27 else 1.0),
                                           'frac': ach * (64.0 / 216.0 if ltaps == 27 else 1.0) / FP32_MFMA_PEAK_TFLOPS},
                         'flops_per_launch': lflops, 'ms_per_launch': k_ms, 'launches_timed': k_n,
                         'dgrad_ms': extra.get('dgrad'), 'wgrad_ms': extra.get('wgrad'),
                         'dgrad_tflops': lflops / (extra['dgrad'] * 1e-3) / 1e12 if extra.get('dgrad') else None,
                         'wgrad_tflops': lflops / (extra['wgrad'] * 1e-3) / 1e12 if extra.get('wgrad') else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                res['cpu_baseline'] = cpu_baseline()
            except Exception as e:  # noqa: BLE001
                res['cpu_baseline'] = {'value': None, 'unit': 'voxels/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                                       'sample': f'failed: {e}'}
        if world == 1 and dist is None and not args.no_predictor:
            try:
                del x, tgt
                torch.cuda.empty_cache()
                res['predictor'] = predictor_leg(dev)
            except Exception as e:  # noqa: BLE001
                res['predictor'] = {'metric': 'Predictor MVox/s', 'value': None, 'note': f'failed: {e}'}
        elif multi_pred is not None:
            res['predictor'] = dict(multi_pred, n_gpus=world, parallelism=f'tile-parallel over {world} ranks, shared-memory output')
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
