#!/usr/bin/env python3
"""Benchmark of the hot path: 3D U-Net training step (forward + backward) on synthetic 64x128x128 crops.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on rank 0.  For N > 1 the driver launches
it under ``python -m torch.distributed.run --nproc-per-node N``; called directly with ``--gpus N`` (no WORLD_SIZE in the
environment) it re-executes itself under that launcher, so ``python bench.py --gpus 8`` never silently benches one GPU.

Workload = BASELINE.json configs[1]: ``UNet(in=1, out=2, n_blocks=4, start_filts=32, normalization='batch')``, fp32, batch 2 per
GPU of 1x64x128x128 crops (N > 1: the same per-GPU batch on every rank = weak scaling, gradients all-reduced with RCCL on a side
stream overlapped with the backward).  ``--dtype bf16`` runs configs[2]'s per-GPU workload instead: the same module cast with
``.to(torch.bfloat16)``, bf16 crops, native bf16 kernels.  A step is forward + loss + backward (+ gradient all-reduce); the
optimizer is excluded.  Inputs are resident in HBM before the timed region.

Extra objects on the JSON line:
  roofline      the forward conv of the heaviest layer (up_convs.2.conv1, 64->32 at full resolution), timed with HIP events on the
                compute stream INSIDE the timed steps (e3_unet_profile_*).  `achieved` = EXECUTED matrix TFLOP/s and `frac` =
                achieved / dense MFMA peak of the dtype (fp32: the Winograd F(2x2x2,3x3x3) kernel executes 64/216 of the direct
                convolution's multiplies; bf16: direct implicit GEMM, executed == algorithmic).  `algorithmic_tflops` (SURVEY 8d's
                2*Cin*Cout*27 per voxel / time) and `hbm_frac` (SURVEY 8d's layer bytes / time / 8 TB/s) are flat keys beside it.
                `traffic` = HBM bytes per launch from the rocprofv3 PMC passes recorded in the file `traffic_source` names.
  cpu_baseline  the reference's ATen op sequence (oracle/torch_ref.py) on the host cores, rank 0, N = 1 only: BASELINE.md section 3's
                protocol (one cfg-2 sample, 1 warm-up + 3 timed forward+backward iterations).
  predictor     BASELINE's second metric ("Predictor MVox/s", configs[4]): N = 1: the FULL 512x2048x2048 volume (726 tiles);
                N > 1: tile-parallel over the ranks on a 288x1152x1152 sub-volume.
"""
import argparse
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CROP = (64, 128, 128)
BATCH_PER_GPU = 2
MFMA_PEAK_TFLOPS = {'f32': 157.3, 'bf16': 2500.0, 'f16': 2500.0}   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16, dense
HBM_PEAK = 8.0e12                                    # B/s (spec)
FWDBWD_FLOP_PER_VOXEL = 1279.9e3                     # SURVEY.md 8d (cfg 2 network)
PMC_FILE = os.path.join('profiles', 'r06_pmc_roofline.json')      # fallback of roofline.traffic when the live PMC passes cannot run


def live_traffic(dtype, kernel_substr, timeout=240):
    """roofline.traffic measured IN this run (it used to be replayed from a committed file): two child runs of this script under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `... WRITE_SIZE` (separate passes, as MI355X_MICROARCH.md prescribes; 1 warm-up + 1 timed + 2 x 2 per-layer
    timing steps each), the dispatches of the roofline kernel grouped by their position in a step, FETCH_SIZE doubled (gfx950 wide-read correction).
    Returns (hbm bytes per launch of the roofline layer, description) or (None, reason)."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, 'rocprofv3 not found'
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    try:
        from pmc_roofline import per_position
    except Exception as e:  # noqa: BLE001
        return None, f'tools/pmc_roofline.py: {e}'
    vals = {}
    with tempfile.TemporaryDirectory(dir='/tmp') as tmp:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(tmp, counter)
            cmd = [exe, '--kernel-trace', '--pmc', counter, '-d', out, '-o', 'run', '--output-format', 'csv', '--', sys.executable, os.path.abspath(__file__),
                   '--no-cpu-baseline', '--no-predictor', '--no-extra-legs', '--no-live-traffic', '--steps', '1', '--warmup', '1', '--dtype', dtype]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=tmp, env={**os.environ, 'TMPDIR': '/tmp'})
                if r.returncode != 0:
                    return None, f'rocprofv3 --pmc {counter} failed: {(r.stderr or r.stdout)[-200:]}'
                vals[counter] = per_position(out, kernel_substr, counter, 6)
            except Exception as e:  # noqa: BLE001
                return None, f'rocprofv3 --pmc {counter}: {e}'
    rd, wr = vals['FETCH_SIZE'], vals['WRITE_SIZE']
    pos = max(range(len(rd)), key=lambda i: rd[i])      # the roofline layer (64 -> 32 at full resolution) has the largest read volume
    return rd[pos] * 1024 * 2 + wr[pos] * 1024, (f'live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two child passes of this run); launch {pos} of {len(rd)} '
                                               f'of {kernel_substr} per step: fetch {rd[pos] * 2048 / 1e6:.1f} MB (FETCH_SIZE x 2), write {wr[pos] * 1024 / 1e6:.1f} MB')


def cpu_baseline(iters=3):
    """The reference's CPU PyTorch path (same ATen op sequence, oracle/torch_ref.py) timed on this box's host cores."""
    from oracle.torch_ref import combined_loss, unet_forward
    from elektronn3_amd.unet import UNet
    torch.manual_seed(0)
    m = UNet(in_channels=1, out_channels=2, n_blocks=4, start_filts=32)
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and 'running' not in k) for k, v in m.state_dict().items()}
    x = torch.randn(1, 1, *CROP)
    t = torch.randint(0, 2, (1, *CROP))
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
            'sample': f'one cfg-2 sample (1x1x{"x".join(map(str, CROP))}) fp32 fwd+bwd, 1 warm-up + {iters} timed iterations (BASELINE.md section 3) '
                      f'of the reference\'s ATen op sequence (oracle/torch_ref.py) with torch {torch.__version__} on the host CPU '
                      f'({os.cpu_count()} logical cores)',
            's_per_step': dt}


def needed_region_flops(tile_in, overlap, n_blocks=4, sf=32, bricks=((4, 4, 16),) * 3, fp32=True):
    """(forward FLOP of one tile, FLOP the needed-region forward skips, matrix FLOP the fp32 kernels EXECUTE for what is not skipped): the Predictor keeps
    the central crop of a tile, so the decoder's 3x3x3 convs only compute the bricks (fp32 Winograd: 4 x 4 x 16 voxels; 16-bit kernels: 4 x 4 x 32 at level
    0, 2 x 8 x 16 below) that crop depends on -- the box grows by one voxel per conv and halves per transposed conv on the way back through the decoder
    (elektronn3_amd/csrc/unet_plan.cpp, e3_unet_forward_roi).  fp32: levels whose per-sample grid has >= 256 workgroup-bricks run the eval-mode forward on
    F(2x2x4) Winograd tiles (csrc/conv_wino4.hip: 96 multiplies per 16 outputs = 96/432 of the direct count; bricks start at multiples of 4 along W and the
    box of a layer in FRONT of such a conv grows to the W tile grid), the others on F(2x2x2) (64/216)."""
    vox = tile_in[0] * tile_in[1] * tile_in[2]
    whole = 427.2e3 * vox                                                 # forward FLOP per input voxel of UNet(n_blocks=4, sf=32)
    lo = [o for o in overlap]; hi = [t - o for t, o in zip(tile_in, overlap)]
    skipped = 0.0
    F222, F224 = 64.0 / 216.0, 96.0 / 432.0

    def factor(lvl):
        d = [t >> lvl for t in tile_in]
        nblk1 = -(-d[0] // 4) * -(-d[1] // 4) * -(-d[2] // 16) * ((sf << lvl) // 32)
        return F224 if (fp32 and nblk1 >= 256) else F222
    conv_flop = {}                                                        # level -> 3x3x3 conv FLOP of the whole tile (encoder + decoder)
    for lvl in range(n_blocks):
        d = [t >> lvl for t in tile_in]
        c = sf << lvl
        pairs = [(c // 2 if lvl else 0, c), (c, c)] + ([(2 * c, c), (c, c)] if lvl < n_blocks - 1 else [])      # (the 1 -> 32 first conv is not a Winograd layer)
        conv_flop[lvl] = sum(2.0 * 27 * ci * co for ci, co in pairs) * d[0] * d[1] * d[2]
    skipped_lvl = {lvl: 0.0 for lvl in range(n_blocks)}
    for lvl in range(n_blocks - 1):
        dims = [t >> lvl for t in tile_in]
        c = sf << lvl
        w4 = factor(lvl) == F224
        for cin in (c, 2 * c):                                            # conv2 (c -> c), then conv1 (concat 2c -> c), walking backwards
            edge = bricks[lvl]
            blo = [l & ~1 for l in lo] if len(set(bricks)) == 1 else list(lo)      # bricks start at the box's low corner (fp32 Winograd: rounded down to an even voxel)
            if w4:
                blo[2] = lo[2] // 4 * 4
            bhi = [min(b + -(-(h - b) // e) * e, d) for b, h, e, d in zip(blo, hi, edge, dims)]
            done = (bhi[0] - blo[0]) * (bhi[1] - blo[1]) * (bhi[2] - blo[2])
            sk = 2.0 * 27 * cin * c * (dims[0] * dims[1] * dims[2] - done)
            skipped += sk; skipped_lvl[lvl] += sk
            if w4:
                lo[2] = lo[2] // 4 * 4; hi[2] = -(-hi[2] // 4) * 4
            lo = [max(0, l - 1) for l in lo]; hi = [min(d, h + 1) for h, d in zip(hi, dims)]
        lo = [l // 2 for l in lo]; hi = [-(-h // 2) for h in hi]          # through the transposed conv
    executed = sum((conv_flop[lvl] - skipped_lvl[lvl]) * factor(lvl) for lvl in range(n_blocks)) + (whole - sum(conv_flop.values())) * 1.0
    return whole, skipped, executed


class GpuSensors:
    """Shader clock, socket power and temperature of THIS process' GPU, and how many of the node's other GPUs are busy, sampled from sysfs (amdgpu hwmon)
    every 50 ms on a thread while a leg runs.  Reporting only: the boxes of the pool are shared nodes, and single runs of the same build differ by a few
    percent (the Predictor leg by up to 11 %) with what the neighbours do -- the line says under which conditions its numbers were taken."""

    def __init__(self, dev):
        self.me, self.others = None, []
        try:
            bus = torch.cuda.get_device_properties(dev).pci_bus_id
            want = f'{int(bus):02x}:' if isinstance(bus, int) else None
            import glob
            for d in sorted(glob.glob('/sys/class/drm/card*/device')):
                hw = glob.glob(os.path.join(d, 'hwmon', 'hwmon*'))
                if not hw or not os.path.exists(os.path.join(hw[0], 'freq1_input')):
                    continue
                addr = os.path.basename(os.path.realpath(d))              # 0000:bb:dd.f
                if want is not None and addr.split(':', 1)[1].startswith(want) and self.me is None:
                    self.me = hw[0]
                else:
                    self.others.append(hw[0])
        except Exception:  # noqa: BLE001
            self.me = None
        self.rows, self._stop, self._thr = [], None, None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except Exception:  # noqa: BLE001
            return None

    def _loop(self):
        while not self._stop.is_set():
            busy = sum(1 for o in self.others if (self._read(os.path.join(o, 'freq1_input')) or 0) > 1.0e9)
            self.rows.append((self._read(os.path.join(self.me, 'freq1_input')), self._read(os.path.join(self.me, 'power1_input')),
                              self._read(os.path.join(self.me, 'temp2_input')), busy))
            self._stop.wait(0.05)

    def __enter__(self):
        if self.me is not None:
            self.rows, self._stop = [], threading.Event()
            self._thr = threading.Thread(target=self._loop, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *exc):
        if self._thr is not None:
            self._stop.set()
            self._thr.join()
            self._thr = None
        return False

    def summary(self):
        if not self.rows:
            return None
        def col(i, scale):
            v = [r[i] * scale for r in self.rows if r[i] is not None]
            return [round(min(v), 1), round(sum(v) / len(v), 1), round(max(v), 1)] if v else None
        return {'samples': len(self.rows), 'sclk_mhz_min_mean_max': col(0, 1e-6), 'socket_power_w_min_mean_max': col(1, 1e-6), 'temp_c_min_mean_max': col(2, 1e-3),
                'power_cap_w': (lambda v: round(v * 1e-6, 1) if v else None)(self._read(os.path.join(self.me, 'power1_cap'))),
                'other_gpus_of_the_node_busy_mean': round(sum(r[3] for r in self.rows) / len(self.rows), 2), 'other_gpus_seen': len(self.others),
                'source': 'amdgpu hwmon in sysfs, 50 ms samples'}


def predictor_leg(dev, shape, tile=(96, 192, 192), overlap=(16, 16, 16), tile_parallel=False, bf16=False, whole_tiles=False):
    """BASELINE.json's second metric ("Predictor MVox/s", configs[4]): tile 96x192x192, overlap 16, eval-mode UNet(n_blocks=4,
    start_filts=32), softmax output, fp32 volume in HOST memory, result back in host memory.  Input voxels / predict() wall time
    incl. H2D and D2H (benchmark/pred_benchmark.py:101).  bf16 (`--dtype bf16`): the module is cast with model.to(torch.bfloat16) after its
    running statistics are set -- the bf16 counterpart of pred_benchmark.py's float16 switch: bf16 tiles, native bf16 kernels, bf16 result volume."""
    from elektronn3_amd.inference import Predictor
    from elektronn3_amd.unet import UNet
    from elektronn3_amd import inference as _inf
    roi_default = _inf._ROI
    if whole_tiles:
        _inf._ROI = False
    torch.manual_seed(0)
    model = UNet(1, 2, n_blocks=4, start_filts=32).to(dev)
    model.train()
    with torch.no_grad():                    # running statistics from 10 warm-up batches (SURVEY 8d cfg 5)
        for _ in range(10):
            model(torch.randn(2, 1, 32, 64, 64, device=dev))
    if bf16:
        model = model.to(torch.float16 if bf16 == 'f16' else torch.bfloat16)
    vol = torch.empty(1, 1, *shape)
    gen = torch.Generator().manual_seed(0)
    for z in range(0, shape[0], 32):         # per-slab generation
        vol[0, 0, z:z + 32].normal_(generator=gen)
    Predictor(model, device=dev, apply_softmax=True).predict(torch.randn(1, 1, *[t + 2 * o for t, o in zip(tile, overlap)]))   # warm-up tile
    pred = Predictor(model, device=dev, tile_shape=tile, overlap_shape=overlap, offset=None, out_shape=(2, *shape), apply_softmax=True,
                     strict_shapes=False, tile_parallel=tile_parallel)
    pred.prepare(vol)            # (the page-locked staging slots of the host <-> device pipeline: once per process, not part of a predict() call)
    torch.cuda.synchronize()
    ms0 = torch.cuda.memory_stats(dev)
    with GpuSensors(dev) as sensors:
        t0 = time.perf_counter()
        out = pred.predict(vol)
        dt = time.perf_counter() - t0
    ms1 = torch.cuda.memory_stats(dev)
    # device allocations / frees of torch's caching allocator inside predict() (a hipFree synchronises the device: a retry after a failed allocation would show as a stalled row)
    alloc_delta = {k: int(ms1.get(k, 0) - ms0.get(k, 0)) for k in ('num_device_alloc', 'num_device_free', 'num_alloc_retries', 'num_ooms')}
    ntiles = 1
    for n, t in zip(shape, tile):
        ntiles *= -(-n // t)
    tile_in = [t + 2 * o for t, o in zip(tile, overlap)]
    tile_flop, skipped, executed = needed_region_flops(tile_in, overlap, bricks=((4, 4, 32), (2, 8, 16), (2, 8, 16)) if bf16 else ((4, 4, 16),) * 3, fp32=not bf16)   # forward FLOP per tile incl. halo (2744 GFLOP)
    roi_on = bool(_inf._ROI)
    done_flop = tile_flop - (skipped if roi_on else 0.0)
    if not roi_on:
        executed = needed_region_flops(tile_in, [0, 0, 0], bricks=((4, 4, 16),) * 3, fp32=not bf16)[2]
    _inf._ROI = roi_default
    return {'metric': 'Predictor MVox/s', 'value': vol.numel() / dt / 1e6, 'unit': 'MVox/s (input voxels / predict() wall time incl. H2D + D2H)',
            'seconds': dt, 'volume': list(shape), 'tile': list(tile), 'overlap': list(overlap), 'tiles': ntiles, 'dtype': (bf16 if isinstance(bf16, str) else 'bf16') if bf16 else 'f32',
            'out_dtype': str(out.dtype).replace('torch.', ''),
            'timing': {k: (round(v, 6 if k.startswith('tile_call') else 4) if isinstance(v, float) else v) for k, v in (getattr(pred, 'last_timing', None) or {}).items()},
            'sensors': sensors.summary(), 'allocator': alloc_delta,
            'finite': bool(torch.isfinite(out[..., ::32, ::32].float()).all()),
            'needed_region': roi_on, 'flop_skipped_frac': (skipped / tile_flop) if roi_on else 0.0,
            'algorithmic_tflops': ntiles * tile_flop / dt / 1e12,
            'mfma_executed_frac': ntiles * (done_flop if bf16 else executed) / dt / 1e12 / MFMA_PEAK_TFLOPS['bf16' if bf16 else 'f32'],
            'note': ('needed_region: as in the fp32 leg; executed fraction = matrix FLOP of the direct 16-bit convs actually run / wall time incl. PCIe / dense bf16 MFMA peak' if bf16 else
                     'needed_region: the decoder convs compute only the Winograd bricks that the kept central crop of a tile depends on (same predict() result; '
                     'E3_PREDICTOR_NO_ROI=1 computes whole tiles); algorithmic_tflops counts whole tiles (what the reference computes), executed fraction = '
                     'matrix FLOP actually executed (un-skipped 3x3x3 conv FLOP x 96/432 on the levels that run F(2x2x4) Winograd tiles, x 64/216 on the F(2x2x2) levels, other layers as they are) / wall time incl. PCIe / fp32 MFMA peak')}


def train_leg(dev, kind, steps=10, warmup=3):
    """Secondary training legs of the DEFAULT run (N = 1), so that the driver records BASELINE's other training configurations too:
      'bf16'  configs[2]'s per-GPU workload: the cfg-2 module cast with model.to(torch.bfloat16), bf16 crops, native bf16 kernels;
      'cfg4'  configs[3]'s per-GPU workload: anisotropic UNet (planar_blocks=(0,1), start_filts=64), fp32, batch 2 of 32x256x256;
      'two_call'  cfg 2 in fp32 through the reference's two calls (out = model(inp); loss = criterion(out, target), trainer.py:520-524) instead
                  of UNet.forward_with_loss (the boundary the headline uses).
    Same step as the headline: forward + CE/Dice loss + backward, optimizer excluded, inputs resident, `warmup` untimed + `steps` timed steps."""
    from elektronn3_amd.unet import UNet
    from elektronn3_amd.loss import CombinedCEDiceLoss
    torch.manual_seed(0)
    if kind == 'cfg4':
        model = UNet(in_channels=1, out_channels=2, n_blocks=4, start_filts=64, planar_blocks=(0, 1), normalization='batch').to(dev).train()
        crop, prof_layer = (32, 256, 256), None
    else:
        model = UNet(in_channels=1, out_channels=2, n_blocks=4, start_filts=32, normalization='batch').to(dev).train()
        crop, prof_layer = CROP, 'up_convs.2.conv1'
    x = torch.randn(BATCH_PER_GPU, 1, *crop, device=dev)
    tgt = torch.randint(0, 2, (BATCH_PER_GPU, *crop), device=dev)
    if kind == 'bf16':
        model = model.to(torch.bfloat16)
        x = x.to(torch.bfloat16)
        if not model._plan().bf16_supported():
            return {'value': None, 'note': 'configuration not on the native bf16 path'}
    criterion = CombinedCEDiceLoss(weight=[0.2653, 0.7347]).to(dev)

    def step():
        if kind == 'two_call':
            loss = criterion(model(x), tgt)
        else:
            _, loss = model.forward_with_loss(x, tgt, criterion)
        for p in model.parameters():
            p.grad = None
        loss.backward()

    li = None
    if prof_layer is not None:
        layers = model.conv_layers()
        li = [l[0] for l in layers].index(prof_layer)
        model.profile_select(li, 0)
    for _ in range(warmup):
        step()
    # these legs start after the GPU has idled through the CPU baseline (~30 s): `warmup` steps of 5-12 ms do not bring the clocks back
    # (measured on one box: two_call 12.6 ms / bf16 5.7 ms right after the idle against 11.2 / 5.1), so more UNTIMED steps follow until the
    # warm-up phase has kept the device busy for half a second; the timed region is unchanged (exactly `steps` steps between two synchronisations)
    torch.cuda.synchronize()
    tw, extra_warm = time.perf_counter(), 0
    while time.perf_counter() - tw < 0.5:
        step()
        torch.cuda.synchronize()
        extra_warm += 1
    if li is not None:
        model.profile_select(li, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    vox = BATCH_PER_GPU * crop[0] * crop[1] * crop[2]
    res = {'ms_per_step': dt * 1e3, 'value': vox / dt, 'unit': 'voxels/s', 'steps': steps, 'warmup': warmup, 'clock_ramp_warmup_steps': extra_warm, 'batch': BATCH_PER_GPU, 'crop': list(crop),
           'dtype': 'bf16' if kind == 'bf16' else 'f32', 'max_memory_gib': torch.cuda.max_memory_allocated(dev) / 2 ** 30}
    if li is not None:
        k_ms, k_n = model.profile_read()
        model.profile_select(-1, 0)
        _, lcin, lcout, ltaps, llevel = layers[li]
        lvox = vox // (8 ** llevel)
        lflops = 2.0 * lcin * lcout * ltaps * lvox
        if kind == 'bf16' and k_ms > 0:
            ach = lflops / (k_ms * 1e-3) / 1e12
            res['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': MFMA_PEAK_TFLOPS['bf16'], 'unit': 'TFLOP/s', 'frac': ach / MFMA_PEAK_TFLOPS['bf16'],
                               'kernel': f'conv_b16_pkernel (persistent direct implicit GEMM, v_mfma_f32_32x32x16_bf16) fwd of {prof_layer} ({lcin}->{lcout}, {ltaps} taps, {lvox} voxels)', 'ms_per_launch': k_ms, 'launches_timed': k_n,
                               'hbm_frac': (lvox * (lcin + lcout) + lcin * lcout * ltaps) * 2 / (k_ms * 1e-3) / HBM_PEAK}
    if kind == 'cfg4':
        res['workload'] = 'BASELINE.json configs[3] per-GPU workload: UNet(n_blocks=4, start_filts=64, planar_blocks=(0,1)) fp32 train fwd+bwd, batch 2 of 1x32x256x256'
    elif kind == 'bf16':
        res['workload'] = 'BASELINE.json configs[2] per-GPU workload: the cfg-2 module cast with model.to(torch.bfloat16), bf16 crops, native bf16 kernels'
    else:
        res['workload'] = 'cfg 2 (the headline) through the two separate calls out = model(x); loss = criterion(out, target) instead of UNet.forward_with_loss'
    del model, x, tgt
    torch.cuda.empty_cache()
    return res


def respawn_under_launcher(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: run the same command one process per GPU."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--dtype', choices=('f32', 'bf16', 'f16'), default='f32', help='f32 = BASELINE configs[1] (the metric\'s config); bf16 = configs[2] per-GPU workload; f16 = the same with model.half() (the reference\'s own mixed-precision type)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-only', action='store_true', help='(internal) time the CPU baseline, print its JSON object, exit')
    ap.add_argument('--no-predictor', action='store_true', help='skip the Predictor MVox/s leg')
    ap.add_argument('--no-live-traffic', action='store_true', help='roofline.traffic from the committed PMC file instead of two rocprofv3 --pmc child passes (N = 1 only)')
    ap.add_argument('--predictor-volume', choices=('full', 'sub', 'tiny'), default=None, help='full = 512x2048x2048 (default for N = 1), sub = 288x1152x1152, tiny = 96x384x384 (dry runs)')
    ap.add_argument('--backend', choices=('nccl', 'gloo'), default='nccl', help='process-group backend for N > 1.  nccl = RCCL (the measured configuration).  gloo is TEST-ONLY: it lets the whole '
                    'N > 1 branch (launcher respawn, barriers, MAX-reduced time, global-batch criterion, GradSync, tile-parallel Predictor leg, rank-0 JSON) run where RCCL cannot, '
                    'e.g. with --share-device on a one-GPU box; the line is then marked "test_only"')
    ap.add_argument('--share-device', action='store_true', help='TEST-ONLY: every rank uses cuda:0 (dry run of the N > 1 branch on a one-GPU box; implies nothing about scaling)')
    ap.add_argument('--profile-layer', default='up_convs.2.conv1')
    ap.add_argument('--no-extra-legs', action='store_true', help='skip the cfg-3 (bf16) / cfg-4 / two-call training legs of the default f32 run')
    ap.add_argument('--dp-overlap', action='store_true', help='N > 1: all-reduce bucket A at the bucket event, overlapped with the rest of the backward '
                    '(GradSync(overlap=True): 16 CUs reserved after the event); default: one all-reduce behind the backward')
    args = ap.parse_args()

    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()), flush=True)
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        respawn_under_launcher(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); pass the same N to both')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU (the HIP path has no CPU fallback)')
    if args.share_device and args.backend == 'nccl' and world > 1:
        raise SystemExit('bench.py: --share-device needs --backend gloo (RCCL wants one GPU per rank)')
    dev_index = 0 if args.share_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    dist = None
    if world > 1 or 'RANK' in os.environ:   # launched by torch.distributed.run (also with a single rank)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.dp_overlap or os.environ.get('E3_DP_OVERLAP') is not None:
            os.environ.setdefault('NCCL_MAX_NCHANNELS', os.environ.get('E3_DP_CU_RESERVE', '16'))     # RCCL's kernel: at most as many workgroups as CUs are reserved
        # RCCL's kernels must get compute units while the (512-register, one-workgroup-per-CU) conv kernels of the backward own the
        # chip: run the collectives on a high-priority stream so that they are dispatched first whenever a CU frees up
        pg_opts = None
        try:
            pg_opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        except Exception:  # noqa: BLE001
            pg_opts = None
        if args.backend == 'gloo':
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev, **({'pg_options': pg_opts} if pg_opts is not None else {}))

    from elektronn3_amd.unet import UNet
    from elektronn3_amd.loss import CombinedCEDiceLoss   # the example's criterion (0.5 CE + 0.5 Dice, class weights) on device

    bf16 = args.dtype in ('bf16', 'f16')
    t16 = torch.float16 if args.dtype == 'f16' else torch.bfloat16
    torch.manual_seed(0)                                   # identical replica on every rank
    model = UNet(in_channels=1, out_channels=2, n_blocks=4, start_filts=32, normalization='batch').to(dev).train()
    if bf16:
        model = model.to(t16)                              # BASELINE configs[2]: "same UNet bf16" (whole-module cast); f16: model.half()
        if not model._plan().bf16_supported():
            raise SystemExit('bench.py --dtype bf16: configuration not on the native bf16 path')
    # N > 1: ONE loss over the global minibatch, as the reference computes on the batch nn.DataParallel gathers (trainer.py:520-524):
    # the ranks exchange the criterion's 2 + 3C sums (one all-reduce of 8 doubles) between forward and backward
    criterion = CombinedCEDiceLoss(weight=[0.2653, 0.7347], global_batch=dist is not None).to(dev)
    sync = None
    if world > 1 or (dist is not None):
        from elektronn3_amd.dataparallel import GradSync
        sync = GradSync(model, overlap=True if args.dp_overlap else None)
    torch.manual_seed(1000 + rank)                         # different synthetic crops per rank
    x = torch.randn(BATCH_PER_GPU, 1, *CROP, device=dev)
    if bf16:
        x = x.to(t16)
    tgt = torch.randint(0, 2, (BATCH_PER_GPU, *CROP), device=dev)

    layers = model.conv_layers()
    names = [l[0] for l in layers]
    li = names.index(args.profile_layer)
    _, lcin, lcout, ltaps, llevel = layers[li]

    def step():
        # the two lines of the reference's training step (out = model(inp); loss = criterion(out, target), trainer.py:520-524) through the boundary
        # that lets the 1x1x1 head evaluate the criterion (UNet.forward_with_loss; it makes exactly those two calls where that does not apply:
        # N > 1, whose criterion exchanges its sums between the ranks first)
        out, loss = model.forward_with_loss(x, tgt, criterion)
        for p in model.parameters():
            p.grad = None
        loss.backward()
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
    with GpuSensors(dev) as step_sensors:
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

    res = None
    if rank == 0:
        vox_per_step = world * BATCH_PER_GPU * CROP[0] * CROP[1] * CROP[2]
        lvox = BATCH_PER_GPU * CROP[0] * CROP[1] * CROP[2] // (8 ** llevel)
        lflops = 2.0 * lcin * lcout * ltaps * lvox                         # algorithmic (direct-convolution) count, SURVEY 8d
        esz = 2 if bf16 else 4
        lbytes = (lvox * (lcin + lcout) + lcin * lcout * ltaps) * esz + lcout * 4   # x + y + w (+ b), SURVEY 8d's per-layer bytes
        wino = (not bf16) and ltaps == 27                                  # fp32 3x3x3: Winograd F(2x2x2,3x3x3) executes 64/216 of the multiplies
        exec_flops = lflops * (64.0 / 216.0 if wino else 1.0)
        sec = k_ms * 1e-3
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        ach = exec_flops / sec / 1e12 if sec > 0 else 0.0
        traffic, tsrc = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, PMC_FILE)))
            ent = pmc.get(args.dtype, {}).get(args.profile_layer)
            if ent:
                traffic, tsrc = float(ent['hbm_bytes_per_launch']), PMC_FILE
        except Exception:  # noqa: BLE001
            pass
        # (the plain default run only: the developer flags of the A/B and profiling scripts -- which may themselves run under rocprofv3 -- keep the committed file)
        if (world == 1 and dist is None and not args.no_live_traffic and not args.no_cpu_baseline and not args.no_extra_legs and args.profile_layer == 'up_convs.2.conv1'
                and os.environ.get('E3_BENCH_NO_LIVE_TRAFFIC') is None):
            lt, why = live_traffic(args.dtype, 'conv_b16_pkernel' if bf16 else 'conv3_wino_pkernel')
            if lt is not None:
                traffic, tsrc = float(lt), why
            else:
                tsrc = f'{tsrc} (live PMC passes unavailable: {why})'
        kern = ('conv_b16_pkernel (persistent direct implicit GEMM, v_mfma_f32_32x32x16_bf16)' if bf16 else
                ('conv3_wino_pkernel (persistent Winograd F(2x2x2,3x3x3), v_mfma_f32_32x32x2_f32)' if wino else 'conv3_v3_kernel (direct, fp32 MFMA)'))
        res = {
            'metric': 'voxels/sec (train fwd+bwd) 3D UNet 64x128x128',
            'value': vox_per_step / (ms_per_step * 1e-3),
            'unit': 'voxels/s',
            'n_gpus': world, 'rccl_ranks': (dist.get_world_size() if (dist is not None and args.backend == 'nccl') else 0),
            'backend': (args.backend if dist is not None else None),
            **({'test_only': f'dry run of the N > 1 branch: backend {args.backend}, ' + ('all ranks on cuda:0' if args.share_device else 'one GPU per rank') + ' -- not a measurement',
                'dist_ranks': dist.get_world_size()} if (dist is not None and (args.backend != 'nccl' or args.share_device)) else {}),
            'dp_mode': (sync.mode if sync is not None else None), 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic', 'sensors': step_sensors.summary(),
            'config': {'workload': (f'BASELINE.json configs[{2 if bf16 else 1}]: UNet(in=1,out=2,n_blocks=4,start_filts=32,bn) '
                                    f'{("float16 (model.half(), float16 crops, native f16 kernels)" if args.dtype == "f16" else "bf16 (model.to(bfloat16), bf16 crops, native bf16 kernels)") if bf16 else "fp32"} train fwd+bwd, '
                                    f'batch {BATCH_PER_GPU}/GPU of 1x64x128x128 random crops, CE+Dice loss, optimizer excluded'),
                       'global_batch': world * BATCH_PER_GPU, 'crop': list(CROP),
                       'parallelism': f'dp{world}' if world > 1 else 'single',
                       'step_tflops': FWDBWD_FLOP_PER_VOXEL * vox_per_step / world / (ms_per_step * 1e-3) / 1e12},
            'roofline': {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
                         'traffic': traffic, 'traffic_source': tsrc,
                         'kernel': f'{kern} fwd of {args.profile_layer} ({lcin}->{lcout}, {ltaps} taps, {lvox} voxels)',
                         'executed_flops_per_launch': exec_flops, 'algorithmic_flops_per_launch': lflops,
                         'algorithmic_tflops': lflops / sec / 1e12 if sec > 0 else 0.0,
                         'algorithmic_bytes_per_launch': lbytes, 'hbm_frac': lbytes / sec / HBM_PEAK if sec > 0 else 0.0,
                         'ms_per_launch': k_ms, 'launches_timed': k_n,
                         'dgrad_ms': extra.get('dgrad'), 'wgrad_ms': extra.get('wgrad'),
                         'dgrad_algorithmic_tflops': lflops / (extra['dgrad'] * 1e-3) / 1e12 if extra.get('dgrad') else None,
                         'wgrad_algorithmic_tflops': lflops / (extra['wgrad'] * 1e-3) / 1e12 if extra.get('wgrad') else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                # in a CHILD process: the 128-thread CPU run leaves thread pools and allocator state behind that slow the host side of the Predictor leg
                # further down (measured: 398 -> 365 MVox/s on the 512x2048x2048 volume when it ran in this process)
                import subprocess
                r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-only'], capture_output=True, text=True, timeout=900,
                                   env={**os.environ, 'HIP_VISIBLE_DEVICES': '', 'CUDA_VISIBLE_DEVICES': ''})
                line = [l for l in r.stdout.splitlines() if l.startswith('{')]
                if r.returncode != 0 or not line:
                    raise RuntimeError((r.stderr or r.stdout)[-300:])
                res['cpu_baseline'] = json.loads(line[-1])
            except Exception as e:  # noqa: BLE001
                res['cpu_baseline'] = {'value': None, 'unit': 'voxels/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                                       'sample': f'failed: {e}'}

    # ---- BASELINE's other training configurations and the two-call boundary (N = 1, default dtype only; ~0.5 s of GPU time)
    if world == 1 and dist is None and args.dtype == 'f32' and not args.no_extra_legs and os.environ.get('E3_BENCH_NO_EXTRA') is None:
        del model, x, tgt
        x = tgt = None
        torch.cuda.empty_cache()
        for key, kind in (('two_call', 'two_call'), ('cfg3_bf16', 'bf16'), ('cfg4', 'cfg4')):
            try:
                res[key] = train_leg(dev, kind)
            except Exception as e:  # noqa: BLE001
                res[key] = {'value': None, 'note': f'failed: {e}'}

    # ---- Predictor leg (all ranks take part when N > 1)
    done = threading.Event()

    def emit():
        if rank == 0 and not done.is_set():
            done.set()
            print(json.dumps(res), flush=True)

    if not args.no_predictor and os.environ.get('E3_BENCH_NO_PREDICTOR') is None:
        x = tgt = None
        torch.cuda.empty_cache()
        vol_kind = args.predictor_volume or ('full' if world == 1 else 'sub')
        shape = {'full': (512, 2048, 2048), 'sub': (288, 1152, 1152), 'tiny': (96, 384, 384)}[vol_kind]
        # the tile-parallel leg has a control-plane exchange (shared-memory name, closing barrier): a rank that dies in it must not
        # take the training line down with it -- after 300 s rank 0 prints what it has and every rank leaves
        watchdog = None
        if world > 1:
            def bail():
                if rank == 0 and res is not None:
                    res['predictor'] = {'metric': 'Predictor MVox/s', 'value': None, 'note': 'tile-parallel leg did not finish within 300 s'}
                emit()
                os._exit(0)
            watchdog = threading.Timer(300.0, bail)
            watchdog.daemon = True
            watchdog.start()
        try:
            b16_leg = args.dtype if args.dtype in ('bf16', 'f16') else False
            p = predictor_leg(dev, shape, tile_parallel=world > 1, bf16=b16_leg)
            if rank == 0:
                if world > 1:
                    p.update(n_gpus=world, parallelism=f'tile-parallel over {world} ranks, shared-memory output')
                res['predictor'] = p
            if world == 1 and p.get('needed_region'):      # the same Predictor on a 288x1152x1152 volume with whole tiles and with the needed region
                sub = (288, 1152, 1152)
                b16 = args.dtype if args.dtype in ('bf16', 'f16') else False
                a = predictor_leg(dev, sub, whole_tiles=True, bf16=b16)
                b = p if tuple(shape) == sub else predictor_leg(dev, sub, bf16=b16)
                p['needed_region_ab'] = {'volume': list(sub), 'whole_tiles_mvox_s': a['value'], 'needed_region_mvox_s': b['value'],
                                         'whole_tiles_compute_s': a['timing'].get('compute_stream_s'), 'needed_region_compute_s': b['timing'].get('compute_stream_s')}
            if world == 1 and not b16_leg and not args.no_extra_legs:
                # the reference's own reduced-precision inference switch: Predictor(float16=True) = model.half() + float16 tiles (inference.py:445-446,
                # benchmark/pred_benchmark.py:55,71), on the native float16 kernels; same tiling on the 288x1152x1152 sub-volume
                try:
                    h = predictor_leg(dev, (288, 1152, 1152), bf16='f16')
                    res['predictor_f16'] = {k: h[k] for k in ('metric', 'value', 'unit', 'seconds', 'volume', 'tile', 'overlap', 'tiles', 'dtype', 'out_dtype', 'timing', 'finite',
                                                              'needed_region', 'mfma_executed_frac')}
                    res['predictor_f16']['workload'] = 'configs[4] tiling on a 288x1152x1152 volume with Predictor(float16=True) semantics: model.half(), float16 tiles, native float16 kernels'
                except Exception as e:  # noqa: BLE001
                    res['predictor_f16'] = {'value': None, 'note': f'failed: {e}'}
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                res['predictor'] = {'metric': 'Predictor MVox/s', 'value': None, 'note': f'failed: {e}'}
        if watchdog is not None:
            watchdog.cancel()
    emit()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
