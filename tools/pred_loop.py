"""The Predictor leg of bench.py (cfg 5: 512x2048x2048 volume, tile 96x192x192, overlap 16) N times in ONE process, one line per predict() with the pipeline's
timing and the GPU's sensors: does a slow run (profiles/r06_predictor_modes.md) belong to the process or to the moment?   python tools/pred_loop.py [N]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from elektronn3_amd.inference import Predictor  # noqa: E402
from elektronn3_amd.unet import UNet  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device('cuda:0')
shape, tile, overlap = (512, 2048, 2048), (96, 192, 192), (16, 16, 16)
torch.manual_seed(0)
model = UNet(1, 2, n_blocks=4, start_filts=32).to(dev)
model.train()
with torch.no_grad():
    for _ in range(10):
        model(torch.randn(2, 1, 32, 64, 64, device=dev))
vol = torch.empty(1, 1, *shape)
gen = torch.Generator().manual_seed(0)
for z in range(0, shape[0], 32):
    vol[0, 0, z:z + 32].normal_(generator=gen)
pred = Predictor(model, device=dev, tile_shape=tile, overlap_shape=overlap, offset=None, out_shape=(2, *shape), apply_softmax=True, strict_shapes=False)
pred.prepare(vol)
for i in range(n):
    torch.cuda.synchronize()
    with bench.GpuSensors(dev) as sensors:
        t0 = time.perf_counter()
        out = pred.predict(vol)
        dt = time.perf_counter() - t0
    t = pred.last_timing
    s = sensors.summary() or {}
    print(f'predict {i}: {vol.numel() / dt / 1e6:.1f} MVox/s ({dt:.3f} s); rows {t["rows_s"]:.3f} s, compute stream {t["compute_stream_s"]:.3f} s, upload worker {t["upload_worker_s"]:.2f} s, '
          f'download worker {t["download_worker_s"]:.2f} s; sclk mean {s.get("sclk_mhz_min_mean_max", [0, 0, 0])[1]} MHz, power mean {s.get("socket_power_w_min_mean_max", [0, 0, 0])[1]} W, '
          f'busy neighbours {s.get("other_gpus_of_the_node_busy_mean")}', flush=True)
    del out
