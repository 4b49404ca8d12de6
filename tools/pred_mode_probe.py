"""A slow Predictor run (profiles/r06_predictor_modes.md): bench.py's sequence in one process -- training steps, the Predictor leg on the full cfg-5 volume -- and then
the tile forward ALONE (tools/bench_tile.py's loop, no copies beside it) on the same module and scratch: is the tile slow too in a process whose predict() was slow?
    python tools/pred_mode_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device('cuda:0')
leg = bench.train_leg(dev, 'two_call', steps=10, warmup=3)
torch.cuda.empty_cache()
from elektronn3_amd import inference as inf  # noqa: E402
keep = {}
orig = inf.Predictor.predict


def spy(self, x):
    keep['pred'] = self
    return orig(self, x)


inf.Predictor.predict = spy
p = bench.predictor_leg(dev, (512, 2048, 2048))
m = keep['pred'].model
t = p['timing']
vol = torch.randn(1, 1, 128, 224, 224 * 3, device=dev)
out = torch.zeros(1, 2, 96, 192, 192 * 3, device=dev)
roi = [(16, 112), (16, 208), (16, 208)]
res = []
with torch.no_grad(), m.frozen_weights():
    for rep in range(2):
        for i in range(3):
            m.forward_tile(vol, (0, 0, 224 * (i % 3)), (128, 224, 224), out, (0, 0, 192 * (i % 3)), roi, softmax=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(60):
            m.forward_tile(vol, (0, 0, 224 * (i % 3)), (128, 224, 224), out, (0, 0, 192 * (i % 3)), roi, softmax=True)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 60 * 1e3)
s = p.get('sensors') or {}
print(f'train step {leg["ms_per_step"]:.3f} ms; Predictor {p["value"]:.1f} MVox/s, rows {t["rows_s"]:.3f} s, upload worker {t["upload_worker_s"]:.2f} s, power mean '
      f'{(s.get("socket_power_w_min_mean_max") or [0, 0, 0])[1]} W; tile alone afterwards {res[0]:.3f} / {res[1]:.3f} ms', flush=True)
