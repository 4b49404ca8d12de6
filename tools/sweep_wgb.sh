#!/bin/bash
# bf16 weight-gradient split policy sweep (E3_WGB_WGS = workgroup target, E3_WGB_MINB = minimum bricks per split): per-layer op times and the whole step
cd ${GRAFT_REPO_ROOT:-.}
B="python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-predictor --no-extra-legs"
J='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["ms_per_step"],3))'
for cfg in "512 1" "512 4" "512 8" "512 16" "256 1" "256 8" "384 1"; do
  set -- $cfg
  echo "== WGS=$1 MINB=$2"
  E3_WGB_WGS=$1 E3_WGB_MINB=$2 python tools/bench_conv.py --dtype bf16 --what wgrad --iters 20 2>&1 | grep -v "^total"
  E3_WGB_WGS=$1 E3_WGB_MINB=$2 $B 2>/dev/null | python -c "$J" "step WGS=$1 MINB=$2"
  E3_WGB_WGS=$1 E3_WGB_MINB=$2 $B 2>/dev/null | python -c "$J" "step WGS=$1 MINB=$2"
done
