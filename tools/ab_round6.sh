#!/bin/bash
# same-box comparison of round 5's library (tools/_bin/libe3unet_r05.so, built from commit 95702ce) with the in-tree build: fp32 step, bf16 step, cfg-5 tile,
# and the Predictor leg on the 288x1152x1152 sub-volume; $1 = rounds
R=${1:-2}; B=$PWD/tools/_bin/libe3unet_r05.so
for i in $(seq 1 $R); do
  for which in r05 r06; do
    if [ $which = r05 ]; then export E3_LIB_PATH=$B; else unset E3_LIB_PATH; fi
    s=$(python bench.py --no-cpu-baseline --no-extra-legs --no-predictor --steps 20 --warmup 20 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    b=$(python bench.py --dtype bf16 --no-cpu-baseline --no-extra-legs --no-predictor --steps 20 --warmup 20 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    t=$(python tools/bench_tile.py 100 2>/dev/null | head -1)
    p=$(python bench.py --no-cpu-baseline --no-extra-legs --predictor-volume sub --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['predictor']; print('%.1f MVox/s (%.3f s)' % (d['value'], d['seconds']))")
    echo "round $i $which: fp32 step $s ms; bf16 step $b ms; $t; Predictor 288x1152x1152 $p"
  done
done
