#!/bin/bash
# fp32 step time and cfg-5 tile time under forced block shapes of the logical brick order (E3_WINO_BLOCK, brick_order.h), interleaved rounds; $1 = rounds, rest = shapes
R=$1; shift
for i in $(seq 1 $R); do
  for sh in "$@"; do
    if [ $sh = auto ]; then unset E3_WINO_BLOCK; else export E3_WINO_BLOCK=$sh; fi
    s=$(python bench.py --no-cpu-baseline --no-extra-legs --no-predictor --steps 30 --warmup 20 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    t=$(python tools/bench_tile.py 100 2>/dev/null | head -1 | sed 's/tile forward alone: //; s/ per tile.*//')
    echo "round $i $sh: step $s ms; tile $t"
  done
done
