#!/bin/bash
# same-box A/B of two builds of the library: tools/_bin/libe3unet_base.so (kept from before a change) against the in-tree build.
# $1 = rounds (default 2).  Prints a result hash of a train step + eval forward (must agree when the change is layout / scheduling only), the step,
# the cfg-5 tile and (with W4=1) the per-layer table of the F(2x2x4) kernel.
R=${1:-2}; B=$PWD/tools/_bin/libe3unet_base.so
echo "== hash base"; E3_LIB_PATH=$B python tools/ab_hash.py | grep HASH
echo "== hash new";  python tools/ab_hash.py | grep HASH
for i in $(seq 1 $R); do
  for which in base new; do
    if [ $which = base ]; then export E3_LIB_PATH=$B; else unset E3_LIB_PATH; fi
    s=$(python bench.py --no-cpu-baseline --no-extra-legs --no-predictor --steps 20 --warmup 20 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    t=$(python tools/bench_tile.py 100 | head -1)
    echo "round $i $which: step $s ms; $t"
  done
done
unset E3_LIB_PATH
