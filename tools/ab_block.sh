#!/bin/bash
# same-box A/B of the logical brick order of the persistent Winograd kernels (brick_order.h): E3_WINO_BLOCK=0 (plain order) against the default (compact blocks),
# same library.  Result hash, fp32 step, cfg-5 tile, and the HBM reads (FETCH_SIZE pass) of the two kernels.  $1 = rounds
R=${1:-2}; ROOT=$PWD; O=$ROOT/gpurun_out/ab_block; mkdir -p $O
for which in plain block; do
  if [ $which = plain ]; then export E3_WINO_BLOCK=0; else unset E3_WINO_BLOCK; fi
  echo "== hash $which: $(python tools/ab_hash.py 2>/dev/null | tail -1)"
done
for i in $(seq 1 $R); do
  for which in plain block; do
    if [ $which = plain ]; then export E3_WINO_BLOCK=0; else unset E3_WINO_BLOCK; fi
    s=$(python bench.py --no-cpu-baseline --no-extra-legs --no-predictor --steps 20 --warmup 20 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    t=$(python tools/bench_tile.py 100 2>/dev/null | head -1)
    echo "round $i $which: fp32 step $s ms; $t"
  done
done
export TMPDIR=/tmp; cd /tmp
PB="python $ROOT/bench.py --no-cpu-baseline --no-predictor --no-extra-legs --steps 1 --warmup 1"
for which in plain block; do
  if [ $which = plain ]; then export E3_WINO_BLOCK=0; else unset E3_WINO_BLOCK; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${which}_$c -o run --output-format csv -- $PB > $O/pmc_${which}_$c.log 2>&1
  done
  (cd $ROOT; python tools/pmc_summary.py $O/pmc_${which}_FETCH_SIZE $O/pmc_${which}_WRITE_SIZE --filter "conv3_wino" -o $O/pmc_$which.md > /dev/null 2>&1; echo "== $which"; cat $O/pmc_$which.md)
done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
