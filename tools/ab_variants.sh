#!/bin/bash
# same-box comparison of library variants: tools/_bin/libe3unet_<name>.so for every name in $2.. against the in-tree build ("tree"); $1 = rounds
R=$1; shift
echo "hash tree: $(python tools/ab_hash.py 2>/dev/null | grep HASH)"
for n in "$@"; do echo "hash $n: $(E3_LIB_PATH=$PWD/tools/_bin/libe3unet_$n.so python tools/ab_hash.py 2>/dev/null | grep HASH)"; done
for i in $(seq 1 $R); do
  for n in tree "$@"; do
    if [ $n = tree ]; then unset E3_LIB_PATH; else export E3_LIB_PATH=$PWD/tools/_bin/libe3unet_$n.so; fi
    s=$(python bench.py ${E3_AB_BENCH_ARGS:-} --no-cpu-baseline --no-extra-legs --no-predictor --steps 20 --warmup 20 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    t=$(python tools/bench_tile.py 100 2>/dev/null | head -1)
    echo "round $i $n: step $s ms; $t"
  done
done
unset E3_LIB_PATH
