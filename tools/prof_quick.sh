#!/bin/bash
# kernel statistics of a short default bench run: $1 = tag, rest = env assignments; prints the rows matching $FILTER
R=$PWD; TAG=$1; shift
O=$R/gpurun_out/prof_$TAG; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
env "$@" rocprofv3 --kernel-trace --stats -d $O/stats -o run -- python $R/bench.py --no-cpu-baseline --no-predictor --no-extra-legs --steps 5 --warmup 2 > $O/stats.log 2>&1
cd $R
python tools/prof_summary.py $O/stats -o $O/kernel_stats.md --title "$TAG" > /dev/null
grep -E "${FILTER:-.}" $O/kernel_stats.md | cut -c1-150
