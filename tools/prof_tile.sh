#!/bin/bash
# kernel statistics of the cfg-5 Predictor tile alone (tools/bench_tile.py: 3 warm-up + N tiles, no copies): $1 = tag
R=$PWD; TAG=${1:-tile}; O=$R/gpurun_out/prof_$TAG; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o run -- python $R/tools/bench_tile.py 47 > $O/stats.log 2>&1
cd $R
python tools/prof_summary.py $O/stats -o $O/kernel_stats.md --title "cfg-5 Predictor tile alone, 50 tiles + the running-statistics warm-up ($TAG)" > /dev/null
head -30 $O/kernel_stats.md | cut -c1-150
find $O -name "*.db" -delete
