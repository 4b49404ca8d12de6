#!/bin/bash
# kernel statistics of the Predictor leg on the 288x1152x1152 sub-volume (fp32): $1 = tag
R=$PWD; TAG=${1:-pred}; O=$R/gpurun_out/prof_$TAG; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o run -- python $R/bench.py --no-cpu-baseline --no-extra-legs --steps 1 --warmup 1 --predictor-volume sub > $O/stats.log 2>&1
cd $R
python tools/prof_summary.py $O/stats -o $O/kernel_stats.md --title "Predictor leg, 288x1152x1152, fp32 ($TAG)" > /dev/null
head -34 $O/kernel_stats.md | cut -c1-130
