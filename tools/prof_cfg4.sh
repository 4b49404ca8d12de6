#!/bin/bash
# kernel statistics of a cfg-4 training step (BASELINE configs[3]: planar_blocks=(0,1), start_filts=64, batch 2 of 32x256x256): $1 = tag, $2 = mode (f32)
R=$PWD; TAG=${1:-cfg4}; O=$R/gpurun_out/prof_$TAG; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o run -- python $R/tools/bench_cfg4.py ${2:-f32} > $O/stats.log 2>&1
cd $R
python tools/prof_summary.py $O/stats -o $O/kernel_stats.md --title "cfg 4 training step (13 steps in the trace), ${2:-f32} ($TAG)" > /dev/null
grep "ms/step" $O/stats.log
head -40 $O/kernel_stats.md | cut -c1-150
find $O -name "*.db" -delete
