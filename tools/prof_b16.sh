#!/bin/bash
# kernel table of the bf16 bench step (rocprofv3 --kernel-trace): bash tools/prof_b16.sh <tag> [grep pattern]
R=$PWD; O=$R/gpurun_out/pb_$1; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python $R/bench.py --dtype ${3:-bf16} --no-predictor --no-cpu-baseline --steps 5 --warmup 2 > $O/prof.log 2>&1
cd $R
python tools/prof_summary.py $O/prof -o $O/summary.md > /dev/null
head -3 $O/summary.md | tail -1 | cut -c1-120
grep -E "${2:-.}" $O/summary.md | cut -c1-110 | head -40
grep -o "ms_per_step\": [0-9.]*" $O/prof.log
find $O -name "*.db" -delete
