#!/bin/bash
# kernel tables of the attention / ResUNet variants of the headline step (rocprofv3 --kernel-trace): bash tools/prof_variants.sh  ->  gpurun_out/pv_<variant>/summary.md
R=$PWD; export TMPDIR=/tmp
for v in 1 3; do
  O=$R/gpurun_out/pv_$v; mkdir -p $O; cd /tmp
  rocprofv3 --kernel-trace --stats -d $O/prof -o run -- python $R/tools/bench_variants.py --only $v --steps 5 > $O/prof.log 2>&1
  cd $R
  python tools/prof_summary.py $O/prof -o $O/summary.md > /dev/null
  echo "== variant $v"; head -30 $O/summary.md | cut -c1-100
  find $O -name "*.db" -delete
done
