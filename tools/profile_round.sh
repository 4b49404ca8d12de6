#!/bin/bash
# Round profile on the GPU box: kernel statistics of the default bench command (fp32 and bf16) and the PMC passes behind
# bench.py's roofline.traffic.  Usage (inside gpurun): bash tools/profile_round.sh r03
set -u
TAG=${1:-r05}
R=$PWD
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-predictor --no-extra-legs --steps 5 --warmup 2"
for dt in f32 bf16; do
  rocprofv3 --kernel-trace --stats -d $O/stats_$dt -o run -- $BENCH --dtype $dt > $O/stats_$dt.log 2>&1
  python $R/tools/prof_summary.py $O/stats_$dt -o $O/kernel_stats_$dt.md --title "rocprofv3 --kernel-trace --stats of: bench.py --dtype $dt --steps 5 --warmup 2 (+ 2 x 2 per-layer timing steps)" > /dev/null
done
PB="python $R/bench.py --no-cpu-baseline --no-predictor --no-extra-legs --steps 1 --warmup 1"
for dt in f32 bf16; do
  for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
    rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${dt}_$c -o run --output-format csv -- $PB --dtype $dt > $O/pmc_${dt}_$c.log 2>&1
  done
done
cd $R
rm -f $O/pmc_roofline.json
python tools/pmc_roofline.py --dtype f32 --kernel conv3_wino_pkernel --fetch $O/pmc_f32_FETCH_SIZE --write $O/pmc_f32_WRITE_SIZE --busy $O/pmc_f32_SQ_VALU_MFMA_BUSY_CYCLES --steps 6 -o $O/pmc_roofline.json --command "$PB --dtype f32" > $O/pmc_roofline_f32.log 2>&1
python tools/pmc_roofline.py --dtype bf16 --kernel "conv_b16_pkernel" --fetch $O/pmc_bf16_FETCH_SIZE --write $O/pmc_bf16_WRITE_SIZE --busy $O/pmc_bf16_SQ_VALU_MFMA_BUSY_CYCLES --steps 6 -o $O/pmc_roofline.json --command "$PB --dtype bf16" > $O/pmc_roofline_bf16.log 2>&1
python tools/pmc_summary.py $O/pmc_bf16_FETCH_SIZE $O/pmc_bf16_WRITE_SIZE $O/pmc_bf16_SQ_VALU_MFMA_BUSY_CYCLES --filter "b16|bn_|wgrad_reduce" -o $O/pmc_bf16.md > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_f32_FETCH_SIZE $O/pmc_f32_WRITE_SIZE $O/pmc_f32_SQ_VALU_MFMA_BUSY_CYCLES -o $O/pmc_f32.md > /dev/null 2>&1
# keep the merged output small: drop the raw databases / csv
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
ls -la $O | head -40
cat $O/pmc_roofline.json | head -60
