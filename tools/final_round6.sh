#!/bin/bash
# the round's final evidence in one gpurun call: bench lines, kernel tables, PMC passes, per-layer tables, SQ counters, tile / Predictor tables
O=gpurun_out/final_r06; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err
python bench.py --dtype f16 --no-cpu-baseline > $O/bench_f16.json 2> $O/bench_f16.err
bash tools/profile_round.sh r06k > $O/profile_round.log 2>&1
python tools/layer_table.py 3 > $O/per_layer_f32.md 2> $O/per_layer_f32.err
python tools/layer_table.py 3 bf16 > $O/per_layer_bf16.md 2> $O/per_layer_bf16.err
bash tools/prof_tile.sh r06k_tile > $O/prof_tile.log 2>&1
bash tools/prof_predictor.sh r06k_pred > $O/prof_pred.log 2>&1
bash tools/pmc_sq.sh r06k f32 . > $O/sq_counters.md 2> $O/sq.err
for f in kernel_stats_f32.md kernel_stats_bf16.md pmc_roofline.json pmc_f32.md pmc_bf16.md; do cp gpurun_out/prof_r06k/$f $O/ 2>/dev/null; done
cp gpurun_out/prof_r06k_tile/kernel_stats.md $O/kernel_stats_tile.md; cp gpurun_out/prof_r06k_pred/kernel_stats.md $O/kernel_stats_predictor.md
rm -rf gpurun_out/prof_r06k gpurun_out/prof_r06k_tile gpurun_out/prof_r06k_pred gpurun_out/sq_r06k
ls -la $O; head -c 300 $O/bench_default.json
