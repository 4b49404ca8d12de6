#!/bin/bash
# copy the evidence of tools/final_round6.sh (gpurun_out/final_r06, or $1) into profiles/r06_*
S=${1:-gpurun_out/final_r06}
for f in bench_default.json bench_bf16.json bench_f16.json kernel_stats_f32.md kernel_stats_bf16.md kernel_stats_tile.md kernel_stats_predictor.md per_layer_f32.md per_layer_bf16.md \
         pmc_f32.md pmc_bf16.md pmc_roofline.json sq_counters.md; do
  [ -s $S/$f ] && cp $S/$f profiles/r06_$f || echo "missing $S/$f"
done
