#!/bin/bash
# timing of the E3_PABL builds of conv_b16_pkernel (tools/build_b16p_variants.sh): bf16 step and the 64->32 level-0 forward launch
cd ${GRAFT_REPO_ROOT:-.}
B="python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-predictor --no-extra-legs"
J='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["ms_per_step"],3), round(1e3*d.get("roofline",{}).get("ms_per_launch"),1))'
$B 2>/dev/null | python -c "$J" tree
for m in "$@"; do
  E3_LIB_PATH=$PWD/tools/_bin/libe3unet_pabl$m.so $B 2>/dev/null | python -c "$J" abl$m
done
E3_LIB_PATH=$PWD/tools/_bin/libe3unet_base.so $B 2>/dev/null | python -c "$J" base
