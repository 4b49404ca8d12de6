#!/bin/bash
# same-box A/B of the bf16 step: the tree's library against tools/_bin/libe3unet_base.so (E3_LIB_PATH), alternating runs
cd ${GRAFT_REPO_ROOT:-.}
B="python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-predictor --no-extra-legs"
J='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["ms_per_step"],3), d.get("roofline",{}).get("ms_per_launch"))'
for i in 1 2 3; do
  E3_LIB_PATH=$PWD/tools/_bin/libe3unet_base.so $B 2>/dev/null | python -c "$J" base
  $B 2>/dev/null | python -c "$J" new
done
