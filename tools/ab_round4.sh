#!/bin/bash
# same-box A/B of round 4's default kernels against the same library with this round's changes switched off
# (E3_NO_LOSS_BWD=1: criterion backward through e3_ce_dice_bwd + conv_final_bwd; E3_WINO_NO_TR=1: dword stores in the persistent Winograd kernel)
B="python bench.py --no-cpu-baseline --no-predictor --no-extra-legs"
J='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["ms_per_step"], 3), "ms", round(d["value"] / 1e6, 1), "MVox/s")'
for i in 1 2 3; do
  $B 2>/dev/null | python -c "$J" "round-4 defaults      "
  E3_NO_LOSS_BWD=1 E3_WINO_NO_TR=1 $B 2>/dev/null | python -c "$J" "round-4 changes off   "
done
E3_BNRED_FUSE=1 $B 2>/dev/null | python -c "$J" "+ E3_BNRED_FUSE=1      "
E3_WINO16=1 $B 2>/dev/null | python -c "$J" "+ E3_WINO16=1          "
