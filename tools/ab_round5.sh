#!/bin/bash
# same-box A/B of round 5's default kernels against the same library with this round's changes switched off
# (E3_NO_CHUNKED=1: the data gradients' input in [voxel][C] rows; E3_WINO4=0: data gradients / inference forwards on the F(2x2x2) persistent kernel; E3_NO_BNRED_FUSE=1: the BatchNorm backward's REDUCE pass as its own kernel)
B="python bench.py --no-cpu-baseline --no-predictor --no-extra-legs"
J='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["ms_per_step"], 3), "ms", round(d["value"] / 1e6, 1), "MVox/s")'
for i in 1 2 3; do
  $B 2>/dev/null | python -c "$J" "round-5 defaults              "
  E3_NO_CHUNKED=1 $B 2>/dev/null | python -c "$J" "  dZ in rows                  "
  E3_NO_CHUNKED=1 E3_NO_BNRED_FUSE=1 $B 2>/dev/null | python -c "$J" "  + REDUCE pass on its own    "
  E3_WINO4=0 E3_NO_BNRED_FUSE=1 $B 2>/dev/null | python -c "$J" "  + F(2x2x2) data gradients   "
done
