#!/bin/bash
# runs tools/w4_check.py's timing on the ablation builds of tools/build_w4_variants.sh: $@ = masks
cd "$(dirname "$0")/.."
export E3_WINO4=2
for m in "$@"; do
  echo "== abl $m"
  if [ "$m" = "0" ]; then timeout 200 python tools/w4_check.py bench1 2>/dev/null | grep -v amdgpu
  else E3_LIB_PATH=tools/_bin/libe3unet_w4abl$m.so timeout 200 python tools/w4_check.py bench1 2>/dev/null | grep -v amdgpu; fi
done
