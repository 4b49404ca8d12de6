#!/bin/bash
# developer builds of libe3unet with pieces of conv3_wino16_kernel left out (E3_W16_ABL bit mask; timing only, wrong results): $@ = masks
set -e
cd "$(dirname "$0")/.."
python -m elektronn3_amd.build > /dev/null
mkdir -p tools/_bin
objs=$(ls elektronn3_amd/build/*.o | grep -v "conv_wino16.hip.o")
for m in "$@"; do
  tag=$m${E3_W16_TAG:-}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -DE3_W16_ABL=$m ${E3_W16_EXTRA:-} -c elektronn3_amd/csrc/conv_wino16.hip -o tools/_bin/conv_wino16_abl$tag.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libe3unet_w16abl$tag.so $objs tools/_bin/conv_wino16_abl$tag.o
  echo tools/_bin/libe3unet_w16abl$tag.so
done
