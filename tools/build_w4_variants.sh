#!/bin/bash
# developer builds of libe3unet with pieces of conv3_wino4_kernel left out (E3_W4_ABL bit mask; timing only, wrong results): $@ = masks
#   1 DMA, 2 weight loads, 4 output stores, 8 H / W passes, 32 MFMAs, 64 window reads, 128 output transform + exchange
set -e
cd "$(dirname "$0")/.."
python -m elektronn3_amd.build > /dev/null
mkdir -p tools/_bin
objs=$(ls elektronn3_amd/build/*.o | grep -v "conv_wino4.hip.o")
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -DE3_W4_ABL=$m ${E3_W4_EXTRA:-} -c elektronn3_amd/csrc/conv_wino4.hip -o tools/_bin/conv_wino4_abl$m.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libe3unet_w4abl$m.so $objs tools/_bin/conv_wino4_abl$m.o
  echo tools/_bin/libe3unet_w4abl$m.so
done
