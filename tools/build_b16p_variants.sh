#!/bin/bash
# developer builds of libe3unet with pieces of conv_b16_pkernel left out (E3_PABL bit mask; timing only, wrong results): $@ = masks
set -e
cd "$(dirname "$0")/.."
python -m elektronn3_amd.build > /dev/null
mkdir -p tools/_bin
objs=$(ls elektronn3_amd/build/*.o | grep -v "bf16_conv.hip.o")
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -DE3_PABL=$m ${E3_PABL_EXTRA:-} -c elektronn3_amd/csrc/bf16_conv.hip -o tools/_bin/bf16_conv_abl$m.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libe3unet_pabl$m.so $objs tools/_bin/bf16_conv_abl$m.o
  echo tools/_bin/libe3unet_pabl$m.so
done
