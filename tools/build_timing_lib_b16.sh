#!/bin/bash
# developer build of libe3unet with the phase stamps of the bf16 conv / wgrad kernels (tools/conv_phases.py); $1 = suffix of the output library
set -e
cd "$(dirname "$0")/.."
python -m elektronn3_amd.build > /dev/null
mkdir -p tools/_bin
for f in bf16_conv bf16_wgrad; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -DE3_CONV_TIMING -c elektronn3_amd/csrc/$f.hip -o tools/_bin/${f}_timing.o 2>/dev/null
done
objs=$(ls elektronn3_amd/build/*.o | grep -v "bf16_conv.hip.o\|bf16_wgrad.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libe3unet_b16timing${1}.so $objs tools/_bin/bf16_conv_timing.o tools/_bin/bf16_wgrad_timing.o
echo tools/_bin/libe3unet_b16timing${1}.so
