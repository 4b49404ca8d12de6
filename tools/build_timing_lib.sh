#!/bin/bash
# developer build of libe3unet with the phase-timing stamps of conv3_wino_pkernel (tools/phase_timing_pwino.py); $1 = ablation mask, $2 = suffix
set -e
cd "$(dirname "$0")/.."
python -m elektronn3_amd.build > /dev/null
mkdir -p tools/_bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -DE3_WINO_TIMING -DE3_WINO_ABL=${1:-0} -c elektronn3_amd/csrc/conv_wino.hip -o tools/_bin/conv_wino_timing.o 2>/dev/null
objs=$(ls elektronn3_amd/build/*.o | grep -v "conv_wino.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libe3unet_timing${2}.so $objs tools/_bin/conv_wino_timing.o
echo tools/_bin/libe3unet_timing${2}.so
