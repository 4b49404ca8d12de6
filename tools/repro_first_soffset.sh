#!/bin/bash
# the scalar-offset buffer_store hazard of conv_first_mfma_kernel.  Builds (1) the stand-alone repro tools/repro_soffset.hip and
# (2) two developer variants of the library with the store's plane offsets in the scalar offset operand (-DE3_FIRST_SOFFSET=1: wave-uniform part only,
# =2: the lane-dependent part too), then -- on a GPU box -- runs each 20 times on one input and counts distinct results.   $1 = build | run
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
if [ "${1:-build}" = build ]; then
  python -m elektronn3_amd.build > /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/_bin/repro_soffset tools/repro_soffset.hip 2>/dev/null
  objs=$(ls elektronn3_amd/build/*.o | grep -v "conv_small.hip.o")
  for v in 1 2; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -DE3_FIRST_SOFFSET=$v -c elektronn3_amd/csrc/conv_small.hip -o tools/_bin/conv_small_soff$v.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libe3unet_soff$v.so $objs tools/_bin/conv_small_soff$v.o
  done
  ls -la tools/_bin/repro_soffset tools/_bin/libe3unet_soff*.so
else
  tools/_bin/repro_soffset
  echo "== in-tree library (plane offsets in the lane offsets)"; python tools/repro_first_soffset.py
  for v in 1 2; do echo "== E3_FIRST_SOFFSET=$v"; E3_LIB_PATH=$PWD/tools/_bin/libe3unet_soff$v.so python tools/repro_first_soffset.py; done
fi
