B="python tools/bench_conv.py --what fwd,dgrad --iters 10 --no-stats --layers L0_32_32,L0_64_32,L1_64_64,L1_128_64,L2_128_128"
for v in "$@"; do echo "== $v"; E3_LIB_PATH=tools/_bin/libe3unet_w16abl$v.so timeout 200 $B 2>/dev/null | grep -v amdgpu; done
echo "== old"; E3_NO_WINO16=1 timeout 200 $B 2>/dev/null | grep -v amdgpu
