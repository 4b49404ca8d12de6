B="python tools/bench_conv.py --what fwd,dgrad --iters 10 --no-stats --layers L0_32_32,L0_64_32,L1_64_64,L1_128_64"
for m in "$@"; do echo "== abl $m"; E3_LIB_PATH=tools/_bin/libe3unet_w16abl$m.so timeout 200 $B 2>/dev/null | grep -v amdgpu; done
