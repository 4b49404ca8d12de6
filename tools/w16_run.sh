mkdir -p gpurun_out/r04b
E3_WINO16_MIN=1 timeout 600 python tools/fuzz_conv.py 50 3 > gpurun_out/r04b/fuzz16.log 2>&1; grep -c "^ok" gpurun_out/r04b/fuzz16.log; grep "BAD" gpurun_out/r04b/fuzz16.log | head -20
B="python tools/bench_conv.py --what fwd,dgrad --iters 10"
echo "== new (stats)"; timeout 300 $B 2>/dev/null | grep -v amdgpu
echo "== old (stats)"; E3_NO_WINO16=1 timeout 300 $B 2>/dev/null | grep -v amdgpu
