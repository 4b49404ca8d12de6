mkdir -p gpurun_out/r04g
E3_WINO_PERSIST_MIN=1 timeout 600 python tools/fuzz_conv.py 40 7 > gpurun_out/r04g/fuzz_tr.log 2>&1; grep -c "^ok" gpurun_out/r04g/fuzz_tr.log; grep "BAD" gpurun_out/r04g/fuzz_tr.log | head
B="python tools/bench_conv.py --what fwd,dgrad --iters 10 --no-stats --layers L0_32_32,L0_64_32,L1_64_64,L1_128_64,L2_128_128"
echo "== TR"; timeout 300 $B 2>/dev/null | grep -v amdgpu
echo "== no TR"; E3_WINO_NO_TR=1 timeout 300 $B 2>/dev/null | grep -v amdgpu
for i in 1 2; do python bench.py --no-cpu-baseline --no-predictor --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('TR', d['ms_per_step'])"; done
for i in 1 2; do E3_WINO_NO_TR=1 python bench.py --no-cpu-baseline --no-predictor --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('noTR', d['ms_per_step'])"; done
