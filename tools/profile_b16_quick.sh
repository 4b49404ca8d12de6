#!/bin/bash
# quick bf16 check on the GPU box: parity tests, step time, traffic of the roofline layer
R=$PWD; O=$R/gpurun_out/q_$1; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bf16_gpu.py -q -x 2>&1 | tail -2
python bench.py --dtype bf16 --no-predictor --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('step ms', round(d['ms_per_step'],3), 'fwd', round(r['ms_per_launch'],4), 'dgrad', round(r['dgrad_ms'],4), 'wgrad', round(r['wgrad_ms'],4))"
cd /tmp
PB="python $R/bench.py --no-cpu-baseline --no-predictor --steps 1 --warmup 1 --dtype bf16"
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o run --output-format csv -- $PB > $O/pmc_$c.log 2>&1; done
cd $R
python tools/pmc_roofline.py --dtype bf16 --kernel "conv_b16_kernel<4, 1, 3, 32, 16>" --fetch $O/pmc_FETCH_SIZE --write $O/pmc_WRITE_SIZE --steps 6 -o $O/roof.json | grep -E "fetch_bytes|write_bytes|all_pos" -A0 | head -3
python -c "
import json; d=json.load(open('$O/roof.json'))['bf16']['up_convs.2.conv1']; print(d['all_positions_MB'])"
