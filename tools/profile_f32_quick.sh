#!/bin/bash
# quick fp32 check on the GPU box: step time and traffic of the persistent Winograd kernel's launches
R=$PWD; O=$R/gpurun_out/qf_$1; mkdir -p $O; export TMPDIR=/tmp
python bench.py --no-predictor --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('step ms', round(d['ms_per_step'],3), 'fwd', round(r['ms_per_launch'],4), 'dgrad', round(r['dgrad_ms'],4), 'wgrad', round(r['wgrad_ms'],4))"
cd /tmp
PB="python $R/bench.py --no-cpu-baseline --no-predictor --steps 1 --warmup 1 --dtype f32"
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o run --output-format csv -- $PB > $O/pmc_$c.log 2>&1; done
cd $R
python tools/pmc_roofline.py --dtype f32 --kernel conv3_wino_pkernel --fetch $O/pmc_FETCH_SIZE --write $O/pmc_WRITE_SIZE --steps 6 -o $O/roof.json > /dev/null
python -c "
import json; d=json.load(open('$O/roof.json'))['f32']['up_convs.2.conv1']; print([(round(a), round(b)) for a, b in d['all_positions_MB']]); print('total fetch MB', round(sum(a for a, b in d['all_positions_MB'])))"
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
