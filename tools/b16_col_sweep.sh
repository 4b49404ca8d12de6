#!/bin/bash
# conv_b16_pkernel under forced cross-sections of its brick columns (E3_B16_COL=a,b: 2^a x 2^b bricks of 4 rows x 32 voxels; default 1,1): HBM reads per launch
# (FETCH_SIZE pass) and the unprofiled bf16 step, interleaved; $1 = timing rounds, rest = shapes
R=$1; shift
ROOT=$PWD; O=$ROOT/gpurun_out/b16_col; mkdir -p $O
for i in $(seq 1 $R); do
  for sh in "$@"; do
    export E3_B16_COL=$sh
    s=$(python bench.py --dtype bf16 --no-cpu-baseline --no-extra-legs --no-predictor --steps 30 --warmup 20 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "round $i $sh: bf16 step $s ms"
  done
done
export TMPDIR=/tmp; cd /tmp
PB="python $ROOT/bench.py --dtype bf16 --no-cpu-baseline --no-predictor --no-extra-legs --steps 1 --warmup 1"
for sh in "$@"; do
  export E3_B16_COL=$sh
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_$sh -o run --output-format csv -- $PB > $O/pmc_$sh.log 2>&1
  python - "$O/pmc_$sh" "$sh" <<'PY'
import csv, glob, os, sys, collections
d, sh = sys.argv[1:3]
cc = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
rows = collections.OrderedDict(); dur = {}
for r in csv.DictReader(open(cc)):
    k = int(r['Dispatch_Id'])
    if 'conv_b16_pkernel' in r['Kernel_Name']:
        rows[k] = rows.get(k, 0.0) + float(r['Counter_Value']); dur[k] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
ks = sorted(rows); n = len(ks) // 6; last = ks[-n:]
print(sh, ' '.join('%.0f/%.0f' % (rows[k] * 2048 / 1e6, dur[k]) for k in last), 'sum %.0f MB %.0f us' % (sum(rows[k] for k in last) * 2048 / 1e6, sum(dur[k] for k in last)))
PY
done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
