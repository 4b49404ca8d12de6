#!/bin/bash
# HBM reads (FETCH_SIZE) of every conv3_wino_pkernel / conv3_wino4_kernel launch of one fp32 step under forced block shapes of the logical brick order
# (E3_WINO_BLOCK="kw,kh,kd", brick_order.h); $@ = shapes ("auto": the launcher's choice, "0": the plain order)
ROOT=$PWD; O=$ROOT/gpurun_out/block_shapes; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
PB="python $ROOT/bench.py --no-cpu-baseline --no-predictor --no-extra-legs --steps 1 --warmup 1"
for sh in "$@"; do
  if [ $sh = auto ]; then unset E3_WINO_BLOCK; else export E3_WINO_BLOCK=$sh; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_$sh -o run --output-format csv -- $PB > $O/pmc_$sh.log 2>&1
  python - "$O/pmc_$sh" "$sh" <<'PY'
import csv, glob, os, sys, collections
d, sh = sys.argv[1:3]
cc = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
rows = collections.OrderedDict(); dur = {}; name = {}
for r in csv.DictReader(open(cc)):
    k = int(r['Dispatch_Id'])
    if 'conv3_wino' in r['Kernel_Name'] and 'pkernel' in r['Kernel_Name'] or 'conv3_wino4' in r['Kernel_Name']:
        rows[k] = rows.get(k, 0.0) + float(r['Counter_Value']); dur[k] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        name[k] = 'p' if 'pkernel' in r['Kernel_Name'] else '4'
ks = sorted(rows); n = len(ks) // 6; last = ks[-n:]
print(sh, ' '.join('%s:%.0f/%.0f' % (name[k], rows[k] * 2048 / 1e6, dur[k]) for k in last))
PY
done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
