#!/bin/bash
# the Predictor leg under rocprofv3 --kernel-trace --memory-copy-trace: blit kernels against SDMA copy records (profiles/r06_predictor_modes.md section 2)
R=$PWD; O=$R/gpurun_out/copy_trace; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for i in 1 2 3; do
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/run$i -o run -- python $R/bench.py --no-cpu-baseline --no-extra-legs --no-live-traffic --steps 1 --warmup 1 --predictor-volume sub > $O/run$i.log 2>&1
python - $O/run$i <<'PY'
import csv, glob, os, sys, collections
d = sys.argv[1]
kt = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
n = 0; tot = 0; big = 0; bigt = 0
for r in csv.DictReader(open(kt)):
    if 'copyBuffer' in r['Kernel_Name']:
        dt = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        n += 1; tot += dt
        if dt > 500: big += 1; bigt += dt
print('copyBuffer kernels:', n, 'total %.1f ms' % (tot / 1e3), '; > 0.5 ms:', big, '%.1f ms' % (bigt / 1e3))
mc = glob.glob(os.path.join(d, '**', '*memory_copy_trace.csv'), recursive=True)
if mc:
    by = collections.defaultdict(lambda: [0, 0.0, 0])
    rows = list(csv.DictReader(open(mc[0])))
    if rows: print('columns:', list(rows[0].keys()))
    for r in rows:
        k = r.get('Direction') or r.get('Name') or '?'
        dt = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        by[k][0] += 1; by[k][1] += dt
    for k, v in by.items(): print('memory copy', k, v[0], 'copies, %.1f ms' % (v[1] / 1e3))
print(open(os.path.join(os.path.dirname(d), os.path.basename(d) + '.log')).read().strip().splitlines()[-1][:0])
PY
grep -o '"predictor": {"metric": "Predictor MVox/s", "value": [0-9.]*' $O/run$i.log | tail -1
done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
