#!/bin/bash
# SQ counters of one conv layer through tools/bench_conv.py (two passes of 8 SQ slots): $1 = tag, $2 = layer, rest = env assignments
R=$PWD; TAG=$1; LAYER=$2; shift 2
O=$R/gpurun_out/pmcconv_$TAG; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
CMD="python $R/tools/bench_conv.py --what fwd --iters 3 --no-stats --layers $LAYER"
env "$@" rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/a -o run --output-format csv -- $CMD > $O/a.log 2>&1
env "$@" rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $O/b -o run --output-format csv -- $CMD > $O/b.log 2>&1
env "$@" rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU -d $O/c -o run --output-format csv -- $CMD > $O/c.log 2>&1
cd $R
python - <<PY
import csv, glob, collections, statistics, re
res = {}
for sub in 'abc':
    cc = glob.glob('$O/%s/**/*counter_collection.csv' % sub, recursive=True)
    kt = glob.glob('$O/%s/**/*kernel_trace.csv' % sub, recursive=True)
    if not cc: continue
    ktr = {r['Dispatch_Id']: r for r in csv.DictReader(open(kt[0]))}
    agg = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc[0])):
        agg[r['Dispatch_Id']][r['Counter_Name']] = agg[r['Dispatch_Id']].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    per = collections.defaultdict(list)
    for d, c in agg.items():
        k = ktr[d]; name = re.sub(r'\(anonymous namespace\)::|void ', '', k['Kernel_Name']); name = re.sub(r'\(.*', '', name)[:40]
        if 'conv3_wino' not in name: continue
        c['dur_us'] = (int(k['End_Timestamp']) - int(k['Start_Timestamp'])) / 1e3
        per[name].append(c)
    for name, L in per.items():
        for cn in L[0]:
            res.setdefault(name, {})[cn] = statistics.median(c[cn] for c in L)
for name, c in res.items():
    print('==', name, '$TAG')
    wc = c.get('SQ_WAVE_CYCLES', 1)
    for cn in sorted(c):
        print(f'  {cn:28s} {c[cn]:16.0f}   /wave_cycles {c[cn] / wc:8.3f}')
PY
