#!/bin/bash
# VALU occupancy of a step's kernels (one pass): instructions issued, cycles the VALUs were busy, wave cycles.  $1 = tag, $2 = dtype, $3 = kernel-name regex
R=$PWD; O=$R/gpurun_out/valu_$1; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $O/pmc -o run --output-format csv -- python $R/bench.py --no-cpu-baseline --no-predictor --no-extra-legs --steps 1 --warmup 1 --dtype ${2:-bf16} > $O/pmc.log 2>&1
cd $R
python - <<PY
import csv, glob, collections, statistics, re
cc = glob.glob('$O/pmc/**/*counter_collection.csv', recursive=True)[0]
kt = glob.glob('$O/pmc/**/*kernel_trace.csv', recursive=True)[0]
ktr = {r['Dispatch_Id']: r for r in csv.DictReader(open(kt))}
agg = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    agg[r['Dispatch_Id']][r['Counter_Name']] = agg[r['Dispatch_Id']].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
per = collections.defaultdict(list)
for d, c in agg.items():
    k = ktr[d]; name = re.sub(r'\(anonymous namespace\)::|void ', '', k['Kernel_Name']); name = re.sub(r'\(.*', '', name)[:48]
    if not re.search('${3:-.}', name): continue
    grid = int(k['Grid_Size_X']) * int(k['Grid_Size_Y']) // max(int(k['Workgroup_Size_X']), 1)
    per[(name, grid)].append((int(k['End_Timestamp']) - int(k['Start_Timestamp']), c))
print('| kernel | WGs | n | us (max) | VALU insts / wave | VALU busy (of SIMD cycles) | waves/SIMD | wait_any | VMEM rd / wr per wave |')
print('|---|---|---|---|---|---|---|---|---|')
for (name, grid), L in sorted(per.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
    d, c = max(L, key=lambda t: t[0])
    cyc = c.get('GRBM_GUI_ACTIVE', 0) / 8
    waves = grid * 4
    wc = c.get('SQ_WAVE_CYCLES', 1)
    print(f"| {name} | {grid} | {len(L)} | {d / 1e3:.1f} | {c.get('SQ_INSTS_VALU', 0) / waves:.0f} | {c.get('SQ_ACTIVE_INST_VALU', 0) / (cyc * 1024):.2f} | {4 * wc / (cyc * 1024):.2f} | {c.get('SQ_WAIT_ANY', 0) / wc:.2f} | {c.get('SQ_INSTS_VMEM_RD', 0) / waves:.0f} / {c.get('SQ_INSTS_VMEM_WR', 0) / waves:.0f} |")
PY
