#!/bin/bash
# SQ-level counters of the bf16 step's kernels (one pass, 8 SQ slots): where do the waves spend their cycles
# (lds_active / lds_conflict = LDS-array cycles per CU cycle: a ds_read_b128 wave-instruction is 4 of them, MI355X_MICROARCH.md "LDS";
#  until round 3 the two columns carried a spurious factor 4)
R=$PWD; O=$R/gpurun_out/sq_$1; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/pmc -o run --output-format csv -- python $R/bench.py --no-cpu-baseline --no-predictor --steps 1 --warmup 1 --dtype ${2:-bf16} > $O/pmc.log 2>&1
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
    k = ktr[d]; name = re.sub(r'\(anonymous namespace\)::|void ', '', k['Kernel_Name']); name = re.sub(r'\(.*', '', name)[:44]
    if not re.search('$3' or '.', name): continue
    grid = int(k['Grid_Size_X']) * int(k['Grid_Size_Y']) // max(int(k['Workgroup_Size_X']), 1)
    dur = int(k['End_Timestamp']) - int(k['Start_Timestamp'])
    per[(name, grid)].append((dur, c))
print('| kernel | WGs | n | us | clk GHz | MFMA busy | waves/SIMD | wait_any | wait_inst | wait_lds | lds_active | lds_conflict |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|')
for (name, grid), L in sorted(per.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
    med = lambda f: statistics.median(f(d, c) for d, c in L)
    cyc = med(lambda d, c: c.get('GRBM_GUI_ACTIVE', 0) / 8)
    us = med(lambda d, c: d / 1e3)
    g = lambda n: med(lambda d, c: c.get(n, 0))
    wc = g('SQ_WAVE_CYCLES')
    print(f"| {name} | {grid} | {len(L)} | {us:.1f} | {cyc / (us * 1e3):.2f} | {g('SQ_VALU_MFMA_BUSY_CYCLES') / (cyc * 1024):.3f} | {4 * wc / (cyc * 1024):.2f} | {g('SQ_WAIT_ANY') / wc:.2f} | {g('SQ_WAIT_INST_ANY') / wc:.2f} | {g('SQ_WAIT_INST_LDS') / wc:.2f} | {g('SQ_LDS_IDX_ACTIVE') / (cyc * 256):.2f} | {g('SQ_LDS_BANK_CONFLICT') / (cyc * 256):.2f} |")
PY
