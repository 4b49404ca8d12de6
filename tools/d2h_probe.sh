#!/bin/bash
# tools/d2h_probe.py under rocprofv3 for every case and three runtime switches: which engine moves a device -> page-locked host copy
R=$PWD; O=$R/gpurun_out/d2h_probe; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
run() {  # name, env..., case
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/$name -o run -- python $R/tools/d2h_probe.py ${name%%-*} > $O/$name.log 2>&1
  python - $O/$name $name <<'PY'
import csv, glob, os, sys
d, name = sys.argv[1:3]
kt = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
nb = sum(1 for r in csv.DictReader(open(kt[0])) if 'copyBuffer' in r['Kernel_Name'] and int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 1e6) if kt else -1
mc = glob.glob(os.path.join(d, '**', '*memory_copy_trace.csv'), recursive=True)
rows = [r for r in csv.DictReader(open(mc[0]))] if mc else []
big = [r for r in rows if int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 1e6]
dirs = {}
for r in big: dirs[r['Direction']] = dirs.get(r['Direction'], 0) + 1
print(name, ': blit kernels > 1 ms:', nb, '; SDMA copies > 1 ms:', dirs)
PY
}
run fresh x=1
run after x=1
run split x=1
run h2d x=1
run after-blit0 GPU_FORCE_BLIT_COPY_SIZE=0
run after-sdmasize HSA_FORCE_SDMA_SIZE=1
run after-engine2 GPU_BLIT_ENGINE_TYPE=2
find $O -name "*.db" -delete
