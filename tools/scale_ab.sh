#!/bin/bash
# Data-parallel A/B on ONE multi-GPU node: bench.py at N = 1, 2, 4, 8 ranks (one process per GPU over RCCL), once with the default serial all-reduce behind the
# backward and once with the bucketed all-reduce overlapped with the rest of the backward (--dp-overlap: GradSync(overlap=True), CU reserve behind the bucket event).
# Prints one line per (N, mode): ms per step, whole-job MVox/s, dp_mode as bench.py reports it; the JSON lines go to gpurun_out/scale_ab/.
# Usage: bash tools/scale_ab.sh [max_gpus=8] [steps=20] [warmup=5]        (weak scaling: batch 2 per rank; the driver's SCALE run uses the same command line)
set -u
MAXN=${1:-8}; STEPS=${2:-20}; WARM=${3:-5}
R="$(cd "$(dirname "$0")/.." && pwd)"; O=$R/gpurun_out/scale_ab; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
HAVE=$(python -c "import torch; print(torch.cuda.device_count())")
J='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print("N=%d %-8s %8.3f ms/step %9.1f MVox/s  dp_mode=%s" % (d["n_gpus"], sys.argv[1], d["ms_per_step"], d["value"] / 1e6, d.get("dp_mode")))'
for N in 1 2 4 8; do
  [ $N -gt $MAXN ] && break
  if [ $N -gt $HAVE ]; then echo "N=$N: only $HAVE GPU(s) visible, skipped"; continue; fi
  for mode in serial overlap; do
    [ $N -eq 1 ] && [ $mode = overlap ] && continue
    extra=""; [ $mode = overlap ] && extra="--dp-overlap"
    args="bench.py --gpus $N --steps $STEPS --warmup $WARM --no-cpu-baseline --no-predictor --no-extra-legs $extra"
    if [ $N -eq 1 ]; then cmd="python $R/$args"; else
      cmd="python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) $R/$args"; fi
    (cd $R && timeout 900 $cmd) > $O/n${N}_$mode.json 2> $O/n${N}_$mode.err
    python -c "$J" $mode < $O/n${N}_$mode.json || { echo "N=$N $mode: failed, see $O/n${N}_$mode.err"; tail -3 $O/n${N}_$mode.err; }
  done
done
