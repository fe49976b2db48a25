#!/bin/bash
# Collect HBM traffic of the conv family with rocprofv3 PMC counters: one pass per counter group (never combined with
# tracing), reduced by tools/pmc_traffic.py.   usage (on the GPU box, repo root):  bash tools/pmc_collect.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc}
REPO=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
for prec in ${PRECS:-f16x3 f32 f16}; do
  for model in unet resnet18; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      d=/tmp/pmc_${prec}_${model}_${ctr}
      rm -rf "$d"
      (cd /tmp && timeout 300 rocprofv3 --pmc $ctr -d "$d" -o r --output-format csv -- python3 "$REPO/tools/pmc_run.py" $prec $model > "$d.log" 2>&1)
    done
    f=$(find /tmp/pmc_${prec}_${model}_FETCH_SIZE -name '*counter_collection.csv' | head -1)
    w=$(find /tmp/pmc_${prec}_${model}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
    python3 tools/pmc_traffic.py "$f" "$w" "$OUT/traffic.json" ${prec}_${model}
  done
done
