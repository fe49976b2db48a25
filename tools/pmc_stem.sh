#!/bin/bash
# SQ counters of the ResNet-18 stem kernel (own counter-only runs, two passes).  usage (GPU box, repo root): bash tools/pmc_stem.sh <outdir> [prec]
set -u
OUT=${1:-gpurun_out/pmc_stem}; PREC=${2:-f16x3}
REPO=$(pwd); mkdir -p "$OUT"; export TMPDIR=/tmp
for pass in A B; do
  d=/tmp/pmc_stem_$pass; rm -rf "$d"
  if [ $pass = A ]; then CTR="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES";
  else CTR="SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES"; fi
  (cd /tmp && timeout 600 rocprofv3 --pmc $CTR -d "$d" -o r --output-format csv -- python3 "$REPO/tools/pmc_run.py" $PREC resnet18 > "$d.log" 2>&1)
  f=$(find "$d" -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = collections.defaultdict(float); n = 0
for r in csv.DictReader(open(sys.argv[1])):
    if "stem_pool" not in r["Kernel_Name"]: continue
    rows[r["Counter_Name"]] += float(r["Counter_Value"])
print({k: v for k, v in rows.items()})
if "SQ_WAVE_CYCLES" in rows:
    wc = rows["SQ_WAVE_CYCLES"]; print({k: round(v / wc, 4) for k, v in rows.items()})
if "GRBM_GUI_ACTIVE" in rows:
    gui = rows["GRBM_GUI_ACTIVE"] / 8; print("mfma_busy_frac", rows["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024), "lds_conflict", rows["SQ_LDS_BANK_CONFLICT"] / max(1, rows["SQ_LDS_IDX_ACTIVE"]),
                                           "valu per wave", rows["SQ_INSTS_VALU"] / max(1, rows["SQ_WAVES"]), "lds per wave", rows["SQ_INSTS_LDS"] / max(1, rows["SQ_WAVES"]))
PY
done
