#!/bin/bash
# SQ-side PMC pass over the single-board shape (UNet B=1 / ResNet-18 B=64): matrix-pipe busy share and LDS conflicts per conv
# kernel instantiation, incl. the split-K launches and their second pass.  CV_GRAPH=0: a graph replay hides kernel names from the
# counter collection.   usage (GPU box, repo root): bash tools/pmc_sq_b1.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc_sq_b1}
REPO=$(pwd); mkdir -p "$OUT"; export TMPDIR=/tmp
for model in unet resnet18; do
  b=1; [ $model = resnet18 ] && b=64
  d=/tmp/pmc_sq_b1_$model; rm -rf "$d"
  (cd /tmp && CV_GRAPH=0 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d "$d" -o r --output-format csv -- python3 "$REPO/tools/pmc_run.py" f16x3 $model $b 20 > "$d.log" 2>&1)
  f=$(find "$d" -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$OUT/sq_b1_f16x3_${model}.json" <<'PY'
import csv, json, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if not any(t in k for t in ("conv3x3_halo", "conv_igemm", "conv_splitk_reduce", "stem_pool", "head_kernel")): continue
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
out = {}
for k, c in rows.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                     # summed over the 8 XCDs
    out[k.replace("void cv::", "").split("(")[0]] = {
        "dispatches": n[k],
        "mfma_busy_frac_of_active": (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024.0)) if gui else None,   # 256 CUs x 4 matrix pipes
        "lds_bank_conflict_frac": (c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None,
        "gpu_active_cycles_per_dispatch": gui / n[k] if n[k] else None}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out.items(): print(k[:100], v)
PY
done
