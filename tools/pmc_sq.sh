#!/bin/bash
# SQ-side PMC pass over the conv kernels of one model (own run, counters only): MFMA busy cycles, LDS bank conflicts,
# wave cycles.  usage (GPU box, repo root): bash tools/pmc_sq.sh <outdir> [prec] [model]
set -u
OUT=${1:-gpurun_out/pmc_sq}; PREC=${2:-f16x3}; MODEL=${3:-unet}
REPO=$(pwd); mkdir -p "$OUT"; export TMPDIR=/tmp
d=/tmp/pmc_sq_${PREC}_${MODEL}; rm -rf "$d"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d "$d" -o r --output-format csv -- python3 "$REPO/tools/pmc_run.py" $PREC $MODEL > "$d.log" 2>&1)
f=$(find "$d" -name '*counter_collection.csv' | head -1)
python3 - "$f" "$OUT/sq_${PREC}_${MODEL}.json" <<'PY'
import csv, json, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "conv3x3_halo" not in k and "conv_igemm" not in k and "inc0_mfma" not in k and "shortcut1x1s2" not in k: continue
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
out = {}
for k, c in rows.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                     # summed over the 8 XCDs
    out[k] = {"dispatches": n[k], "counters": dict(c),
              # MFMA busy cycles are summed over all SIMDs' matrix pipes: 256 CUs x 4
              "mfma_busy_frac_of_active": (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024.0)) if gui else None,
              "lds_bank_conflict_frac": (c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE") else None}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out.items(): print(k[:90], v["dispatches"], v["mfma_busy_frac_of_active"], v["lds_bank_conflict_frac"])
PY
