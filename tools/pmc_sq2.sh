#!/bin/bash
# SQ wave-state breakdown of the conv kernels (own counter-only run): where the wave cycles go.
#   usage (GPU box, repo root): bash tools/pmc_sq2.sh <outdir> [prec] [model]
set -u
OUT=${1:-gpurun_out/pmc_sq2}; PREC=${2:-f16x3}; MODEL=${3:-unet}
REPO=$(pwd); mkdir -p "$OUT"; export TMPDIR=/tmp
d=/tmp/pmc_sq2_${PREC}_${MODEL}; rm -rf "$d"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES -d "$d" -o r --output-format csv -- python3 "$REPO/tools/pmc_run.py" $PREC $MODEL > "$d.log" 2>&1)
f=$(find "$d" -name '*counter_collection.csv' | head -1)
python3 - "$f" "$OUT/sq2_${PREC}_${MODEL}.json" <<'PY'
import csv, json, sys, collections, re
rows = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "conv3x3_halo" not in k and "conv_igemm" not in k and "inc0_mfma" not in k: continue
    k = re.sub(r"cv::|\(cv::ConvParams\)|void ", "", k)
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
out = {}
for k, c in rows.items():
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    out[k] = {"dispatches": n[k], **{name.replace("SQ_", "").lower() + "_frac_of_wave_cycles": round(v / wc, 4) for name, v in c.items() if name not in ("SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES")},
              "wave_cycles": wc, "mfma_busy_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES")}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out.items(): print(k[:80], {a: b for a, b in v.items() if a.endswith("cycles") is False})
PY
