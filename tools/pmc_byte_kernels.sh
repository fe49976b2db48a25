#!/bin/bash
# PMC passes over the SURVEY section 8f byte kernels (resize_area_u8, extract_squares_u8) on 256 boards: HBM traffic (separate
# FETCH_SIZE / WRITE_SIZE passes, gfx950-corrected as tools/pmc_traffic.py) and vector instructions per wave (own pass).
#   usage (GPU box, repo root): bash tools/pmc_byte_kernels.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc_bytes}
REPO=$(pwd); mkdir -p "$OUT"; export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE"; do
  tag=$(echo $ctr | tr ' ' '_')
  d=/tmp/pmc_bytes_$tag; rm -rf "$d"
  (cd /tmp && timeout 300 rocprofv3 --pmc $ctr -d "$d" -o r --output-format csv -- python3 "$REPO/tools/byte_kernels.py" > "$d.log" 2>&1)
done
python3 - "$OUT/byte_kernels_pmc.json" <<'PY'
import collections, csv, glob, json, sys
def load(tag):
    f = glob.glob(f"/tmp/pmc_bytes_{tag}/**/*counter_collection.csv", recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
res = {}
kern = lambda n: "resize_area_2x2c3" in n or "extract_squares_u8" in n
def short(n): return n.split("(")[0].replace("void cv::", "")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU_SQ_WAVES_GRBM_GUI_ACTIVE"):
    for r in load(tag):
        if kern(r["Kernel_Name"]): acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    n = max(len(v) for v in c.values())
    mean = lambda key: (sum(c[key]) / len(c[key])) if c.get(key) else None
    fetch, write = mean("FETCH_SIZE"), mean("WRITE_SIZE")
    res[k] = {"dispatches": n,
              "read_bytes_per_launch": None if fetch is None else 2.0 * fetch * 1024,       # gfx950 wide-read correction (see pmc_traffic.py)
              "read_bytes_per_launch_uncorrected": None if fetch is None else fetch * 1024,
              "write_bytes_per_launch": None if write is None else write * 1024,
              "valu_insts_per_wave": (mean("SQ_INSTS_VALU") / mean("SQ_WAVES")) if mean("SQ_WAVES") else None,
              "waves_per_launch": mean("SQ_WAVES"),
              "note": "256 boards per launch (tools/byte_kernels.py); one wave = 256 output pixels (extract) / 256 output pixels (resize)"}
json.dump(res, open(sys.argv[1], "w"), indent=1)
print(json.dumps(res, indent=1))
PY
