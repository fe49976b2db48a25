#!/bin/bash
# One MI355X pass over everything profiles/ quotes: GPU tests, PMC traffic, the three bench lines, the rocprofv3 kernel
# summary of the default bench, per-layer profiles and single-board latency.
#   usage (repo root, on the GPU box):  bash tools/collect_evidence.sh gpurun_out/final
set -u
OUT=${1:-gpurun_out/final}
REPO=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3 > "$OUT/tests.txt"
bash tools/pmc_collect.sh "$OUT" > "$OUT/pmc.log" 2>&1
[ -s "$OUT/traffic.json" ] && cp "$OUT/traffic.json" profiles/r01_pmc_traffic.json
for d in f16x3 f32 f16; do
  python3 bench.py --dtype $d > "$OUT/bench_$d.json" 2> "$OUT/bench_$d.err"
done
rm -rf /tmp/rp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o r1 -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$REPO/$OUT/bench_f16x3_rocprof.json" 2> "$REPO/$OUT/rocprof.err")
find /tmp/rp -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
for d in f16x3 f32 f16; do
  python3 tools/layer_profile.py --prec $d > "$OUT/layer_profile_$d.txt" 2>&1
done
python3 tools/latency.py > "$OUT/latency.txt" 2>&1
cat "$OUT/tests.txt"; cut -c1-200 "$OUT"/bench_f16x3.json; head -5 "$OUT/kernel_stats.csv" | cut -c1-160; cat "$OUT/latency.txt"
