#!/bin/bash
# One MI355X pass over everything profiles/ quotes for this round: PMC traffic, the rocprofv3 kernel trace of the default bench,
# the SQ counters, per-layer profiles, single-board latency, the bilinear bench line.
#   usage (repo root, on the GPU box):  bash tools/collect_evidence.sh gpurun_out/final
set -u
OUT=${1:-gpurun_out/final}
mkdir -p "$OUT"
export TMPDIR=/tmp
PRECS="f16x3 f32" bash tools/pmc_collect.sh "$OUT" > "$OUT/pmc.log" 2>&1
bash tools/rocprof_bench.sh "$OUT" > "$OUT/rocprof_reduce.log" 2>&1
bash tools/pmc_sq.sh "$OUT" f16x3 unet > "$OUT/pmc_sq_unet.log" 2>&1
bash tools/pmc_sq.sh "$OUT" f16x3 resnet18 > "$OUT/pmc_sq_resnet18.log" 2>&1
bash tools/pmc_sq.sh "$OUT" f16r resnet18 > "$OUT/pmc_sq_f16r_resnet18.log" 2>&1
bash tools/pmc_sq2.sh "$OUT" f16x3 unet > "$OUT/pmc_sq2_unet.log" 2>&1
for d in f16x3 f32 f16; do
  python3 tools/layer_profile.py --prec $d > "$OUT/layer_profile_$d.txt" 2>&1
done
python3 bench.py --unet-variant bilinear --no-cpu-baseline --no-extras > "$OUT/bench_bilinear.json" 2> "$OUT/bench_bilinear.err"
python3 tools/latency.py > "$OUT/latency.txt" 2>&1
python3 tools/e2e_profile.py > "$OUT/e2e_profile.txt" 2>&1
# round 4: the single-board shape (what process_image runs) and the byte kernels
for d in f16x3 f32 f16; do
  python3 tools/layer_profile.py --prec $d --unet-batch 1 --squares 64 --iters 20 > "$OUT/layer_profile_b1_$d.txt" 2>&1
done
python3 tools/process_image_latency.py --prec f16x3,f16x3+f16r,f32 > "$OUT/process_image_latency.txt" 2>&1
python3 tools/process_image_breakdown.py > "$OUT/process_image_breakdown.txt" 2>&1
python3 tools/byte_kernels.py > "$OUT/byte_kernels.txt" 2>&1
bash tools/pmc_byte_kernels.sh "$OUT" > "$OUT/pmc_byte_kernels.log" 2>&1
bash tools/pmc_sq_b1.sh "$OUT" > "$OUT/pmc_sq_b1.log" 2>&1
rm -rf /tmp/rp_b1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_b1 -o b1 -- python3 "$(cd "$OLDPWD" && pwd)/tools/b1_trace.py" f16x3 50 > /tmp/rp_b1.log 2>&1)
find /tmp/rp_b1 -name '*kernel_stats.csv' -exec cp {} "$OUT/b1_kernel_stats.csv" \;
python3 tools/b1_trace_summary.py "$(find /tmp/rp_b1 -name '*kernel_trace.csv' | head -1)" > "$OUT/b1_kernel_trace_summary.txt" 2>&1
# round 5: one process_image call as a host / copy / kernel timeline
rm -rf /tmp/rp_pi
(cd /tmp && rocprofv3 --kernel-trace --hip-trace --memory-copy-trace --output-format csv -d /tmp/rp_pi -o pi -- python3 "$(cd "$OLDPWD" && pwd)/tools/process_image_latency.py" --iters 60 > /tmp/rp_pi.log 2>&1)
python3 tools/process_image_timeline.py /tmp/rp_pi 30 > "$OUT/process_image_timeline.txt" 2>&1
python3 tests/dev/python_overhead.py > "$OUT/python_overhead.txt" 2>&1
# round 5: the classifier's fp16 mode launch by launch (chained layer1; both forms of the chain, the four-launch schedule for scale)
python3 tests/dev/chain_ab.py > "$OUT/f16r_layer_profile.txt" 2>&1
CV_RESNET_CHAIN=0 python3 tests/dev/chain_ab.py > "$OUT/f16r_layer_profile_unchained.txt" 2>&1
CV_CHAIN_WG=1 python3 tests/dev/chain_ab.py > "$OUT/f16r_layer_profile_chain_1wg.txt" 2>&1
python3 tests/dev/determinism_probe.py f16r 200 128 > "$OUT/f16r_determinism.txt" 2>&1
cat "$OUT/rocprof_reduce.log" | tail -3; cat "$OUT/latency.txt" | tail -5
