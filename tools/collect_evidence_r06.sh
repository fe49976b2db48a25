#!/bin/bash
# Round 6: everything profiles/r06_* quotes beyond the common pass of tools/collect_evidence.sh, on ONE box.
#   usage (repo root, on the GPU box):  bash tools/collect_evidence_r06.sh gpurun_out/final6
set -u
OUT=${1:-gpurun_out/final6}
mkdir -p "$OUT"
export TMPDIR=/tmp
# the default bench line (20 timed steps) -- the figures of profiles/README.md's first r06 row
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_f16x3.json" 2> "$OUT/bench_f16x3_layers.txt"
# position-major launches: same box, off / on, per launch (section 1 of r06_tuning.md)
for prec in f16r f16x3 f32; do
  for pos in 0 1; do
    CV_POS=$pos python3 tools/layer_profile.py --prec $prec --unet-batch 1 --chunk 2 2>&1 | grep -v amdgpu.ids > "$OUT/exp_pos_${prec}_pos$pos.txt"
  done
done
# the fp16 classifier's worst case over seeds (section 2)
python3 tests/dev/f16r_seed_search.py --seeds 8 --precs f16r,f16 > "$OUT/exp_seed_search_bias_corrected.jsonl" 2>/dev/null
CV_BIAS_CORR=0 python3 tests/dev/f16r_seed_search.py --seeds 8 --precs f16r,f16 > "$OUT/exp_seed_search_before.jsonl" 2>/dev/null
# request slots (section 5): thread sweep, then one and four threads under a kernel trace
python3 tests/dev/concurrent_probe.py sweep 200 2>&1 | grep -v amdgpu.ids > "$OUT/concurrent_sweep.txt"
for T in 1 4; do
  rm -rf /tmp/ct$T
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/ct$T -- python3 "$OLDPWD/tests/dev/concurrent_probe.py" trace $T 100 > /tmp/ct$T.log 2>&1)
  python3 tests/dev/concurrent_overlap.py "$(find /tmp/ct$T -name '*kernel_trace.csv' | head -1)" > "$OUT/concurrent_overlap_T$T.txt" 2>&1
done
tail -2 "$OUT/concurrent_sweep.txt"; tail -c 400 "$OUT/bench_f16x3.json"
