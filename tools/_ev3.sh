set -u
OUT=gpurun_out/ev3; mkdir -p $OUT; export TMPDIR=/tmp
PRECS="f16x3 f32" bash tools/pmc_collect.sh "$OUT" > "$OUT/pmc.log" 2>&1
bash tools/rocprof_bench.sh "$OUT" > "$OUT/rocprof_reduce.log" 2>&1
bash tools/pmc_sq.sh "$OUT" f16x3 unet > "$OUT/pmc_sq_unet.log" 2>&1
bash tools/pmc_sq.sh "$OUT" f16x3 resnet18 > "$OUT/pmc_sq_resnet18.log" 2>&1
bash tools/pmc_sq2.sh "$OUT" f16x3 unet > "$OUT/pmc_sq2_unet.log" 2>&1
tail -2 $OUT/rocprof_reduce.log
