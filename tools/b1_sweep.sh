mkdir -p gpurun_out/r04m
for cfg in "" "CV_SPLITK_HALO_STAGE_NS=300" "CV_SPLITK_HALO_STAGE_NS=800" "CV_SPLITK_TARGET=1024" "CV_SPLITK_TARGET=256" "CV_SPLITK_MAX_TILES=384" "CV_HALO_TH8_MAX_TILES=640" "CV_HALO_TH8_MAX_TILES=200"; do
  echo "== $cfg" >> gpurun_out/r04m/sweep.txt
  env $cfg python tools/latency.py 2>&1 | grep f16x3 >> gpurun_out/r04m/sweep.txt
done
cat gpurun_out/r04m/sweep.txt
