set -u
O=gpurun_out/r2_sweep6; mkdir -p $O
run() { tag=$1; shift; env "$@" python3 tools/layer_profile.py --prec f16x3 --unet-batch 128 --squares 64 --sq-chunk 64 > $O/$tag.txt 2>&1; grep -E "unet \[|inc.double|pack_input" $O/$tag.txt | sed "s/^/$tag: /"; }
run fused A=1
run generic CV_INC0=0
run fused2 A=1
python -m pytest tests/test_gpu_models.py tests/test_gpu_numerics.py tests/test_gpu_golden.py tests/test_gpu_e2e.py -m gpu -q -x 2>&1 | tail -5
