set -u
O=gpurun_out/r2_sweep8; mkdir -p $O
run() { tag=$1; shift; env "$@" python3 tools/layer_profile.py --prec f16x3 --unet-batch 128 --squares 64 --sq-chunk 64 > $O/$tag.txt 2>&1; grep -E "unet \[|inc.double" $O/$tag.txt | sed "s/^/$tag: /"; }
python -m pytest tests/test_gpu_models.py -m gpu -q -x 2>&1 | tail -5
run fused A=1
run apart CV_FUSE_INC=0
run fused2 A=1
