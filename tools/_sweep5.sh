set -u
O=gpurun_out/r2_sweep5; mkdir -p $O
L=$PWD/chessvision-3lc_amd/lib/ab
run() { tag=$1; shift; env "$@" python3 tools/layer_profile.py --prec f16x3 --unet-batch 128 --squares 16384 > $O/$tag.txt 2>&1; grep -E "unet \[|resnet18 \[|inc.double_conv.3|up4.conv|layer1" $O/$tag.txt | sed "s/^/$tag: /"; }
run base A=1
run th16 CHESSVISION_HIP_LIB=$L/libcv_th16.so
run th16ns4 CHESSVISION_HIP_LIB=$L/libcv_th16ns4.so
run base2 A=1
run th16b CHESSVISION_HIP_LIB=$L/libcv_th16.so
run th16ns4b CHESSVISION_HIP_LIB=$L/libcv_th16ns4.so
CHESSVISION_HIP_LIB=$L/libcv_th16.so python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x 2>&1 | tail -3
CHESSVISION_HIP_LIB=$L/libcv_th16ns4.so python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x 2>&1 | tail -3
