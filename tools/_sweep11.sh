set -u
O=gpurun_out/r2_sweep11; mkdir -p $O
L=$PWD/chessvision-3lc_amd/lib/ab
run() { tag=$1; shift; env "$@" python3 tools/layer_profile.py --prec f16x3 --unet-batch 128 --squares 16384 > $O/$tag.txt 2>&1; grep -E "unet \[|resnet18 \[" $O/$tag.txt | sed "s/^/$tag: /"; }
run base A=1
run ns4 CHESSVISION_HIP_LIB=$L/libcv_ns4.so
run base2 A=1
run ns4b CHESSVISION_HIP_LIB=$L/libcv_ns4.so
