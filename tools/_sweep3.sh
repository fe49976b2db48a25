set -u
O=gpurun_out/r2_sweep3; mkdir -p $O
run() { tag=$1; shift; env "$@" python3 tools/layer_profile.py --prec f16x3 --unet-batch 128 --squares 16384 > $O/$tag.txt 2>&1; grep -E "unet \[|resnet18 \[|conv" $O/$tag.txt | grep -v "up[1-4].up\|downsample\|layer[34]" | sed "s/^/$tag: /"; }
run base A=1
run ns4 CHESSVISION_HIP_LIB=$PWD/chessvision-3lc_amd/lib/ab/libcv_ns4.so
run base2 A=1
run ns4b CHESSVISION_HIP_LIB=$PWD/chessvision-3lc_amd/lib/ab/libcv_ns4.so
