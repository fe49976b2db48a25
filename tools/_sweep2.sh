set -u
O=gpurun_out/r2_sweep2; mkdir -p $O
run() { tag=$1; shift; env "$@" python3 tools/layer_profile.py --prec f16x3 --unet-batch 128 --squares 16384 > $O/$tag.txt 2>&1; grep -E "unet \[|resnet18 \[|inc.double_conv.3|up4.conv|layer1.0.conv1|down3.*conv.3|up1.conv.double_conv.0|layer2.1.conv1" $O/$tag.txt | sed "s/^/$tag: /"; }
run t0 CV_TUNE=0
run t1 CV_TUNE=1
run t2 CV_TUNE=2
run t4 CV_TUNE=4
run t8 CV_TUNE=8
run t0b CV_TUNE=0
