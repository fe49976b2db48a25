set -u
O=gpurun_out/r2_sweep9; mkdir -p $O
run() { tag=$1; shift; env "$@" python3 tools/layer_profile.py --prec f16x3 --unet-batch 128 --squares 64 --sq-chunk 64 > $O/$tag.txt 2>&1; grep -E "unet \[|conv" $O/$tag.txt | grep -v "\.up \|inc\.\|up4\|layer" | sed "s/^/$tag: /"; }
run base A=1
run c256 CV_CT64_MAXROWS=256
run c512 CV_CT64_MAXROWS=512
run c1024 CV_CT64_MAXROWS=1024
run base2 A=1
