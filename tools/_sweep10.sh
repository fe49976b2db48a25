set -u
O=gpurun_out/r2_sweep10; mkdir -p $O
run() { tag=$1; shift; env "$@" python3 tools/layer_profile.py --prec f16x3 --unet-batch 64 --squares 16384 > $O/$tag.txt 2>&1; grep -E "resnet18 \[|layer2" $O/$tag.txt | sed "s/^/$tag: /"; }
run base A=1
run img8_64 CV_HALO_IMG8_64=1
run base2 A=1
run img8_64b CV_HALO_IMG8_64=1
CV_HALO_IMG8_64=1 python -m pytest tests/test_gpu_models.py tests/test_gpu_ops.py -m gpu -q -x -k "resnet or conv or chunk" 2>&1 | tail -3
