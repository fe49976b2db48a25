set -u
python -m pytest tests/test_gpu_models.py tests/test_gpu_numerics.py tests/test_gpu_golden.py -m gpu -q -x -k "resnet or golden or u8 or chunks" 2>&1 | tail -5
python3 tools/layer_profile.py --prec f16x3 --unet-batch 2 --chunk 2 --squares 16384 2>&1 | grep -E "resnet18 \[|stem|head"
python3 tools/layer_profile.py --prec f16 --unet-batch 2 --chunk 2 --squares 16384 2>&1 | grep -E "resnet18 \[|stem|head"
