set -x
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_knobs.py -x -q -m gpu -k "position_major" > gpurun_out/r06/pos_test.txt 2>&1; tail -5 gpurun_out/r06/pos_test.txt
for prec in f16r f16x3 f32; do
for pos in 0 1; do
CV_POS=$pos timeout 600 python tools/layer_profile.py --prec $prec --unet-batch 1 --chunk 2 > gpurun_out/r06/lp_${prec}_pos$pos.txt 2>&1
done; done
grep -h "resnet18\|layer[234]" gpurun_out/r06/lp_f16r_pos0.txt gpurun_out/r06/lp_f16r_pos1.txt | cut -c1-120
