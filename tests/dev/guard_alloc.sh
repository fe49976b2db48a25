#!/bin/bash
# Run the serve paths of every precision with guard pages behind (1) and before (2) every engine buffer.
out=gpurun_out/guard; mkdir -p $out
for mode in 1 2; do
  for p in f16x3 f32 f16 f16r; do
    CV_GUARD_ALLOC=$mode timeout 300 python tests/dev/guard_probe.py $p 40 > $out/${p}_$mode.log 2>&1
    echo "mode=$mode $p rc=$? :: $(grep -v amdgpu.ids $out/${p}_$mode.log | tail -n 2 | tr '\n' '|' | cut -c1-250)"
  done
done
