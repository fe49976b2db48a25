for prec in f16r f16x3; do
 for kn in "" "CV_CONV_PT=128" "CV_CONV_PT=128 CV_CONV_NS=2" "CV_CONV_W8=0" "CV_CT256=0"; do
  echo "== $prec [$kn]"
  env $kn timeout 200 python tests/dev/chain_ab.py --prec $prec 2>&1 | grep -E "ms/forward|layer2.0.conv1|layer3.0.conv1|layer3.1.conv1|layer4.1.conv1|layer4.0.conv1"
 done
done
