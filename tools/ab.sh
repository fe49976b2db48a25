#!/bin/bash
# Same-box A/B of two builds of the library: `bash tools/ab.sh <base.so> <new.so> [grep pattern] [layer_profile args...]`
# runs tools/layer_profile.py alternately (base new base new) and prints the matching per-layer lines of each run.
BASE=$1; NEW=$2; PAT=${3:-"=="}; shift 3 || true
for lib in "$BASE" "$NEW" "$BASE" "$NEW"; do
  echo "--- $lib"
  CHESSVISION_HIP_LIB=$(pwd)/$lib python3 tools/layer_profile.py --prec f16x3 "$@" 2>&1 | grep -E "$PAT"
done
