#!/bin/bash
cd "$(dirname "$0")/../.."
for th in 8 16 32; do
  ODET_EXTRA_HIPCC_FLAGS="-DRG_TH=$th" python -c "
from tf_eager_object_detection_amd import _build
import os
os.utime(os.path.join(_build.CSRC,'stem.hip'))
_build.build()" > /dev/null 2>&1
  echo "RG_TH=$th"; python tools/exp/rgb_conv_time.py 2>&1 | tail -2
done
