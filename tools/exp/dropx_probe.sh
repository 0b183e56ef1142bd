#!/bin/bash
cd "$(dirname "$0")/../.."
echo product; python tools/exp/dropx_probe.py 2>&1 | tail -6
ODET_EXTRA_HIPCC_FLAGS="-DODET_C3_DROPX" python -c "
from tf_eager_object_detection_amd import _build
import os
os.utime(os.path.join(_build.CSRC,'conv3x3.hip'))
_build.build()" > /dev/null 2>&1
echo "two of three pixel copies dropped"; python tools/exp/dropx_probe.py 2>&1 | tail -6
