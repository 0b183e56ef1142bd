#!/bin/bash
# round 4: the float32 carrying RoI walk with the next bin's loads in flight, as a side library built from
# tools/exp/roi_carry_prefetch.patch (applied to a temporary copy of csrc/; the build fails if it does not apply)
#   tools/r04/roi_carry_prefetch_build.sh  ->  tools/exp/libodet_roi_pf.so   (tools select it with ODET_LIB_PATH, tools/_diag.py)
set -e
cd "$(dirname "$0")/../.."
python - <<'PY'
import tools._diag as d
print(d.build_variant('tools/exp/libodet_roi_pf.so', ['-DODET_ROI_CARRY_PREFETCH'], patch='tools/exp/roi_carry_prefetch.patch'))
PY
