#!/bin/bash
# diagnostic builds of conv_x3.hip timed on two layers: what each part of the K loop costs.
#   tools/r05/x3_diag.sh build   (in the build container: tools/exp/libodet_x3_<variant>.so)
#   tools/r05/x3_diag.sh         (on the GPU box)
cd "$(dirname "$0")/../.."
VARIANTS="NOLOADA NOW NOSPLIT RAWSTORE ALL3"
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    python3 - <<PY
import tools._diag as d
flags = ['-DX3_DIAG_NOLOADA', '-DX3_DIAG_NOW', '-DX3_DIAG_NOSPLIT'] if '$v' == 'ALL3' else ['-DX3_DIAG_$v']
print(d.build_variant('tools/exp/libodet_x3_$v.so', flags, only=['conv_x3.hip']))
PY
  done
  exit 0
fi
echo "== product"; python3 tools/r05/x3_time.py
for v in $VARIANTS; do
  echo "== $v"; ODET_LIB_PATH=$PWD/tools/exp/libodet_x3_$v.so python3 tools/r05/x3_time.py
done
