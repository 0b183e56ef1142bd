#!/bin/bash
# what each part of the split-precision K loop costs, two-limb (x2) and three-limb (x3) form, 128 x 256 tile: diagnostic builds
# (tools/exp/conv_x3_diag_switches.patch: -DX3_DIAG_NOLOADA / _NOW / _NOSPLIT switch off the pixel loads, the weight DMA, the
# split + LDS stores) timed on the RpnHead's P2 level and fc1 at 15 images.  Results are WRONG numbers by construction.
#   tools/r05/x2_diag.sh build     (build container)      tools/r05/x2_diag.sh        (GPU box)
cd "$(dirname "$0")/../.."
VARIANTS="${VARIANTS:-NOLOADA NOW NOSPLIT ALL3}"
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    python3 - <<PY
import tools._diag as d
flags = {'ALL3': ['-DX3_DIAG_NOLOADA', '-DX3_DIAG_NOW', '-DX3_DIAG_NOSPLIT']}.get('$v', ['-DX3_DIAG_$v'])
print(d.build_variant('tools/exp/libodet_x3_$v.so', flags, only=['conv_x3.hip'], patch='tools/exp/conv_x3_diag_switches.patch'))
PY
  done
  exit 0
fi
for form in x2 x3; do
  echo "== $form product"; python3 tools/r05/x3_time.py 4 4 1 $form
  for v in $VARIANTS; do
    echo "== $form $v"; ODET_LIB_PATH=$PWD/tools/exp/libodet_x3_$v.so python3 tools/r05/x3_time.py 4 4 1 $form
  done
done
