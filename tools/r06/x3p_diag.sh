#!/bin/bash
# (needs tools/exp/conv_x3_limb_planes.patch applied first; conv_x3p_diag_switches.patch goes on top of it)
# what each part of the limb-plane (PRE) 3x3 K loop costs: diagnostic builds (tools/exp/conv_x3p_diag_switches.patch:
# -DX3P_DIAG_NOX / _NOW switch off the pixel / weight LDS-DMA copies, _ALLEARLY / _ALLLATE move every wave's copies right behind
# the barrier / behind the second MFMA group) timed on the RpnHead's P2 level at 15 images.  WRONG numbers by construction.
#   tools/r06/x3p_diag.sh build     (build container)      tools/r06/x3p_diag.sh        (GPU box)
cd "$(dirname "$0")/../.."
VARIANTS="${VARIANTS:-NOX NOW NOXW ALLEARLY ALLLATE}"
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    python3 - <<PY &
import tools._diag as d
flags = {'NOXW': ['-DX3P_DIAG_NOX', '-DX3P_DIAG_NOW']}.get('$v', ['-DX3P_DIAG_$v'])
print(d.build_variant('tools/exp/libodet_x3p_$v.so', flags, only=['conv_x3.hip'], patch='tools/exp/conv_x3p_diag_switches.patch'))
PY
  done
  wait
  exit 0
fi
for form in x3 x2; do
  echo "== $form product"; python3 tools/r06/x3p_time.py $form
  for v in $VARIANTS; do
    echo "== $form $v"; ODET_LIB_PATH=$PWD/tools/exp/libodet_x3p_$v.so python3 tools/r06/x3p_time.py $form
  done
done
