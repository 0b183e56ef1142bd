#!/bin/bash
# diagnostic builds of conv_x3.hip (tools/exp/historical/conv_x3_wave_specialised.patch applied to a temporary copy: it carries
# the wave-specialised 256 x 128 tile and the X3_DIAG_* switches) timed on two layers: what each part of the K loop costs.
# The patch was cut against commit 1602e9b (before the K split went in): check that commit out to rebuild the variants.
#   tools/r05/x3_diag.sh build   (in the build container: tools/exp/libodet_x3_<variant>.so)
#   X3_TILE="8 2" tools/r05/x3_diag.sh         (on the GPU box; X3_TILE = the tile forced through odet_debug_x3_tile)
# round 5, rpn P2 / fc1 at 15 images, TFLOP/s-equivalent: product (256 x 128 tile) 217 / 212; no pixel loads 204 / 202; no weight
# DMA 233 / 228; no split + store 235 / 231; MFMA skeleton alone 309 / 296.  Wave-specialised tile: 221 / 219; its producers
# idle 286 / 280, without the weight DMA 231 / 228, without split + store 253 / 253.
cd "$(dirname "$0")/../.."
VARIANTS="${VARIANTS:-BASE NOLOADA NOW NOSPLIT ALL3 WS_IDLE WS_NOW WS_NOSTORE}"
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    python3 - <<PY
import tools._diag as d
flags = {'ALL3': ['-DX3_DIAG_NOLOADA', '-DX3_DIAG_NOW', '-DX3_DIAG_NOSPLIT'], 'BASE': []}.get('$v', ['-DX3_DIAG_$v'])
print(d.build_variant('tools/exp/libodet_x3_$v.so', flags, only=['conv_x3.hip'], patch='tools/exp/historical/conv_x3_wave_specialised.patch'))
PY
  done
  exit 0
fi
for v in $VARIANTS; do
  echo "== $v"; ODET_LIB_PATH=$PWD/tools/exp/libodet_x3_$v.so python3 tools/r05/x3_time.py $X3_TILE
done
