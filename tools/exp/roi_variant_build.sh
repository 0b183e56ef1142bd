#!/bin/bash
# Diagnostic / experimental builds of the RoI kernel, outside the product tree:
#   tools/exp/roi_variant_build.sh <name> <extra hipcc flags...>   -> tools/exp/libodet_<name>.so
# The product source has ONE code path; the switches live in tools/exp/roi_diag_switches.patch, applied here to a
# temporary copy of csrc/:  -DODET_ROI_ABLATE=1 (no stores) | 2 (loads, no lerps) | 3 (only the first 2 x 2 cells of a
# bin are loaded) | 4 (no loads);  -DODET_ROI_LOAD_AUX=<cache policy bits of the cell loads: 1 sc0, 2 nt, 16 sc1>, -DODET_ROI_STORE_AUX=<the same
# for the float32 feature stores, product: 2>, -DODET_ROI_CARRY_COLS=<0, 1: columns a bin may take from its left neighbour, product 2>;
# at run time ODET_ROI_DEV_LDS=<bytes> (dynamic LDS to limit workgroups per CU), ODET_ROI_DEV_NOPIN=1 (no image ->
# XCD pinning in batched launches).  Same ABI: select with ODET_LIB_PATH (tools/roi_bench.py, tools/pmc_roi.sh).
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
python -m tf_eager_object_detection_amd._build > /dev/null
T=$(mktemp -d)
mkdir -p $T/pkg/csrc $T/include
cp tf_eager_object_detection_amd/csrc/*.hip tf_eager_object_detection_amd/csrc/*.h $T/pkg/csrc/
cp include/*.h $T/include/
(cd $T/pkg/csrc && patch -s -p0 roi.hip < "$OLDPWD/tools/exp/roi_diag_switches.patch")
sed -i 's#"../../include/odet.h"#"'$T'/include/odet.h"#' $T/pkg/csrc/odet_internal.h
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable"
hipcc $FLAGS ${ROI_SLP:--fno-slp-vectorize} "$@" -c $T/pkg/csrc/roi.hip -o $T/roi.o &
hipcc $FLAGS "$@" -c $T/pkg/csrc/roi_half.hip -o $T/roi_half.o &
wait
O=tf_eager_object_detection_amd/csrc/_obj
objs=""
for f in $O/*.hip.o; do case $f in */roi.hip.o|*/roi_half.hip.o) ;; *) objs="$objs $f";; esac; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs $T/roi.o $T/roi_half.o -lpthread -o tools/exp/libodet_$name.so
rm -rf $T
echo tools/exp/libodet_$name.so
