#!/bin/bash
# round 4: the float32 carrying RoI walk with the next bin's loads in flight (-DODET_ROI_CARRY_PREFETCH) as a side library
#   tools/r04/roi_carry_prefetch_build.sh  ->  tools/exp/libodet_roi_pf.so   (select with ODET_LIB_PATH)
set -e
cd "$(dirname "$0")/../.."
python -m tf_eager_object_detection_amd._build > /dev/null
O=tf_eager_object_detection_amd/csrc/_obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable -Iinclude"
hipcc $FLAGS -fno-slp-vectorize -DODET_ROI_CARRY_PREFETCH -c tf_eager_object_detection_amd/csrc/roi.hip -o /tmp/roi_pf.o
objs=""; for f in $O/*.hip.o; do case $f in */roi.hip.o) ;; *) objs="$objs $f";; esac; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/roi_pf.o -lpthread -o tools/exp/libodet_roi_pf.so
ls -la tools/exp/libodet_roi_pf.so
