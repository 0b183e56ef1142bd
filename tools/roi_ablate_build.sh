#!/bin/bash
# Diagnostic builds of the library with roi.hip recompiled under extra flags:
#   tools/roi_ablate_build.sh                    -> tools/exp/_ablate/libodet_hip_a{1..5}.so (-DODET_ROI_ABLATE=k)
#   tools/roi_ablate_build.sh NAME FLAGS...      -> tools/exp/_ablate/libodet_hip_NAME.so
set -e
cd "$(dirname "$0")/.."
python -m tf_eager_object_detection_amd._build >/dev/null
O=tf_eager_object_detection_amd/csrc/_obj
D=tools/exp/_ablate
mkdir -p $D
one() {   # name flags...
  local name=$1; shift
  /opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
    -c tf_eager_object_detection_amd/csrc/roi.hip -o $D/roi_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/boxes.hip.o $O/sort.hip.o $O/nms.hip.o $D/roi_$name.o \
    $O/postops.hip.o $O/neck.hip.o $O/epilogue.hip.o $O/executor.hip.o -lpthread -o $D/libodet_hip_$name.so
  rm -f $D/roi_$name.o
}
if [ $# -ge 1 ]; then one "$@"; else for k in 1 2 3 4 5; do one a$k -DODET_ROI_ABLATE=$k & done; wait; fi
ls $D
