#!/bin/bash
# diagnostic builds of csrc/conv1x1.hip: tools/exp/_ablate/libodet_hip_c1a{1,2,3}.so
# (the product source has ONE code path; the switches live in tools/exp/conv1x1_diag_switches.patch, applied to a temporary copy)
#   -DODET_C1_ABLATE=1 no output stores | 2 weight group loaded once | 3 bias / shortcut loaded once |
#   4 LDS reads but no MFMAs | 5 no k-loop at all
set -e
cd "$(dirname "$0")/../.."
python -c "import __graft_entry__ as g; g.build()" >/dev/null
O=tf_eager_object_detection_amd/csrc/_obj
mkdir -p tools/exp/_ablate
T=$(mktemp -d)
mkdir -p $T/pkg/csrc $T/include
cp tf_eager_object_detection_amd/csrc/conv1x1.hip tf_eager_object_detection_amd/csrc/*.h $T/pkg/csrc/
cp include/*.h $T/include/
(cd $T/pkg/csrc && patch -s -p0 conv1x1.hip < "$OLDPWD/tools/exp/conv1x1_diag_switches.patch")
sed -i 's#"../../include/odet.h"#"'$T'/include/odet.h"#' $T/pkg/csrc/odet_internal.h
for k in ${ODET_C1_VARIANTS:-1 2 3 4 5}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude \
      -DODET_C1_ABLATE=$k -c $T/pkg/csrc/conv1x1.hip -o /tmp/conv1x1_a$k.o
  objs=$(ls $O/*.hip.o | grep -v conv1x1)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/conv1x1_a$k.o -lpthread -o tools/exp/_ablate/libodet_hip_c1a$k.so
done
rm -rf $T
echo built
