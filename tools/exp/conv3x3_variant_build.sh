#!/bin/bash
# Experimental builds of csrc/conv3x3.hip with extra hipcc flags: tools/exp/conv3x3_variant_build.sh <name> <flags...>
# -> tools/exp/libodet_<name>.so (select with ODET_LIB_PATH).
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
python -m tf_eager_object_detection_amd._build > /dev/null
O=tf_eager_object_detection_amd/csrc/_obj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable "$@" \
  -c tf_eager_object_detection_amd/csrc/conv3x3.hip -o /tmp/conv3x3_$name.o
objs=""
for f in $O/*.hip.o; do case $f in */conv3x3.hip.o) ;; *) objs="$objs $f";; esac; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/conv3x3_$name.o -lpthread -o tools/exp/libodet_$name.so
echo tools/exp/libodet_$name.so
