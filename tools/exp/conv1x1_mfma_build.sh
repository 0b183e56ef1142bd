#!/bin/bash
# builds tools/exp/_ablate/libodet_hip_conv1x1.so = the product objects + the experimental MFMA 1x1 convolution
set -e
cd "$(dirname "$0")/../.."
python -c "import __graft_entry__ as g; g.build()" >/dev/null
O=tf_eager_object_detection_amd/csrc/_obj
mkdir -p tools/exp/_ablate
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude \
    -Itf_eager_object_detection_amd/csrc -c tools/exp/conv1x1_mfma.hip -o /tmp/conv1x1_mfma.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/*.hip.o /tmp/conv1x1_mfma.o -lpthread \
    -o tools/exp/_ablate/libodet_hip_conv1x1.so
echo built tools/exp/_ablate/libodet_hip_conv1x1.so
