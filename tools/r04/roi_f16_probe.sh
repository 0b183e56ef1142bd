#!/bin/bash
# round 4 probe: what bounds the float16-map RoI launch (config 5 arrangement: 164 us for 540 MB)?  Variants of the kernel built
# outside the product tree, timed with tools/roi_forms.py --form 1 (ODET_LIB_PATH selects the library):
#   wide    the cell loads fetch 16 bytes per lane instead of 8 (same instruction count, twice the bytes per instruction)
#   nostore the feature stores are skipped behind a condition no value meets (the arithmetic stays)
#   half    only the first two cell columns of a bin are loaded (the others re-use them): fewer load instructions
set -e
cd "$(dirname "$0")/../.."
python -m tf_eager_object_detection_amd._build > /dev/null
O=tf_eager_object_detection_amd/csrc/_obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable"
for v in wide nostore half; do
  T=$(mktemp -d); mkdir -p $T/csrc $T/include
  cp tf_eager_object_detection_amd/csrc/*.hip tf_eager_object_detection_amd/csrc/*.h $T/csrc/; cp include/*.h $T/include/
  sed -i 's#"../../include/odet.h"#"'$T'/include/odet.h"#' $T/csrc/odet_internal.h
  python3 - $T/csrc/roi.hip $v <<'PY'
import sys
p, v = sys.argv[1], sys.argv[2]
s = open(p).read()
if v == 'wide':
    old = "    return __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);"
    new = "    const u4v w_ = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0); raw_t o_; o_.x = w_.x ^ (w_.z & 0u); o_.y = w_.y ^ (w_.w & 0u); return o_;"
elif v == 'nostore':
    old = "    __builtin_amdgcn_raw_buffer_store_b64(u, r, (int)voff, (int)soff, 0);   // (float16 features are read back by the RoI head)"
    new = "    if (v.x == 12345.678f) __builtin_amdgcn_raw_buffer_store_b64(u, r, (int)voff, (int)soff, 0);"
else:
    old = "      blk[i][j] = Cell<FT>::load_raw(rc.feat, voff[j], soff[i]);"
    new = "      blk[i][j] = (j < 2) ? Cell<FT>::load_raw(rc.feat, voff[j], soff[i]) : blk[i][j - 2];"
assert old in s, v
open(p, 'w').write(s.replace(old, new))
PY
  hipcc $FLAGS -c $T/csrc/roi_half.hip -o $T/roi_half.o
  objs=""; for f in $O/*.hip.o; do case $f in */roi_half.hip.o) ;; *) objs="$objs $f";; esac; done
  hipcc --offload-arch=gfx950 -shared -fPIC $objs $T/roi_half.o -lpthread -o tools/exp/libodet_f16_$v.so
  rm -rf $T
done
ls -la tools/exp/libodet_f16_*.so
