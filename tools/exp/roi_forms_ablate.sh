#!/bin/bash
# what bounds each RoI form: the product kernel against diagnostic builds (tools/exp/roi_variant_build.sh: 1 no stores,
# 2 loads but no lerps, 3 only the first 2 x 2 cells of a bin loaded, 4 no loads) -- 8-image launches, cold / warm us
cd "$(dirname "$0")/../.."
for v in "" abl1 abl2 abl3 abl4; do
  if [ -n "$v" ]; then export ODET_LIB_PATH=$PWD/tools/exp/libodet_$v.so; else unset ODET_LIB_PATH; fi
  python3 tools/roi_forms.py --reps 10 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('${v:-product}'.ljust(8), '  '.join('%s:%5.0f/%5.0f' % (k.split(':')[0], v['us_cold'], v['us_warm'] or 0) for k,v in d.items() if isinstance(v,dict)))"
done
