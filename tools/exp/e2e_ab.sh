#!/bin/bash
# same-box A/B of the end-to-end float16 rate: each argument is an environment assignment list ("" = defaults), e.g.
#   tools/exp/e2e_ab.sh "" "ODET_PW=0" "ODET_PW_OFF=fc"      (three runs, interleaved twice)
cd "$(dirname "$0")/../.."
export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0
for rep in 1 2; do
  for cfg in "$@"; do
    r=$(env $cfg python3 tools/e2e_bench.py --dtype ${DT:-fp16} --batch ${BATCH:-8} --steps ${STEPS:-60} --miopen-find 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f img/s  backbone+neck %.2f ms  rpn %.2f ms' % (d['value'], d['ms_backbone_neck_per_batch'], d['ms_rpn_head_per_batch']))")
    echo "[$cfg] $r"
  done
done
