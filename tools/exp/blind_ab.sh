export MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD=0
for bc in 3 2 3 2; do
  python tools/e2e_bench.py --dtype fp16 --batch 30 --steps 20 --miopen-find --blind-chunks $bc 2>/dev/null | tail -1 > /tmp/o.json
  python -c "
import json
d=json.load(open('/tmp/o.json')); print('blind $bc', round(d['value'],1), sum(d['nms_done']), d['detections_image0'])"
done
