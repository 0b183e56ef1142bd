for fl in write read; do
for v in intree st0 st1 st16 st17; do
  if [ $v = intree ]; then unset ODET_LIB_PATH; else export ODET_LIB_PATH=tools/exp/libodet_$v.so; fi
  python tools/roi_bench.py --tag $v --flush $fl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['tag'], d['flush'], 'cold %.1f warm %.1f one %.1f/%.1f' % (d['batch_cold']['median_us'], d['batch_warm']['median_us'], d['one_cold']['median_us'], d['one_warm']['median_us']), d['features_sha256'][:12])"
done; done
