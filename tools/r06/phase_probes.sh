#!/bin/bash
# (GPU box) average launch time of the probed kernel per diagnostic build of tools/r06/phase_probes.py, and of the product
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for v in product po_p1 po_p2 po_p3 po_p4 rank_r0 rank_r1 scan_staged scan_walked scan_outputs; do
  if [ $v != product ]; then export ODET_LIB_PATH=$PWD/tools/exp/libodet_probe_$v.so; else unset ODET_LIB_PATH; fi
  rm -rf /tmp/pp; timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d /tmp/pp --output-format csv -- python3 tools/r06/stage_time.py > /dev/null 2>&1
  f=$(find /tmp/pp -name "*_kernel_stats.csv" | head -1)
  case $v in po_*) k=k_postops;; rank_*) k=k_sel_rank;; scan_*) k="k_nms_scan<true>";; *) k="k_postops\|k_sel_rank\|k_nms_scan<true>\|k_roi_pool\|k_rp_prepare\|k_sel_hist2\|k_sel_compact\|k_nms_mask";; esac
  echo "== $v"; python3 - "$f" "$k" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2].replace('\\|', '|'), r['Name']):
        print('  %-44s calls %5s avg %7.1f us min %7.1f max %7.1f' % (r['Name'][:44], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
done
