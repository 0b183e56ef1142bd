cd /root/repo
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" "TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $set -d /tmp/pmc_tlb/p$i --output-format csv -- python3 tools/roi_ablate.py > /tmp/pmc_tlb_p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob('/tmp/pmc_tlb/p*/*/*_counter_collection.csv')):
    rows=[r for r in csv.DictReader(open(f)) if 'k_roi_pool' in r['Kernel_Name']]
    by=collections.defaultdict(list)
    for r in rows: by[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for k,v in by.items():
        v.sort(); vals=[x for _,x in v]; n=len(vals)//2
        print('%-50s warm %.4g cold %.4g' % (k, sum(vals[2:n])/len(vals[2:n]), sum(vals[n+1:])/len(vals[n+1:])))
PY
