#!/usr/bin/env python3
"""MFMA-busy of the dense path (the library convolutions / GEMMs of the assembled detector) from one
`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE` pass over
tools/e2e_bench.py.  Only the steady-state tail of the run is used (MIOpen's find mode benchmarks every
solver first): the last `--tail` fraction of the dispatches.

    python tools/mfma_busy.py <counter_collection.csv> --out profiles/<name>.json"""
import argparse, collections, csv, json

ap = argparse.ArgumentParser()
ap.add_argument('csv')
ap.add_argument('--tail', type=float, default=0.25)
ap.add_argument('--out', required=True)
a = ap.parse_args()
per = collections.defaultdict(dict)
names = {}
with open(a.csv) as f:
    for r in csv.DictReader(f):
        d = int(r['Dispatch_Id'])
        per[d][r['Counter_Name']] = per[d].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        names[d] = r['Kernel_Name']
ids = sorted(per)
ids = ids[int(len(ids) * (1.0 - a.tail)):]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
tot = collections.defaultdict(float)
for d in ids:
    n = names[d]
    fam = ('odet_mfma' if any(k in n for k in ('k_conv1x1', 'k_rpn_tail', 'k_conv3x3', 'k_stem', 'k_pointwise')) else
           'mfma_conv_gemm' if any(k in n for k in ('igemm', 'ck16', 'Cijk', 'gemm', 'conv', 'Conv')) and 'naive' not in n else
           'odet_hip' if (n.startswith('k_') or 'k_roi' in n or 'k_rp_' in n or 'k_nms' in n or 'k_fpn' in n or 'k_bias' in n) else 'other')
    for c, v in per[d].items():
        agg[fam][c] += v
        tot[c] += v
    agg[fam]['dispatches'] += 1
out = {'source': a.csv, 'tail_fraction': a.tail, 'dispatches_used': len(ids), 'families': {}}
for fam, c in agg.items():
    busy, cu = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), c.get('SQ_BUSY_CU_CYCLES', 0.0)
    out['families'][fam] = {'dispatches': int(c['dispatches']), 'SQ_VALU_MFMA_BUSY_CYCLES': busy, 'SQ_BUSY_CU_CYCLES': cu,
                            'GRBM_GUI_ACTIVE': c.get('GRBM_GUI_ACTIVE', 0.0), 'MFMA_MOPS_F16': c.get('SQ_INSTS_VALU_MFMA_MOPS_F16', 0.0),
                            'mfma_busy_over_cu_busy': (busy / cu) if cu else None,
                            'mfma_busy_fraction_of_simd_cycles': (busy / cu / 4.0) if cu else None}
out['note'] = ('SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD, SQ_BUSY_CU_CYCLES per CU (4 SIMDs), both summed over the chip: '
               'ratio / 4 = fraction of the SIMD cycles of busy CUs with the MFMA pipe busy (gfx94x derivation; ROCm 7.2 ships '
               'no gfx950 derived-counter section)')
json.dump(out, open(a.out, 'w'), indent=1)
print(json.dumps(out['families'], indent=1))
