import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, torch
RUNS = [(f.split(':')[0], int(f.split(':')[1])) for f in sys.argv[1].split(',')] if len(sys.argv) > 1 else \
    [('x3', 30), ('exact', 30), ('x3', 15), ('x3', 8), ('x3', 1)]
for form, b in RUNS:
    r = bench.e2e_record('fp32', b, budget_s=4.0, f32_form=form)
    print(form, b, round(r['value'], 1), 'img/s', r['nms_done'], r['detections_image0'], flush=True)
