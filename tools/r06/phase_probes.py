#!/usr/bin/env python3
"""round 6: where the latency arrangement's launches spend their time -- early-exit diagnostic builds of k_postops, k_sel_rank and
k_nms_scan<true> (a `return` inserted at a marker line of a temporary copy of the source; results are WRONG by construction).
    python tools/r06/phase_probes.py build        (build container)  ->  tools/exp/libodet_probe_<name>.so
    tools/r06/phase_probes.sh                     (GPU box)          ->  average launch time per variant"""
import os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tf_eager_object_detection_amd import _build
import tools._diag as d

PROBES = {
    # name: (source, marker line, early exit inserted BEFORE the marker)
    'po_p1': ('postops.hip', "  // 2. sort: score desc, RoI index asc; rejected rows (key = ~0) go last\n",
              "  if (p.K != 123456) { if (mykey == 12345ull) cls_count[blockIdx.x] = (int)mykey; return; }\n"),
    'po_p2': ('postops.hip', "  // 3. greedy NMS in rounds of PO_ROUND sorted candidates\n",
              "  if (p.K != 123456) { if (mykey == 12345ull) cls_count[blockIdx.x] = (int)mykey; return; }\n"),
    'po_p3': ('postops.hip', "  // 4. publish this class's list, draw a ticket; the last workgroup of the image merges.", "  if (p.K != 123456) return;\n"),
    'po_p4': ('postops.hip', "  if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"agent\");\n", "  if (p.K != 123456) return;\n"),
    'rank_r1': ('nms.hip', "  part[w][lane] = c;\n", "  if (n != -123456) { if (c == 123456789) sel_idx[0] = 1u; return; }\n"),
    'rank_r0': ('nms.hip', "  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;\n  const int i = blockIdx.x * 64 + lane;\n", "  if (n != -123456) return;\n"),
    'scan_staged': ('nms.hip', "    if (b_own < nblk) {\n      for (int b = 0; b < b_own; ++b) {\n        if (stop) break;\n        // words of (source block b -> own blocks)", "    if (K != -123456) return;\n"),
    'scan_walked': ('nms.hip', "  // outputs: kept candidates in score order\n", "  if (K != -123456) return;\n"),
    'scan_outputs': ('nms.hip', "  if (fused_lv) {\n    // _assign_levels (base_fpn_model.py:303-324) as a stable partition by level without re-reading", "  if (K != -123456) return;\n"),
}

if __name__ == '__main__':
    for name, (src, marker, stop) in PROBES.items():
        path = os.path.join(_build.CSRC, src)
        bak = '/tmp/probe_' + src
        shutil.copy(path, bak)
        s = open(path).read()
        assert s.count(marker) >= 1, (name, marker[:40])
        open(path, 'w').write(s.replace(marker, stop + marker, 1))
        try:
            print(d.build_variant('tools/exp/libodet_probe_%s.so' % name, [], only=[src]))
        finally:
            shutil.copy(bak, path)
