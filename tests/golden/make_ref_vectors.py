"""Generates tests/golden/ref_numpy_vectors.npz by EXECUTING the reference's own TF-free code.

Runs only in the build container (needs /root/reference); the GPU box and the test-suite
only read the committed .npz.  Nothing from the reference is copied into the repo: the
functions are pulled out of the reference files with `ast` at run time, executed with numpy,
and only their inputs / outputs are stored.

What can be executed without TensorFlow:
  * utils/anchor_generator.py: generate_anchor_base (+ _whctrs/_mkanchors/_ratio_enum/
    _scale_enum) and generate_by_anchor_base_np  -- module imports TF on line 2, so the
    numpy-only function defs are extracted by name.
  * utils/bbox_np.py (pure numpy; y,x,y,x convention, +1 areas): pairwise_iou, used to pin
    the +1 IoU formula of utils/bbox_tf.py:pairwise_iou on transposed-coordinate inputs.
  * evaluation/detectron_pascal_evaluation_utils.py: voc_ap (numpy only).
"""
import ast
import os
import sys

import numpy as np

REF = os.environ.get('ODET_REFERENCE', '/root/reference')


def _extract(path, names, extra_globals=None):
    src = open(path).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    mod = ast.Module(body=keep, type_ignores=[])
    g = {'np': np, 'range': range}
    if extra_globals:
        g.update(extra_globals)
    exec(compile(mod, path, 'exec'), g)
    return g


def main():
    out = {}
    ag = _extract(os.path.join(REF, 'object_detection/utils/anchor_generator.py'),
                  {'generate_anchor_base', '_whctrs', '_mkanchors', '_ratio_enum', '_scale_enum',
                   'generate_by_anchor_base_np'})
    cases = [(16, [0.5, 1, 2], [8, 16, 32]),
             (16, [0.5, 1, 2], [4, 8, 16, 32]),
             (32, [0.5, 1.0, 2.0], [1]),
             (8, [0.25, 0.5, 1, 2, 4], [2, 4]),
             (17, [0.333, 1, 3], [1, 2.5])]
    for i, (b, r, s) in enumerate(cases):
        out['ab%d_base' % i] = np.float64(b)
        out['ab%d_ratios' % i] = np.asarray(r, dtype=np.float64)
        out['ab%d_scales' % i] = np.asarray(s, dtype=np.float64)
        out['ab%d_out' % i] = ag['generate_anchor_base'](b, r, np.array(s))
    base = ag['generate_anchor_base'](16, [0.5, 1, 2], np.array([8, 16, 32]))
    for i, (h, w, st) in enumerate([(32, 48, 16), (600, 800, 16), (37, 50, 8)]):
        out['np%d_hws' % i] = np.asarray([h, w, st], dtype=np.int64)
        out['np%d_out' % i] = ag['generate_by_anchor_base_np'](base, st, h, w)

    bn = _extract(os.path.join(REF, 'object_detection/utils/bbox_np.py'),
                  {'area', 'intersection', 'pairwise_iou'})
    rng = np.random.default_rng(7)
    def boxes(n):
        c = rng.uniform(0, 300, size=(n, 2))
        wh = rng.uniform(1, 120, size=(n, 2))
        return np.concatenate([c - wh / 2, c + wh / 2], axis=1).astype(np.float32)
    b1, b2 = boxes(37), boxes(23)
    b2[3] = b1[5]                      # identical pair
    b2[4] = b1[6] + np.float32(500)    # disjoint
    out['iou_b1'] = b1
    out['iou_b2'] = b2
    out['iou_out'] = bn['pairwise_iou'](b1, b2).astype(np.float32)

    ev = _extract(os.path.join(REF, 'object_detection/evaluation/detectron_pascal_evaluation_utils.py'),
                  {'voc_ap'})
    for i in range(4):
        n = 5 + 11 * i
        tp = rng.integers(0, 2, size=n)
        fp = 1 - tp
        ctp, cfp = np.cumsum(tp), np.cumsum(fp)
        rec = ctp / float(max(tp.sum() + i, 1))
        prec = ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
        out['ap%d_rec' % i] = rec
        out['ap%d_prec' % i] = prec
        out['ap%d_07' % i] = np.float64(ev['voc_ap'](rec, prec, True))
        out['ap%d_area' % i] = np.float64(ev['voc_ap'](rec, prec, False))

    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_numpy_vectors.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, os.path.getsize(dst), 'bytes', len(out), 'arrays')


if __name__ == '__main__':
    sys.exit(main())
