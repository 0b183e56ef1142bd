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
  * evaluation/detectron_pascal_evaluation_utils.py: voc_ap (numpy only) and voc_eval (numpy + xml + pickle):
    the reference's own VOC matching / precision-recall / AP code, run on a synthetic dataset written to a
    temporary directory in the devkit's file formats (XML annotations, image-set list, per-class result
    files).  `np.bool` (removed from numpy >= 1.24, the reference predates that) is supplied as the alias of
    `bool` it always was.
  * config/fpn_config.py, config/faster_rcnn_config.py (plain Python): the default hyper-parameter dictionaries,
    written to tests/golden/ref_configs.json -- pins the defaults of the hot-path classes.
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


class _NumpyWithBoolAlias:
    """numpy, plus the `np.bool` alias of Python's bool that numpy < 1.24 had (the reference uses it)."""
    bool = bool

    def __getattr__(self, name):
        return getattr(np, name)


def _voc_eval_vectors(rng):
    """Runs the reference's voc_eval (detectron_pascal_evaluation_utils.py:86-222) per class on a synthetic
    dataset in the VOC devkit's file formats; stores the dataset as arrays and (rec, prec, ap) per class."""
    import logging
    import pickle
    import tempfile
    import xml.etree.ElementTree as ET
    ev = _extract(os.path.join(REF, 'object_detection/evaluation/detectron_pascal_evaluation_utils.py'),
                  {'parse_rec', 'voc_ap', 'voc_eval'},
                  {'np': _NumpyWithBoolAlias(), 'os': os, 'ET': ET, 'pickle': pickle,
                   'logger': logging.getLogger('ref_voc_eval')})
    classes = ['aeroplane', 'bicycle', 'bird']
    n_img = 14
    out = {'ve_num_images': np.int64(n_img), 've_num_classes': np.int64(len(classes))}
    gts, dets = [], {c: [] for c in range(len(classes))}
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, 'Annotations'))
        names = ['%06d' % (i + 1) for i in range(n_img)]
        with open(os.path.join(tmp, 'test.txt'), 'w') as f:
            f.write('\n'.join(names) + '\n')
        for i, name in enumerate(names):
            root = ET.Element('annotation')
            for _ in range(int(rng.integers(0, 6))):
                c = int(rng.integers(0, len(classes)))
                x1, y1 = int(rng.integers(0, 300)), int(rng.integers(0, 200))
                w, h = int(rng.integers(20, 180)), int(rng.integers(20, 150))
                diff = int(rng.random() < 0.25)
                gts.append((i, c, x1, y1, x1 + w, y1 + h, diff))
                o = ET.SubElement(root, 'object')
                ET.SubElement(o, 'name').text = classes[c]
                ET.SubElement(o, 'pose').text = 'Unspecified'
                ET.SubElement(o, 'truncated').text = '0'
                ET.SubElement(o, 'difficult').text = str(diff)
                b = ET.SubElement(o, 'bndbox')
                for tag, v in zip(('xmin', 'ymin', 'xmax', 'ymax'), (x1, y1, x1 + w, y1 + h)):
                    ET.SubElement(b, tag).text = str(v)
            ET.ElementTree(root).write(os.path.join(tmp, 'Annotations', name + '.xml'))
        # detections: jittered copies of ground truth (hits, duplicates) + random boxes, distinct confidences
        conf = rng.permutation(4000)[:600].astype(np.float64) / 4000.0
        k = 0
        for (i, c, x1, y1, x2, y2, diff) in gts:
            for _ in range(int(rng.integers(0, 3))):
                j = rng.normal(0, 6, 4)
                dets[c].append((i, conf[k], x1 + j[0], y1 + j[1], x2 + j[2], y2 + j[3])); k += 1
        for c in range(len(classes)):
            for _ in range(25):
                i = int(rng.integers(0, n_img))
                x1, y1 = rng.uniform(0, 350), rng.uniform(0, 250)
                dets[c].append((i, conf[k], x1, y1, x1 + rng.uniform(10, 150), y1 + rng.uniform(10, 150))); k += 1
        for c, cname in enumerate(classes):
            rows = sorted(dets[c], key=lambda r: r[0])              # file order: by image
            with open(os.path.join(tmp, 'det_%s.txt' % cname), 'w') as f:
                for (i, s, x1, y1, x2, y2) in rows:
                    f.write('%s %r %r %r %r %r\n' % (names[i], float(s), float(x1), float(y1), float(x2), float(y2)))
            out['ve_dets_%d' % c] = np.asarray([[r[0], r[2], r[3], r[4], r[5], r[1]] for r in rows], dtype=np.float64)
            for metric, flag in (('07', True), ('area', False)):
                cache = os.path.join(tmp, 'cache_%s_%s' % (cname, metric))
                rec, prec, ap = ev['voc_eval'](os.path.join(tmp, 'det_{:s}.txt'), os.path.join(tmp, 'Annotations', '{:s}.xml'),
                                               os.path.join(tmp, 'test.txt'), cname, cache, ovthresh=0.5, use_07_metric=flag)
                out['ve_rec_%d' % c] = np.asarray(rec, dtype=np.float64)
                out['ve_prec_%d' % c] = np.asarray(prec, dtype=np.float64)
                out['ve_ap%s_%d' % (metric, c)] = np.float64(ap)
    out['ve_gts'] = np.asarray(gts, dtype=np.int64).reshape(-1, 7)      # image, class, x1, y1, x2, y2, difficult
    return out


def _write_configs():
    import json
    cfg = {}
    for name, path, fns in (('fpn', 'object_detection/config/fpn_config.py', ['get_default_pascal_faster_rcnn_config']),
                            ('faster_rcnn', 'object_detection/config/faster_rcnn_config.py',
                             ['get_default_pascal_faster_rcnn_config', 'get_default_coco_faster_rcnn_config'])):
        g = _extract(os.path.join(REF, path), set(fns))
        for fn in fns:
            cfg['%s.%s' % (name, fn)] = g[fn]()
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_configs.json')
    json.dump(cfg, open(dst, 'w'), indent=1, sort_keys=True)
    print('wrote', dst)


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

    out.update(_voc_eval_vectors(rng))
    _write_configs()

    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_numpy_vectors.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, os.path.getsize(dst), 'bytes', len(out), 'arrays')


if __name__ == '__main__':
    sys.exit(main())
