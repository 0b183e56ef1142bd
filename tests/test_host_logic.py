"""CPU tests of the product's host side: the numpy-level anchor tables against the reference's
golden vectors, the C-ABI library (loads, exports every symbol include/odet.h declares, ctypes
table in sync with the header) and the fail-loudly rule (no CPU path)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_anchor_base_matches_reference_golden(golden):
    from tf_eager_object_detection_amd.utils.anchor_generator import generate_anchor_base
    for i in range(5):
        got = generate_anchor_base(float(golden['ab%d_base' % i]), golden['ab%d_ratios' % i],
                                   golden['ab%d_scales' % i])
        np.testing.assert_array_equal(got, golden['ab%d_out' % i])
    assert generate_anchor_base().shape == (9, 4)


def test_product_anchor_by_base_np_matches_reference_golden(golden):
    from tf_eager_object_detection_amd.utils.anchor_generator import generate_anchor_base, generate_by_anchor_base_np
    base = generate_anchor_base(16, [0.5, 1, 2], [8, 16, 32])
    for i in range(3):
        h, w, st = golden['np%d_hws' % i]
        np.testing.assert_array_equal(generate_by_anchor_base_np(base, int(st), int(h), int(w)),
                                      golden['np%d_out' % i])


def test_wh_table_is_float32_and_swapped():
    from tf_eager_object_detection_amd.utils.anchor_generator import _wh_table
    wh = _wh_table(32, (1.,), (0.5, 1.0, 2.0))
    assert wh.dtype == np.float32 and wh.shape == (3, 2)
    # ratio 0.5 -> w = 32*sqrt(.5) = 22.627, h = 32/sqrt(.5) = 45.25  (w/h = ratio)
    np.testing.assert_allclose(wh[0], [22.627417, 45.254833], rtol=1e-6)
    np.testing.assert_array_equal(wh[1], np.float32([32, 32]))


def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'odet.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(odet_[a-z0-9_]+)\s*\(', text)))


def test_library_loads_and_exports_every_header_symbol():
    from tf_eager_object_detection_amd import _lib
    handle = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 20
    for name in syms:
        assert hasattr(handle, name), 'libodet_hip.so does not export %s' % name
    assert handle.odet_version() == 103


def test_shipped_library_has_no_debug_hooks():
    """the tile-forcing diagnostics are not in the product ABI -- not in include/odet.h, not exported by
    libodet_hip.so, not in the ctypes table; they live in include/odet_diag.h and the separate -DODET_DIAG build"""
    from tf_eager_object_detection_amd import _build, _lib
    assert not [n for n in _header_symbols() if 'debug' in n]
    assert not [n for n in _lib.SIGNATURES if 'debug' in n]
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert 'debug' not in out
    exported = sorted(set(re.findall(r'\b(odet_[a-z0-9_]+)\b', out)))
    assert exported == _header_symbols()                  # the library exports exactly what the header declares
    diag = open(os.path.join(ROOT, 'include', 'odet_diag.h')).read()
    assert 'odet_debug_conv_tile' in diag and 'odet_debug_x3_tile' in diag
    lib = _build.build_diag()
    out = subprocess.run(['nm', '-D', '--defined-only', lib], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert 'odet_debug_conv_tile' in out and 'odet_debug_x3_tile' in out


def test_f32_form_is_per_thread_context():
    """the float32 form is ambient state of the calling CONTEXT (contextvars), not a process global: two threads
    entering and leaving f32_form blocks in the interleaving A-enter, B-enter, A-exit, B-exit each keep their own form, and
    nothing is left switched afterwards"""
    import threading
    from tf_eager_object_detection_amd import ops
    seen, steps = {}, [threading.Event() for _ in range(4)]

    def a():
        with ops.f32_form('x3'):
            steps[0].set(); steps[1].wait(5)
            seen['a_inside'] = ops.current_f32_form()
        steps[2].set()
        seen['a_after'] = ops.current_f32_form()

    def b():
        steps[0].wait(5)
        with ops.f32_form('x2'):
            steps[1].set(); steps[2].wait(5)
            seen['b_inside'] = ops.current_f32_form()
        seen['b_after'] = ops.current_f32_form()

    ts = [threading.Thread(target=a), threading.Thread(target=b)]
    [t.start() for t in ts]
    [t.join(10) for t in ts]
    assert seen == {'a_inside': 'x3', 'a_after': 'exact', 'b_inside': 'x2', 'b_after': 'exact'}
    assert ops.current_f32_form() == 'exact'
    with pytest.raises(ValueError):
        ops.f32_form('x4')


def test_ctypes_table_matches_header():
    from tf_eager_object_detection_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_symbols()
    text = open(os.path.join(ROOT, 'include', 'odet.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r'\b%s\s*\(([^;]*?)\)\s*;' % name, text, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ('', 'void') else len(params.split(','))
        assert n == len(args), '%s: header has %d parameters, ctypes table %d' % (name, n, len(args))


def test_argument_errors_are_reported_through_the_abi():
    from tf_eager_object_detection_amd import _lib
    L = _lib.lib()
    rc = L.odet_roi_pool(None, 1, 256, None, None, 4, None, 0, 0, 0, 7, 1, None, None)
    assert rc == -1 and b'null pointer' in L.odet_last_error()
    rc = L.odet_anchors_fpn(99, 3, None, None, None, None, None, None)
    assert rc == -1
    assert L.odet_nms_workspace_bytes(267069, 1000) > 267069 * 16
    assert L.odet_post_ops_workspace_bytes(21, 50) > 20 * 50 * 20


def test_no_cpu_fallback():
    from tf_eager_object_detection_amd import _lib
    from tf_eager_object_detection_amd.utils.bbox_transform import decode_bbox_with_mean_and_std
    from tf_eager_object_detection_amd.model.region_proposal import RegionProposal
    a = torch.zeros(4, 4)
    with pytest.raises(_lib.OdetError):
        decode_bbox_with_mean_and_std(a, a, [0, 0, 0, 0], [1, 1, 1, 1])
    with pytest.raises(_lib.OdetError):
        RegionProposal()((a, a, torch.zeros(4), [32, 32]), training=False)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'tf_eager_object_detection_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert 'oracle/' not in src and 'liboracle' not in src, f


def test_synthetic_shapes():
    from tf_eager_object_detection_amd import synthetic as syn
    assert syn.num_fpn_anchors((800, 1333)) == 267069
    assert syn.num_fpn_anchors((1333, 1333)) == 446118
    assert syn.fpn_level_shapes((800, 1333)) == [(200, 334), (100, 167), (50, 84), (25, 42), (13, 21)]
    rng = np.random.default_rng(0)
    s = syn.scores_distinct(1000, rng)
    assert len(np.unique(s)) == 1000 and s.dtype == np.float32


def test_hot_path_defaults_pinned_by_reference_configs():
    """PIN: the default hyper-parameters of the hot-path classes against the reference's own config dictionaries
    (config/fpn_config.py, config/faster_rcnn_config.py, dumped by tests/golden/make_ref_vectors.py)."""
    import inspect, json, os
    from tf_eager_object_detection_amd import synthetic as syn
    from tf_eager_object_detection_amd.pipeline import FpnHotPath, FrcnnHotPath
    cfg = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'ref_configs.json')))
    fpn = cfg['fpn.get_default_pascal_faster_rcnn_config']
    d = {k: v.default for k, v in inspect.signature(FpnHotPath.__init__).parameters.items()}
    assert d['num_classes'] == fpn['num_classes'] and d['num_proposals'] == fpn['rpn_proposal_test_after_nms_sample_number']
    assert d['channels'] == fpn['top_down_dims'] == fpn['resnet_roi_feature_size'][2] and d['pool_size'] == fpn['roi_pooling_size']
    assert d['rpn_nms_iou'] == fpn['rpn_proposal_nms_iou_threshold']
    assert list(d['rpn_means']) == fpn['rpn_proposal_means'] and list(d['rpn_stds']) == fpn['rpn_proposal_stds']
    assert list(d['roi_means']) == fpn['roi_proposal_means'] and list(d['roi_stds']) == fpn['roi_proposal_stds']
    assert d['max_per_class'] == fpn['max_objects_per_class_per_image'] and d['max_per_image'] == fpn['max_objects_per_image']
    assert d['nms_iou'] == fpn['prediction_nms_iou_threshold'] and d['score_threshold'] == fpn['prediction_score_threshold']
    assert d['min_level'] == fpn['min_level'] and d['max_level'] == fpn['max_level']
    assert list(syn.FPN_STRIDES) == fpn['anchor_stride_list'] and list(syn.FPN_BASE_SIZES) == fpn['base_anchor_size_list']
    assert list(syn.FPN_RATIOS) == fpn['ratios'] and list(syn.FPN_SCALES) == fpn['scales']
    assert fpn['roi_pooling_max_pooling_flag'] is True              # FpnHotPath: 14x14 crop + 2x2 max
    fr = cfg['faster_rcnn.get_default_pascal_faster_rcnn_config']
    d = {k: v.default for k, v in inspect.signature(FrcnnHotPath.__init__).parameters.items()}
    assert d['num_classes'] == fr['num_classes'] and d['num_proposals'] == fr['rpn_proposal_test_after_nms_sample_number']
    assert d['pool_size'] == fr['roi_pooling_size'] and d['extractor_stride'] == fr['extractor_stride']
    assert list(d['ratios']) == fr['ratios'] and list(d['scales']) == fr['scales']
    assert d['rpn_nms_iou'] == fr['rpn_proposal_nms_iou_threshold']
    assert list(d['rpn_means']) == fr['rpn_proposal_means'] and list(d['rpn_stds']) == fr['rpn_proposal_stds']
    assert list(d['roi_means']) == fr['roi_proposal_means'] and list(d['roi_stds']) == fr['roi_proposal_stds']
    assert d['max_per_class'] == fr['max_objects_per_class_per_image'] and d['max_per_image'] == fr['max_objects_per_image']
    assert d['nms_iou'] == fr['prediction_nms_iou_threshold'] and d['score_threshold'] == fr['prediction_score_threshold']
    assert d['max_pooling_flag'] == fr['resnet_roi_pooling_max_pooling_flag']
    from tf_eager_object_detection_amd.model.frcnn_detector import ResNetC4Detector, Vgg16Detector
    assert ResNetC4Detector(50, 21, (64, 64), 10)._hot_kwargs['max_pooling_flag'] == fr['resnet_roi_pooling_max_pooling_flag']
    assert Vgg16Detector(21, (64, 64), 10)._hot_kwargs['max_pooling_flag'] == fr['vgg16_roi_pooling_max_pooling_flag']


def test_bench_self_launch_starts_torchrun_children(monkeypatch):
    """`python bench.py --gpus N` without a launcher's environment starts `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>` as a child process (never an exec) and returns
    its exit code; with WORLD_SIZE set (the driver's torchrun form) it does not."""
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run'] and '--nnodes=1' in cmd
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-7:] == [os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '3', '--warmup', '1']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert 'os.exec' not in src and 'execv' not in src


def test_product_reads_no_environment_switches():
    """nothing in the package selects a library, compiler flags or a code path from the environment"""
    import glob
    for f in glob.glob(os.path.join(ROOT, 'tf_eager_object_detection_amd', '**', '*.py'), recursive=True):
        src = open(f).read()
        assert 'ODET_LIB_PATH' not in src and 'ODET_EXTRA_HIPCC_FLAGS' not in src, f
        for line in src.splitlines():
            if 'os.environ' in line:
                assert 'HIPCC' in line and 'ODET' not in line, (f, line)      # (_build.py: the compiler's location only)
