"""The split-precision kernels (csrc/conv_x3.hip) count their copies in ISSUE order: the counted `s_waitcnt vmcnt(N)` before a
K-step's barrier assumes that, inside every K-step, a wave issues its LDS-DMA weight copies BEFORE its pixel-slot loads.  The
slot loads are plain buffer loads the compiler may move; the source pins the order with compiler barriers.  This test compiles
the file with the product's flags and checks the order in the ISA of every kernel -- a compiler that reorders them would make
the kernels read weight rows that have not landed (caught on the GPU only by the random-data tests, and only sometimes)."""
import os
import re
import shutil
import subprocess

import pytest

from tf_eager_object_detection_amd import _build


def _device_isa(tmp_path):
    hipcc = _build._hipcc()
    if shutil.which(hipcc) is None and not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    out = os.path.join(str(tmp_path), 'conv_x3.s')
    cmd = [hipcc] + _build.HIPCC_FLAGS + _build.PER_SOURCE_FLAGS.get('conv_x3.hip', []) + \
          ['--cuda-device-only', '-S', os.path.join(_build.CSRC, 'conv_x3.hip'), '-o', out]
    subprocess.check_call(cmd)
    return open(out).read()


def test_split_precision_kernels_issue_weight_copies_before_slot_loads(tmp_path):
    isa = _device_isa(tmp_path)
    kernels = re.split(r'\n(?=_Z\d+k_(?:conv3x3|pointwise)_x[23]I)', isa)[1:]
    assert len(kernels) == 18, len(kernels)                       # 5 tiles x 2 (three limbs) + 4 tiles x 2 (two limbs)
    for k in kernels:
        name = k.split(':', 1)[0]
        body = k[:k.find('s_endpgm')]
        seq = []
        for line in body.split('\n'):
            t = line.strip()
            if t.startswith('buffer_load_dwordx4'):
                seq.append('D' if t.endswith(' lds') else 'A')     # LDS-DMA copy / slot load into registers
            elif t.startswith('s_barrier'):
                seq.append('|')
            elif 'Loop Header' in t:
                seq.append('{')
        seq = ''.join(seq)
        assert seq.count('D') >= 8 and seq.count('A') >= 8 and '{' in seq, (name, seq[:80])
        # a K-step = what lies between two barriers; what ends at a loop header is a prologue (slots and weights interleaved
        # by design: the order the steps -NA .. -1 would have issued them in)
        for seg, end in re.findall(r'([DA]*)([|{]|$)', seq):
            if end == '|':
                assert re.fullmatch(r'D*A*', seg), (name, seg)
        # no spills in the K loop's kernels (the tile list is chosen for that)
    spills = [int(v) for v in re.findall(r'\.vgpr_spill_count:\s+(\d+)', isa)]
    assert max(spills) <= 8, spills
