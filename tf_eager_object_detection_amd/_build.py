"""Builds libodet_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is
git-ignored but travels to the GPU box inside the gpurun snapshot."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
INCLUDE = os.path.join(os.path.dirname(PKG), 'include')
OBJ_DIR = os.path.join(CSRC, '_obj')
LIB = os.path.join(PKG, 'libodet_hip.so')
SOURCES = ['boxes.hip', 'sort.hip', 'nms.hip', 'roi.hip', 'roi_half.hip', 'postops.hip', 'neck.hip', 'epilogue.hip', 'conv1x1.hip', 'conv3x3.hip', 'conv_f32.hip', 'conv_x3.hip', 'rpn_tail.hip',
           'calib.hip', 'stem.hip', 'executor.hip']
HEADERS = [os.path.join(CSRC, 'odet_internal.h'), os.path.join(CSRC, 'conv_f32_common.h'), os.path.join(INCLUDE, 'odet.h')]

# -ffp-contract=off: the parity contract is "one IEEE float32 operation per reference
# operation"; an FMA would change low bits of box coordinates and bilinear taps.
HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
               '-fno-fast-math', '-Wall', '-Wno-unused-function', '-Wno-unused-variable']


# per-source flags: the RoI walk is bound by vector-instruction issue and gfx950's packed float32 operations
# (which the SLP vectoriser produces, plus the v_mov shuffles to feed them) issue slower than two plain ones:
# -1.5 % kernel time, +1.2 % throughput (same-box A/B); results are the same IEEE operations either way
PER_SOURCE_FLAGS = {'roi.hip': ['-fno-slp-vectorize']}
# sources that #include another source: rebuilt when that one changes
EXTRA_DEPS = {'roi_half.hip': ['roi.hip']}


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, src + '.o')
        objs.append(o)
        if force or _stale(o, [s] + [os.path.join(CSRC, d) for d in EXTRA_DEPS.get(src, [])] + HEADERS + [os.path.abspath(__file__)]):
            cmd = [hipcc] + HIPCC_FLAGS + PER_SOURCE_FLAGS.get(src, []) + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % src)
    if force or procs or _stale(LIB, objs):
        tmp = LIB + '.tmp.%d' % os.getpid()
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-lpthread', '-o', tmp]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    return LIB


# ---- the DIAGNOSTIC build: the same sources, the two files with tile-forcing hooks compiled with -DODET_DIAG (include/odet_diag.h).
# Test / tool infrastructure (tools/_diag.py loads it explicitly); the product never loads it and the shipped library has no hook.
DIAG_LIB = os.path.join(os.path.dirname(PKG), 'tools', 'libodet_hip_diag.so')
DIAG_SOURCES = ['conv3x3.hip', 'conv_x3.hip']


def build_diag(force=False, verbose=False):
    build(force=False, verbose=verbose)               # (the other objects are the product's)
    hipcc = _hipcc()
    objs, procs = [], []
    for src in SOURCES:
        if src not in DIAG_SOURCES:
            objs.append(os.path.join(OBJ_DIR, src + '.o'))
            continue
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, src + '.diag.o')
        objs.append(o)
        if force or _stale(o, [s] + HEADERS + [os.path.join(INCLUDE, 'odet_diag.h'), os.path.abspath(__file__)]):
            cmd = [hipcc, '-DODET_DIAG'] + HIPCC_FLAGS + PER_SOURCE_FLAGS.get(src, []) + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s (diagnostic build)' % src)
    if force or procs or _stale(DIAG_LIB, objs):
        tmp = DIAG_LIB + '.tmp.%d' % os.getpid()
        # -Bsymbolic: the library's own calls bind to its own definitions even when the product library (same symbol names,
        # RTLD_GLOBAL) is loaded in the same process
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-Wl,-Bsymbolic'] + objs + ['-lpthread', '-o', tmp])
        os.replace(tmp, DIAG_LIB)
    return DIAG_LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
    if '--diag' in sys.argv:
        print(build_diag(force='--force' in sys.argv, verbose=True))
