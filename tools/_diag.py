"""Diagnostics only (tools/): points the package's ctypes loader at a side build of the same C ABI and builds such side
libraries with extra compiler flags.  The product (`tf_eager_object_detection_amd/_lib.py`, `_build.py`) reads no
environment switch; tools that compare builds import this module FIRST:

    import tools._diag            # honours ODET_LIB_PATH=<path to a libodet_*.so of the same ODET_VERSION>
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tf_eager_object_detection_amd import _build, _lib          # noqa: E402


def use_library(path):
    """the library every later _lib.lib() call of this process loads (must be called before the first one)"""
    if _lib._lib is not None:
        raise RuntimeError('the library is already loaded')
    _lib.LIB_PATH = os.path.abspath(path)


def build_variant(out_path, extra_flags=(), sources=None, patch=None, only=None):
    """libodet with `extra_flags` (e.g. -D switches a patch under tools/exp introduces) into `out_path`; `patch` is applied
    with `git apply` to a temporary copy of csrc/ first and the build FAILS if it does not apply.  `only`: the sources to
    recompile with the flags; the other objects are the product build's (csrc/_obj, built first if stale).  Variants that
    force tiles (odet_debug_*) pass '-DODET_DIAG' in `extra_flags` and are loaded with ODET_LIB_PATH + DIAG_SIGNATURES."""
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix='odet_variant_')
    try:
        pkg = os.path.join(tmp, 'tf_eager_object_detection_amd')
        shutil.copytree(os.path.join(ROOT, 'tf_eager_object_detection_amd', 'csrc'), os.path.join(pkg, 'csrc'),
                        ignore=shutil.ignore_patterns('_obj'))
        shutil.copytree(os.path.join(ROOT, 'include'), os.path.join(tmp, 'include'))
        if patch:
            subprocess.check_call(['git', 'apply', '--verbose', os.path.abspath(patch)], cwd=tmp)
        objs = []
        if only:
            _build.build()
        for src in (sources or _build.SOURCES):
            if only and src not in only:
                objs.append(os.path.join(_build.OBJ_DIR, src + '.o'))
                continue
            o = os.path.join(tmp, src + '.o')
            cmd = [_build._hipcc()] + list(extra_flags) + _build.HIPCC_FLAGS + _build.PER_SOURCE_FLAGS.get(src, []) + \
                  ['-I', os.path.join(tmp, 'include'), '-c', os.path.join(pkg, 'csrc', src), '-o', o]
            subprocess.check_call(cmd)
            objs.append(o)
        subprocess.check_call([_build._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-lpthread', '-o', out_path])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out_path


# ---- the diagnostic build of the library (tools/libodet_hip_diag.so: _build.build_diag(), include/odet_diag.h) ----------------
DIAG_SIGNATURES = {
    'odet_debug_conv_tile': (_lib._i, [_lib._i] * 5),
    'odet_debug_x3_tile': (_lib._i, [_lib._i] * 3),
}
_diag_handle = None


def diag_handle():
    """ctypes handle of the diagnostic library (every product signature + the odet_debug_* hooks); built on demand where hipcc
    exists, otherwise it must have travelled with the snapshot (__graft_entry__.build() builds it)"""
    global _diag_handle
    if _diag_handle is None:
        import ctypes as C
        import torch  # noqa: F401  (its HIP runtime first, as _lib.lib() does)
        path = _build.DIAG_LIB
        if not os.path.exists(path):
            _build.build_diag()
        h = C.CDLL(path, mode=C.RTLD_LOCAL)
        for name, (res, args) in list(_lib.SIGNATURES.items()) + list(DIAG_SIGNATURES.items()):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        if h.odet_version() != _lib.ODET_VERSION:
            raise _lib.OdetError('libodet_hip_diag.so version mismatch: %d' % h.odet_version())
        _diag_handle = h
    return _diag_handle


class diag_library:
    """with tools._diag.diag_library() as lib:  -- every C-ABI call of the block (ops.*, _lib.call) goes to the DIAGNOSTIC build,
    which alone has odet_debug_conv_tile / odet_debug_x3_tile; any forced tile is cleared on the way out and the product
    library is back afterwards.  Both libraries are stateless apart from that override, so buffers made by one work with the other."""

    def __enter__(self):
        _lib.lib()                                    # (the product library first: it stays the process's RTLD_GLOBAL one)
        self.prev = _lib._lib
        _lib._lib = diag_handle()
        return _lib._lib

    def __exit__(self, *exc):
        try:
            _lib._lib.odet_debug_conv_tile(0, 0, 0, 0, 0)
            _lib._lib.odet_debug_conv_tile(1, 0, 0, 0, 0)
            _lib._lib.odet_debug_x3_tile(0, 0, 0)
        finally:
            _lib._lib = self.prev
        return False


def use_diag_build():
    """(scripts) the diagnostic build for the WHOLE process: call before the first _lib.lib()"""
    use_library(_build.build_diag() if not os.path.exists(_build.DIAG_LIB) else _build.DIAG_LIB)
    _lib.SIGNATURES.update(DIAG_SIGNATURES)


if os.environ.get('ODET_LIB_PATH'):
    use_library(os.environ['ODET_LIB_PATH'])
