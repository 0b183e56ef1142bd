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
    recompile with the flags; the other objects are the product build's (csrc/_obj, built first if stale)."""
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


if os.environ.get('ODET_LIB_PATH'):
    use_library(os.environ['ODET_LIB_PATH'])
