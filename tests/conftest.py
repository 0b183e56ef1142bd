import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session', autouse=True)
def _built_libraries():
    """Both shared libraries are build artefacts (git-ignored).  Build them if absent/stale; on the
    GPU box they arrive prebuilt with the snapshot and this is a no-op."""
    from oracle import build as oracle_build
    from tf_eager_object_detection_amd import _build as odet_build
    oracle_build.build()
    odet_build.build()


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'ref_numpy_vectors.npz'))
