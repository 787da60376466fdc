import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """The shared libraries are build artefacts (git-ignored): build them when a fresh checkout runs the tests.
    hipcc cross-compiles gfx950 without a GPU (~1-2 min once); the oracle's C NMS compiles in a second."""
    lib = os.path.join(ROOT, 'celldetection_amd', 'libcpn_hip.so')
    if not os.path.isfile(lib):
        from celldetection_amd import build
        build.build(verbose=False)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
