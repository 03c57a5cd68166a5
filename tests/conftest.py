import os
import sys

# (tacotron_amd/lib.py: HIP reads this at its first API call; the test modules import torch before the package)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def built_lib():
    """libtaco_hip.so must exist (built in-tree by __graft_entry__.build()); build it if a fresh checkout lacks it."""
    path = os.path.join(ROOT, 'tacotron_amd', 'libtaco_hip.so')
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    from tacotron_amd import lib
    return lib
