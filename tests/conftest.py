import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# Several GPU tests run the reference's op sequence with PyTorch-ROCm (MIOpen) as a second opinion.  MIOpen's default exhaustive
# kernel search costs minutes on a fresh box (no find-db persists); the fast mode keeps the whole -m gpu suite under a minute.
# (libe3unet itself never calls MIOpen.)
os.environ.setdefault('MIOPEN_FIND_MODE', '2')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # a plain `pytest` on a machine without a GPU skips the GPU tests instead of failing them
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason='needs a ROCm GPU (MI355X)')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
