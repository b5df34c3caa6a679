import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need an MI355X and the built library: skip (not fail) them on a CPU-only box, so that a plain
    `pytest tests` is green there.  On the GPU box nothing is skipped: a missing libtsc.so fails loudly."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason='needs a GPU (MI355X): run with -m gpu on the GPU box')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session', autouse=True)
def _few_cpu_threads():
    """The float64 oracle is thousands of tiny torch-CPU ops: on a many-core GPU host the default thread pool makes
    each of them slower (bench.py's cpu_baseline note), so cap it."""
    try:
        import torch
        torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    except Exception:
        pass
    yield


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
