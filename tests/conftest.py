import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pytest_collection_modifyitems(config, items):
    """-m gpu tests must never silently skip on the GPU box; on a GPU-less host
    they are skipped unless explicitly selected."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLD, name + '.npz')))


def load_full():
    with open(os.path.join(GOLD, 'lss_full.json')) as f:
        return {e['tag']: e for e in json.load(f)}


@pytest.fixture(scope='session')
def full_cases():
    return load_full()
