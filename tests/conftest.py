import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'ground-plane-polling_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session')
def oracle_lib():
    """ The C restatement of the polling stage (oracle/polling.c), built on demand. """
    import ctypes
    import subprocess
    path = os.path.join(ROOT, 'oracle', 'liboracle_polling.so')
    if not os.path.isfile(path):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')])
    return ctypes.CDLL(path)
