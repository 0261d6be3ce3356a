import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'ground-plane-polling_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session', autouse=True)
def shared_tile_choices(tmp_path_factory):
    """ One per-layer tile-choice file for the whole session (GPP_TUNE_CACHE): the dozens of models the GPU tests build time a layer's
    candidate tiles once per (backbone, type, layer, batch, image size) instead of once per model.  A tile never changes a result
    (test_every_tile_gives_identical_results, test_random_tiles_never_change_a_byte draw theirs at random, past this file); child
    processes inherit the variable. """
    if 'GPP_TUNE_CACHE' not in os.environ:
        os.environ['GPP_TUNE_CACHE'] = str(tmp_path_factory.mktemp('tiles') / 'tile_choices.json')
    yield


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
