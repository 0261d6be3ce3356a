import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'ground-plane-polling_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


# the driver gives `pytest -m gpu` 900 s on its box; the default selection is meant to fit with a margin.  Going over it is REPORTED (the ten
# slowest tests are named) and changes the exit status only when asked to: GPP_ENFORCE_SUITE_BUDGET=1 (the builder's own collection runs set it),
# because a green suite on a slower or loaded box, on a cold tune cache or after a first-time library build must stay green
GPU_SUITE_BUDGET_S = float(os.environ.get('GPP_SUITE_BUDGET_S', '600'))


def pytest_addoption(parser):
    parser.addoption('--run-slow', action='store_true', default=False,
                     help='also run the soak-style parametrisations marked `slow` (tools/collect_r5.sh does; the default -m gpu run does not)')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: soak-style parametrisation (preemption loops, the fixture frames beyond the first 16, the less used '
                                       'types and plan variants): deselected unless --run-slow or -m names `slow`')
    config._gpp_durations = []
    config._gpp_t0 = None


def pytest_collection_modifyitems(config, items):
    if config.getoption('--run-slow') or 'slow' in (config.getoption('-m') or ''):
        return
    keep, drop = [], []
    for item in items:
        (drop if item.get_closest_marker('slow') else keep).append(item)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


def pytest_sessionstart(session):
    import time
    session.config._gpp_t0 = time.time()


def pytest_runtest_logreport(report):
    if report.when == 'call' or (report.when == 'setup' and report.duration > 1.0):
        _DURATIONS.append((report.duration, report.nodeid, report.when))


_DURATIONS = []


def pytest_sessionfinish(session, exitstatus):
    """ the default GPU selection should stay inside its budget (a suite that outgrows the driver's step limit turns every parity row red):
    over budget -> a warning with the ten slowest tests; a failure only under GPP_ENFORCE_SUITE_BUDGET=1 """
    import time
    marker = session.config.getoption('-m') or ''
    if 'gpu' not in marker or 'not gpu' in marker or session.config.getoption('--run-slow') or 'slow' in marker:
        return
    if len(_DURATIONS) < 200:                         # a hand-picked subset, not the suite
        return
    elapsed = time.time() - (session.config._gpp_t0 or time.time())
    if elapsed > GPU_SUITE_BUDGET_S:
        worst = sorted(_DURATIONS, reverse=True)[:10]
        enforce = os.environ.get('GPP_ENFORCE_SUITE_BUDGET', '0') == '1'
        print('\n{}: GPU suite took {:.0f} s, budget {:.0f} s.  Ten slowest:'.format('ERROR' if enforce else 'WARNING', elapsed, GPU_SUITE_BUDGET_S))
        for d, node, when in worst:
            print('  {:7.1f} s  {}  ({})'.format(d, node, when))
        if enforce:
            session.exitstatus = 1


def pytest_terminal_summary(terminalreporter):
    """ the parity margins the full-size ledger tests measured, in the log itself (a passed test prints nothing otherwise) """
    mod = sys.modules.get('test_fullsize_golden_gpu')
    lines = getattr(mod, 'LEDGER_LINES', None) if mod else None
    if lines:
        terminalreporter.write_sep('=', 'parity ledger: HIP path against the float64 oracle fixtures (utils/ledger.REFERENCE_BARS)')
        for line in lines:
            terminalreporter.write_line(line)


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
