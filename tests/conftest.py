import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_T0 = time.time()
_LOG = None


def _host_threads():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import host_threads

    return host_threads(16)


def pytest_configure(config):
    # Thread pools sized for the cores this process may USE (affinity mask / cgroup quota), not for the machine: torch and
    # OpenMP default to os.cpu_count(), and on a shared GPU box (128+ cores reported, a fraction granted) the oracle's
    # GEMMs then run 12x slower (BENCH_r03.json cpu_baseline) -- that, not the GPU, is what made the round-3 suite miss the
    # driver's limit.  Child processes (DataLoader workers, torch.distributed ranks, the CLI tests) inherit the environment.
    n = str(_host_threads())
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ.setdefault(var, n)
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "long: tens of seconds or more (full-size parity, training runs): collected LAST, so a "
                                       "killed run cannot hide the cheap tests behind a slow one")


def pytest_collection_modifyitems(config, items):
    """Stable partition: everything not marked `long` first, in file order; then the long tests, cheapest file first."""
    rank = {"test_gpu_train.py": 1, "test_gpu_parity_full.py": 2}
    short = [it for it in items if it.get_closest_marker("long") is None]
    long_ = [it for it in items if it.get_closest_marker("long") is not None]
    long_.sort(key=lambda it: rank.get(os.path.basename(str(it.fspath)), 0))
    items[:] = short + long_
    # every test has a timeout: the marked ones their own (the long tests; with -x only ONE test can ever run into its limit, and the
    # largest mark -- 120 s -- plus the ~3 minutes of the whole suite stays far inside the driver's 900 s),
    # everything else 60 s (they take 0.01-5 s)
    for it in items:
        if it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(60))


def _duration_log():
    """gpurun_out/pytest_durations.log when that directory exists (it is merged back from the GPU box), else nothing."""
    global _LOG
    if _LOG is None:
        d = os.path.join(ROOT, "gpurun_out")
        try:
            _LOG = open(os.path.join(d, "pytest_durations.log"), "a") if os.path.isdir(d) and os.access(d, os.W_OK) else False
        except OSError:
            _LOG = False
    return _LOG


def pytest_runtest_logreport(report):
    """`seconds  elapsed  nodeid` after every test that took a second or more, flushed to the real stderr (pytest -q shows
    only dots; a driver that kills the run keeps the tail, and the tail then names where the time went) and, for every
    test, to gpurun_out/pytest_durations.log."""
    if report.when != "call":
        return
    line = f"{report.duration:7.1f}s  t+{time.time() - _T0:6.0f}s  {report.outcome:7s} {report.nodeid}\n"
    f = _duration_log()
    if f:
        f.write(line)
        f.flush()
    if report.duration >= 1.0:
        sys.__stderr__.write("\n" + line)
        sys.__stderr__.flush()


@pytest.fixture(scope="session", autouse=True)
def _thread_pools():
    import torch

    torch.set_num_threads(_host_threads())
    yield


@pytest.fixture(scope="session")
def oracle_maps():
    from oracle import maps

    maps.build()
    maps.set_threads(_host_threads())
    return maps
