import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    import __graft_entry__ as g
    pkg_lib = os.path.join(ROOT, "project3-cuda-path-tracer-1_amd", "csrc", "libptk.so")
    orc_lib = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(pkg_lib) and os.path.exists(orc_lib)):
        g.build()


@pytest.fixture(scope="session")
def ptk():
    _ensure_built()
    return importlib.import_module("project3-cuda-path-tracer-1_amd")


@pytest.fixture(scope="session")
def orc():
    _ensure_built()
    import oracle_lib
    return oracle_lib


@pytest.fixture(scope="session")
def scenes_dir():
    return os.path.join(ROOT, "scenes")


def host_threads():
    """CPU threads the oracle may really use here: the affinity mask cut by the cgroup quota (a GPU box shows every core of the host but
    grants 16), never more than 16 — oversubscribed OpenMP loops are far slower than undersubscribed ones."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        tok = open("/sys/fs/cgroup/cpu.max").read().split()
        if tok[0] != "max":
            n = min(n, max(1, int(int(tok[0]) / int(tok[1]))))
    except Exception:
        pass
    return max(1, min(n, 16))
