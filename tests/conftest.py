import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The whole suite runs with the append path's copy guard on (csrc/table.cpp: host_to_device -- every host -> device copy of
# caller memory is digested in HBM and compared with the host bytes; child processes inherit it).  It exists because of one
# unexplained wrong-keys event in round 5: a repeat fails at the copy, loudly.
os.environ.setdefault("SYBL_VERIFY_COPIES", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "eager_only: not repeated with SYBL_LAZY_ROWS=1 by the rows_mode fixture")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(params=["default", "lazy"])
def rows_mode(request, monkeypatch):
    """Runs a test twice: as is, and with SYBL_LAZY_ROWS=1 -- every result, whatever its size, builds its rows on first
    access (result.cpp: result_ensure_rows), so the lazy path sees every query shape of the file that asks for this fixture
    (pytestmark usefixtures in the hash / CLI / loghist files)."""
    if request.param == "lazy":
        if request.node.get_closest_marker("eager_only"):
            pytest.skip("a minute-long test whose results are lazy anyway (>= 2048 rows): run once")
        monkeypatch.setenv("SYBL_LAZY_ROWS", "1")
    return request.param
