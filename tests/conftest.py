import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# every workspace is filled with NaN patterns when it is handed out (mtl_ssl_amd/ops.py:workspace): an entry point
# that skips part of its work cannot pass on what an earlier call left behind
os.environ.setdefault("MTLSSL_POISON_WS", "1")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_terminal_summary(terminalreporter):
    from tests import parity_report
    if parity_report.LINES:
        terminalreporter.section("observed parity (GPU path vs CPU oracle)")
        for line in parity_report.LINES:
            terminalreporter.write_line(line)
