import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A test that hangs (a GPU box that stalls, a collective that never completes) must fail by itself instead of
    # holding the whole session: 15 minutes per test where pytest-timeout is installed (no test needs a tenth of that;
    # the multi-process ones carry their own, shorter limits).
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 900


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
