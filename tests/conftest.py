import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: soak-like parametrisations; deselected unless MFFT_TEST_SLOW=1 (or -m names 'slow')")
    # A test that hangs (a GPU box that stalls, a collective that never completes) must fail by itself instead of
    # holding the whole session: 5 minutes per test where pytest-timeout is installed (the slowest takes 90 s; the
    # multi-process ones carry their own limits of at most 240 s and give their workers MFFT_LOCAL_TIMEOUT = 30 s).
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 300


def pytest_collection_modifyitems(config, items):
    """`-m gpu` must finish well inside the driver's step limit (round 4: 656 s of 1200) so that one stalled subprocess is
    a named failure and not a timeout of the whole step: the soak-like parametrisations carry the `slow` marker and run
    only with MFFT_TEST_SLOW=1 (scripts/multi_gpu_check.sh, the end-of-round script) or when -m asks for them.  What stays
    covers every kernel family, rank count and pipeline flavour at least once (the reference gates its own matrix by rank
    count the same way, tests/test_FFT.py:27-34)."""
    if os.environ.get("MFFT_TEST_SLOW", "0") not in ("", "0") or "slow" in (config.option.markexpr or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("slow") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
