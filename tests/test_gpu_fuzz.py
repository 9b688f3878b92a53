"""Seeded random sweep over meshes (radix, chirp-z and mixed lengths), rank counts, decompositions, precisions and
dealias modes against the oracle (scripts/fuzz_parity.py)."""
import os
import subprocess
import sys

import pytest

from gpu_util import have_gpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_configurations_match_the_oracle(seed):
    if not have_gpu():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "60", str(seed)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and "60 cases, 0 failures" in out, out[-3000:]
