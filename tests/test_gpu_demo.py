"""GPU: the Taylor-Green known-answer test of the reference demo
(demo/spectral_dns_solver.py:103-105), with the FFT class swapped for
mpifft4py_amd, against the value the REAL reference produced
(tests/golden/taylor_green.json)."""
import json
import os
import sys

import pytest

from gpu_util import have_gpu, run_ranks

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not have_gpu():
        pytest.fail("no GPU visible")


@pytest.mark.parametrize("dealias", ["3/2-rule", "2/3-rule", None])
@pytest.mark.parametrize("decomp,P", [("slab", 1), ("slab", 2), ("slab", 4), ("pencil", 4)])
def test_taylor_green_known_answer(decomp, P, dealias, golden_dir):
    import spectral_dns_solver as demo
    gold = json.load(open(os.path.join(golden_dir, "taylor_green.json")))
    ks = run_ranks(P, lambda comm: demo.solve(comm, dealias=dealias, decomposition=decomp))
    k = ks[0]
    assert round(k - gold["k_expected_demo"], 7) == 0            # the demo's own assertion
    ref = gold["k_P1_%s" % dealias]
    assert abs(k - ref) < 1e-12, (k, ref)
