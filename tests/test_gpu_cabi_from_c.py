"""The C ABI driven from a plain C program (tests/cabi/roundtrip.c, built with gcc against include/mpifft4py_amd.h):
analytic plane-wave spectrum, round trip, input preserved, error path."""
import os
import subprocess

import pytest

from gpu_util import have_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi", "roundtrip.c")


def _build(tmp_path):
    exe = str(tmp_path / "roundtrip")
    libdir = os.path.join(ROOT, "mpifft4py_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L", libdir, "-lmpifft4py_amd", "-lm", "-Wl,-rpath," + libdir])
    return exe


def test_c_program_builds_and_fails_loudly_without_gpu(tmp_path):
    exe = _build(tmp_path)
    if have_gpu():
        pytest.skip("a GPU is present")
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"C_ABI_ROUNDTRIP_OK" not in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mesh", [(32, 48, 64), (64, 64, 64), (36, 100, 50), (7, 9, 22), (256, 256, 256)])
def test_c_program_roundtrip(tmp_path, mesh):
    if not have_gpu():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    exe = _build(tmp_path)
    p = subprocess.run([exe] + [str(m) for m in mesh], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    out = p.stdout.decode()
    assert p.returncode == 0 and "C_ABI_ROUNDTRIP_OK" in out, out
