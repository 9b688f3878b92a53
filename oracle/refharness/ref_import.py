"""Import the real reference (DEV CONTAINER ONLY).

The mounted reference is read-only and its Cython helper is not built, so a
scratch copy is made under /tmp, `setup.py build_ext --inplace` is run there,
and that copy is put on sys.path *after* the fake mpi4py and the py3/numpy-2
shims have been installed.  Nothing is copied into this repository.
"""
import os
import shutil
import subprocess
import sys

REF = "/root/reference"
SCRATCH = "/tmp/mpifft4py_ref_scratch"


def have_reference():
    return os.path.isdir(os.path.join(REF, "mpiFFT4py"))


def import_reference():
    from . import fake_mpi
    fake_mpi.install()
    if not os.path.isdir(os.path.join(SCRATCH, "mpiFFT4py")):
        os.makedirs(SCRATCH, exist_ok=True)
        for name in ("mpiFFT4py", "setup.py", "README.rst"):
            src = os.path.join(REF, name)
            dst = os.path.join(SCRATCH, name)
            if os.path.isdir(src):
                shutil.copytree(src, dst)
            else:
                shutil.copy(src, dst)
    import glob
    if not glob.glob(os.path.join(SCRATCH, "mpiFFT4py", "cython", "maths*.so")):
        env = dict(os.environ)
        subprocess.check_call([sys.executable, "setup.py", "build_ext", "--inplace"],
                              cwd=SCRATCH, env=env,
                              stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
    if SCRATCH not in sys.path:
        sys.path.insert(0, SCRATCH)
    import mpiFFT4py  # noqa: F401
    return mpiFFT4py
